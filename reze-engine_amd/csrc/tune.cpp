// tune.cpp — the launch-shape search (rz_autotune*) and the tuning keys (ctx.h).
#include "ctx.h"

using namespace rzi;

extern "C" {

int rz_autotune_measure(rz_ctx *c, uint32_t frames, rz_tune_entry *table, int cap, int *count)
{
    if (int r = use(c)) return r;
    if (!table || cap < 1 || !count) return fail(RZ_ERR_INVALID, "rz_autotune_measure: bad table");
    *count = 0;
    if (int r = check_ready(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    if (frames == 0) frames = 100;
    frames = std::min<uint32_t>(frames, 1000);      // a search, not a benchmark
    // Entry 0 = the heuristic plan. Then: morph split x workgroups per CU (single mesh), poses per workgroup x workgroups per
    // CU (instanced). Every candidate is a legal plan; the search only picks among the variants the parity tests already cover.
    std::vector<rz_tune_entry> cands;
    auto add = [&](int split, int capv, int loop) {
        rz_tune_entry e;
        memset(&e, 0, sizeof e);
        e.morph_split = split; e.grid_cap = capv; e.inst_loop = loop; e.same_as = -1;
        cands.push_back(e);
    };
    const int ncu = c->n_cu;
    const int keep_split = c->t_split, keep_cap = c->t_grid_cap, keep_loop = c->t_instloop;
    const bool keep_tuned = c->tuned_by_search;
    InstShape is;
    const bool instanced = inst_shape(c, &is);      // with the caller's inst_loop: 0 (crowd kernel off) and 9 (register form) are not searched over
    c->t_split = 0; c->t_grid_cap = 0; if (instanced) c->t_instloop = -1;
    add(0, 0, -1);
    if (instanced) {
        // total workgroups: one or two rounds of what the CUs hold at once (two 256-thread workgroups or one larger one);
        // with bone subsets the palettes of 16 poses still fit, which halves the number of workgroup fronts
        for (int loop : {8, 4, 16})
            for (int capv : {ncu, 2 * ncu, 3 * ncu, 4 * ncu}) add(0, capv, loop);
    } else {
        const int smax = c->morph_mode == 1 ? (int)std::min<uint32_t>(8, std::max<uint32_t>(1, c->M)) : 4;
        for (int sp = 1; sp <= smax; sp <<= 1)
            for (int capv : {ncu, 2 * ncu, 4 * ncu}) add(sp, (int)(capv * c->I), 0);
    }
    if ((int)cands.size() > cap) cands.resize(cap);
    const int n = (int)cands.size();
    auto restore = [&]() { c->t_split = keep_split; c->t_grid_cap = keep_cap; c->t_instloop = keep_loop; c->tuned_by_search = keep_tuned; };
    std::vector<Plan> plans(n);
    for (int i = 0; i < n; ++i) {
        c->t_split = cands[i].morph_split; c->t_grid_cap = cands[i].grid_cap; c->t_instloop = instanced ? cands[i].inst_loop : keep_loop;
        if (int r = frame_plan(c, &plans[i])) { restore(); return r; }
        const Plan &pl = plans[i];
        cands[i].eff_split = pl.v.S; cands[i].eff_grid = (int)pl.grid_x; cands[i].eff_inst_group = pl.inst_group;
        for (int k = 0; k < i && cands[i].same_as < 0; ++k)      // different requests often resolve to the same launch
            if (plans[k].grid_x == pl.grid_x && (pl.inst_group > 0 || plans[k].quads_per_wave == pl.quads_per_wave) && plans[k].v.S == pl.v.S &&
                plans[k].inst_group == pl.inst_group && plans[k].verts_per_wg == pl.verts_per_wg && plans[k].subsets == pl.subsets &&
                plans[k].inst_block == pl.inst_block)
                cands[i].same_as = cands[k].same_as >= 0 ? cands[k].same_as : k;
    }
    constexpr int kRounds = 5;
    std::vector<float> t((size_t)n * kRounds, 0.f);
    int rc = RZ_OK;
    {
        // the GPU reaches its sustained clocks only after a while of work (measured: the first timed round of the first
        // candidate came out 15-20 % slow on a cold device, which is enough to move a median of three): run the heuristic
        // plan for ~0.25 s first, untimed
        c->t_split = cands[0].morph_split; c->t_grid_cap = cands[0].grid_cap; c->t_instloop = instanced ? cands[0].inst_loop : keep_loop;
        Plan warm;
        rc = frame_plan(c, &warm);      // (a crowd's run lists belong to ONE launch shape: bring them back to entry 0's)
        if (rc == RZ_OK) rc = set_overlap(c, want_overlap(c, warm));
        const auto t0 = std::chrono::steady_clock::now();
        while (rc == RZ_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(250)) {
            for (uint32_t f = 0; f < 64 && rc == RZ_OK; ++f) rc = run_frame(c, warm);
            if (rc == RZ_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(RZ_ERR_HIP, "rz_autotune_measure: warm-up failed");
        }
    }
    for (int round = -1; round < kRounds && rc == RZ_OK; ++round) {          // round -1 warms every variant up, untimed
        for (int i = 0; i < n && rc == RZ_OK; ++i) {
            if (cands[i].same_as >= 0) continue;
            c->t_split = cands[i].morph_split; c->t_grid_cap = cands[i].grid_cap; c->t_instloop = instanced ? cands[i].inst_loop : keep_loop;
            Plan pl;
            if ((rc = frame_plan(c, &pl)) != RZ_OK) break;       // crowd shapes alternate: the run lists follow (a rebuild, untimed)
            if ((rc = set_overlap(c, want_overlap(c, pl))) != RZ_OK) break;
            const uint32_t nf = round < 0 ? 8 : frames;
            for (uint32_t f = 0; f < 4 && rc == RZ_OK; ++f) rc = run_frame(c, pl);
            if (rc != RZ_OK) break;
            hipError_t he = hipEventRecord(c->ev0, c->stream);
            for (uint32_t f = 0; f < nf && rc == RZ_OK; ++f) rc = run_frame(c, pl);
            if (rc != RZ_OK) break;
            if (he == hipSuccess) he = hipEventRecord(c->ev1, c->stream);
            if (he == hipSuccess) he = hipEventSynchronize(c->ev1);
            float ms = 0.f;
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, c->ev0, c->ev1);
            if (he != hipSuccess) { rc = fail(RZ_ERR_HIP, "rz_autotune_measure: %s", hipGetErrorString(he)); break; }
            if (round >= 0) t[(size_t)i * kRounds + round] = ms / nf;
        }
    }
    restore();
    if (rc != RZ_OK) return rc;
    for (int i = 0; i < n; ++i) {
        const int src = cands[i].same_as >= 0 ? cands[i].same_as : i;
        float v[kRounds];
        for (int k = 0; k < kRounds; ++k) v[k] = t[(size_t)src * kRounds + k];
        std::sort(v, v + kRounds);
        cands[i].ms = v[kRounds / 2]; cands[i].ms_min = v[0]; cands[i].ms_max = v[kRounds - 1];
        table[i] = cands[i];
    }
    *count = n;
    return RZ_OK;
}

int rz_autotune_pick(const rz_tune_entry *table, int count)
{
    if (!table || count < 1) return 0;
    // entry 0 (the heuristics) stays unless something is clearly faster: median >= 2 % lower AND, when the table carries the
    // per-round spread, its slowest round still under the heuristic's fastest one (overlapping ranges are box noise, and a pick that
    // follows noise differs from run to run); among the qualifying entries, the lowest median
    int best = 0;
    float best_ms = table[0].ms * 0.98f;
    const bool spread = table[0].ms_min > 0.f;
    for (int i = 1; i < count; ++i) {
        if (table[i].same_as == 0 || !(table[i].ms > 0.f) || !(table[i].ms < best_ms)) continue;
        if (spread && table[i].ms_max > 0.f && !(table[i].ms_max < table[0].ms_min)) continue;
        best = i; best_ms = table[i].ms;
    }
    return best;
}

int rz_autotune_apply(rz_ctx *c, const rz_tune_entry *e)
{
    if (int r = use(c)) return r;
    if (!e) return fail(RZ_ERR_INVALID, "rz_autotune_apply: null entry");
    const int sp = e->morph_split;
    if (sp != 0 && sp != 1 && sp != 2 && sp != 4 && sp != 8) return fail(RZ_ERR_INVALID, "rz_autotune_apply: morph_split %d", sp);
    if (e->grid_cap < 0 || e->inst_loop < -1 || e->inst_loop == 1 || e->inst_loop > 64) return fail(RZ_ERR_INVALID, "rz_autotune_apply: bad entry");
    if (e->inst_loop == 9 && !rz_has_all_variants())
        return fail(RZ_ERR_UNSUPPORTED, "rz_autotune_apply: inst_loop = 9 selects the register-resident crowd kernel, which the product library does not carry");
    // a caller who switched the crowd kernel off (inst_loop = 0) or chose the register form (9) keeps that choice: the entry's
    // pose-group size only applies where the search itself would have used one
    InstShape is;
    const bool instanced = inst_shape(c, &is);
    c->t_split = sp; c->t_grid_cap = e->grid_cap;
    if (instanced) c->t_instloop = e->inst_loop;
    c->tuned_by_search = true;
    return RZ_OK;
}

int rz_autotune(rz_ctx *c, uint32_t frames)
{
    rz_tune_entry table[32];
    int n = 0;
    if (int r = rz_autotune_measure(c, frames, table, 32, &n)) return r;
    if (n < 1) return RZ_OK;
    return rz_autotune_apply(c, &table[rz_autotune_pick(table, n)]);
}

int rz_set_tuning(rz_ctx *c, const char *key, int value)
{
    if (!c || !key) return fail(RZ_ERR_INVALID, "null argument");
    // variants that were measured slower everywhere are compiled into the tools-only build (make variants), not the product
    if (!rz_has_all_variants() && ((!strcmp(key, "unroll") && value == 4) || (!strcmp(key, "geo_lds") && value != 0) ||
                                   (!strcmp(key, "nontemporal") && value == 0) || (!strcmp(key, "inst_loop") && value == 9)))
        return fail(RZ_ERR_UNSUPPORTED, "%s = %d selects a kernel variant the product library does not carry (tools-only build: make -C reze-engine_amd/csrc variants)", key, value);
    if (!strcmp(key, "morph_split") || !strcmp(key, "grid_cap") || !strcmp(key, "inst_loop")) c->tuned_by_search = false;   // the caller owns the shape now
    if (!strcmp(key, "morph_split")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(RZ_ERR_INVALID, "morph_split must be 0 (auto),1,2,4,8");
        c->t_split = value;
    } else if (!strcmp(key, "unroll")) {
        if (value != 0 && value != 4 && value != 8) return fail(RZ_ERR_INVALID, "unroll must be 0 (auto), 4 or 8");
        c->t_unroll = value;
    } else if (!strcmp(key, "grid_cap")) {
        if (value < 0) return fail(RZ_ERR_INVALID, "grid_cap must be >= 0");
        c->t_grid_cap = value;
    } else if (!strcmp(key, "nontemporal")) {
        c->t_nt = value ? 1 : 0;
    } else if (!strcmp(key, "geo_lds")) {
        c->t_geo = value ? 1 : 0;
    } else if (!strcmp(key, "nt_store")) {
        c->t_nts = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(key, "out_cap")) {
        if (value < -1 || value > 2048) return fail(RZ_ERR_INVALID, "out_cap must be -1 (auto), 0 (off) or 64..2048 vertices per wave");
        c->t_outcap = value;
    } else if (!strcmp(key, "graph")) {
        if (value < 0 || value > 1) return fail(RZ_ERR_INVALID, "graph must be 0 or 1");
        c->t_graph = value;
    } else if (!strcmp(key, "dbg")) {
#ifdef RZ_ABLATE
        c->t_dbg = value;
#else
        // ablation modes (they make the kernels skip work, i.e. emit garbage) are compiled into the tools-only build only
        return fail(RZ_ERR_INVALID, "tuning key 'dbg' does not exist in the product library (tools-only build: make -C reze-engine_amd/csrc ablate)");
#endif
    } else if (!strcmp(key, "inst_loop")) {
        if (value < -1 || value == 1 || value > 64) return fail(RZ_ERR_INVALID, "inst_loop must be -1 (auto), 0 (off), 2..8 / 10..64 (poses per workgroup, LDS form) or 9 (register form)");
        c->t_instloop = value;
    } else if (!strcmp(key, "pose_prefetch")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "pose_prefetch must be -1 (auto = on), 0 (a zero-copy frame never stages the next pose) or 1");
        c->t_prefetch = value;
    } else if (!strcmp(key, "inst_subsets")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "inst_subsets must be -1 (auto = on), 0 (crowd frames always stage the whole palette) or 1");
        c->t_subsets = value;
    } else if (!strcmp(key, "fuse_fk")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "fuse_fk must be -1 (auto), 0 (always rz_fk_kernel in front) or 1 (every device-animated single character)");
        c->t_fusefk = value;
    } else if (!strcmp(key, "zero_copy")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "zero_copy must be -1 (auto = on for one character), 0 (every pose is copied to the device) or 1");
        c->t_zerocopy = value;
    } else if (!strcmp(key, "fuse_fk_plain")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "fuse_fk_plain must be -1 (auto = on), 0 (the fused frame always runs the generic hierarchy solve) or 1");
        c->t_fkplain = value;
    } else if (!strcmp(key, "pose_pull")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "pose_pull must be -1 (auto: a crowd's world matrices are pulled, local rotations copied), 0 (every pose is copied by hipMemcpyAsync as it was handed over) or 1 (every pose of more than 256 KB is pulled)");
        c->t_pull = value;
    } else if (!strcmp(key, "overlap")) {
        if (value < -1 || value > 1) return fail(RZ_ERR_INVALID, "overlap must be -1 (auto = off), 0 (off) or 1 (crowds: front kernels on the upload stream)");
        c->t_overlap = value;
    } else if (!strcmp(key, "inst_order")) {
        if (value != 0 && value != 1) return fail(RZ_ERR_INVALID, "inst_order must be 0 (an XCD takes one vertex run of every pose group) or 1 (every vertex run of its pose groups)");
        c->t_instorder = value;
    } else if (!strcmp(key, "inst_block")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(RZ_ERR_INVALID, "inst_block must be 0 (auto), 256, 512 or 1024 threads per workgroup");
        c->t_instblock = value;
        c->tuned_by_search = false;
    } else if (!strcmp(key, "fast")) {
        c->t_fast = value;        // -1 auto, 0 never (always prep kernel), 1 when possible
    } else {
        return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
    }
    return RZ_OK;
}

int rz_get_tuning(rz_ctx *c, const char *key, int *value)
{
    if (!c || !key || !value) return fail(RZ_ERR_INVALID, "null argument");
    if (!strncmp(key, "effective_", 10)) {
        // (an unknown key is refused BEFORE the work below: it drains the stream and may rebuild the run lists)
        static const char *const known[] = { "nt", "nt_store", "geo", "prep", "split", "unroll", "fast", "variant", "fk_kind", "fuse_fk", "closure_bones",
                                             "overlap", "inst_block", "out_cap", "inst_group", "poses_per_wg", "grid", "subsets", "subset_bones", "inst_lds" };
        bool ok = false;
        for (const char *k : known) ok = ok || !strcmp(key + 10, k);
        if (!ok) return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
        // what the NEXT frame will launch: a crowd's plan depends on the run lists of its launch shape, so bring them up to
        // date first (as every entry point that launches frames does) instead of describing the whole-palette fallback
        if (int r = use(c)) return r;
        if (c->V && c->B) {
            if (int r = ensure_run_subsets(c)) return r;
            if (int r = ensure_subfk(c)) return r;
        }
    }
    if (!strcmp(key, "morph_split")) *value = c->t_split;
    else if (!strcmp(key, "unroll")) *value = c->t_unroll;
    else if (!strcmp(key, "grid_cap")) *value = c->t_grid_cap;
    else if (!strcmp(key, "nontemporal")) *value = c->t_nt;
    else if (!strcmp(key, "geo_lds")) *value = c->t_geo;
    else if (!strcmp(key, "bones")) *value = (int)c->B;
    else if (!strcmp(key, "morphs")) *value = (int)c->M;
    else if (!strcmp(key, "instances")) *value = (int)c->I;
    else if (!strcmp(key, "verts")) *value = (int)c->V;
    else if (!strcmp(key, "nt_store")) *value = c->t_nts;
    else if (!strcmp(key, "fast")) *value = c->t_fast;
    else if (!strcmp(key, "morph_mode")) *value = c->morph_mode;
    else if (!strcmp(key, "effective_nt")) *value = make_plan(c).v.nt && c->morph_mode == 1 ? 1 : 0;
    else if (!strcmp(key, "effective_nt_store")) *value = make_plan(c).v.nts ? 1 : 0;
    else if (!strcmp(key, "effective_geo")) *value = make_plan(c).v.geo ? 1 : 0;
    else if (!strcmp(key, "effective_prep")) { const Plan pl = make_plan(c); *value = ((pl.prep || c->pose_local) && !pl.fuse_fk && !pl.subfk) ? 1 : 0; }
    else if (!strcmp(key, "effective_split")) *value = make_plan(c).v.S;
    else if (!strcmp(key, "effective_unroll")) *value = make_plan(c).v.U;
    else if (!strcmp(key, "effective_fast")) *value = make_plan(c).v.fast ? 1 : 0;
    else if (!strcmp(key, "inst_loop")) *value = c->t_instloop;
    else if (!strcmp(key, "inst_block")) *value = c->t_instblock;
    else if (!strcmp(key, "inst_order")) *value = c->t_instorder;
    else if (!strcmp(key, "overlap")) *value = c->t_overlap;
    else if (!strcmp(key, "zero_copy")) *value = c->t_zerocopy;
    else if (!strcmp(key, "pose_pull")) *value = c->t_pull;
    else if (!strcmp(key, "fuse_fk_plain")) *value = c->t_fkplain;
    else if (!strcmp(key, "effective_variant")) {
        // the last template argument of the single-mesh frame kernel the next frame launches (kernels/deform_small.hip / deform_dense.hip:
        // launch_one): 0 everything compiled in, 3 without the fused consumers, 1 / 2 without them and with the specialised solve
        Plan pl;
        if (int r = frame_plan(c, &pl)) return r;
        const RzDeformParams dp = deform_params(c, pl);
        const bool shapes = !pl.v.geo && (pl.v.mode != 1 || (pl.v.nt && pl.v.U == 8));
        *value = (!shapes || dp.edge || dp.aabb) ? 0 : ((!pl.v.fast && dp.fk_on && (dp.fk_kind == 1 || dp.fk_kind == 2)) ? dp.fk_kind : 3);
    }
    else if (!strcmp(key, "effective_fk_kind")) { Plan pl; if (int r = frame_plan(c, &pl)) return r; *value = deform_params(c, pl).fk_kind; }
    else if (!strcmp(key, "pose_pulled")) *value = c->last_upload_pulled ? 1 : 0;      // the most recent copied pose came down by rz_pull_pose_kernel ...
    else if (!strcmp(key, "pose_rows")) *value = c->last_upload_rows ? 1 : 0;          // ... its world matrices as three rows per bone
    else if (!strcmp(key, "fuse_fk")) *value = c->t_fusefk;
    else if (!strcmp(key, "effective_fuse_fk")) { const Plan pl = make_plan(c); *value = (pl.fuse_fk || pl.subfk) ? 1 : 0; }
    else if (!strcmp(key, "effective_closure_bones")) *value = make_plan(c).subfk ? (int)c->subfk_stride : 0;
    else if (!strcmp(key, "pose_resident")) *value = (c->zc_cur < 0 || (c->world_resident && c->mw_resident && c->local_resident)) ? 1 : 0;
    else if (!strcmp(key, "effective_overlap")) *value = want_overlap(c, make_plan(c)) ? 1 : 0;
    else if (!strcmp(key, "effective_inst_block")) *value = make_plan(c).inst_block;
    else if (!strcmp(key, "out_cap")) *value = c->t_outcap;
    else if (!strcmp(key, "graph")) *value = c->t_graph;
    else if (!strcmp(key, "effective_out_cap")) *value = (int)make_plan(c).out_cap;
    else if (!strcmp(key, "effective_inst_group")) *value = make_plan(c).inst_group;
    else if (!strcmp(key, "effective_poses_per_wg")) *value = make_plan(c).poses_per_wg;
    else if (!strcmp(key, "effective_grid")) *value = (int)make_plan(c).grid_x;
    else if (!strcmp(key, "inst_subsets")) *value = c->t_subsets;
    else if (!strcmp(key, "all_variants")) *value = rz_has_all_variants() ? 1 : 0;
    else if (!strcmp(key, "pose_prefetch")) *value = c->t_prefetch;
    else if (!strcmp(key, "pose_staged")) {
        // did the helper of an earlier frame stage the CURRENT pose in device memory? (synchronises; for tests and tools)
        *value = 0;
        if (c->zc_tag && c->zc_seq_cur) {
            uint64_t tags[2] = {0, 0};
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(tags, c->zc_tag, sizeof tags, hipMemcpyDeviceToHost));
            *value = tags[c->pose_slot] == c->zc_seq_cur ? 1 : 0;
        }
    }
    else if (!strncmp(key, "addr_", 5)) {
        // diagnostics (tools/archive/placement.py): where the allocator put a buffer — bits [12, 43) of its device address, i.e. the address in
        // 4 KB pages (where a buffer lands relative to the 2 MB large-page frame moves the C5 frame by 3 %: NOTEBOOK.md R5.3)
        const void *ptr = !strcmp(key, "addr_dense") ? (const void *)c->dense : !strcmp(key, "addr_out") ? (const void *)c->out_pos :
                          !strcmp(key, "addr_nrm") ? (const void *)c->out_nrm : !strcmp(key, "addr_geom") ? (const void *)c->geom : nullptr;
        if (!ptr && strcmp(key, "addr_dense") && strcmp(key, "addr_out") && strcmp(key, "addr_nrm") && strcmp(key, "addr_geom")) return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
        *value = (int)(((uintptr_t)ptr >> 12) & 0x7fffffffu);
    }
    else if (!strcmp(key, "effective_subsets")) *value = make_plan(c).subsets ? 1 : 0;
    else if (!strcmp(key, "effective_subset_bones")) *value = (int)make_plan(c).sub_bones;
    else if (!strcmp(key, "effective_inst_lds")) *value = (int)make_plan(c).inst_lds;
    else return fail(RZ_ERR_INVALID, "unknown tuning key '%s'", key);
    return RZ_OK;
}

}  // extern "C"
