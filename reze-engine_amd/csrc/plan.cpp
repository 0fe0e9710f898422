// plan.cpp — launch shapes (Plan) and the parameter blocks of the kernels (ctx.h).
#include "ctx.h"

using namespace rzi;

namespace rzi {

namespace {

int auto_split(const rz_ctx *c)
{
    if (c->morph_mode != 1) return 1;
    // S lanes share a quad, so waves = quads * S / 64, and a wave step is 256 / S vertices whose skin phase (and output stores) follow
    // its morph loads. Measured on MI355X, round 6 (tools/plan_sweep.py: 15 mesh sizes x {S} x {whole steps per wave}, round-robin
    // medians, profiles/r6_plan_sweep.txt): the 1 M-vertex mesh streams 3 % faster with two lanes per quad than with one and S = 2 stays
    // best down to ~365 k vertices (see below); under ~300 k a wave at S = 2 is left with ONE long step — every wave loads first and skins last, nothing
    // overlaps (250 k vertices: 37.4 us, against 34.7 us as two steps at S = 4) — so S = 4 down to ~95 k vertices, where the same
    // happens to it, and S = 8 below (C3). A dense frame never runs at S = 1, which also keeps rz_autotune's pick on the heuristic plan
    // instead of flipping between two near-equal candidates from run to run.
    // The S = 2 / S = 4 boundary sits where S = 4 stops filling three whole 64-vertex steps per wave of the persistent grid (~367 k
    // vertices): between 300 k and 365 k vertices S = 2 left every wave with 1.25-1.5 steps and ran 6-12 % behind S = 4 in every one of
    // 24 fresh processes (tools/fresh_plans.py, profiles/r6_fresh_plans_mid.txt: 313 856 vertices 47.2-47.8 us against 41.1-41.9).
    const uint64_t quads = (uint64_t)c->Vp / 4 * c->I;
    int S = 2;
    if (quads * 2 / 64 < 2870) S = 4;
    if (quads * 4 / 64 < 1500) S = 8;
    while (S > 1 && (uint32_t)S > c->M) S >>= 1;
    return S;
}

// Quads per wave of a frame's persistent grid: `cap` workgroups of four waves, every wave one contiguous run (a multiple of 8 quads =
// 128 B per plane). `rules` = the heuristics of a dense single-mesh frame whose launch shape the caller left to the library.
uint32_t run_per_wave(uint32_t n_quads, uint32_t cap, int S, bool rules, bool one_mesh)
{
    const uint32_t waves_per_wg = 4, qpw_step = 64 / (uint32_t)S;
    const uint32_t max_useful = (n_quads + waves_per_wg * qpw_step - 1) / (waves_per_wg * qpw_step);
    const uint32_t gx = std::max<uint32_t>(1, std::min(std::max<uint32_t>(1, cap), max_useful));
    uint32_t per_wave = (n_quads + gx * waves_per_wg - 1) / (gx * waves_per_wg);
    per_wave = std::max<uint32_t>(8, round_up(per_wave, 8));
    if (!rules) return per_wave;
    // whole steps — where they are cheap. A run of 3.5 steps ends every wave on a step with half its lanes idle; rounding the
    // run up to 4 (875 k vertices: 489 -> 428 workgroups) was 4.5 % faster. But the grid must stay close to two workgroups on EVERY CU:
    // at 282 k vertices 2.5 -> 3 steps (440 -> 367 workgroups: 111 CUs with two workgroups, 145 with one) was 9 % SLOWER, in six fresh
    // processes out of six (profiles/r6_fresh_plans.txt). So the run is rounded up only when that costs at most one eighth of the
    // workgroups (k >= 0.875 ceil(k)); shorter runs keep the 8-quad grain.
    if (per_wave > qpw_step) {
        const uint32_t whole = round_up(per_wave, qpw_step);
        if ((uint64_t)per_wave * 8 >= (uint64_t)whole * 7) per_wave = whole;
    }
    // ... and a run of between one and two steps becomes ONE step per wave on as many workgroups as that takes — the hardware deals the
    // second round out as slots free up. S = 2 (what is left of 367-395 k vertices once S = 4 took its share, auto_split): 5 % faster than
    // 1.5 steps per wave in 8 fresh processes of 8 (375 k: 51.2 -> 48.4 us, 390 k: 53.8 -> 51.0; profiles/r6_fresh_plans_mid.txt). S = 4
    // (132-190 k vertices): 1-4 % faster in 16 sizes of 16 and in 8 fresh processes of 8 (profiles/r6_onestep_sweep.txt, r6_fresh_plans_mid.txt).
    if (one_mesh && (S == 2 || S == 4) && per_wave > qpw_step && per_wave < 2 * qpw_step) per_wave = qpw_step;
    return per_wave;
}

void inst_runs(const rz_ctx *c, int G, int blk, bool for_subsets, uint32_t *per, uint32_t *runs)
{
    // 256 threads = two workgroups per CU, 512 / 1024 = one whose 8 / 16 waves share one staged palette group — ONE round of
    // workgroups. With bone subsets (15-30 KB of LDS) two 512-thread workgroups fit a CU and more, shorter runs name fewer
    // bones (36 -> 17 per run at 1024 workgroups), but whether that pays depends on the box: tools/archive/c4_subsets.py measured
    // 256 / 512 / 768 / 1024 workgroups at 33.4 / 34.3 / 32.9 / 32.6 us on one MI355X and 33.1-33.4 / 42 / 42 / 42 us on two
    // others (profiles/archive/r3_c4_subsets.txt). One workgroup per CU is the shape that is good everywhere, so it is the default;
    // rz_autotune tries the others on the box it runs on.
    (void)for_subsets;
    const uint32_t wg_per_cu = blk == 256 ? 2u : 1u;
    const uint32_t groups = (c->I + G - 1) / G;
    const uint32_t total = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : wg_per_cu * (uint32_t)c->n_cu;
    const uint32_t gxi = std::max<uint32_t>(1, total / groups);
    *per = round_up((c->V + gxi - 1) / gxi, 64);
    *runs = (c->V + *per - 1) / *per;
    if (*runs > 0xffffu) {                  // the crowd kernel takes its launch shape as two 16-bit fields of one preloaded argument
        *per = round_up((c->V + 0xfffeu) / 0xffffu, 64);
        *runs = (c->V + *per - 1) / *per;
    }
}

}  // namespace

RzDeformParams deform_params(const rz_ctx *c, const Plan &pl)
{
    RzDeformParams p;
    memset(&p, 0, sizeof p);
    p.geom = c->geom; p.joints01 = c->j01; p.joints23 = c->j23; p.weights = c->wq;
    p.palette = c->palette; p.world = src_world(c); p.inv_bind = c->inv_bind; p.dense = c->dense;
    p.act_idx = c->act_idx; p.act_w = c->act_w; p.act_count = c->act_count; p.morph_w = src_morph_w(c);
    if (pl.v.fast && c->zc_cur >= 0) {            // the one-launch kernel's workgroup 0 makes the pose resident
        if (!c->world_resident && !c->zc_local) p.world_copy = c->world;
        if (!c->mw_resident && pl.v.mode == 2) p.morph_w_copy = c->morph_w;
        // Pose prefetch, first frame of a zero-copy world pose only. This frame: was the pose staged by the previous frame's
        // helper? Next frame: a helper workgroup looks at the slot the next upload will use (deform_kernels.h: pf_*).
        if (pl.pf && p.world_copy && c->zc_tag && c->zc_seq_cur) {
            p.st_tag = c->zc_tag + c->pose_slot; p.st_expect = c->zc_seq_cur;
            p.st_world = c->world; p.st_morph_w = (pl.v.mode == 2 && c->M > 0) ? c->morph_w : nullptr;
            const int nxt = (c->zc_cur + 1) % rz_ctx::kZcSlots;
            p.pf_src = static_cast<const float *>(c->zc_dev[nxt]);
            p.pf_src_seq = reinterpret_cast<const uint64_t *>(static_cast<const char *>(c->zc_dev[nxt]) + c->zc_hdr_off);
            p.pf_dst = c->pose_blk[c->pose_slot ^ 1];
            p.pf_tag = c->zc_tag + (c->pose_slot ^ 1);
            p.pf_expect = zc_seq(c, c->zc_uploads + 1, c->zc_kind);
            p.pf_bytes = (uint32_t)((c->zc_total + 15) / 16 * 16);
        }
    }
    p.sp_ptr = c->sp_ptr; p.sp_entries = c->sp_entries;
    p.out_pos = c->ext_pos ? c->ext_pos : c->out_pos; p.out_nrm = c->ext_nrm ? c->ext_nrm : c->out_nrm;
    p.edge = c->edge; p.out_hull = c->out_hull; p.aabb = c->aabb_on ? c->aabb : nullptr; p.aabb_slot = c->aabb_slot;
    p.n_verts = c->V;
    p.Vp = c->Vp; p.n_quads = pl.n_quads; p.quads_per_wave = pl.quads_per_wave; p.dma = pl.dma ? 1 : 0;
    p.B = (int)c->B; p.M = (int)c->M; p.Mpad = (int)c->Mpad;
    p.inst_order = c->t_instorder;
#ifdef RZ_ABLATE
    p.dbg = c->t_dbg == 100 ? 0 : c->t_dbg;
    p.tl = c->t_dbg == 100 ? c->tl : nullptr;      // (allocated by rz_debug_timeline_arm)
#endif
    p.out_cap = pl.out_cap;
    p.sp_cap = pl.sp_cap;
    if (pl.subsets) { p.rj01 = c->rj01; p.rj23 = c->rj23; p.sub_list = c->sub_list; p.sub_count = c->sub_count; p.sub_stride = (int)c->sub_B; p.sub_max = (int)c->sub_max; }
    if (pl.subfk) p.fk = fk_params(c);          // (the crowd kernel's front reads the pose, the motion and the bone records through it)
    if (pl.fuse_fk) {
        p.fk = fk_params(c); p.fk_on = 1;
        // the specialised variants (fk_solve<true, KIND>): no physics overrides, two bones per thread, one morph per thread;
        // "fuse_fk_plain" = 0 keeps the generic kernel (A/B, tests)
        const bool plain = c->t_fkplain != 0 && !p.fk.ovr_off && c->B <= 512 && c->M <= 256 && (!c->pose_sampled || c->an_M <= 256);
        p.fk_kind = plain ? (c->pose_sampled ? 2 : 1) : 0;
        if (c->zc_cur >= 0 && c->zc_local && !c->pose_sampled && (!c->local_resident || !c->mw_resident)) {
            // zero-copy local pose, first frame: workgroup 0 makes it resident (every later frame of this pose reads the device
            // block instead of pulling the pose over the host link in every workgroup), and — like the one-launch frame of a
            // world pose — the frame looks for a copy the previous frame's helper staged and carries a helper for the next upload
            float *blk_t = c->pose_local_t ? reinterpret_cast<float *>(c->local_q + c->B) : nullptr;
            p.fk.copy_q = c->local_q; p.fk.copy_t = blk_t;
            if (c->M > 0) p.morph_w_copy = c->morph_w;
            if (pl.pf && c->zc_tag && c->zc_seq_cur) {
                p.st_tag = c->zc_tag + c->pose_slot; p.st_expect = c->zc_seq_cur; p.fk.st_expect = c->zc_seq_cur;
                p.fk.st_local_q = c->local_q; p.fk.st_local_t = blk_t; p.st_morph_w = c->morph_w;
                const int nxt = (c->zc_cur + 1) % rz_ctx::kZcSlots;
                p.pf_src = static_cast<const float *>(c->zc_dev[nxt]);
                p.pf_src_seq = reinterpret_cast<const uint64_t *>(static_cast<const char *>(c->zc_dev[nxt]) + c->zc_hdr_off);
                p.pf_dst = c->pose_blk[c->pose_slot ^ 1] + (c->morph_w - c->pose_blk[c->pose_slot]);     // the local range of the other block
                p.pf_tag = c->zc_tag + (c->pose_slot ^ 1);
                p.pf_expect = zc_seq(c, c->zc_uploads + 1, c->zc_kind);
                p.pf_bytes = (uint32_t)((c->zc_total + 15) / 16 * 16);
            }
        }
    }
    return p;
}

// The shape the caller ASKS for (inst_loop / inst_block / grid_cap or the defaults), before LDS limits the group size; when
// bone subsets are allowed it is the shape of the SUBSET form (the whole-palette fallback sizes its own grid).
bool inst_shape(const rz_ctx *c, InstShape *s)
{
    const bool epilogues = c->edge != nullptr || c->aabb_on;   // only the generic kernel carries the fused consumers
    if (!(c->morph_mode == 0 && c->I > 1 && c->t_instloop != 0 && c->t_instloop != 9 && !epilogues) || c->B == 0 || c->V == 0) return false;
    s->want_in_kernel = c->t_fast != 0 && !c->pose_local;
    // workgroup size. Whole palettes: 512 threads for the one-launch frame (one workgroup of 8 waves per CU shares the 102 KB
    // group), 256 behind rz_prep_kernel / rz_fk_kernel (two workgroups of 80 KB each: measured best in round 2). Bone subsets
    // (30 KB): 512 threads in both forms — with finished rows staged the 512-thread kernel runs C4 in 32.1 us against 34.5 us
    // for 256 threads (tools/archive/c4_subsets.py, fast = 0 rows of profiles/archive/r3_c4_subsets.txt).
    const bool forced = c->t_instblock == 256 || c->t_instblock == 512 || c->t_instblock == 1024;
    s->blk_full = forced ? c->t_instblock : (s->want_in_kernel ? 512 : 256);
    s->blk = forced ? c->t_instblock : (c->t_subsets != 0 ? 512 : s->blk_full);
    s->G = (int)std::min<uint32_t>(c->t_instloop > 0 ? (uint32_t)c->t_instloop : 8u, c->I);
    if (s->G < 2) return false;
    inst_runs(c, s->G, s->blk, c->t_subsets != 0, &s->per, &s->runs);
    return true;
}

bool subfk_wanted(const rz_ctx *c);

Plan make_plan(const rz_ctx *c)
{
    Plan pl;
    memset(&pl, 0, sizeof pl);          // padding too: frame_signature() hashes the struct
    RzVariant &v = pl.v;
    v.mode = c->morph_mode;
    // Without dense targets S only sets the size of a wave step: S = 4 makes it 64 vertices instead of 256, so a small
    // mesh (one 30 k-vertex character is 118 wave steps at S = 1) reaches four times as many CUs, and a region where
    // every vertex carries dozens of sparse entries (a face) spreads over four times as many waves. 1 or 4 there.
    v.S = (v.mode == 1) ? (c->t_split > 0 ? c->t_split : auto_split(c))
                        : (c->t_split > 0 ? (c->t_split >= 4 ? 4 : 1)
                                          : ((uint64_t)c->V * c->I <= (v.mode == 2 ? (256u << 10) : (64u << 10)) ? 4 : 1));   // measured: 30 k verts 5.5 -> 4.5 us, 126 k 6.0 -> 6.6
    v.U = c->t_unroll > 0 ? c->t_unroll : 8;    // 24 loads in flight per lane: best or tied at every size measured
    v.nt = c->t_nt != 0;
    // streaming stores pay once the frame's output no longer fits the L2s (measured: 1 M verts yes, 126 k no)
    // and the morph stream is flowing too; a morph-free instanced frame (184 MB of output, MALL-absorbed) is faster
    // with plain stores: 6.9 vs 5.7 TB/s in tools/membench
    v.nts = c->t_nts < 0 ? (v.mode == 1 && (uint64_t)c->V * c->I * 24 >= (16u << 20)) : c->t_nts != 0;
    v.geo = c->t_geo != 0;
    // one-launch frame: single instance, and (dense) the active list fits the kernel arguments
    // Device-animated single character: the hierarchy solve (and the motion sampling) runs as the prologue of every
    // workgroup of the deform kernel — one launch per frame, no rz_fk_kernel / rz_prep_kernel in front of it. Measured
    // (tools/archive/fk_fuse_bench.py, 30 k vertices / 200 bones): sampled poses 11.5-17.3 -> 9.2-13.5 us per frame in every morph
    // mode (and 27.7 -> 24.5 us on a 1/8 shard of C5); local poses 12.7-14.4 -> 9.8-12.8 us without dense morphs, no gain
    // with them (there the three-kernel frame keeps its kernel-argument morph list and streams from its first instruction).
    // Automatic mode follows that; "fuse_fk" = 0 / 1 forces it.
    pl.fuse_fk = c->I == 1 && c->pose_local && c->has_topology && (size_t)c->B * 48 + rz_fk_scratch_bytes((int)c->B) + (size_t)c->M * 12 + 4096 <= 160 * 1024 &&
                 (c->t_fusefk == 1 || (c->t_fusefk < 0 && (c->pose_sampled || c->morph_mode != 1)));
    const bool can_fast = c->I == 1 && !pl.fuse_fk && (v.mode != 1 || c->ml.count >= 0);
    v.fast = can_fast && c->t_fast != 0;
    pl.dma = false;
    pl.inst_group = 0;
    pl.verts_per_wg = 0;
    pl.poses_per_wg = 0;
    pl.out_cap = 0;
    pl.n_quads = (c->V + 3) / 4;
    pl.quads_per_wave = 8;
    pl.grid_x = 1;
    pl.prep = !v.fast && !pl.fuse_fk;
    // persistent, balanced grid: `cap` workgroups in total, every wave owns an equal contiguous run of quads
    const uint32_t waves_per_wg = 4;
    // measured (profiles/archive/r1_*sweep*): 2 workgroups per CU for one big mesh, 8 per instance when instanced
    uint32_t cap = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : std::max(2u * (uint32_t)c->n_cu, 8u * c->I);
    // dense frames at S = 8 (under ~95 k vertices): ONE 32-vertex step per wave even where that takes a second round of workgroups —
    // two steps per wave on half as many waves were 11-14 % slower at 63 k and 94 k vertices (profiles/r6_plan_sweep.txt)
    if (v.mode == 1 && v.S == 8 && c->t_grid_cap <= 0 && c->I == 1) cap = std::max(cap, 4u * (uint32_t)c->n_cu);
    // Pose prefetch: the first frame of a zero-copy world pose carries one helper workgroup that stages the NEXT pose (if the
    // host has written it already) — it takes one of the grid's slots, the workers share the mesh among cap - 1.
    pl.pf = c->I == 1 && c->t_prefetch != 0 && c->zc_cur >= 0 && c->zc_seq_cur != 0 && c->zc_tag &&
            ((v.fast && !c->zc_local && !c->world_resident) ||
             (pl.fuse_fk && c->zc_local && !c->pose_sampled && (!c->local_resident || !c->mw_resident)));
    if (pl.pf && cap > 1) cap -= 1;
    const bool dense_rules = v.mode == 1 && c->t_grid_cap <= 0, one_mesh = c->I == 1;
    const uint32_t cap_i = std::max<uint32_t>(1, cap / std::max<uint32_t>(1, c->I));
    uint32_t per_wave = run_per_wave(pl.n_quads, cap_i, v.S, dense_rules, one_mesh);
    // S = 2 pays where it leaves a wave WHOLE steps (1 M: 4, 750 k: 3, 500 k: 2 — there it is 1-3 % ahead of S = 4). Where it does not
    // (2.25 / 2.5 / 3.25 steps of 128 vertices) the same grid at S = 4 — 64-vertex steps, the ragged last one half the size — was faster in
    // every one of 36 fresh processes: 530-580 k vertices 7-8 %, 596-613 k 3 %, 797-845 k 4 % (profiles/r6_fresh_plans_hi.txt). Its own
    // runs keep the 8-quad grain: rounded up to whole 64-vertex steps on fewer workgroups they were 0.4-4 % slower in 28 processes of 28.
    if (dense_rules && one_mesh && c->t_split <= 0 && v.S == 2 && c->M >= 4 && per_wave > 32 && per_wave % 32 != 0) {
        v.S = 4;
        per_wave = run_per_wave(pl.n_quads, cap_i, v.S, false, true);
    }
    pl.quads_per_wave = per_wave;
    pl.grid_x = std::max<uint32_t>(1, (pl.n_quads + per_wave * waves_per_wg - 1) / (per_wave * waves_per_wg));
    // LDS write batching. Measured (tools/archive/ablate_c5.py): parking a wave's WHOLE run and writing it once at the end
    // takes C5 from 127.6 to 124.5 us (the stores leave the read stream alone until the kernel's tail); flushing every
    // step gains nothing. So automatic mode turns it on exactly when the run fits the buffer (<= 640 vertices per wave).
    {
        const uint32_t step = 256u / (uint32_t)v.S;
        const uint32_t run = round_up(pl.quads_per_wave * 4, 64);
        if (c->t_outcap > 0) pl.out_cap = std::max(std::min<uint32_t>(round_up((uint32_t)c->t_outcap, 64), 640), step);
        else if (c->t_outcap < 0 && run <= 640) pl.out_cap = std::max(run, step);
        // a skeleton near the LDS limit (3 242 bones = 152 KB of palette) leaves no room for the write-batching buffer: do without
        if (pl.out_cap) {
            RzDeformParams q;
            memset(&q, 0, sizeof q);
            q.B = (int)c->B; q.M = (int)c->M; q.Mpad = (int)c->Mpad; q.out_cap = pl.out_cap; q.fk_on = pl.fuse_fk ? 1 : 0;
            if (rz_deform_lds_bytes(q, v) > 160 * 1024) pl.out_cap = 0;
        }
    }
    // Sparse targets: every wave stages its step's piece of the CSR in LDS (deform_kernels.hip, MODE 2). The buffer takes what the
    // launch leaves of the LDS — a frame of at most one workgroup per CU (a single character: the demo model is 113 workgroups)
    // has the CU's 160 KB to itself, larger frames plan for two workgroups per CU — up to a whole step of M-entry rows or 2 048
    // entries (32 KB per wave); longer ranges go through it in pieces.
    if (v.mode == 2) {
        RzDeformParams q;
        memset(&q, 0, sizeof q);
        q.B = (int)c->B; q.M = (int)c->M; q.Mpad = (int)c->Mpad; q.out_cap = pl.out_cap; q.fk_on = pl.fuse_fk ? 1 : 0;
        const size_t base = rz_deform_lds_bytes(q, v);
        const size_t budget = ((uint64_t)pl.grid_x * c->I <= (uint64_t)c->n_cu ? 160u : 80u) * 1024u;
        // (a skeleton near the LDS limit leaves less than the planned share: take what the CU's 160 KB still hold, down to a quarter
        // burst of 16 entries per wave — the piece loop works with any multiple of 16; nothing at all left = "skeleton too large" below)
        const size_t hard = 160u * 1024u > base ? (160u * 1024u - base) / (waves_per_wg * 16) : 0;
        const size_t avail = budget > base + 4096 ? (budget - base - 1024) / (waves_per_wg * 16) : hard;
        const uint32_t want = std::min<uint32_t>(2048u, (256u / (uint32_t)v.S) * std::max<uint32_t>(c->M, 1u));
        pl.sp_cap = avail >= 64 ? std::max<uint32_t>(64u, std::min<uint32_t>(round_up(want, 256), (uint32_t)(avail / 64 * 64)))         // whole 1 KiB bursts (the kernel issues them in groups of four)
                                : std::max<uint32_t>(16u, (uint32_t)(avail / 16 * 16));
    }
    // instanced, morph-free frames: G poses per workgroup share one decode of each vertex, their palettes live in LDS.
    // Where the palettes come from (measured on C4, tools/archive/ablate_c4.py, frame = everything a frame launches):
    //   prep kernel + 16-byte LDS-DMA of finished palettes (default, and always behind the on-device FK, which writes
    //       the palettes itself): 38.7-39.5 us (kernel 34 + 2.7 us prep + launch boundary);
    //   in-kernel (fast = 1): the skin kernel stages the group's world matrices (64-byte slots, same LDS-DMA) and
    //       multiplies by the inverse bind matrices in place — one launch per frame, 39.6-40 us: the staging costs
    //       what the extra launch did, so it is opt-in.
    const bool epilogues = c->edge != nullptr || c->aabb_on;   // only the generic kernel carries the fused consumers
    InstShape is;
    if (inst_shape(c, &is)) {
        // Where the palettes come from. World-matrix poses (rz_set_pose): the skin kernel forms them itself — ONE launch per
        // frame (fast = 0 forces rz_prep_kernel in front). Device-solved poses: rz_fk_kernel has written the palettes already,
        // the skin kernel copies them in (48-byte rows, LDS-DMA).
        // What is staged: by default only the bones the workgroup's vertex run names (bone-subset form) — on C4
        // ~34 of 200 bones, 30 KB of LDS instead of 102 KB and a front of 1.3 us instead of 2.9 us per workgroup. It needs the
        // run lists of exactly this launch shape (ensure_run_subsets, called by every entry point that launches frames) and
        // is a gain only when the largest list is shorter than the skeleton; otherwise the whole palette is staged: 64-byte
        // slots re-packed in place for the one-launch frame, which wants B <= block threads and, at 8 poses x 200 bones =
        // 102 KB, one 512-thread workgroup per CU (measured 35.5 us on C4 against 39.0 us for prep kernel + launch boundary +
        // 256-thread skin kernel, and 39.5 us for two 256-thread workgroups of 6 poses).
        const int blk = is.blk;
        const uint32_t lds_budget = (blk == 256 ? 80u : 156u) * 1024u;
        bool sub = c->t_subsets != 0 && c->sub_valid && c->sub_per == is.per && c->sub_runs == is.runs && c->sub_B == c->B &&
                   c->sub_max < c->B && c->sub_max <= (uint32_t)blk;
        if (sub) {
            const size_t lds = rz_skin_instances_lds_bytes(is.G, c->sub_max, !is.want_in_kernel, true);
            if (lds > lds_budget) sub = false;
            else {
                pl.subsets = true; pl.sub_bones = c->sub_max; pl.inst_lds = lds;
                pl.inst_group = is.G; pl.inst_block = blk; pl.verts_per_wg = is.per; pl.grid_x = is.runs;
                pl.prep = !is.want_in_kernel; pl.dma = !is.want_in_kernel;
                // device-animated poses: the hierarchy solved in the front of the skin kernel — one launch per frame — when the closure
                // records of these very lists are there, a workgroup's (pose, closure bone) items fit two per thread, and the two matrix
                // buffers + the palettes fit the LDS budget; else rz_fk_kernel in front, as before
                if (subfk_wanted(c) && c->subfk_valid && c->subfk_sub_gen == c->sub_gen && c->subfk_fk_gen == c->fk_gen &&
                    (size_t)is.G * c->subfk_stride <= 2 * (size_t)blk) {
                    const size_t lf = rz_skin_instances_fk_lds_bytes(is.G, c->subfk_stride, c->sub_max);
                    if (lf <= lds_budget) { pl.subfk = true; pl.prep = false; pl.dma = false; pl.inst_lds = lf; }
                }
            }
        }
        if (!sub) {
            const int blk_f = is.blk_full;
            const uint32_t budget_f = (blk_f == 256 ? 80u : 156u) * 1024u;
            const bool in_kernel = is.want_in_kernel && c->B <= (uint32_t)blk_f;   // the in-place product gives every bone its own thread
            const uint32_t g_lds = budget_f / (c->B * (in_kernel ? 64u : 48u));
            int G = (int)std::min<uint32_t>((uint32_t)is.G, g_lds);
            if (G >= 2) {
                uint32_t per = 0, runs = 0;
                inst_runs(c, G, blk_f, false, &per, &runs);
                pl.inst_group = G; pl.inst_block = blk_f; pl.verts_per_wg = per; pl.grid_x = runs;
                pl.prep = !in_kernel; pl.dma = !in_kernel;
                pl.inst_lds = rz_skin_instances_lds_bytes(G, c->B, !in_kernel, false);
            }
        }
    }
    // register-resident instanced form: 2048-vertex runs, pose ranges sized for ~2 WGs per CU
    // (measured slower than the LDS pose-group form on C4 — 37 vs 34 us — so it is opt-in: inst_loop = 9)
    if (v.mode == 0 && c->I > 1 && c->t_instloop == 9 && c->B * 3 <= 65535u && !epilogues) {
        const uint32_t runs = (c->V + 2047) / 2048;
        uint32_t total = c->t_grid_cap > 0 ? (uint32_t)c->t_grid_cap : 2u * (uint32_t)c->n_cu;
        uint32_t ranges = std::max<uint32_t>(1, std::min<uint32_t>(c->I, total / std::max<uint32_t>(1, runs)));
        pl.poses_per_wg = (int)((c->I + ranges - 1) / ranges);
        pl.inst_group = 0;
        pl.grid_x = runs;
        pl.prep = true;
        pl.dma = true;
    }
    return pl;
}

#ifdef RZ_ALL_VARIANTS
}  // namespace rzi
// tools-only build (make variants), test hook — not part of the C ABI: the launch shape make_plan picks for the one-launch dense frame of
// one mesh of these sizes on a device of n_cu compute units (no GPU involved: the plan is a pure function of the sizes), so that the CPU
// suite can hold the heuristics of DESIGN.md 4.4 to their table.
extern "C" __attribute__((visibility("default"))) int rz_debug_plan_dense(uint32_t V, uint32_t B, uint32_t M, int n_cu, int *split, int *grid, int *quads_per_wave, int *out_cap)
{
    if (V == 0 || M == 0 || n_cu <= 0 || !split || !grid || !quads_per_wave) return RZ_ERR_INVALID;
    rz_ctx c;
    c.n_cu = n_cu; c.V = V; c.Vp = rzi::round_up(V, rzi::kVertPad); c.B = B; c.M = M; c.I = 1; c.morph_mode = 1;
    c.ml.count = M <= (uint32_t)kKargMorphs ? (int)M : -1;
    const rzi::Plan pl = rzi::make_plan(&c);
    *split = pl.v.S; *grid = (int)pl.grid_x; *quads_per_wave = (int)pl.quads_per_wave;
    if (out_cap) *out_cap = (int)pl.out_cap;
    return RZ_OK;
}
namespace rzi {
#endif

// Bring the run lists of the bone-subset crowd frame in line with the launch shape the next frame asks for. One small kernel
// and one readback of `runs` counters, only when the shape, the mesh or the skeleton changed — never per frame.
int ensure_run_subsets(rz_ctx *c)
{
    InstShape is;
    if (c->t_subsets == 0 || !inst_shape(c, &is)) return RZ_OK;
    if (c->sub_valid && c->sub_per == is.per && c->sub_runs == is.runs && c->sub_B == c->B) return RZ_OK;
    c->sub_valid = false;
    HIP_TRY(hipStreamSynchronize(c->stream));       // a frame in flight may still read the old lists
    drop_graph(c);
    if (!c->rj01) HIP_TRY(hipMalloc(&c->rj01, (size_t)c->Vp * 4));
    if (!c->rj23) HIP_TRY(hipMalloc(&c->rj23, (size_t)c->Vp * 4));
    const size_t need_list = (size_t)is.runs * c->B;
    if (need_list > c->sub_list_alloc) {
        dfree(c->sub_list);
        HIP_TRY(hipMalloc(&c->sub_list, need_list * sizeof(uint16_t)));
        c->sub_list_alloc = need_list;
    }
    if (is.runs > c->sub_count_alloc) {
        dfree(c->sub_count);
        HIP_TRY(hipMalloc(&c->sub_count, (size_t)is.runs * sizeof(uint32_t)));
        c->sub_count_alloc = is.runs;
    }
    const uint32_t v_lim = (c->V + 3) / 4 * 4;      // what the skin kernel walks: whole quads
    HIP_TRY(hipMemsetAsync(c->rj01, 0, (size_t)c->Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->rj23, 0, (size_t)c->Vp * 4, c->stream));
    HIP_TRY(rz_launch_run_subsets(c->j01, c->j23, v_lim, is.per, is.runs, c->B, c->sub_list, c->sub_count, c->rj01, c->rj23, c->stream));
    std::vector<uint32_t> counts(is.runs);
    HIP_TRY(hipMemcpyAsync(counts.data(), c->sub_count, (size_t)is.runs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t mx = 0;
    for (uint32_t n : counts) mx = std::max(mx, n);
    c->sub_per = is.per; c->sub_runs = is.runs; c->sub_B = c->B; c->sub_max = mx;
    c->sub_valid = true;
    c->sub_gen++;
    return RZ_OK;
}

// Plan of the next frame: the run lists first (the plan only takes the subset form when they match its shape).
// Can the next crowd frame solve its hierarchy in the skin kernel's front? A device-animated crowd in the bone-subset form, nothing
// acting on the solved pose between the solve and the palette (bone morphs fold weights in, physics overrides replace world matrices:
// both stay with rz_fk_kernel), "fuse_fk" not switched off.
bool subfk_wanted(const rz_ctx *c)
{
    return c->I > 1 && c->pose_local && c->has_topology && c->t_fusefk != 0 && c->t_subsets != 0 && c->sub_valid && c->sub_max < c->B &&
           !(c->bm_count && c->M) && c->ovr_count == 0 && c->fk_host.size() == (size_t)c->B * 4;
}

// The closure records of the current run lists (see ctx.h). Host work + one readback of the lists, only when the lists or the
// hierarchy's static data changed — never per frame.
int ensure_subfk(rz_ctx *c)
{
    if (!subfk_wanted(c)) return RZ_OK;
    if (c->subfk_sub_gen == c->sub_gen && c->subfk_fk_gen == c->fk_gen) return RZ_OK;      // built, or found unfit, for exactly these lists and this hierarchy
    c->subfk_valid = false;
    c->subfk_sub_gen = c->sub_gen; c->subfk_fk_gen = c->fk_gen;         // (also remembers a refusal: no retry per frame)
    const uint32_t runs = c->sub_runs, B = c->B;
    std::vector<uint16_t> lists((size_t)runs * B);
    std::vector<uint32_t> counts(runs);
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    HIP_TRY(hipMemcpy(lists.data(), c->sub_list, lists.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counts.data(), c->sub_count, (size_t)runs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const bool motion = c->has_animation && c->an_host_range.size() == B;
    auto parent = [&](uint32_t b) { return (int32_t)c->fk_host[4 * (size_t)b].x; };
    std::vector<std::vector<uint32_t>> closures(runs);
    std::vector<uint32_t> depth(B, 0);
    uint32_t stride = 0, max_depth = 0;
    std::vector<uint8_t> in(B);
    for (uint32_t r = 0; r < runs; ++r) {
        std::fill(in.begin(), in.end(), 0);
        for (uint32_t k = 0; k < counts[r]; ++k)
            for (int32_t b = lists[(size_t)r * B + k]; b >= 0 && !in[b]; b = parent((uint32_t)b)) in[b] = 1;
        for (uint32_t b = 0; b < B; ++b)
            if (in[b]) closures[r].push_back(b);
        stride = std::max<uint32_t>(stride, (uint32_t)closures[r].size());
        for (uint32_t b : closures[r]) {
            uint32_t d = 0;
            for (int32_t a = parent(b); a >= 0; a = parent((uint32_t)a)) ++d;
            depth[b] = d; max_depth = std::max(max_depth, d);
        }
    }
    uint32_t rounds = 0;
    for (uint32_t span = 1; span < max_depth + 1; span *= 4) ++rounds;
    if (rounds > 3 || stride > 0xfffu || stride == 0) return RZ_OK;      // deeper than 64 levels / wider than the record stride field: rz_fk_kernel keeps these
    std::vector<uint4> rec((size_t)runs * stride * 5, make_uint4(0u, 0u, 0u, 0u));
    std::vector<uint32_t> ccount(runs);
    std::vector<uint32_t> slot(B), pslot(B);
    for (uint32_t r = 0; r < runs; ++r) {
        const std::vector<uint32_t> &cl = closures[r];
        ccount[r] = (uint32_t)cl.size();
        std::fill(pslot.begin(), pslot.end(), 0xffffu);
        for (uint32_t k = 0; k < counts[r]; ++k) pslot[lists[(size_t)r * B + k]] = k;       // the run lists are ascending: k is the slot the joints were rewritten to
        for (uint32_t k = 0; k < cl.size(); ++k) slot[cl[k]] = k;
        auto anc_slot = [&](uint32_t b, uint32_t d) -> uint32_t {
            int32_t cur = (int32_t)b;
            for (uint32_t k = 0; k < d && cur >= 0; ++k) cur = parent((uint32_t)cur);
            return cur < 0 ? 0xffffu : slot[(uint32_t)cur];
        };
        for (uint32_t k = 0; k < cl.size(); ++k) {
            const uint32_t b = cl[k];
            uint4 *w = &rec[((size_t)r * stride + k) * 5];
            const uint4 t = c->fk_host[4 * (size_t)b], bind = c->fk_host[4 * (size_t)b + 1];
            w[0] = make_uint4(b | (pslot[b] << 16), t.y, t.z, t.w);
            w[1] = bind;
            uint32_t a[3][2] = { { 0xffffffffu, 0xffffu }, { 0xffffffffu, 0xffffu }, { 0xffffffffu, 0xffffu } };
            uint32_t span = 1;
            for (uint32_t q = 0; q < rounds; ++q, span *= 4) {
                a[q][0] = anc_slot(b, span) | (anc_slot(b, 2 * span) << 16);
                a[q][1] = anc_slot(b, 3 * span);
            }
            w[2] = make_uint4(a[0][0], a[0][1], a[1][0], a[1][1]);
            w[3] = make_uint4(a[2][0], a[2][1], 0u, 0u);
            w[4] = motion ? c->an_host_range[b] : make_uint4(0u, 0u, 0u, 0u);
        }
        // Padding up to the longest closure: the kernel does not read a run's closure length (one dependent load less in its front), it
        // works through `stride` records per pose. A padding record poses bone 0 as a root with no palette slot, no append parent, no
        // ancestors and no track: its matrix lands in its own LDS slot and nobody reads it.
        for (uint32_t k = (uint32_t)cl.size(); k < stride; ++k) {
            uint4 *w = &rec[((size_t)r * stride + k) * 5];
            w[0] = make_uint4(0u | (0xffffu << 16), 0xffffffffu, 0u, 0u);
            w[1] = make_uint4(0u, 0u, 0u, 0u);
            w[2] = make_uint4(0xffffffffu, 0xffffu, 0xffffffffu, 0xffffu);
            w[3] = make_uint4(0xffffffffu, 0xffffu, 0u, 0u);
            w[4] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (rec.size() > c->subfk_rec_alloc) {
        dfree(c->subfk_rec);
        HIP_TRY(hipMalloc(&c->subfk_rec, rec.size() * sizeof(uint4)));
        c->subfk_rec_alloc = rec.size();
    }
    if (runs > c->subfk_count_alloc) {
        dfree(c->subfk_count);
        HIP_TRY(hipMalloc(&c->subfk_count, (size_t)runs * sizeof(uint32_t)));
        c->subfk_count_alloc = runs;
    }
    HIP_TRY(hipMemcpy(c->subfk_rec, rec.data(), rec.size() * sizeof(uint4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->subfk_count, ccount.data(), (size_t)runs * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->subfk_stride = stride; c->subfk_rounds = rounds;
    c->subfk_valid = true;
    return RZ_OK;
}

RzSubFk subfk_params(const rz_ctx *c)
{
    RzSubFk f;
    f.count = c->subfk_count; f.rec = c->subfk_rec; f.stride = c->subfk_stride; f.rounds = c->subfk_rounds;
    return f;
}

int frame_plan(rz_ctx *c, Plan *pl)
{
    if (int r = ensure_run_subsets(c)) return r;
    if (int r = ensure_subfk(c)) return r;
    *pl = make_plan(c);
    return RZ_OK;
}

RzPrepParams prep_params(const rz_ctx *c)
{
    RzPrepParams p;
    memset(&p, 0, sizeof p);
    p.world = c->world; p.inv_bind = c->inv_bind; p.palette = c->palette; p.morph_w = c->morph_w;
    p.act_idx = c->act_idx; p.act_w = c->act_w; p.act_count = c->act_count;
    p.B = (int)c->B; p.M = (int)c->M; p.Mpad = (int)c->Mpad;
    return p;
}

RzFkParams fk_params(const rz_ctx *c)
{
    RzFkParams p;
    memset(&p, 0, sizeof p);            // padding too: frame_signature() hashes the struct
    p.local_q = src_local_q(c);
    p.local_t = c->pose_local_t ? reinterpret_cast<const float *>(p.local_q + (size_t)c->I * c->B) : nullptr;
    p.bone_rec = c->fk_rec; p.anc_more = c->fk_anc_more; p.inv_bind = c->inv_bind;
    p.world = c->world; p.palette = c->palette; p.B = (int)c->B; p.n_rounds = c->fk_rounds;
    if (c->ovr_count) { p.ovr_off = c->ovr_off; p.ovr_bone = c->ovr_bone; p.ovr_world = c->ovr_world; }
    if (c->bm_count && c->M) {
        p.bm_off = c->bm_off; p.bm_morph = c->bm_morph; p.bm_rot = c->bm_rot; p.bm_tr = c->bm_tr;
        p.bm_w = src_morph_w(c); p.bm_M = (int)c->M;
    }
    if (c->pose_sampled) {
        RzSampleParams &q = p.sample;
        q.frames = c->frames_inline ? nullptr : c->an_frames; q.frames_inline = c->frames_inline ? 1 : 0; q.frame0 = c->frame0;
        q.key_frame = c->an_key_frame;
        q.key_rot = c->an_key_rot; q.key_pos = c->an_key_pos; q.key_interp = c->an_key_interp;
        q.mkey_frame = c->an_mkey_frame; q.mkey_weight = c->an_mkey_weight;
        q.feed_off = c->an_feed_off; q.feed_range = c->an_feed_range; q.feed_ratio = c->an_feed_ratio;
        q.morph_w = c->morph_w; q.M = (int)c->M;
    }
    return p;
}

uint64_t algorithmic_bytes(const rz_ctx *c)
{
    // 36 B read + 24 B written per vertex, 12*M B of dense morph targets per vertex,
    // world + inverse-bind matrices, morph weights. The static mesh is counted once for instances.
    const uint64_t V = c->V, I = c->I, B = c->B, M = c->M;
    uint64_t bytes = V * 36 + I * (V * 24 + B * 64) + B * 64;
    if (c->morph_mode == 1) bytes += I * (V * 12 * M + M * 4);
    if (c->morph_mode == 2) bytes += V * 4 + I * (c->sp_count * 16 + M * 4);
    return bytes;
}

}  // namespace rzi
