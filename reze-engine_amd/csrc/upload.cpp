// upload.cpp — static data of a context: mesh shard, skeleton, topology, morph targets, bone morphs, motion, edge scale (ctx.h).
#include "ctx.h"

using namespace rzi;

namespace rzi {

// The device block behind fk_rec: the bone records with the CURRENT motion's tracks filled in, then the motion's vertex-morph records.
// Called when either side changes (the caller has drained the stream); a context without a topology has no block.
int rebuild_fk_static(rz_ctx *c)
{
    if (!c->has_topology || c->fk_host.size() != (size_t)c->B * 4) return RZ_OK;
    const bool motion = c->has_animation && c->an_host_range.size() == c->B;
    const size_t nm = motion ? c->an_host_mrec.size() / 2 : 0;
    std::vector<uint4> blk(c->fk_host);
    blk.resize((size_t)c->B * 4 + nm * 2);
    for (uint32_t b = 0; b < c->B; ++b) blk[4 * (size_t)b + 2] = motion ? c->an_host_range[b] : make_uint4(0u, 0u, 0u, 0u);
    for (size_t k = 0; k < nm * 2; ++k) blk[(size_t)c->B * 4 + k] = c->an_host_mrec[k];
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    dfree(c->fk_rec);
    c->fk_gen++;                        // (the closure records of crowd frames carry copies of these: plan.cpp ensure_subfk)
    return to_device(&c->fk_rec, blk.data(), blk.size());
}

}  // namespace rzi

namespace {

int upload_skinning(rz_ctx *c, uint32_t V, const uint16_t *joints4, const uint8_t *weights4)
{
    Scratch<uint16_t> dj;
    Scratch<uint8_t> dw;
    HIP_TRY(dj.alloc((size_t)V * 4));
    HIP_TRY(dw.alloc((size_t)V * 4));
    HIP_TRY(hipMemcpy(dj.p, joints4, (size_t)V * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dw.p, weights4, (size_t)V * 4, hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_pack_skinning(dj.p, dw.p, V, c->j01, c->j23, c->wq, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RZ_OK;
}

int alloc_mesh(rz_ctx *c, uint32_t V)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_direct_gather(c);                // shard sizes are about to change: back to private output buffers
    dfree(c->geom); dfree(c->j01); dfree(c->j23); dfree(c->wq); dfree(c->edge);
    dfree(c->rj01); dfree(c->rj23); c->sub_valid = false;      // the run lists name this mesh's joints
    free_morphs(c);                       // morph targets are per-vertex: a new mesh invalidates them
    c->V = V;
    c->Vp = round_up(V, kVertPad);
    const size_t Vp = c->Vp;
    HIP_TRY(hipMalloc(&c->geom, 6 * Vp * sizeof(float)));
    HIP_TRY(hipMalloc(&c->j01, Vp * 4));
    HIP_TRY(hipMalloc(&c->j23, Vp * 4));
    HIP_TRY(hipMalloc(&c->wq, Vp * 4));
    // padding vertices: zero position/normal, joint 0, weights 0 (takes the (1,0,0,0) branch)
    HIP_TRY(hipMemsetAsync(c->geom, 0, 6 * Vp * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->j01, 0, Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->j23, 0, Vp * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->wq, 0, Vp * 4, c->stream));
    return RZ_OK;
}

}  // namespace

extern "C" {

int rz_shard_range(uint32_t v_total, int nranks, int rank, uint32_t *begin, uint32_t *count)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !begin || !count)
        return fail(RZ_ERR_INVALID, "bad shard query (nranks=%d rank=%d)", nranks, rank);
    const uint64_t per = ((uint64_t)v_total + nranks - 1) / nranks;
    const uint64_t chunk = (per + kShardGrain - 1) / kShardGrain * kShardGrain;
    uint64_t b = std::min<uint64_t>(v_total, chunk * (uint64_t)rank);
    uint64_t n = std::min<uint64_t>(chunk, v_total - b);
    *begin = (uint32_t)b;
    *count = (uint32_t)n;
    return RZ_OK;
}

int rz_gather_chunk(uint32_t v_total, int nranks, uint32_t *chunk)
{
    if (!chunk) return fail(RZ_ERR_INVALID, "null chunk");
    uint32_t b0 = 0, n0 = 0;
    if (int r = rz_shard_range(v_total, nranks, 0, &b0, &n0)) return r;
    const uint64_t ch = ((uint64_t)n0 + kShardGrain - 1) / kShardGrain * kShardGrain;
    if (ch > 0xffffffffull) return fail(RZ_ERR_INVALID, "a shard of %u vertices rounds up past 2^32: no gathered buffer for this mesh", n0);
    *chunk = (uint32_t)ch;
    return RZ_OK;
}

int rz_instance_range(uint32_t instances, int nranks, int rank, uint32_t *begin, uint32_t *count)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !begin || !count)
        return fail(RZ_ERR_INVALID, "bad instance-shard query (nranks=%d rank=%d)", nranks, rank);
    const uint64_t per = ((uint64_t)instances + nranks - 1) / nranks;
    const uint64_t b = std::min<uint64_t>(instances, per * (uint64_t)rank);
    *begin = (uint32_t)b;
    *count = (uint32_t)std::min<uint64_t>(per, instances - b);
    return RZ_OK;
}

int rz_upload_mesh(rz_ctx *c, uint32_t V, const float *interleaved8, const uint16_t *joints4, const uint8_t *weights4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_mesh")) return r;
    if (V == 0 || !interleaved8 || !joints4 || !weights4) return fail(RZ_ERR_INVALID, "rz_upload_mesh: empty mesh or null array");
    if (int r = alloc_mesh(c, V)) return r;
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)V * 8));
    float *tmp = scratch.p;
    HIP_TRY(hipMemcpy(tmp, interleaved8, (size_t)V * 8 * sizeof(float), hipMemcpyHostToDevice));
    const size_t Vp = c->Vp;
    HIP_TRY(rz_launch_deinterleave(tmp, 8, 0, V, c->geom, c->geom + Vp, c->geom + 2 * Vp, c->stream));
    HIP_TRY(rz_launch_deinterleave(tmp, 8, 3, V, c->geom + 3 * Vp, c->geom + 4 * Vp, c->geom + 5 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int r = upload_skinning(c, V, joints4, weights4)) return r;
    return ensure_outputs(c);
}

int rz_upload_mesh_soa(rz_ctx *c, uint32_t V, const float *pos3, const float *nrm3, const uint16_t *joints4,
                       const uint8_t *weights4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_mesh_soa")) return r;
    if (V == 0 || !pos3 || !nrm3 || !joints4 || !weights4) return fail(RZ_ERR_INVALID, "rz_upload_mesh_soa: empty mesh or null array");
    if (int r = alloc_mesh(c, V)) return r;
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)V * 3));
    float *tmp = scratch.p;
    const size_t Vp = c->Vp;
    HIP_TRY(hipMemcpy(tmp, pos3, (size_t)V * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_deinterleave(tmp, 3, 0, V, c->geom, c->geom + Vp, c->geom + 2 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(tmp, nrm3, (size_t)V * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(rz_launch_deinterleave(tmp, 3, 0, V, c->geom + 3 * Vp, c->geom + 4 * Vp, c->geom + 5 * Vp, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int r = upload_skinning(c, V, joints4, weights4)) return r;
    return ensure_outputs(c);
}

int rz_upload_skeleton(rz_ctx *c, uint32_t B, const float *inverse_bind16)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_skeleton")) return r;
    if (B == 0 || !inverse_bind16) return fail(RZ_ERR_INVALID, "rz_upload_skeleton: model has no bones");
    if ((size_t)B * 48 + 8192 > 160 * 1024) return fail(RZ_ERR_UNSUPPORTED, "more than %d bones do not fit the LDS palette", (160 * 1024 - 8192) / 48);
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->ovr_count = 0;
    dfree(c->inv_bind);
    HIP_TRY(hipMalloc(&c->inv_bind, (size_t)B * 16 * sizeof(float)));
    HIP_TRY(hipMemcpy(c->inv_bind, inverse_bind16, (size_t)B * 16 * sizeof(float), hipMemcpyHostToDevice));
    c->B = B;
    c->zc_epoch++; c->zc_seq_cur = 0;   // a pose staged for the old skeleton must never match
    c->sub_valid = false;               // joints are clamped to the bone count when the run lists are built
    c->palette_stale = false;
    c->pose_set = false;
    c->has_topology = false;            // belongs to the previous skeleton ...
    dfree(c->fk_rec); dfree(c->fk_anc_more);        // ... and so do its records (sized for the old bone count: nothing may read them for the new one)
    c->fk_host.clear(); c->fk_rounds = 0; c->fk_gen++;
    c->fk_stale = false; c->subfk_valid = false;
    free_bone_morphs(c);                // ... as do bone morphs (their entries name its bones)
    free_animation(c);                  // ... and so does an uploaded motion (its tracks name bones of that skeleton)
    return ensure_pose_buffers(c);
}

int rz_upload_morphs_dense(rz_ctx *c, uint32_t M, const float *deltas)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_morphs_dense")) return r;
    if (c->V == 0) return fail(RZ_ERR_INVALID, "upload the mesh before its morph targets");
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_morphs(c);
    if (M == 0) return ensure_pose_buffers(c);
    if (!deltas) return fail(RZ_ERR_INVALID, "null morph deltas");
    const size_t Vp = c->Vp, V = c->V;
    size_t dense_off = 0;
#ifdef RZ_ALL_VARIANTS
    // tools-only build: place the morph planes `RZ_DENSE_OFFSET` bytes into a larger allocation (tools/archive/placement.py: how much of the C5
    // frame's box-to-box spread is WHERE the 768 MB land). The offset is lost on free: hipFree gets the shifted pointer — leak, tools only.
    if (const char *e = getenv("RZ_DENSE_OFFSET")) dense_off = (size_t)strtoull(e, nullptr, 0) / 256 * 256;
#endif
    HIP_TRY(hipMalloc(&c->dense, (size_t)M * 3 * Vp * sizeof(float) + dense_off));
    c->dense += dense_off / sizeof(float);
    if (Vp != V) HIP_TRY(hipMemsetAsync(c->dense, 0, (size_t)M * 3 * Vp * sizeof(float), c->stream));
    // stream the host array through a bounded device staging buffer, re-laying each morph into planes
    const uint32_t batch = (uint32_t)std::max<size_t>(1, std::min<size_t>(M, (64u << 20) / (V * 12)));
    Scratch<float> scratch;
    HIP_TRY(scratch.alloc((size_t)batch * V * 3));
    float *tmp = scratch.p;
    for (uint32_t m0 = 0; m0 < M; m0 += batch) {
        const uint32_t nb = std::min(batch, M - m0);
        HIP_TRY(hipMemcpy(tmp, deltas + (size_t)m0 * V * 3, (size_t)nb * V * 3 * sizeof(float), hipMemcpyHostToDevice));
        for (uint32_t k = 0; k < nb; ++k) {
            float *pl = c->dense + (size_t)(m0 + k) * 3 * Vp;
            HIP_TRY(rz_launch_deinterleave(tmp + (size_t)k * V * 3, 3, 0, (uint32_t)V, pl, pl + Vp, pl + 2 * Vp, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->morph_mode = 1;
    c->M = M;
    c->Mpad = round_up(M + 8, 4);
    return ensure_pose_buffers(c);
}

int rz_upload_morphs_sparse(rz_ctx *c, uint32_t M, const uint32_t *morph_off, const uint32_t *vert_idx, const float *delta3)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_morphs_sparse")) return r;
    if (c->V == 0) return fail(RZ_ERR_INVALID, "upload the mesh before its morph targets");
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_morphs(c);
    if (M == 0) return ensure_pose_buffers(c);
    if (!morph_off) return fail(RZ_ERR_INVALID, "null morph offsets");
    const uint32_t E = morph_off[M];
    if (E > 0 && (!vert_idx || !delta3)) return fail(RZ_ERR_INVALID, "null morph entries");
    for (uint32_t m = 0; m < M; ++m)
        if (morph_off[m] > morph_off[m + 1]) return fail(RZ_ERR_INVALID, "morph offsets must be non-decreasing");
    // transpose morph-major (PMX file order) into a per-vertex CSR; a vertex's entries keep
    // ascending morph order (then file order), which is the oracle's accumulation order
    const size_t Vp = c->Vp;
    std::vector<uint32_t> ptr(Vp + 1, 0);
    for (uint32_t e = 0; e < E; ++e)
        if (vert_idx[e] < c->V) ptr[vert_idx[e] + 1]++;
    for (size_t v = 0; v < Vp; ++v) ptr[v + 1] += ptr[v];
    const uint32_t kept = ptr[Vp];
    std::vector<float4> ent(std::max<uint32_t>(kept, 1));
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (uint32_t m = 0; m < M; ++m)
        for (uint32_t e = morph_off[m]; e < morph_off[m + 1]; ++e) {
            const uint32_t v = vert_idx[e];
            if (v >= c->V) continue;
            float4 x;
            x.x = delta3[(size_t)e * 3]; x.y = delta3[(size_t)e * 3 + 1]; x.z = delta3[(size_t)e * 3 + 2];
            memcpy(&x.w, &m, 4);
            ent[cur[v]++] = x;
        }
    HIP_TRY(hipMalloc(&c->sp_ptr, (Vp + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->sp_entries, ent.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(c->sp_ptr, ptr.data(), (Vp + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->sp_entries, ent.data(), ent.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->sp_count = kept;
    c->morph_mode = 2;
    c->M = M;
    c->Mpad = round_up(M + 8, 4);
    return ensure_pose_buffers(c);
}

int rz_set_instances(rz_ctx *c, uint32_t I)
{
    if (int r = use(c)) return r;
    if (I == 0 || I > 65535) return fail(RZ_ERR_INVALID, "instance count must be 1..65535");
    if (I > 1 && (c->comm || c->gather_root)) return fail(RZ_ERR_UNSUPPORTED, "instancing and vertex sharding are exclusive");
    if (I != c->I) {
        forget_search(c);
        drop_graph(c);
        c->aabb_rearm = true;
        c->ovr_count = 0;                 // overrides name (instance, bone) pairs of the old crowd
        // The host-compacted active-morph list is only maintained while I == 1 (upload_pose). Coming back to one instance
        // from a crowd it is stale (zeroed): let the prep kernel compact instance 0's weights, which are still on the device.
        if (c->M > 0 && c->morph_mode == 1) c->ml.count = -1;
        // A crowd larger than the one the resident pose was uploaded for has no pose for its new members (and a
        // single-character pose may still sit in its pinned slot, which holds exactly one instance): ask for a new one.
        if (I > c->pose_I) c->pose_set = false;
        c->zc_epoch++; c->zc_seq_cur = 0;
    }
    c->I = I;
    if (int r = ensure_pose_buffers(c)) return r;
    return ensure_outputs(c);
}

int rz_upload_skeleton_topology(rz_ctx *c, uint32_t B, const int32_t *parents, const float *bind_translation3,
                                const int32_t *append_parent, const float *append_ratio, const uint8_t *append_move)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_skeleton_topology")) return r;
    if (B == 0 || B != c->B) return fail(RZ_ERR_INVALID, "topology has %u bones but the uploaded skeleton has %u", B, c->B);
    if (!parents || !bind_translation3) return fail(RZ_ERR_INVALID, "null topology arrays");
    // hierarchy levels (parents may come in any order, like the reference's recursive solve; cycles are an error)
    std::vector<int> level(B, -1);
    for (uint32_t b = 0; b < B; ++b) {
        if (parents[b] >= (int32_t)B) return fail(RZ_ERR_INVALID, "bone %u parent %d out of range", b, parents[b]);
        std::vector<uint32_t> chain;
        uint32_t cur = b;
        while (level[cur] < 0) {
            chain.push_back(cur);
            if (chain.size() > B) return fail(RZ_ERR_INVALID, "bone hierarchy has a cycle through bone %u", b);
            if (parents[cur] < 0) { level[cur] = 0; chain.pop_back(); break; }
            cur = (uint32_t)parents[cur];
        }
        for (size_t k = chain.size(); k-- > 0;) level[chain[k]] = level[(uint32_t)parents[chain[k]]] + 1;
    }
    int n_levels = 0;
    for (uint32_t b = 0; b < B; ++b) n_levels = std::max(n_levels, level[b] + 1);
    // ancestors by level: up[b][d] for d = 1, 2, ... (parents may come in any order)
    auto ancestor = [&](uint32_t b, uint32_t d) -> uint32_t {
        int32_t cur = (int32_t)b;
        for (uint32_t k = 0; k < d && cur >= 0; ++k) cur = parents[cur];
        return cur < 0 ? 0xffffu : (uint32_t)cur;
    };
    int n_rounds = 0;
    for (int span = 1; span < n_levels; span *= 4) ++n_rounds;          // radix-4 pointer doubling: ceil(log4(depth)) rounds
    // 64-byte bone records (kernels/fk.hip.h): topology | bind translation | the motion's track (rebuild_fk_static) | ancestors of rounds 0, 1
    std::vector<uint4> rec((size_t)B * 4);
    std::vector<uint2> more((size_t)std::max(0, n_rounds - 2) * B);
    for (uint32_t b = 0; b < B; ++b) {
        const int32_t ap = (append_parent && append_parent[b] >= 0 && append_parent[b] < (int32_t)B) ? append_parent[b] : -1;
        const float ratio = append_ratio ? append_ratio[b] : 1.0f;
        uint32_t rb, bx, by, bz;
        memcpy(&rb, &ratio, 4);
        memcpy(&bx, bind_translation3 + (size_t)b * 3, 4); memcpy(&by, bind_translation3 + (size_t)b * 3 + 1, 4); memcpy(&bz, bind_translation3 + (size_t)b * 3 + 2, 4);
        rec[4 * b] = make_uint4((uint32_t)(parents[b] < 0 ? -1 : parents[b]), (uint32_t)ap, rb, (append_move && append_move[b]) ? 1u : 0u);
        rec[4 * b + 1] = make_uint4(bx, by, bz, 0u);
        rec[4 * b + 2] = make_uint4(0u, 0u, 0u, 0u);
        uint32_t span = 1, w[4] = { 0xffffffffu, 0xffffu, 0xffffffffu, 0xffffu };
        for (int r = 0; r < n_rounds; ++r, span *= 4) {
            const uint32_t a1 = ancestor(b, span), a2 = ancestor(b, 2 * span), a3 = ancestor(b, 3 * span);
            if (r < 2) { w[2 * r] = a1 | (a2 << 16); w[2 * r + 1] = a3; }
            else more[(size_t)(r - 2) * B + b] = make_uint2(a1 | (a2 << 16), a3);
        }
        rec[4 * b + 3] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->ovr_count = 0;
    c->fk_stale = false;                // (a crowd frame solved under the old topology: nothing of it can be formed on demand any more)
    dfree(c->fk_anc_more);
    if (!more.empty())
        if (int r = to_device(&c->fk_anc_more, more.data(), more.size())) return r;
    c->fk_host = std::move(rec);
    c->fk_rounds = n_rounds;
    c->has_topology = true;
    return rebuild_fk_static(c);
}

int rz_upload_bone_morphs(rz_ctx *c, uint32_t n, const uint32_t *morph, const uint32_t *bone, const float *translation3, const float *rotation4)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_bone_morphs")) return r;
    if (n == 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        free_bone_morphs(c);
        return RZ_OK;
    }
    if (!c->has_topology) return fail(RZ_ERR_INVALID, "bone morphs act on device-solved poses: call rz_upload_skeleton_topology first");
    if (c->M == 0) return fail(RZ_ERR_INVALID, "upload the morph set first (rz_upload_morphs_*): bone-morph entries name its morphs");
    if (!morph || !bone || !translation3 || !rotation4) return fail(RZ_ERR_INVALID, "null bone-morph arrays");
    for (uint32_t k = 0; k < n; ++k) {
        if (morph[k] >= c->M) return fail(RZ_ERR_INVALID, "bone-morph entry %u names morph %u of %u", k, morph[k], c->M);
        if (bone[k] >= c->B) return fail(RZ_ERR_INVALID, "bone-morph entry %u names bone %u of %u", k, bone[k], c->B);
        for (int j = 0; j < 7; ++j) {
            const float x = j < 3 ? translation3[(size_t)k * 3 + j] : rotation4[(size_t)k * 4 + j - 3];
            if (!(x == x) || x - x != 0.0f) return fail(RZ_ERR_INVALID, "bone-morph entry %u is not finite", k);
        }
    }
    // group by bone; inside a bone ascending morph index, file order among equal morphs (a stable sort of the entry list)
    std::vector<uint32_t> idx(n);
    for (uint32_t k = 0; k < n; ++k) idx[k] = k;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return bone[a] != bone[b] ? bone[a] < bone[b] : morph[a] < morph[b]; });
    std::vector<uint32_t> off(c->B + 1, 0), mo(n);
    std::vector<float4> rot(n), tr(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t e = idx[k];
        off[bone[e] + 1]++;
        mo[k] = morph[e];
        rot[k] = make_float4(rotation4[(size_t)e * 4], rotation4[(size_t)e * 4 + 1], rotation4[(size_t)e * 4 + 2], rotation4[(size_t)e * 4 + 3]);
        tr[k] = make_float4(translation3[(size_t)e * 3], translation3[(size_t)e * 3 + 1], translation3[(size_t)e * 3 + 2], 0.0f);
    }
    for (uint32_t b = 0; b < c->B; ++b) off[b + 1] += off[b];
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_bone_morphs(c);
    drop_graph(c);
    if (int r = to_device(&c->bm_off, off.data(), off.size())) return r;
    if (int r = to_device(&c->bm_morph, mo.data(), n)) return r;
    if (int r = to_device(&c->bm_rot, rot.data(), n)) return r;
    if (int r = to_device(&c->bm_tr, tr.data(), n)) return r;
    c->bm_count = n;
    return RZ_OK;
}

int rz_upload_animation(rz_ctx *c, const rz_animation *a)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_animation")) return r;
    if (!a) return fail(RZ_ERR_INVALID, "null animation");
    if (c->B == 0) return fail(RZ_ERR_INVALID, "upload the skeleton before a motion");
    const uint32_t n = a->n_bone_tracks, mt = a->n_morph_tracks;
    if (n && (!a->track_bone || !a->key_off || !a->key_frame || !a->key_rot4 || !a->key_pos3)) return fail(RZ_ERR_INVALID, "null bone-track arrays");
    if (mt && (!a->mkey_off || !a->mkey_frame || !a->mkey_weight)) return fail(RZ_ERR_INVALID, "null morph-track arrays");
    if (c->M && mt && (!a->feed_off || (a->feed_off[c->M] && (!a->feed_track || !a->feed_ratio)))) return fail(RZ_ERR_INVALID, "null morph feeds");
    std::vector<int> bone_track(c->B, -1);
    const uint32_t K = n ? a->key_off[n] : 0;
    for (uint32_t t = 0; t < n; ++t) {
        if (a->key_off[t] > a->key_off[t + 1]) return fail(RZ_ERR_INVALID, "key offsets must be non-decreasing");
        const int32_t b = a->track_bone[t];
        if (b < 0 || (uint32_t)b >= c->B) continue;                      // a motion may key bones this model lacks
        if (bone_track[b] >= 0) return fail(RZ_ERR_INVALID, "bone %d is driven by two tracks", b);
        for (uint32_t k = a->key_off[t] + 1; k < a->key_off[t + 1]; ++k)
            // equal frames are legal (real VMD files carry duplicate keys; host/vmd-sampler.js keeps them too): the span search
            // lands on the last key <= frame and the first key > frame, so a zero-length span is never divided by
            if (!(a->key_frame[k] >= a->key_frame[k - 1])) return fail(RZ_ERR_INVALID, "track %u: key frames must not descend", t);
        bone_track[b] = (int)t;
    }
    const uint32_t Km = mt ? a->mkey_off[mt] : 0;
    for (uint32_t t = 0; t < mt; ++t) {
        if (a->mkey_off[t] > a->mkey_off[t + 1]) return fail(RZ_ERR_INVALID, "morph key offsets must be non-decreasing");
        for (uint32_t k = a->mkey_off[t] + 1; k < a->mkey_off[t + 1]; ++k)
            if (!(a->mkey_frame[k] >= a->mkey_frame[k - 1])) return fail(RZ_ERR_INVALID, "morph track %u: key frames must not descend", t);
    }
    std::vector<uint32_t> feed_off(c->M + 1, 0);
    uint32_t F = 0;
    if (c->M && mt) {
        for (uint32_t m = 0; m <= c->M; ++m) feed_off[m] = a->feed_off[m];
        F = feed_off[c->M];
        for (uint32_t m = 0; m < c->M; ++m)
            if (feed_off[m] > feed_off[m + 1]) return fail(RZ_ERR_INVALID, "feed offsets must be non-decreasing");
        for (uint32_t f = 0; f < F; ++f)
            if (a->feed_track[f] < 0 || (uint32_t)a->feed_track[f] >= mt) return fail(RZ_ERR_INVALID, "feed %u names morph track %d of %u", f, a->feed_track[f], mt);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_animation(c);
    // per bone / per morph feed: the key range itself, so the sampler's chain of dependent loads starts one level lower
    std::vector<uint4> bone_range(c->B), feed_range(F);
    auto record = [](const uint32_t *off, const float *kf, int t) {
        uint4 r; r.x = r.y = r.z = r.w = 0;
        if (t < 0 || off[t + 1] == off[t]) return r;
        r.x = off[t]; r.y = off[t + 1];
        memcpy(&r.z, &kf[off[t]], 4);
        memcpy(&r.w, &kf[off[t + 1] - 1], 4);
        return r;
    };
    for (uint32_t b = 0; b < c->B; ++b) bone_range[b] = record(a->key_off, a->key_frame, bone_track[b]);
    for (uint32_t f = 0; f < F; ++f) feed_range[f] = record(a->mkey_off, a->mkey_frame, a->feed_track[f]);
    if (int r = to_device(&c->an_key_frame, a->key_frame, K)) return r;
    if (int r = to_device(&c->an_key_rot, a->key_rot4, K)) return r;
    if (int r = to_device(&c->an_key_pos, a->key_pos3, (size_t)K * 3)) return r;
    if (a->key_interp16 && K)
        if (int r = to_device(&c->an_key_interp, a->key_interp16, K)) return r;
    if (int r = to_device(&c->an_mkey_frame, a->mkey_frame, Km)) return r;
    if (int r = to_device(&c->an_mkey_weight, a->mkey_weight, Km)) return r;
    if (int r = to_device(&c->an_feed_off, feed_off.data(), (size_t)c->M + 1)) return r;
    if (int r = to_device(&c->an_feed_range, feed_range.data(), F)) return r;
    if (int r = to_device(&c->an_feed_ratio, a->feed_ratio, F)) return r;
    // per vertex morph: the key range of its FIRST feed and (first feed, end of feeds, bits(its ratio)) — the sampler's one record per morph
    std::vector<uint4> mrec((size_t)c->M * 2, make_uint4(0u, 0u, 0u, 0u));
    for (uint32_t m = 0; m < c->M && F; ++m) {
        const uint32_t f0 = feed_off[m], f1 = feed_off[m + 1];
        if (f1 > f0) {
            uint32_t rb;
            memcpy(&rb, &a->feed_ratio[f0], 4);
            mrec[2 * m] = feed_range[f0];
            mrec[2 * m + 1] = make_uint4(f0, f1, rb, 0u);
        }
    }
    c->an_host_range = std::move(bone_range);
    c->an_host_mrec = std::move(mrec);
    c->an_M = c->M;
    c->has_animation = true;
    return rebuild_fk_static(c);
}

int rz_upload_edge_scale(rz_ctx *c, uint32_t V, const float *edge_size)
{
    if (int r = use(c)) return r;
    if (int r = static_unlocked(c, "rz_upload_edge_scale")) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (!edge_size) { dfree(c->edge); return RZ_OK; }
    if (V != c->V || V == 0) return fail(RZ_ERR_INVALID, "edge scale has %u entries but the mesh has %u vertices", V, c->V);
    dfree(c->edge);
    HIP_TRY(hipMalloc(&c->edge, (size_t)c->Vp * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->edge, 0, (size_t)c->Vp * sizeof(float), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->edge, edge_size, (size_t)V * sizeof(float), hipMemcpyHostToDevice));
    return ensure_outputs(c);
}

int rz_enable_aabb(rz_ctx *c, int enable)
{
    if (int r = use(c)) return r;
    drop_graph(c);
    c->aabb_on = enable != 0;
    c->aabb_rearm = true;
    return ensure_outputs(c);
}

}  // extern "C"
