// core.cpp — context life cycle, the buffers every unit sizes, error state (part of libreze_deform.so's host side: ctx.h).
//
// One rz_ctx = one MI355X: a HIP stream, the static mesh shard re-laid-out as planar SoA, the skeleton, optional morph targets,
// per-frame pose staging, output buffers, and (optionally) an RCCL communicator for the all-gather of deformed positions. Every entry
// point cites, in the header, the reference call site it replaces; the host side is only plumbing around the kernels in kernels/.
// There is NO CPU fallback: without a working HIP device every call fails.
#include "ctx.h"

#include <cctype>

using namespace rzi;

namespace rzi {

static thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const char *last_error() { return g_err.c_str(); }

int use(rz_ctx *c)
{
    if (!c) return fail(RZ_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return RZ_OK;
}

// static data shared between a context and its forks cannot be replaced
int static_unlocked(const rz_ctx *c, const char *what)
{
    if (c->lender) return fail(RZ_ERR_INVALID, "%s on a fork: static data belongs to the context it was forked from", what);
    if (c->n_forks) return fail(RZ_ERR_INVALID, "%s while %d fork(s) of this context share its static data: destroy them first", what, c->n_forks);
    return RZ_OK;
}

// Poll an event (no sleep: a blocking wait wakes tens of microseconds late, which starves a GPU whose frames are 16 us long
// — measured: 36 us per frame). Bounded: a GPU that stops making progress turns into an error after 10 s, not a hang.
int poll_event(hipEvent_t ev, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return RZ_OK;
        if (q != hipErrorNotReady) return fail(RZ_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
            return fail(RZ_ERR_HIP, "%s: the GPU made no progress for 10 s", what);
    }
}

// A captured hipGraph bakes device pointers and launch shapes in. The replay key (frame_signature) covers all of them; on
// top of that every entry point that frees or re-shapes something a frame reads drops the graph outright.
void drop_graph(rz_ctx *c)
{
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    c->graph_sig = 0;
}

int ensure_outputs(rz_ctx *c)
{
    // one instance: room for a whole all-gather chunk; instances are strided by Vp
    const size_t need = (c->I == 1) ? std::max<size_t>(c->Vp, c->chunk) * 3 : (size_t)c->I * c->Vp * 3;
    if (need == 0) return RZ_OK;
    if (need > c->out_alloc_floats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->out_pos);
        dfree(c->out_nrm);
        HIP_TRY(hipMalloc(&c->out_pos, need * sizeof(float)));
        HIP_TRY(hipMalloc(&c->out_nrm, need * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->out_pos, 0, need * sizeof(float), c->stream));
        HIP_TRY(hipMemsetAsync(c->out_nrm, 0, need * sizeof(float), c->stream));
        c->out_alloc_floats = need;
    }
    if (c->edge && need > c->hull_alloc_floats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->out_hull);
        HIP_TRY(hipMalloc(&c->out_hull, need * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->out_hull, 0, need * sizeof(float), c->stream));
        c->hull_alloc_floats = need;
    }
    // Bounding-box keys: a frame accumulates into one slot and re-arms the other FOR THE INSTANCES IT LAUNCHES, so the
    // buffer is armed from scratch whenever the reduction is switched on or the instance count changes (aabb_rearm) —
    // otherwise an instance that sat out some frames would come back onto a slot still holding its old extents.
    if (c->aabb_on && (c->I > c->aabb_alloc_inst || c->aabb_rearm)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->I > c->aabb_alloc_inst) {
            dfree(c->aabb);
            HIP_TRY(hipMalloc(&c->aabb, (size_t)c->I * 12 * sizeof(uint32_t)));
            c->aabb_alloc_inst = c->I;
        }
        std::vector<uint32_t> init((size_t)c->aabb_alloc_inst * 12);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
        HIP_TRY(hipMemcpy(c->aabb, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->aabb_slot = 0;
        c->aabb_rearm = false;
    }
    return RZ_OK;
}

void set_ring(rz_ctx *c, int slot)
{
    c->ring_slot = slot;
    c->palette = c->palette_ring[slot]; c->act_idx = c->act_idx_ring[slot]; c->act_w = c->act_w_ring[slot]; c->act_count = c->act_count_ring[slot];
}

// Lay the CURRENT instance / bone / morph counts out over pose slot k and make it the current slot.
void point_pose_at(rz_ctx *c, float *block)
{
    const size_t I = c->I, B = c->B, Mq = std::max<uint32_t>(c->M, 1);
    c->mw_pad = (I * Mq + 3) / 4 * 4;
    c->world = block;
    c->morph_w = block + I * B * 16;
    c->local_q = reinterpret_cast<float4 *>(block + I * B * 16 + c->mw_pad);
}

void point_pose_slot(rz_ctx *c, int k)
{
    c->pose_slot = k;
    point_pose_at(c, c->pose_blk[k]);
}

// (callers have drained both streams)
void free_big_ring(rz_ctx *c)
{
    for (int k = 0; k < rz_ctx::kBigBlocks; ++k) dfree(c->big_blk[k]);
    c->big_floats = 0;
    c->big_uploads = 0;
    c->big_ev_seq[0] = c->big_ev_seq[1] = ~0ull;
}

int ensure_pose_buffers(rz_ctx *c)
{
    if (c->B == 0) return RZ_OK;
    const uint32_t Mq = std::max<uint32_t>(c->M, 1);
    if (c->I <= c->pose_alloc_I && c->B <= c->pose_alloc_B && Mq <= c->pose_alloc_M && c->world) return RZ_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    for (int k = 0; k < 2; ++k) dfree(c->pose_blk[k]);
    free_big_ring(c);
    c->world = nullptr; c->morph_w = nullptr; c->local_q = nullptr;
    for (int k = 0; k < 2; ++k) { dfree(c->palette_ring[k]); dfree(c->act_idx_ring[k]); dfree(c->act_w_ring[k]); dfree(c->act_count_ring[k]); c->skin_recorded[k] = false; }
    c->palette = nullptr; c->act_idx = nullptr; c->act_w = nullptr; c->act_count = nullptr;
    const size_t I = c->I, B = c->B;
    const size_t Mpad = round_up(Mq + 8, 4);
    const size_t blk_floats = I * B * 16 + ((I * Mq + 3) / 4 * 4) + I * B * 7 + 4;      // + 4: the prefetch helper copies whole 16-byte cells
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&c->pose_blk[k], blk_floats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->pose_blk[k], 0, blk_floats * sizeof(float), c->stream));
    }
    point_pose_slot(c, 0);
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&c->palette_ring[k], I * B * 3 * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->act_idx_ring[k], I * Mpad * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->act_w_ring[k], I * Mpad * sizeof(float)));
        HIP_TRY(hipMalloc(&c->act_count_ring[k], I * sizeof(int)));
        HIP_TRY(hipMemsetAsync(c->palette_ring[k], 0, I * B * 3 * sizeof(float4), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_idx_ring[k], 0, I * Mpad * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_w_ring[k], 0, I * Mpad * sizeof(float), c->stream));
        HIP_TRY(hipMemsetAsync(c->act_count_ring[k], 0, I * sizeof(int), c->stream));
    }
    set_ring(c, 0);
    if (!c->zc_tag) HIP_TRY(hipMalloc(&c->zc_tag, 2 * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(c->zc_tag, 0, 2 * sizeof(uint64_t), c->stream));
    c->zc_epoch++;                      // poses staged under the old layout must never match again
    c->zc_seq_cur = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->pose_alloc_I = I; c->pose_alloc_B = B; c->pose_alloc_M = Mq;
    c->pose_set = false;
    return RZ_OK;
}

void free_animation(rz_ctx *c)
{
    drop_graph(c);
    dfree(c->an_feed_range); dfree(c->an_feed_off);
    c->an_host_range.clear(); c->an_host_mrec.clear();
    dfree(c->an_key_frame); dfree(c->an_key_pos); dfree(c->an_mkey_frame); dfree(c->an_mkey_weight); dfree(c->an_feed_ratio);
    dfree(c->an_key_rot); dfree(c->an_key_interp);
    c->has_animation = false;
    if (c->pose_sampled) { c->pose_sampled = false; c->pose_set = false; }
}

void free_bone_morphs(rz_ctx *c)
{
    if (!c->bm_off) return;
    drop_graph(c);
    dfree(c->bm_off); dfree(c->bm_morph); dfree(c->bm_rot); dfree(c->bm_tr);
    c->bm_count = 0;
}

// A launch shape found by rz_autotune belongs to the workload it was timed on.
void forget_search(rz_ctx *c)
{
    if (!c->tuned_by_search) return;
    c->t_split = 0; c->t_grid_cap = 0; c->t_instloop = -1;
    c->tuned_by_search = false;
}

void free_morphs(rz_ctx *c)
{
    forget_search(c);
    drop_graph(c);
    dfree(c->dense); dfree(c->sp_ptr); dfree(c->sp_entries);
    free_bone_morphs(c);                  // their entries name morphs of the old set
    c->morph_mode = 0; c->M = 0; c->Mpad = 12; c->sp_count = 0;
    c->zc_epoch++; c->zc_seq_cur = 0;     // ... and so does a pose staged ahead of its frame
    c->pose_set = false;                  // morph weights belong to the old target set
}

}  // namespace rzi

extern "C" {

const char *rz_last_error(void) { return last_error(); }
int rz_abi_version(void) { return RZ_ABI_VERSION; }

int rz_device_count(int *count)
{
    if (!count) return fail(RZ_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RZ_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RZ_OK;
}

int rz_device_numa_node(int device, int *node)
{
    if (!node) return fail(RZ_ERR_INVALID, "null node");
    *node = -1;
    char bus[32] = {0};
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        (void)hipGetLastError();        // (the runtime keeps a failed call's error for the next hipGetLastError(): a launch check would report it)
        return fail(RZ_ERR_NO_DEVICE, "no device %d (%d visible)", device, n_dev);
    }
    hipError_t e = hipDeviceGetPCIBusId(bus, (int)sizeof bus, device);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(RZ_ERR_NO_DEVICE, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e)); }
    for (char *p = bus; *p; ++p) *p = (char)tolower((unsigned char)*p);      // sysfs spells the address in lower case
    char path[128];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    if (FILE *f = fopen(path, "r")) {
        int n = -1;
        if (fscanf(f, "%d", &n) == 1) *node = n;
        fclose(f);
    }
    return RZ_OK;
}

int rz_create(int device, rz_ctx **out)
{
    if (!out) return fail(RZ_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(RZ_ERR_NO_DEVICE, "no HIP device: %s", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RZ_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RZ_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device,
                    prop.gcnArchName);
    rz_ctx *c = new rz_ctx();
    memset(&c->ml, 0, sizeof c->ml);
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && se == hipSuccess; ++k) {
        se = hipEventCreateWithFlags(&c->big_ev[k], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_front[k], hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_skin[k], hipEventDisableTiming);
    }
    if (se == hipSuccess) se = hipEventCreate(&c->ev0);
    if (se == hipSuccess) se = hipEventCreate(&c->ev1);
    for (int i = 0; i < kStageSlots && se == hipSuccess; ++i)
        se = hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming);
    if (se != hipSuccess) {
        rz_destroy(c);
        return fail(RZ_ERR_HIP, "context setup failed: %s", hipGetErrorString(se));
    }
    *out = c;
    return RZ_OK;
}

int rz_destroy(rz_ctx *c)
{
    if (!c) return RZ_OK;
    if (c->n_forks) return fail(RZ_ERR_INVALID, "%d fork(s) still borrow this context's static data: destroy them first", c->n_forks);
    (void)hipSetDevice(c->device);
#ifdef RZ_ALL_VARIANTS
    if (c->gate_host) *reinterpret_cast<volatile uint32_t *>(c->gate_host) = 1u;      // a test died with the gate closed: open it before draining the stream
#endif
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->lender) {                      // a fork frees nothing it borrowed
        c->geom = nullptr; c->j01 = c->j23 = c->wq = nullptr; c->inv_bind = nullptr;
        c->fk_rec = nullptr; c->fk_anc_more = nullptr;
        c->an_feed_range = nullptr; c->an_feed_off = nullptr;
        c->an_key_frame = c->an_key_pos = c->an_mkey_frame = c->an_mkey_weight = c->an_feed_ratio = nullptr; c->an_key_rot = nullptr; c->an_key_interp = nullptr;
        c->bm_off = c->bm_morph = nullptr; c->bm_rot = c->bm_tr = nullptr;
        c->dense = nullptr; c->sp_ptr = nullptr; c->sp_entries = nullptr; c->edge = nullptr;
        c->lender->n_forks--;
        c->lender = nullptr;
    }
    drop_direct_gather(c);
    drop_graph(c);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    dfree(c->geom); dfree(c->j01); dfree(c->j23); dfree(c->wq); dfree(c->inv_bind);
    dfree(c->rj01); dfree(c->rj23); dfree(c->sub_list); dfree(c->sub_count); dfree(c->zc_tag);
    dfree(c->subfk_rec); dfree(c->subfk_count);
    dfree(c->fk_rec); dfree(c->fk_anc_more);
    free_animation(c); dfree(c->an_frames);
    dfree(c->pose_blk[0]); dfree(c->pose_blk[1]);
    free_big_ring(c);
    dfree(c->ovr_off); dfree(c->ovr_bone); dfree(c->ovr_world);
    free_bone_morphs(c);
    free_morphs(c);
    for (int k = 0; k < 2; ++k) {
        if (c->big_ev[k]) (void)hipEventDestroy(c->big_ev[k]);
    }
    for (int k = 0; k < 2; ++k) {
        dfree(c->palette_ring[k]); dfree(c->act_idx_ring[k]); dfree(c->act_w_ring[k]); dfree(c->act_count_ring[k]);
        if (c->ev_front[k]) (void)hipEventDestroy(c->ev_front[k]);
        if (c->ev_skin[k]) (void)hipEventDestroy(c->ev_skin[k]);
    }
    dfree(c->out_pos); dfree(c->out_nrm); dfree(c->g_pos); dfree(c->g_nrm);
#ifdef RZ_ABLATE
    dfree(c->tl);
#endif
#ifdef RZ_ALL_VARIANTS
    if (c->gate_host) { (void)hipHostFree(c->gate_host); c->gate_host = nullptr; }
#endif
    dfree(c->edge); dfree(c->out_hull); dfree(c->aabb);
    for (int i = 0; i < kStageSlots; ++i) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
    }
    for (int i = 0; i < rz_ctx::kZcSlots; ++i)
        if (c->zc_host[i]) (void)hipHostFree(c->zc_host[i]);
    for (int e = 0; e < 2; ++e)
        if (c->zc_ev[e]) (void)hipEventDestroy(c->zc_ev[e]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RZ_OK;
}

int rz_fork(rz_ctx *parent, rz_ctx **out)
{
    if (!out) return fail(RZ_ERR_INVALID, "null out");
    *out = nullptr;
    if (int r = use(parent)) return r;
    if (parent->lender) return fail(RZ_ERR_INVALID, "rz_fork of a fork: fork the context that owns the static data");
    if (parent->V == 0 || !parent->geom || parent->B == 0 || !parent->inv_bind) return fail(RZ_ERR_INVALID, "rz_fork needs a mesh and a skeleton (rz_upload_mesh, rz_upload_skeleton)");
    if (parent->comm || parent->gather_root) return fail(RZ_ERR_UNSUPPORTED, "rz_fork of a context that takes part in a gather");
    HIP_TRY(hipStreamSynchronize(parent->stream));        // every static upload of the lender has landed (before anything is created: nothing to undo on failure)
    rz_ctx *c = nullptr;
    if (int r = rz_create(parent->device, &c)) return r;
    c->V = parent->V; c->Vp = parent->Vp; c->geom = parent->geom; c->j01 = parent->j01; c->j23 = parent->j23; c->wq = parent->wq;
    c->B = parent->B; c->inv_bind = parent->inv_bind;
    c->has_topology = parent->has_topology; c->fk_rec = parent->fk_rec; c->fk_anc_more = parent->fk_anc_more; c->fk_rounds = parent->fk_rounds;
    // ... and the host-side mirrors the plan reads (plan.cpp: subfk_wanted / ensure_subfk build a crowd's closure records from them): without
    // them a fork of a device-animated crowd never took the one-launch frame and a context and its fork alternated two frame shapes
    c->fk_host = parent->fk_host; c->an_host_range = parent->an_host_range; c->an_host_mrec = parent->an_host_mrec; c->fk_gen = parent->fk_gen;
    c->has_animation = parent->has_animation; c->an_feed_range = parent->an_feed_range;
    c->an_feed_off = parent->an_feed_off; c->an_key_frame = parent->an_key_frame; c->an_key_pos = parent->an_key_pos;
    c->an_mkey_frame = parent->an_mkey_frame; c->an_mkey_weight = parent->an_mkey_weight; c->an_feed_ratio = parent->an_feed_ratio;
    c->an_key_rot = parent->an_key_rot; c->an_key_interp = parent->an_key_interp; c->an_M = parent->an_M;
    c->bm_off = parent->bm_off; c->bm_morph = parent->bm_morph; c->bm_rot = parent->bm_rot; c->bm_tr = parent->bm_tr; c->bm_count = parent->bm_count;
    c->morph_mode = parent->morph_mode; c->M = parent->M; c->Mpad = parent->Mpad; c->dense = parent->dense;
    c->sp_ptr = parent->sp_ptr; c->sp_entries = parent->sp_entries; c->sp_count = parent->sp_count;
    c->edge = parent->edge; c->aabb_on = parent->aabb_on; c->aabb_rearm = parent->aabb_on;
    c->I = parent->I;
    c->t_split = parent->t_split; c->t_unroll = parent->t_unroll; c->t_grid_cap = parent->t_grid_cap; c->t_nt = parent->t_nt; c->t_nts = parent->t_nts;
    c->t_geo = parent->t_geo; c->t_fast = parent->t_fast; c->t_instloop = parent->t_instloop; c->t_outcap = parent->t_outcap; c->t_instblock = parent->t_instblock;
    c->t_instorder = parent->t_instorder; c->t_overlap = parent->t_overlap; c->t_zerocopy = parent->t_zerocopy; c->t_pull = parent->t_pull; c->t_fkplain = parent->t_fkplain; c->t_fusefk = parent->t_fusefk;
    c->t_graph = parent->t_graph; c->tuned_by_search = parent->tuned_by_search; c->t_subsets = parent->t_subsets; c->t_prefetch = parent->t_prefetch;
    c->lender = parent;
    parent->n_forks++;
    int rc = ensure_pose_buffers(c);
    if (rc == RZ_OK) rc = ensure_outputs(c);
    if (rc != RZ_OK) { rz_destroy(c); return rc; }
    *out = c;
    return RZ_OK;
}

int rz_sync(rz_ctx *c)
{
    if (int r = use(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->up_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RZ_OK;
}

int rz_output_ptrs(rz_ctx *c, void **pos, void **nrm, uint32_t *v_padded)
{
    if (int r = use(c)) return r;
    if (int r = ensure_outputs(c)) return r;
    if (pos) *pos = c->ext_pos ? c->ext_pos : c->out_pos;
    if (nrm) *nrm = c->ext_nrm ? c->ext_nrm : c->out_nrm;
    if (v_padded) *v_padded = c->Vp;
    return RZ_OK;
}

}  // extern "C"
