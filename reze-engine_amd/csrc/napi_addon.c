/*
 * napi_addon.c — raw N-API (node_api.h, N-API <= 8, plain C) shim over the C ABI of
 * include/reze_deform.h. Builds reze_deform.node next to libreze_deform.so.
 *
 * It is the Node-side binding of the reference's de-facto GPU boundary: where class Engine calls
 * device.createBuffer / queue.writeBuffer / dispatchWorkgroups (reference engine/src/engine.ts:
 * 1734-1817, 2383-2401), host/engine.js calls these functions instead. Typed arrays are passed
 * zero-copy (napi_get_typedarray_info); every failure becomes a JS exception carrying
 * rz_last_error(). No threads are created here; calls are serialised by Node's main thread.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/reze_deform.h"

#define MAX_ARGS 8

static napi_value throw_msg(napi_env env, const char *msg)
{
    napi_throw_error(env, "REZE_DEFORM", msg);
    return NULL;
}

static napi_value throw_rz(napi_env env, int code)
{
    char buf[640];
    snprintf(buf, sizeof buf, "reze_deform error %d: %s", code, rz_last_error());
    napi_throw_error(env, "REZE_DEFORM", buf);
    return NULL;
}

#define ARGS(n_min)                                                                       \
    size_t argc = MAX_ARGS;                                                               \
    napi_value argv[MAX_ARGS];                                                            \
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < (n_min)) \
        return throw_msg(env, "wrong number of arguments")

static int get_ctx(napi_env env, napi_value v, rz_ctx **out)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) return 0;
    *out = *(rz_ctx **)p;
    return *out != NULL;
}

#define CTX(i)                                              \
    rz_ctx *ctx = NULL;                                     \
    if (!get_ctx(env, argv[i], &ctx)) return throw_msg(env, "invalid or destroyed deform context")

/* typed array of an exact element type (or null/undefined when `optional`); returns element count */
static int get_ta(napi_env env, napi_value v, napi_typedarray_type want, int optional, void **data, size_t *len)
{
    napi_valuetype t;
    *data = NULL;
    *len = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return optional;
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return 0;
    napi_typedarray_type tt;
    napi_value ab;
    size_t off;
    if (napi_get_typedarray_info(env, v, &tt, len, data, &ab, &off) != napi_ok) return 0;
    return tt == want;
}

/* numbers are validated, not wrapped: napi_get_value_uint32(-1) would quietly hand back 4294967295 (frames, vertices ...) */
static int get_u32(napi_env env, napi_value v, uint32_t *out)
{
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok || !(d >= 0.0) || d > 4294967295.0 || d != (double)(uint32_t)d) return 0;
    *out = (uint32_t)d;
    return 1;
}
static int get_i32(napi_env env, napi_value v, int32_t *out)
{
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok || !(d >= -2147483648.0) || d > 2147483647.0 || d != (double)(int32_t)d) return 0;
    *out = (int32_t)d;
    return 1;
}

static napi_value undef(napi_env env)
{
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* What an external handle points at. `ctx` comes first, so a (rz_ctx **) view of the slot still reads the context.
 * A fork's slot holds a strong reference on its lender's external: the garbage collector finalizes in no particular order,
 * and rz_destroy() refuses a lender whose forks are alive — with the reference the lender cannot be collected (and so cannot
 * be leaked by a refused destroy) before its forks are gone. */
typedef struct ctx_slot {
    rz_ctx *ctx;
    napi_ref lender;        /* forks only */
    napi_ref mapped[2];     /* the ArrayBuffers mapPose handed out over the pinned ring slot (matrices, morph weights): detached by commitPose */
} ctx_slot;

/* A view handed out by mapPose must not outlive the mapping: the slot goes back to the ring and the GPU reads it. Detaching the
 * ArrayBuffers turns every later write through a stale Float32Array into a no-op on a zero-length buffer instead of a race. */
static void detach_mapped(napi_env env, ctx_slot *slot)
{
    for (int k = 0; k < 2; ++k) {
        if (!slot->mapped[k]) continue;
        napi_value ab;
        if (napi_get_reference_value(env, slot->mapped[k], &ab) == napi_ok && ab) napi_detach_arraybuffer(env, ab);
        napi_delete_reference(env, slot->mapped[k]);
        slot->mapped[k] = NULL;
    }
}

static void release_lender(napi_env env, ctx_slot *slot)
{
    if (slot->lender) {
        napi_delete_reference(env, slot->lender);
        slot->lender = NULL;
    }
}

static void finalize_ctx(napi_env env, void *data, void *hint)
{
    (void)hint;
    ctx_slot *slot = (ctx_slot *)data;
    if (slot) {
        detach_mapped(env, slot);
        if (slot->ctx) rz_destroy(slot->ctx);     /* a fork, or a lender whose forks are gone (they held it alive until now) */
        release_lender(env, slot);
        free(slot);
    }
}

static napi_value fn_abi_version(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value v;
    napi_create_int32(env, rz_abi_version(), &v);
    return v;
}

static napi_value fn_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    int n = 0;
    rz_device_count(&n);
    napi_value v;
    napi_create_int32(env, n, &v);
    return v;
}

/* deviceNumaNode(device) -> node, -1 = unknown (rz_device_numa_node: start the process on that node's cores, e.g. under numactl) */
static napi_value fn_device_numa_node(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], v;
    int32_t device = 0;
    int node = -1;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (argc >= 1) napi_get_value_int32(env, argv[0], &device);
    const int rc = rz_device_numa_node(device, &node);
    if (rc != RZ_OK) return throw_rz(env, rc);
    napi_create_int32(env, node, &v);
    return v;
}

static napi_value fn_create(napi_env env, napi_callback_info info)
{
    ARGS(1);
    int32_t dev = 0;
    if (!get_i32(env, argv[0], &dev)) return throw_msg(env, "create(device: number)");
    rz_ctx *c = NULL;
    int rc = rz_create(dev, &c);
    if (rc) return throw_rz(env, rc);
    ctx_slot *slot = (ctx_slot *)calloc(1, sizeof *slot);
    if (!slot) { rz_destroy(c); return throw_msg(env, "out of memory"); }
    slot->ctx = c;
    napi_value ext;
    if (napi_create_external(env, slot, finalize_ctx, NULL, &ext) != napi_ok) {
        rz_destroy(c);
        free(slot);
        return throw_msg(env, "napi_create_external failed");
    }
    return ext;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info)
{
    ARGS(1);
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return throw_msg(env, "invalid context");
    ctx_slot *slot = (ctx_slot *)p;
    if (slot->ctx) {
        /* rz_destroy refuses a context whose forks are alive: the handle then STAYS valid (throwing away the pointer would
         * leak the mesh and morph targets for good) and the caller hears about it */
        detach_mapped(env, slot);            /* the ring a mapped view points into goes with the context */
        int rc = rz_destroy(slot->ctx);
        if (rc) return throw_rz(env, rc);
        slot->ctx = NULL;
        release_lender(env, slot);
    }
    return undef(env);
}

static napi_value fn_shard_range(napi_env env, napi_callback_info info)
{
    ARGS(3);
    uint32_t vt, b = 0, n = 0;
    int32_t nr, r;
    if (!get_u32(env, argv[0], &vt) || !get_i32(env, argv[1], &nr) || !get_i32(env, argv[2], &r))
        return throw_msg(env, "shardRange(vTotal, nranks, rank)");
    int rc = rz_shard_range(vt, nr, r, &b, &n);
    if (rc) return throw_rz(env, rc);
    napi_value arr, vb, vn;
    napi_create_array_with_length(env, 2, &arr);
    napi_create_uint32(env, b, &vb);
    napi_create_uint32(env, n, &vn);
    napi_set_element(env, arr, 0, vb);
    napi_set_element(env, arr, 1, vn);
    return arr;
}

static napi_value fn_gather_chunk(napi_env env, napi_callback_info info)
{
    ARGS(2);
    uint32_t vt, c = 0;
    int32_t nr;
    if (!get_u32(env, argv[0], &vt) || !get_i32(env, argv[1], &nr)) return throw_msg(env, "gatherChunk(vTotal, nranks)");
    int rc = rz_gather_chunk(vt, nr, &c);
    if (rc) return throw_rz(env, rc);
    napi_value v;
    napi_create_uint32(env, c, &v);
    return v;
}

static napi_value fn_upload_mesh(napi_env env, napi_callback_info info)
{
    ARGS(4);
    CTX(0);
    void *v, *j, *w;
    size_t nv, nj, nw;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &v, &nv) || !get_ta(env, argv[2], napi_uint16_array, 0, &j, &nj) ||
        !get_ta(env, argv[3], napi_uint8_array, 0, &w, &nw))
        return throw_msg(env, "uploadMesh(ctx, Float32Array vertices8, Uint16Array joints4, Uint8Array weights4)");
    if (nv % 8 || nj != nv / 8 * 4 || nw != nv / 8 * 4) return throw_msg(env, "uploadMesh: array lengths disagree");
    int rc = rz_upload_mesh(ctx, (uint32_t)(nv / 8), (const float *)v, (const uint16_t *)j, (const uint8_t *)w);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_mesh_soa(napi_env env, napi_callback_info info)
{
    ARGS(5);
    CTX(0);
    void *p, *n, *j, *w;
    size_t np, nn, nj, nw;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &p, &np) || !get_ta(env, argv[2], napi_float32_array, 0, &n, &nn) ||
        !get_ta(env, argv[3], napi_uint16_array, 0, &j, &nj) || !get_ta(env, argv[4], napi_uint8_array, 0, &w, &nw))
        return throw_msg(env, "uploadMeshSoa(ctx, Float32Array pos3, Float32Array nrm3, Uint16Array joints4, Uint8Array weights4)");
    if (np % 3 || nn != np || nj != np / 3 * 4 || nw != np / 3 * 4) return throw_msg(env, "uploadMeshSoa: array lengths disagree");
    int rc = rz_upload_mesh_soa(ctx, (uint32_t)(np / 3), (const float *)p, (const float *)n, (const uint16_t *)j,
                                (const uint8_t *)w);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_skeleton(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    void *ib;
    size_t n;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &ib, &n) || n % 16)
        return throw_msg(env, "uploadSkeleton(ctx, Float32Array inverseBind /* B*16 */)");
    int rc = rz_upload_skeleton(ctx, (uint32_t)(n / 16), (const float *)ib);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_morphs_dense(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    uint32_t M;
    void *d;
    size_t n;
    if (!get_u32(env, argv[1], &M) || !get_ta(env, argv[2], napi_float32_array, 1, &d, &n))
        return throw_msg(env, "uploadMorphsDense(ctx, M, Float32Array deltas /* M*V*3 */)");
    // the C ABI copies M*V*3 floats out of `deltas`: the typed array must be exactly that long for THIS context's mesh
    // (a shorter one — wrong shard, stale model — would be read past its end)
    int V = 0;
    if (rz_get_tuning(ctx, "verts", &V)) return throw_rz(env, RZ_ERR_INVALID);
    if (M && (!d || n != (size_t)M * (size_t)V * 3))
        return throw_msg(env, "uploadMorphsDense: deltas length must be M*V*3 for the uploaded mesh");
    int rc = rz_upload_morphs_dense(ctx, M, (const float *)d);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_morphs_sparse(napi_env env, napi_callback_info info)
{
    ARGS(4);
    CTX(0);
    void *off, *idx, *d;
    size_t noff, nidx, nd;
    if (!get_ta(env, argv[1], napi_uint32_array, 0, &off, &noff) || !get_ta(env, argv[2], napi_uint32_array, 0, &idx, &nidx) ||
        !get_ta(env, argv[3], napi_float32_array, 0, &d, &nd) || noff < 1)
        return throw_msg(env, "uploadMorphsSparse(ctx, Uint32Array morphOffsets /* M+1 */, Uint32Array vertexIndex, Float32Array delta3)");
    const uint32_t *o = (const uint32_t *)off;
    if (o[noff - 1] > nidx || nd < (size_t)o[noff - 1] * 3) return throw_msg(env, "uploadMorphsSparse: entries shorter than the offsets claim");
    int rc = rz_upload_morphs_sparse(ctx, (uint32_t)(noff - 1), o, (const uint32_t *)idx, (const float *)d);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_set_instances(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    uint32_t I;
    if (!get_u32(env, argv[1], &I)) return throw_msg(env, "setInstances(ctx, count)");
    int rc = rz_set_instances(ctx, I);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_set_pose(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    void *w, *mw = NULL;
    size_t nw, nmw = 0;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &w, &nw)) return throw_msg(env, "setPose(ctx, Float32Array world, Float32Array|null morphWeights)");
    if (argc > 2 && !get_ta(env, argv[2], napi_float32_array, 1, &mw, &nmw)) return throw_msg(env, "setPose: morphWeights must be a Float32Array or null");
    /* lengths are validated against the uploaded skeleton / morph counts by querying them back */
    int B = 0, M = 0, I = 0;
    if (rz_get_tuning(ctx, "bones", &B) || rz_get_tuning(ctx, "morphs", &M) || rz_get_tuning(ctx, "instances", &I))
        return throw_rz(env, RZ_ERR_INVALID);
    if (nw != (size_t)I * B * 16) return throw_msg(env, "setPose: world must hold instances*bones*16 floats");
    if (mw && nmw != (size_t)I * M) return throw_msg(env, "setPose: morphWeights must hold instances*morphs floats");
    int rc = rz_set_pose(ctx, (const float *)w, (const float *)mw);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_instance_range(napi_env env, napi_callback_info info)
{
    ARGS(3);
    uint32_t it, b = 0, n = 0;
    int32_t nr, r;
    if (!get_u32(env, argv[0], &it) || !get_i32(env, argv[1], &nr) || !get_i32(env, argv[2], &r))
        return throw_msg(env, "instanceRange(instances, nranks, rank)");
    int rc = rz_instance_range(it, nr, r, &b, &n);
    if (rc) return throw_rz(env, rc);
    napi_value arr, vb, vn;
    napi_create_array_with_length(env, 2, &arr);
    napi_create_uint32(env, b, &vb);
    napi_create_uint32(env, n, &vn);
    napi_set_element(env, arr, 0, vb);
    napi_set_element(env, arr, 1, vn);
    return arr;
}

/* mapPose(ctx, layout) -> { matrices: Float32Array, morphWeights: Float32Array | null } — views OVER the pinned ring slot the next pose
 * upload would have copied into (rz_map_pose): the pose solve writes its matrices in place (layout 0: 16 floats per bone, math.ts's
 * layout; 1: 12 floats per bone, the four columns' x y z — crowds only), commitPose(ctx) hands them to the GPU without a copy and
 * detaches the views. Replaces queue.writeBuffer(worldMatrixBuffer)  engine/src/engine.ts:2383-2389. */
static napi_value fn_map_pose(napi_env env, napi_callback_info info)
{
    ARGS(1);
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p || !((ctx_slot *)p)->ctx) return throw_msg(env, "invalid or destroyed deform context");
    ctx_slot *slot = (ctx_slot *)p;
    int32_t layout = RZ_POSE_WORLD16;
    if (argc > 1 && !get_i32(env, argv[1], &layout)) return throw_msg(env, "mapPose(ctx, layout = 0 | 1)");
    detach_mapped(env, slot);            /* a mapping that was never committed */
    float *mats = NULL, *mw = NULL;
    int rc = rz_map_pose(slot->ctx, layout, &mats, &mw);
    if (rc) return throw_rz(env, rc);
    int B = 0, M = 0, I = 0;
    if (rz_get_tuning(slot->ctx, "bones", &B) || rz_get_tuning(slot->ctx, "morphs", &M) || rz_get_tuning(slot->ctx, "instances", &I))
        return throw_rz(env, RZ_ERR_INVALID);
    const size_t nm = (size_t)I * B * (layout == RZ_POSE_ROWS12 ? 12 : 16), nw = (size_t)I * M;
    napi_value out, ab, ta, nul;
    napi_create_object(env, &out);
    napi_get_null(env, &nul);
    if (napi_create_external_arraybuffer(env, mats, nm * sizeof(float), NULL, NULL, &ab) != napi_ok ||
        napi_create_typedarray(env, napi_float32_array, nm, ab, 0, &ta) != napi_ok)
        return throw_msg(env, "mapPose: could not wrap the ring slot");
    napi_create_reference(env, ab, 1, &slot->mapped[0]);
    napi_set_named_property(env, out, "matrices", ta);
    if (mw && nw) {
        if (napi_create_external_arraybuffer(env, mw, nw * sizeof(float), NULL, NULL, &ab) != napi_ok ||
            napi_create_typedarray(env, napi_float32_array, nw, ab, 0, &ta) != napi_ok)
            return throw_msg(env, "mapPose: could not wrap the ring slot");
        napi_create_reference(env, ab, 1, &slot->mapped[1]);
        napi_set_named_property(env, out, "morphWeights", ta);
    } else {
        napi_set_named_property(env, out, "morphWeights", nul);
    }
    return out;
}

static napi_value fn_commit_pose(napi_env env, napi_callback_info info)
{
    ARGS(1);
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p || !((ctx_slot *)p)->ctx) return throw_msg(env, "invalid or destroyed deform context");
    ctx_slot *slot = (ctx_slot *)p;
    int rc = rz_commit_pose(slot->ctx);
    detach_mapped(env, slot);
    return rc ? throw_rz(env, rc) : undef(env);
}

/* timeSpan(ctx, forkCtx | null, frames, lead = 0) -> ms between two events on the stream around `frames` back-to-back frames, `lead` untimed
 * frames in front of the opening event (rz_time_span). */
static napi_value fn_time_span(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    rz_ctx *other = NULL;
    napi_valuetype t;
    uint32_t n, lead = 0;
    if (napi_typeof(env, argv[1], &t) != napi_ok || !get_u32(env, argv[2], &n)) return throw_msg(env, "timeSpan(ctx, forkCtx | null, frames, lead = 0)");
    if (argc > 3 && !get_u32(env, argv[3], &lead)) return throw_msg(env, "timeSpan: lead is a frame count");
    if (t != napi_null && t != napi_undefined && !get_ctx(env, argv[1], &other)) return throw_msg(env, "timeSpan: the second argument is a fork or null");
    double ms = 0.0;
    int rc = rz_time_span(ctx, other, lead, n, &ms);
    if (rc) return throw_rz(env, rc);
    napi_value v;
    napi_create_double(env, ms, &v);
    return v;
}

static napi_value fn_upload_topology(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    void *par, *bind, *ap = NULL, *ar = NULL, *am = NULL;
    size_t npar, nbind, nap = 0, nar = 0, nam = 0;
    if (!get_ta(env, argv[1], napi_int32_array, 0, &par, &npar) || !get_ta(env, argv[2], napi_float32_array, 0, &bind, &nbind) || nbind != npar * 3)
        return throw_msg(env, "uploadSkeletonTopology(ctx, Int32Array parents, Float32Array bindTranslation /* B*3 */, Int32Array|null appendParent, Float32Array|null appendRatio, Uint8Array|null appendMove)");
    if (argc > 3 && !get_ta(env, argv[3], napi_int32_array, 1, &ap, &nap)) return throw_msg(env, "appendParent must be an Int32Array or null");
    if (argc > 4 && !get_ta(env, argv[4], napi_float32_array, 1, &ar, &nar)) return throw_msg(env, "appendRatio must be a Float32Array or null");
    if (argc > 5 && !get_ta(env, argv[5], napi_uint8_array, 1, &am, &nam)) return throw_msg(env, "appendMove must be a Uint8Array or null");
    if ((ap && nap != npar) || (ar && nar != npar) || (am && nam != npar)) return throw_msg(env, "uploadSkeletonTopology: array lengths disagree");
    int rc = rz_upload_skeleton_topology(ctx, (uint32_t)npar, (const int32_t *)par, (const float *)bind, (const int32_t *)ap, (const float *)ar,
                                         (const uint8_t *)am);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_set_pose_local(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    void *q, *mw = NULL, *tr = NULL;
    size_t nq, nmw = 0, ntr = 0;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &q, &nq)) return throw_msg(env, "setPoseLocal(ctx, Float32Array localRotations, Float32Array|null morphWeights, Float32Array|null localTranslations)");
    if (argc > 2 && !get_ta(env, argv[2], napi_float32_array, 1, &mw, &nmw)) return throw_msg(env, "setPoseLocal: morphWeights must be a Float32Array or null");
    if (argc > 3 && !get_ta(env, argv[3], napi_float32_array, 1, &tr, &ntr)) return throw_msg(env, "setPoseLocal: localTranslations must be a Float32Array or null");
    int B = 0, M = 0, I = 0;
    if (rz_get_tuning(ctx, "bones", &B) || rz_get_tuning(ctx, "morphs", &M) || rz_get_tuning(ctx, "instances", &I)) return throw_rz(env, RZ_ERR_INVALID);
    if (nq != (size_t)I * B * 4) return throw_msg(env, "setPoseLocal: localRotations must hold instances*bones*4 floats");
    if (mw && nmw != (size_t)I * M) return throw_msg(env, "setPoseLocal: morphWeights must hold instances*morphs floats");
    if (tr && ntr != (size_t)I * B * 3) return throw_msg(env, "setPoseLocal: localTranslations must hold instances*bones*3 floats");
    int rc = rz_set_pose_local(ctx, (const float *)q, (const float *)tr, (const float *)mw);
    return rc ? throw_rz(env, rc) : undef(env);
}

/* typed-array property of an object argument (or absent / null when optional) */
static int get_prop_ta(napi_env env, napi_value obj, const char *name, napi_typedarray_type want, int optional, void **data, size_t *len)
{
    napi_value v;
    bool has = false;
    *data = NULL; *len = 0;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok) return 0;
    if (!has) return optional;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    return get_ta(env, v, want, optional, data, len);
}

static napi_value fn_upload_animation(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    napi_valuetype vt;
    if (napi_typeof(env, argv[1], &vt) != napi_ok || vt != napi_object)
        return throw_msg(env, "uploadAnimation(ctx, { trackBone, keyOff, keyFrame, keyRot, keyPos, keyInterp?, mkeyOff?, mkeyFrame?, mkeyWeight?, feedOff?, feedTrack?, feedRatio? })");
    void *tb, *ko, *kf, *kr, *kp, *ki, *mo, *mf, *mw, *fo, *ft, *fr;
    size_t ntb, nko, nkf, nkr, nkp, nki, nmo, nmf, nmw, nfo, nft, nfr;
    napi_value o = argv[1];
    if (!get_prop_ta(env, o, "trackBone", napi_int32_array, 0, &tb, &ntb) || !get_prop_ta(env, o, "keyOff", napi_uint32_array, 0, &ko, &nko) ||
        !get_prop_ta(env, o, "keyFrame", napi_float32_array, 0, &kf, &nkf) || !get_prop_ta(env, o, "keyRot", napi_float32_array, 0, &kr, &nkr) ||
        !get_prop_ta(env, o, "keyPos", napi_float32_array, 0, &kp, &nkp) || !get_prop_ta(env, o, "keyInterp", napi_uint8_array, 1, &ki, &nki) ||
        !get_prop_ta(env, o, "mkeyOff", napi_uint32_array, 1, &mo, &nmo) || !get_prop_ta(env, o, "mkeyFrame", napi_float32_array, 1, &mf, &nmf) ||
        !get_prop_ta(env, o, "mkeyWeight", napi_float32_array, 1, &mw, &nmw) || !get_prop_ta(env, o, "feedOff", napi_uint32_array, 1, &fo, &nfo) ||
        !get_prop_ta(env, o, "feedTrack", napi_int32_array, 1, &ft, &nft) || !get_prop_ta(env, o, "feedRatio", napi_float32_array, 1, &fr, &nfr))
        return throw_msg(env, "uploadAnimation: a field has the wrong typed-array type");
    int M = 0;
    if (rz_get_tuning(ctx, "morphs", &M)) return throw_rz(env, RZ_ERR_INVALID);
    const size_t n = ntb;
    if (nko != n + 1) return throw_msg(env, "uploadAnimation: keyOff must hold trackBone.length + 1 offsets");
    const size_t K = n ? ((const uint32_t *)ko)[n] : 0;
    if (nkf < K || nkr < K * 4 || nkp < K * 3 || (ki && nki < K * 16)) return throw_msg(env, "uploadAnimation: key arrays are shorter than keyOff says");
    const size_t mt = mo ? (nmo ? nmo - 1 : 0) : 0;
    const size_t Km = mt ? ((const uint32_t *)mo)[mt] : 0;
    if (mt && (!mf || !mw || nmf < Km || nmw < Km)) return throw_msg(env, "uploadAnimation: morph key arrays are shorter than mkeyOff says");
    if (mt && M > 0) {
        if (!fo || nfo != (size_t)M + 1) return throw_msg(env, "uploadAnimation: feedOff must hold morphs + 1 offsets");
        const size_t F = ((const uint32_t *)fo)[M];
        if (F && (!ft || !fr || nft < F || nfr < F)) return throw_msg(env, "uploadAnimation: feed arrays are shorter than feedOff says");
    }
    rz_animation an;
    memset(&an, 0, sizeof an);
    an.n_bone_tracks = (uint32_t)n; an.track_bone = (const int32_t *)tb; an.key_off = (const uint32_t *)ko; an.key_frame = (const float *)kf;
    an.key_rot4 = (const float *)kr; an.key_pos3 = (const float *)kp; an.key_interp16 = (const uint8_t *)ki;
    an.n_morph_tracks = (uint32_t)mt; an.mkey_off = (const uint32_t *)mo; an.mkey_frame = (const float *)mf; an.mkey_weight = (const float *)mw;
    an.feed_off = (const uint32_t *)fo; an.feed_track = (const int32_t *)ft; an.feed_ratio = (const float *)fr;
    int rc = rz_upload_animation(ctx, &an);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_set_pose_sampled(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    void *f;
    size_t nf;
    int I = 0;
    if (!get_ta(env, argv[1], napi_float32_array, 0, &f, &nf)) return throw_msg(env, "setPoseSampled(ctx, Float32Array frames /* one per instance */)");
    if (rz_get_tuning(ctx, "instances", &I) || nf != (size_t)I) return throw_msg(env, "setPoseSampled: frames must hold one float per instance");
    int rc = rz_set_pose_sampled(ctx, (const float *)f);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_override_world(napi_env env, napi_callback_info info)
{
    ARGS(4);
    CTX(0);
    void *b = NULL, *w = NULL, *in = NULL;
    size_t nb = 0, nw = 0, ni = 0;
    if (!get_ta(env, argv[1], napi_uint32_array, 1, &b, &nb) || !get_ta(env, argv[2], napi_float32_array, 1, &w, &nw) ||
        !get_ta(env, argv[3], napi_uint32_array, 1, &in, &ni))
        return throw_msg(env, "overrideWorld(ctx, Uint32Array|null bones, Float32Array|null world16 /* n*16 */, Uint32Array|null instances)");
    if (nw != nb * 16 || (in && ni != nb)) return throw_msg(env, "overrideWorld: array lengths disagree");
    int rc = rz_override_world(ctx, (uint32_t)nb, (const uint32_t *)in, (const uint32_t *)b, (const float *)w);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_fork(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    rz_ctx *f = NULL;
    int rc = rz_fork(ctx, &f);
    if (rc) return throw_rz(env, rc);
    ctx_slot *slot = (ctx_slot *)calloc(1, sizeof *slot);
    if (!slot) { rz_destroy(f); return throw_msg(env, "out of memory"); }
    slot->ctx = f;
    napi_value ext;
    if (napi_create_reference(env, argv[0], 1, &slot->lender) != napi_ok ||
        napi_create_external(env, slot, finalize_ctx, NULL, &ext) != napi_ok) {
        rz_destroy(f);
        release_lender(env, slot);
        free(slot);
        return throw_msg(env, "napi_create_external failed");
    }
    return ext;
}

static napi_value fn_deform_pair(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    rz_ctx *other = NULL;
    uint32_t n;
    if (!get_ctx(env, argv[1], &other) || !get_u32(env, argv[2], &n)) return throw_msg(env, "deformPair(ctx, forkCtx, frames)");
    int rc = rz_deform_pair(ctx, other, n);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_bone_morphs(napi_env env, napi_callback_info info)
{
    ARGS(5);
    CTX(0);
    void *m = NULL, *b = NULL, *t = NULL, *q = NULL;
    size_t nm = 0, nb = 0, nt = 0, nq = 0;
    if (!get_ta(env, argv[1], napi_uint32_array, 1, &m, &nm) || !get_ta(env, argv[2], napi_uint32_array, 1, &b, &nb) ||
        !get_ta(env, argv[3], napi_float32_array, 1, &t, &nt) || !get_ta(env, argv[4], napi_float32_array, 1, &q, &nq))
        return throw_msg(env, "uploadBoneMorphs(ctx, Uint32Array|null morph, Uint32Array|null bone, Float32Array|null translation3, Float32Array|null rotation4)");
    if (nb != nm || nt != nm * 3 || nq != nm * 4) return throw_msg(env, "uploadBoneMorphs: array lengths disagree");
    int rc = rz_upload_bone_morphs(ctx, (uint32_t)nm, (const uint32_t *)m, (const uint32_t *)b, (const float *)t, (const float *)q);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read_world(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    uint32_t inst;
    void *o;
    size_t n;
    int B = 0;
    if (!get_u32(env, argv[1], &inst) || !get_ta(env, argv[2], napi_float32_array, 0, &o, &n)) return throw_msg(env, "readWorld(ctx, instance, Float32Array out /* B*16 */)");
    if (rz_get_tuning(ctx, "bones", &B) || n < (size_t)B * 16) return throw_msg(env, "readWorld: output array too small");
    int rc = rz_read_world(ctx, inst, (float *)o);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_upload_edge_scale(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    void *e = NULL;
    size_t n = 0;
    if (!get_ta(env, argv[1], napi_float32_array, 1, &e, &n)) return throw_msg(env, "uploadEdgeScale(ctx, Float32Array|null edgeSizePerVertex)");
    int rc = rz_upload_edge_scale(ctx, (uint32_t)n, (const float *)e);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read_hull(napi_env env, napi_callback_info info)
{
    ARGS(5);
    CTX(0);
    uint32_t inst, v0, n;
    void *o;
    size_t no;
    if (!get_u32(env, argv[1], &inst) || !get_u32(env, argv[2], &v0) || !get_u32(env, argv[3], &n) ||
        !get_ta(env, argv[4], napi_float32_array, 0, &o, &no) || no < (size_t)n * 3)
        return throw_msg(env, "readHull(ctx, instance, v0, n, Float32Array out /* n*3 */)");
    int rc = rz_read_hull(ctx, inst, v0, n, (float *)o);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_enable_aabb(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    bool on = false;
    napi_get_value_bool(env, argv[1], &on);
    int rc = rz_enable_aabb(ctx, on ? 1 : 0);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read_aabb(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    uint32_t inst;
    void *o;
    size_t no;
    if (!get_u32(env, argv[1], &inst) || !get_ta(env, argv[2], napi_float32_array, 0, &o, &no) || no < 6)
        return throw_msg(env, "readAabb(ctx, instance, Float32Array out /* 6 */)");
    int rc = rz_read_aabb(ctx, inst, (float *)o);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_deform(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    int rc = rz_deform(ctx);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_deform_n(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    uint32_t n;
    if (!get_u32(env, argv[1], &n)) return throw_msg(env, "deformN(ctx, frames)");
    int rc = rz_deform_n(ctx, n);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_sync(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    int rc = rz_sync(ctx);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read(napi_env env, napi_callback_info info)
{
    ARGS(6);
    CTX(0);
    uint32_t inst, v0, n;
    void *p, *nr;
    size_t np, nn;
    if (!get_u32(env, argv[1], &inst) || !get_u32(env, argv[2], &v0) || !get_u32(env, argv[3], &n) ||
        !get_ta(env, argv[4], napi_float32_array, 1, &p, &np) || !get_ta(env, argv[5], napi_float32_array, 1, &nr, &nn))
        return throw_msg(env, "read(ctx, instance, v0, n, Float32Array|null pos3, Float32Array|null nrm3)");
    if ((p && np < (size_t)n * 3) || (nr && nn < (size_t)n * 3)) return throw_msg(env, "read: output arrays too small");
    int rc = rz_read(ctx, inst, v0, n, (float *)p, (float *)nr);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read_palette(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    uint32_t inst;
    void *o;
    size_t n;
    int B = 0;
    if (!get_u32(env, argv[1], &inst) || !get_ta(env, argv[2], napi_float32_array, 0, &o, &n))
        return throw_msg(env, "readPalette(ctx, instance, Float32Array out /* B*12 */)");
    if (rz_get_tuning(ctx, "bones", &B) || n < (size_t)B * 12) return throw_msg(env, "readPalette: output array too small");
    int rc = rz_read_palette(ctx, inst, (float *)o);
    return rc ? throw_rz(env, rc) : undef(env);
}

static void set_num(napi_env env, napi_value obj, const char *k, double v)
{
    napi_value nv;
    napi_create_double(env, v, &nv);
    napi_set_named_property(env, obj, k, nv);
}

static napi_value fn_time_frames(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    uint32_t n;
    if (!get_u32(env, argv[1], &n)) return throw_msg(env, "timeFrames(ctx, frames)");
    rz_timing t;
    int rc = rz_time_frames(ctx, n, &t);
    if (rc) return throw_rz(env, rc);
    napi_value o;
    napi_create_object(env, &o);
    set_num(env, o, "frameMs", t.frame_ms);
    set_num(env, o, "deformKernelMs", t.deform_kernel_ms);
    set_num(env, o, "prepKernelMs", t.prep_kernel_ms);
    set_num(env, o, "vertsPerFrame", (double)t.verts_per_frame);
    set_num(env, o, "algorithmicBytesPerFrame", (double)t.algorithmic_bytes_per_frame);
    set_num(env, o, "frames", t.frames);
    return o;
}

static napi_value fn_set_tuning(napi_env env, napi_callback_info info)
{
    ARGS(3);
    CTX(0);
    char key[64];
    size_t kl = 0;
    int32_t v;
    if (napi_get_value_string_utf8(env, argv[1], key, sizeof key, &kl) != napi_ok || !get_i32(env, argv[2], &v))
        return throw_msg(env, "setTuning(ctx, key, value)");
    int rc = rz_set_tuning(ctx, key, v);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_get_tuning(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    char key[64];
    size_t kl = 0;
    if (napi_get_value_string_utf8(env, argv[1], key, sizeof key, &kl) != napi_ok) return throw_msg(env, "getTuning(ctx, key)");
    int v = 0;
    int rc = rz_get_tuning(ctx, key, &v);
    if (rc) return throw_rz(env, rc);
    napi_value nv;
    napi_create_int32(env, v, &nv);
    return nv;
}

static napi_value fn_comm_unique_id(napi_env env, napi_callback_info info)
{
    (void)info;
    char id[128];
    int rc = rz_comm_unique_id(id);
    if (rc) return throw_rz(env, rc);
    napi_value buf;
    void *dst = NULL;
    if (napi_create_buffer_copy(env, 128, id, &dst, &buf) != napi_ok) return throw_msg(env, "buffer alloc failed");
    return buf;
}

static napi_value fn_rccl_info(napi_env env, napi_callback_info info)
{
    (void)info;
    char path[512];
    int ver = 0, reused = 0;
    int rc = rz_rccl_info(path, sizeof path, &ver, &reused);
    if (rc) return throw_rz(env, rc);
    napi_value o, v;
    if (napi_create_object(env, &o) != napi_ok) return throw_msg(env, "object alloc failed");
    napi_create_string_utf8(env, path, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "path", v);
    napi_create_int32(env, ver, &v); napi_set_named_property(env, o, "version", v);
    napi_get_boolean(env, reused != 0, &v); napi_set_named_property(env, o, "reused", v);
    return o;
}

static napi_value fn_comm_init(napi_env env, napi_callback_info info)
{
    ARGS(5);
    CTX(0);
    int32_t nr, r;
    uint32_t vt;
    void *id = NULL;
    size_t idl = 0;
    bool is_buf = false;        /* napi_get_buffer_info aborts the process on a non-Buffer in Node 12: ask first */
    if (!get_i32(env, argv[1], &nr) || !get_i32(env, argv[2], &r) || napi_is_buffer(env, argv[3], &is_buf) != napi_ok || !is_buf ||
        napi_get_buffer_info(env, argv[3], &id, &idl) != napi_ok || idl != 128 || !get_u32(env, argv[4], &vt))
        return throw_msg(env, "commInit(ctx, nranks, rank, Buffer id /* 128 B */, vTotal)");
    int rc = rz_comm_init(ctx, nr, r, (const char *)id, vt);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_allgather(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    bool wn = false;
    if (argc > 1) napi_get_value_bool(env, argv[1], &wn);
    int rc = rz_allgather(ctx, wn ? 1 : 0);
    return rc ? throw_rz(env, rc) : undef(env);
}

/* JS array of context handles -> rz_ctx*[] (at most 64) */
static int get_ctx_list(napi_env env, napi_value arr, rz_ctx **out, int *n)
{
    bool is_arr = false;
    uint32_t len = 0;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, arr, &len) != napi_ok || len < 1 || len > 64) return 0;
    for (uint32_t i = 0; i < len; ++i) {
        napi_value v;
        if (napi_get_element(env, arr, i, &v) != napi_ok || !get_ctx(env, v, &out[i])) return 0;
    }
    *n = (int)len;
    return 1;
}

static napi_value fn_comm_init_all(napi_env env, napi_callback_info info)
{
    ARGS(2);
    rz_ctx *list[64];
    int n = 0;
    uint32_t vt;
    if (!get_ctx_list(env, argv[0], list, &n) || !get_u32(env, argv[1], &vt)) return throw_msg(env, "commInitAll(ctx[], vTotal)");
    int rc = rz_comm_init_all(list, n, vt);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_allgather_all(napi_env env, napi_callback_info info)
{
    ARGS(1);
    rz_ctx *list[64];
    int n = 0;
    bool wn = false;
    if (!get_ctx_list(env, argv[0], list, &n)) return throw_msg(env, "allgatherAll(ctx[], withNormals?)");
    if (argc > 1) napi_get_value_bool(env, argv[1], &wn);
    int rc = rz_allgather_all(list, n, wn ? 1 : 0);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_autotune(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    uint32_t frames = 0;
    if (argc > 1 && !get_u32(env, argv[1], &frames)) return throw_msg(env, "autotune(ctx, frames?)");
    int rc = rz_autotune(ctx, frames);
    return rc ? throw_rz(env, rc) : undef(env);
}

/* rz_autotune_measure -> array of { morphSplit, gridCap, instLoop, effSplit, effGrid, effInstGroup, sameAs, ms, msMin, msMax };
 * entry 0 is the heuristic plan. rz_autotune_pick / rz_autotune_apply take such an array / one such object. */
static const char *const kTuneInts[7] = { "morphSplit", "gridCap", "instLoop", "effSplit", "effGrid", "effInstGroup", "sameAs" };
static const char *const kTuneFloats[3] = { "ms", "msMin", "msMax" };

static int entry_from_js(napi_env env, napi_value o, rz_tune_entry *e)
{
    int *ints[7] = { &e->morph_split, &e->grid_cap, &e->inst_loop, &e->eff_split, &e->eff_grid, &e->eff_inst_group, &e->same_as };
    float *floats[3] = { &e->ms, &e->ms_min, &e->ms_max };
    napi_valuetype t;
    if (napi_typeof(env, o, &t) != napi_ok || t != napi_object) return 0;
    for (int k = 0; k < 7; ++k) {
        napi_value v;
        int32_t x;
        if (napi_get_named_property(env, o, kTuneInts[k], &v) != napi_ok || !get_i32(env, v, &x)) return 0;
        *ints[k] = x;
    }
    for (int k = 0; k < 3; ++k) {
        napi_value v;
        double d;
        if (napi_get_named_property(env, o, kTuneFloats[k], &v) != napi_ok || napi_get_value_double(env, v, &d) != napi_ok) return 0;
        *floats[k] = (float)d;
    }
    return 1;
}

static napi_value fn_autotune_measure(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    uint32_t frames = 0;
    if (argc > 1 && !get_u32(env, argv[1], &frames)) return throw_msg(env, "autotuneMeasure(ctx, frames?)");
    rz_tune_entry tab[32];
    int n = 0;
    int rc = rz_autotune_measure(ctx, frames, tab, 32, &n);
    if (rc) return throw_rz(env, rc);
    napi_value arr;
    if (napi_create_array_with_length(env, (size_t)n, &arr) != napi_ok) return throw_msg(env, "array alloc failed");
    for (int i = 0; i < n; ++i) {
        const int ints[7] = { tab[i].morph_split, tab[i].grid_cap, tab[i].inst_loop, tab[i].eff_split, tab[i].eff_grid, tab[i].eff_inst_group, tab[i].same_as };
        const float floats[3] = { tab[i].ms, tab[i].ms_min, tab[i].ms_max };
        napi_value o, v;
        if (napi_create_object(env, &o) != napi_ok) return throw_msg(env, "object alloc failed");
        for (int k = 0; k < 7; ++k) { napi_create_int32(env, ints[k], &v); napi_set_named_property(env, o, kTuneInts[k], v); }
        for (int k = 0; k < 3; ++k) { napi_create_double(env, (double)floats[k], &v); napi_set_named_property(env, o, kTuneFloats[k], v); }
        napi_set_element(env, arr, (uint32_t)i, o);
    }
    return arr;
}

static napi_value fn_autotune_pick(napi_env env, napi_callback_info info)
{
    ARGS(1);
    bool is_arr = false;
    uint32_t len = 0;
    if (napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &len) != napi_ok || len < 1 || len > 32)
        return throw_msg(env, "autotunePick(table: entry[1..32])");
    rz_tune_entry tab[32];
    for (uint32_t i = 0; i < len; ++i) {
        napi_value o;
        if (napi_get_element(env, argv[0], i, &o) != napi_ok || !entry_from_js(env, o, &tab[i])) return throw_msg(env, "autotunePick: malformed entry");
    }
    napi_value v;
    napi_create_int32(env, rz_autotune_pick(tab, (int)len), &v);
    return v;
}

static napi_value fn_autotune_apply(napi_env env, napi_callback_info info)
{
    ARGS(2);
    CTX(0);
    rz_tune_entry e;
    if (!entry_from_js(env, argv[1], &e)) return throw_msg(env, "autotuneApply(ctx, entry)");
    int rc = rz_autotune_apply(ctx, &e);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_comm_info(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    int n = 0, u = -1;
    int rc = rz_comm_info(ctx, &n, &u);
    if (rc) return throw_rz(env, rc);
    napi_value o, v;
    if (napi_create_object(env, &o) != napi_ok) return throw_msg(env, "object alloc failed");
    napi_create_int32(env, n, &v); napi_set_named_property(env, o, "commCount", v);
    napi_create_int32(env, u, &v); napi_set_named_property(env, o, "commUserRank", v);
    return o;
}

static napi_value fn_gather_direct(napi_env env, napi_callback_info info)
{
    ARGS(3);
    rz_ctx *list[64];
    int n = 0;
    uint32_t vt, root;
    if (!get_ctx_list(env, argv[0], list, &n) || !get_u32(env, argv[1], &vt) || !get_u32(env, argv[2], &root))
        return throw_msg(env, "gatherDirect(ctx[], vTotal, root)");
    int rc = rz_gather_direct(list, n, vt, (int)root);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_gather_fence(napi_env env, napi_callback_info info)
{
    ARGS(1);
    CTX(0);
    int rc = rz_gather_fence(ctx);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value fn_read_gathered(napi_env env, napi_callback_info info)
{
    ARGS(5);
    CTX(0);
    uint32_t v0, n;
    void *p, *nr;
    size_t np, nn;
    if (!get_u32(env, argv[1], &v0) || !get_u32(env, argv[2], &n) || !get_ta(env, argv[3], napi_float32_array, 1, &p, &np) ||
        !get_ta(env, argv[4], napi_float32_array, 1, &nr, &nn))
        return throw_msg(env, "readGathered(ctx, v0, n, Float32Array|null pos3, Float32Array|null nrm3)");
    if ((p && np < (size_t)n * 3) || (nr && nn < (size_t)n * 3)) return throw_msg(env, "readGathered: output arrays too small");
    int rc = rz_read_gathered(ctx, v0, n, (float *)p, (float *)nr);
    return rc ? throw_rz(env, rc) : undef(env);
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } table[] = {
        { "abiVersion", fn_abi_version }, { "deviceCount", fn_device_count }, { "deviceNumaNode", fn_device_numa_node }, { "create", fn_create },
        { "destroy", fn_destroy }, { "shardRange", fn_shard_range }, { "gatherChunk", fn_gather_chunk }, { "uploadMesh", fn_upload_mesh },
        { "uploadMeshSoa", fn_upload_mesh_soa }, { "uploadSkeleton", fn_upload_skeleton },
        { "uploadMorphsDense", fn_upload_morphs_dense }, { "uploadMorphsSparse", fn_upload_morphs_sparse },
        { "setInstances", fn_set_instances }, { "setPose", fn_set_pose }, { "uploadSkeletonTopology", fn_upload_topology },
        { "setPoseLocal", fn_set_pose_local }, { "readWorld", fn_read_world }, { "uploadEdgeScale", fn_upload_edge_scale },
        { "readHull", fn_read_hull }, { "enableAabb", fn_enable_aabb }, { "readAabb", fn_read_aabb }, { "deform", fn_deform },
        { "deformN", fn_deform_n }, { "sync", fn_sync }, { "read", fn_read }, { "readPalette", fn_read_palette },
        { "timeFrames", fn_time_frames }, { "setTuning", fn_set_tuning }, { "getTuning", fn_get_tuning },
        { "commUniqueId", fn_comm_unique_id }, { "rcclInfo", fn_rccl_info }, { "commInit", fn_comm_init }, { "allgather", fn_allgather },
        { "readGathered", fn_read_gathered }, { "commInitAll", fn_comm_init_all }, { "allgatherAll", fn_allgather_all },
        { "autotune", fn_autotune }, { "autotuneMeasure", fn_autotune_measure }, { "autotunePick", fn_autotune_pick }, { "autotuneApply", fn_autotune_apply }, { "commInfo", fn_comm_info }, { "uploadAnimation", fn_upload_animation }, { "setPoseSampled", fn_set_pose_sampled }, { "overrideWorld", fn_override_world }, { "uploadBoneMorphs", fn_upload_bone_morphs }, { "fork", fn_fork }, { "deformPair", fn_deform_pair }, { "gatherDirect", fn_gather_direct }, { "gatherFence", fn_gather_fence },
        { "instanceRange", fn_instance_range }, { "mapPose", fn_map_pose }, { "commitPose", fn_commit_pose }, { "timeSpan", fn_time_span },
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i) {
        napi_value f;
        if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, table[i].name, f);
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
