/*
 * reze_deform.h — C ABI of libreze_deform.so: the MI355X-native replacement for the reference's
 * per-frame PMX morph + skin GPU path.
 *
 * The reference (AmyangXYZ/reze-engine, TypeScript + WGSL) has no FFI; its de-facto boundary is
 * the set of WebGPU calls class Engine makes for this path. Every entry point below names the
 * reference call site it replaces (paths relative to the reference repo root). The N-API addon
 * (reze-engine_amd/csrc/napi_addon.c) and the Python ctypes binding (reze-engine_amd/capi.py) are
 * thin shims over exactly these symbols — see INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative rz_status;
 *     rz_last_error() returns a thread-local human-readable message for the last failure.
 *   - "host" pointers are caller-owned and are copied (or fully consumed) before the call returns.
 *   - one rz_ctx drives ONE GPU (one process — or one context — per GPU); calls on a context are
 *     serialised by the caller (Node's main thread). Work is enqueued on the context's own HIP
 *     streams (compute, plus an upload stream for large per-frame inputs) and is asynchronous
 *     unless stated; rz_sync() drains it.
 *   - matrices are column-major float[16] exactly as engine/src/math.ts stores them.
 *   - a context owns a contiguous vertex shard [v_begin, v_begin + V) of a mesh of v_total
 *     vertices (v_begin = 0, v_total = V on a single GPU).
 */
#ifndef REZE_DEFORM_H
#define REZE_DEFORM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 5): shards are cut at 256 vertices (4: 1 024 before round 4's change, which should have bumped this), so the chunk stride of the
 *    gathered buffer changed; rz_gather_chunk exports it instead of making callers re-derive it.
 * 6 (round 5): + rz_device_numa_node.
 * 7 (round 6): + rz_map_pose / rz_commit_pose (caller-written poses), rz_time_span (event-timed K-step span), rz_instance_range
 *    (crowds sharded along the instance axis). Nothing removed or changed: a binding written against 5 or 6 keeps working. */
#define RZ_ABI_VERSION 7

typedef struct rz_ctx rz_ctx;

typedef enum rz_status {
    RZ_OK = 0,
    RZ_ERR_INVALID = -1,      /* bad argument / call order                       */
    RZ_ERR_HIP = -2,          /* a HIP runtime call failed (message has details) */
    RZ_ERR_NO_DEVICE = -3,    /* no usable gfx950 device                         */
    RZ_ERR_RCCL = -4,         /* an RCCL call failed                             */
    RZ_ERR_OOM = -5,
    RZ_ERR_UNSUPPORTED = -6
} rz_status;

/* Timing of the most recent rz_time_frames() / per-frame statistics (getStats() additions). */
typedef struct rz_timing {
    double frame_ms;          /* average wall ms per frame over the timed run (HIP events)     */
    double deform_kernel_ms;  /* average ms of the fused morph+skin kernel alone               */
    double prep_kernel_ms;    /* average ms of the palette / active-morph prep kernel          */
    uint64_t verts_per_frame; /* instances x vertices of this shard                            */
    uint64_t algorithmic_bytes_per_frame; /* SURVEY §8d formula for this shard                 */
    uint32_t frames;
    uint32_t reserved;
} rz_timing;

const char *rz_last_error(void);
int rz_abi_version(void);

/* Number of visible HIP devices. */
int rz_device_count(int *count);
/* The NUMA node of the host the device hangs off, from its PCI address (sysfs); *node = -1 when the system does not say (one socket,
 * no sysfs). No reference counterpart (a browser tab has no say in where it runs). Per-frame inputs cross the host link: a thread
 * that feeds a GPU from the OTHER socket pays for it twice — every doorbell / queue write / signal read of a HIP call crosses the
 * socket link (+14 us of host time per rz_set_pose of a 256-character crowd on 2 x EPYC 9575F), and so does every byte the GPU pulls
 * out of pinned memory that thread first touched (2.5 MB: 57 -> 77 us). The library never moves the caller's threads; bench.py binds
 * each rank to its GPU's node with this (profiles/r5_crowd_upload_numa.txt), a Node host would be started under numactl. */
int rz_device_numa_node(int device, int *node);

/* Engine.init()  engine/src/engine.ts:157-185 (requestAdapter/requestDevice): bind to one GPU,
 * create the stream, events and staging memory. */
int rz_create(int device, rz_ctx **out);
/* Engine.dispose()  engine/src/engine.ts:1692-1701. */
int rz_destroy(rz_ctx *ctx);

/* Pure helper: contiguous shard of rank `rank` of `nranks` over v_total vertices.
 * Shards are equal-sized (a multiple of 256 vertices) except the last; *count may be 0. */
int rz_shard_range(uint32_t v_total, int nranks, int rank, uint32_t *begin, uint32_t *count);
/* The uniform stride, in vertices, between two ranks' blocks of the gathered buffers (rz_allgather / rz_gather_direct /
 * rz_read_gathered): rank r's vertices land at [r * chunk, r * chunk + count_r). It is rank 0's shard size rounded up to the shard
 * grain — ask for it here rather than re-deriving the rule (the grain changed from 1 024 to 256 vertices with ABI 5). */
int rz_gather_chunk(uint32_t v_total, int nranks, uint32_t *chunk);

/* Crowds shard along the INSTANCE axis (last sentence): rank `rank` of `nranks` poses instances [begin, begin + count)
 * of a crowd of `instances` characters — equal contiguous ranges of ceil(I / N), the last may be shorter, *count may be 0. Every rank
 * holds the whole static mesh (rz_upload_mesh of all V vertices) and calls rz_set_instances(count); there is no collective at all,
 * and no communicator (rz_comm_init refuses a crowd). The reference draws one model per engine (engine/src/engine.ts:1704-1721). */
int rz_instance_range(uint32_t instances, int nranks, int rank, uint32_t *begin, uint32_t *count);

/* setupModelBuffers()  engine/src/engine.ts:1734-1765: vertex buffer in the reference's
 * interleaved layout (8 floats/vertex: pos3 nrm3 uv2, engine.ts:340-347), joints Uint16x4
 * (:348-352), weights Unorm8x4 (:353-356). De-interleaved to planar SoA on upload.
 * The arrays describe ONLY this context's shard (V vertices). */
int rz_upload_mesh(rz_ctx *ctx, uint32_t V, const float *interleaved8, const uint16_t *joints4,
                   const uint8_t *weights4);
/* Same, from separate packed position / normal arrays ([V][3] each). */
int rz_upload_mesh_soa(rz_ctx *ctx, uint32_t V, const float *pos3, const float *nrm3,
                       const uint16_t *joints4, const uint8_t *weights4);

/* inverseBindMatrixBuffer + boneCountBuffer  engine/src/engine.ts:1767-1804. */
int rz_upload_skeleton(rz_ctx *ctx, uint32_t B, const float *inverse_bind16);

/* Vertex-morph targets — NEW capability, no reference counterpart (the reference skips the PMX
 * morph section, engine/src/pmx-loader.ts:450-553). Dense: deltas[M][V][3] for this shard.
 * Sparse: PMX on-disk form (pmx-loader.ts:483-488) — morph m owns entries
 * [morph_off[m], morph_off[m+1]) of (vert_idx relative to this shard, delta xyz).
 * Passing M = 0 removes morphs. */
int rz_upload_morphs_dense(rz_ctx *ctx, uint32_t M, const float *deltas);
int rz_upload_morphs_sparse(rz_ctx *ctx, uint32_t M, const uint32_t *morph_off,
                            const uint32_t *vert_idx, const float *delta3);

/* Instancing — NEW (the reference draws one model): I poses of the same static mesh. Shrinking the crowd keeps the
 * resident pose (instances 0 .. I-1 of it); growing it beyond the count the pose was set for needs a new rz_set_pose*. */
int rz_set_instances(rz_ctx *ctx, uint32_t I);

/* updateModelPose()  engine/src/engine.ts:2383-2389: queue.writeBuffer(worldMatrixBuffer).
 * world = I x B x 16 floats; morph_weights = I x M floats or NULL (all zero). Asynchronous H2D
 * through pinned staging; the data is consumed by the next rz_deform(). */
int rz_set_pose(rz_ctx *ctx, const float *world, const float *morph_weights);

/* Caller-written poses (ABI 7). updateModelPose()  engine/src/engine.ts:2383-2389 is ONE queue.writeBuffer of the matrices the pose
 * solve left in Model's own Float32Array. rz_set_pose costs a crowd a copy on top of that: one host thread lays 3.3 MB out in the
 * pinned ring (44 us for 256 x 200 bones) before the GPU can pull it. rz_map_pose hands the ring slot itself to the caller — *matrices
 * points at room for I x B matrices in `layout`, *morph_weights (may be passed as NULL) at I x M floats, zero-filled — the caller (its
 * pose solve, its worker threads) writes them in place, and rz_commit_pose does what is left of rz_set_pose: no pack, no copy.
 *   RZ_POSE_WORLD16  64 B per bone: column-major float[16], math.ts's layout — what rz_set_pose takes
 *   RZ_POSE_ROWS12   48 B per bone: the four columns' x y z (c0.xyz c1.xyz c2.xyz c3.xyz), i.e. the matrix without its bottom row, which
 *                    for the affine matrices a pose solve produces is 0 0 0 1 and is written back as such on the device. Only for
 *                    poses of more than 256 KB (a crowd's: rz_pull_pose_kernel expands them); a smaller pose is read in place by its
 *                    frame and must be mapped as WORLD16 (RZ_ERR_UNSUPPORTED otherwise).
 * The pointers are valid until rz_commit_pose or the next pose call on the context (rz_set_pose* cancels a mapping); the memory is
 * pinned, cacheable host memory — plain stores. One mapping at a time per context (a fork has its own ring: two frames in flight map
 * alternately). A slot comes back to the caller kStageSlots (8) / 32 commits later, after the GPU is done reading it (polled here). */
enum { RZ_POSE_WORLD16 = 0, RZ_POSE_ROWS12 = 1 };
int rz_map_pose(rz_ctx *ctx, int layout, float **matrices, float **morph_weights);
int rz_commit_pose(rz_ctx *ctx);

/* ---- forward kinematics on the device (optional) ----
 * rz_upload_skeleton_topology hands over what Model.computeWorldMatrices (engine/src/model.ts:330-420) reads from
 * the skeleton: parents[B] (-1 = root), bind_translation[B*3] (Bone.bindTranslation), and the append data of
 * model.ts:355-393: append_parent[B] (-1 or NULL = no append rotation), append_ratio[B] (NULL = all 1) and
 * append_move[B] (NULL = none; non-zero = the append parent's local translation * ratio is appended as well).
 * rz_set_pose_local then replaces rz_set_pose: it uploads local rotations (I x B x 4, x y z w — the
 * SkeletonRuntime.localRotations array, model.ts:55) and, optionally, local translations (I x B x 3 —
 * SkeletonRuntime.localTranslations, model.ts:56; NULL = all zero, which is all the reference ever has: nothing
 * writes that array there, VMD bone translations are this build's row f2) instead of world matrices, and the frame
 * computes L = T(bind + t) * R * T(add), W = W_parent * L and the palette on the GPU (f32; the host solves in
 * doubles with f32 stores, so results agree to ~1e-6, not bit for bit; the device resolves the parent chains by pointer
 * doubling, so products are associated as (L0 L1)(L2 L3) rather than ((L0 L1) L2) L3). 4x less per-frame upload; hierarchy
 * solve for all instances in one launch; one character's local pose is zero-copy like a world pose (resident after its first
 * frame, prefetched by the helper workgroup). */
int rz_upload_skeleton_topology(rz_ctx *ctx, uint32_t B, const int32_t *parents, const float *bind_translation3,
                                const int32_t *append_parent, const float *append_ratio, const uint8_t *append_move);
int rz_set_pose_local(rz_ctx *ctx, const float *local_rotations4, const float *local_translations3, const float *morph_weights);
/* PMX bone morphs (morph type 2; SURVEY §8f rank 3 "then 2 bone-morph"). The reference's loader only skips the section
 * (engine/src/pmx-loader.ts:489-497), so the semantics are this build's, the usual MMD ones: a morph with weight w adds
 * w * translation to the bone's local translation and right-multiplies its local rotation by slerp(identity, rotation, w)
 * (Quat.slerp math.ts:156-189, Quat.multiply math.ts:77-85), entries of one bone folded in ascending morph order, BEFORE
 * append rotation / the hierarchy solve (so an append child follows a morphed append parent). n entries, each naming a morph
 * of the uploaded morph set (rz_upload_morphs_* first; a model whose only morphs are bone morphs uploads an empty sparse
 * set) and a bone; the weights are the pose's morph weights (rz_set_pose_local's morph_weights, or the sampled tracks of
 * rz_set_pose_sampled — group morphs feed bone morphs like vertex morphs). Acts on device-solved poses only: with
 * rz_set_pose the host owns the world matrices and folds bone morphs itself (host/model.js). n = 0 clears. Dropped by a
 * new skeleton or morph set. */
int rz_upload_bone_morphs(rz_ctx *ctx, uint32_t n, const uint32_t *morph, const uint32_t *bone, const float *translation3,
                          const float *rotation4);

/* ---- motion sampling on the device (SURVEY §8f ranks 1 + 2 combined; optional) ----
 * The caller of the path in the reference is Engine.playAnimation (engine/src/engine.ts:1515-1553) fed by
 * VMDLoader (engine/src/vmd-loader.ts:95-160, which drops the key's position and interpolation bytes, :129-140).
 * rz_upload_animation hands a whole flattened motion to the GPU once; rz_set_pose_sampled then replaces
 * rz_set_pose_local: per frame the host sends ONE float per instance — the (fractional, 30 fps) frame that instance
 * is posed at — and the frame samples every bone (slerp warped by the later key's R Bezier curve, per-axis lerp
 * warped by the X / Y / Z curves) and every vertex morph (linear keys; group-morph tracks feed their children by
 * ratio) on the device, then solves the hierarchy (needs rz_upload_skeleton_topology) and deforms. Bones and morphs
 * the motion does not key are at rest. Same arithmetic as host/vmd-sampler.js in f32.
 *   bone tracks : track_bone[n] (bone index, each bone at most once), key_off[n+1], and per key (ascending frame
 *                 inside a track) key_frame, key_rot4 (x y z w), key_pos3, key_interp16 (the first 16 of the 64
 *                 interpolation bytes: X_x1 Y_x1 Z_x1 R_x1 | ..y1 | ..x2 | ..y2; NULL = linear)
 *   morph tracks: mkey_off[m+1], mkey_frame, mkey_weight; feeds per VERTEX MORPH of the uploaded morph set:
 *                 feed_off[M+1], feed_track, feed_ratio — its own track (ratio 1) first, then the group-morph
 *                 tracks that include it, ascending (the order Model.getEffectiveMorphWeights adds them in). */
typedef struct rz_animation {
    uint32_t n_bone_tracks;
    const int32_t *track_bone;
    const uint32_t *key_off;
    const float *key_frame;
    const float *key_rot4;
    const float *key_pos3;
    const uint8_t *key_interp16;
    uint32_t n_morph_tracks;
    const uint32_t *mkey_off;
    const float *mkey_frame;
    const float *mkey_weight;
    const uint32_t *feed_off;
    const int32_t *feed_track;
    const float *feed_ratio;
} rz_animation;
int rz_upload_animation(rz_ctx *ctx, const rz_animation *anim);
int rz_set_pose_sampled(rz_ctx *ctx, const float *frames);
/* ---- physics hand-off for device-solved poses ----
 * updateModelPose()  engine/src/engine.ts:2375-2381: between evaluatePose() and the world-matrix upload the reference
 * lets physics overwrite the world matrices of rigid-body-driven bones IN PLACE (physics.ts:715-751,
 * boneWorldMatrices.set(values, boneIndex * 16)): children keep the matrices solved from the un-overridden parent.
 * With rz_set_pose the host owns the world matrices and does exactly that before the call. When the hierarchy is
 * solved on the GPU (rz_set_pose_local / rz_set_pose_sampled) this entry point is the same hook: `n` world matrices
 * (column-major 4x4, affine) replace the solved ones of (instance[k], bone[k]) after the hierarchy solve and before
 * the palette product, in every following frame until the next call; n = 0 clears. instance = NULL means instance 0.
 * Several entries for one bone: the last wins (successive set() calls). Asynchronous H2D through pinned staging.
 * Physics itself (Bullet via @fred3d/ammo) stays out of scope: this only carries its result. */
int rz_override_world(rz_ctx *ctx, uint32_t n, const uint32_t *instance, const uint32_t *bone, const float *world16);
/* Blocking readback of one instance's world matrices (B x 16, column-major) as the frame used them. */
int rz_read_world(rz_ctx *ctx, uint32_t instance, float *world16);

/* computeSkinMatrices() dispatch  engine/src/engine.ts:2393-2402 (WGSL :906-930) followed by the
 * per-vertex body of vs()  engine/src/engine.ts:253-272 (copies :440-443/:700-703), executed
 * once per frame instead of once per draw pass. Enqueues the frame; asynchronous. */
int rz_deform(rz_ctx *ctx);
/* Enqueue `frames` consecutive frames of the resident pose without returning to the host
 * in between (steady-state replay; same kernels as rz_deform). Asynchronous. */
int rz_deform_n(rz_ctx *ctx, uint32_t frames);

/* ---- frames in flight ----
 * WebGPU queues frames: while the GPU drains frame f the application is already encoding frame f + 1 into its own command
 * buffer and the renderer reads frame f's output. A context has ONE compute stream and ONE set of output buffers, so its
 * frames run strictly one after the other and every frame pays its own launch ramp and tail — 1 us of a 16 us frame on a
 * 1/8 shard of C5, 2 us of a 6 us frame on a 30 k-vertex character. rz_fork makes a second context on the same GPU that
 * BORROWS every static device buffer of `ctx` (mesh, skeleton, topology, morph targets, bone morphs, motion, edge scale — no
 * copy, no extra HBM) and owns everything per-frame (streams, pose slots, palettes, outputs, tuning state copied from `ctx`
 * at fork time). Alternate frames between the two (rz_set_pose* + rz_deform on one while the other's frame is in flight, or
 * rz_deform_pair for a replay) and the tail of frame f overlaps the ramp of frame f + 1: measured 16.3 -> 15.2 us per frame
 * on the 1/8 shard, 6.3 -> 4.4 us on the character (tools/archive/shard_two_streams.py). While forks exist, static uploads fail on
 * both sides with RZ_ERR_INVALID; destroy the forks before the context they were forked from (rz_destroy refuses otherwise).
 * A fork cannot be forked and takes no part in gathers. */
int rz_fork(rz_ctx *ctx, rz_ctx **fork_out);
/* `frames` frames of the resident poses, alternating a, b, a, b ... (each on its own stream, into its own outputs). */
int rz_deform_pair(rz_ctx *a, rz_ctx *b, uint32_t frames);

int rz_sync(rz_ctx *ctx);

/* Blocking readback of deformed positions / normals of instance `instance`, vertices
 * [v0, v0+n) of this shard, packed [n][3]. Either pointer may be NULL. (getDeformed()) */
int rz_read(rz_ctx *ctx, uint32_t instance, uint32_t v0, uint32_t n, float *pos3, float *nrm3);
/* Blocking readback of the skin-matrix palette of one instance as B x 12 floats (rows 0..2 of
 * world*inverseBind, row-major 3x4) — the skinMatrixBuffer of engine.ts:1770-1774. */
int rz_read_palette(rz_ctx *ctx, uint32_t instance, float *rows3x4);

/* ---- fused consumers of the deformed mesh (optional) ----
 * rz_upload_edge_scale: per-vertex outline edge size (the material's edgeSize of the draw call that owns the
 * vertex, engine/src/engine.ts:415-421). Once set, every frame also writes the outline pass's inverted hull
 * expandedPos = worldPos + worldNormal * edgeSize * 0.01 (engine.ts:458-461, vertex shader :431-463) so the
 * outline pipeline no longer re-skins; NULL turns it off. rz_read_hull reads it back.
 * rz_enable_aabb: every frame also reduces the axis-aligned bounding box of the deformed positions of each
 * instance inside the skin kernel (no extra pass over the mesh); rz_read_aabb returns min xyz, max xyz of the
 * most recent frame. */
int rz_upload_edge_scale(rz_ctx *ctx, uint32_t V, const float *edge_size);
int rz_read_hull(rz_ctx *ctx, uint32_t instance, uint32_t v0, uint32_t n, float *pos3);
int rz_enable_aabb(rz_ctx *ctx, int enable);
int rz_read_aabb(rz_ctx *ctx, uint32_t instance, float min_max6[6]);

/* Benchmark helper: enqueue `frames` back-to-back frames of the resident pose between two HIP
 * events on the context's stream, then (mode 1) time the deform kernel alone and (mode 2) the
 * prep kernel alone the same way. Blocking. */
int rz_time_frames(rz_ctx *ctx, uint32_t frames, rz_timing *out);

/* Benchmark helper ("hipEvent pairs around >= 200 back-to-back frames after 20 warm-up frames"): `lead` untimed frames, then
 * `frames` timed frames of the resident pose exactly as rz_deform_n enqueues them — or, with a fork in `b`, as rz_deform_pair alternates
 * them — the timed ones between two events on the context's stream; blocks until they have drained and returns the span between the
 * events in ms. The lead frames sit in front of the opening event in the SAME call, so the timed frames are enqueued behind a busy GPU and
 * run back to back from the first one (lead = 0: the first timed frame starts on an idle GPU and waits ~5 us for its own launch). The
 * host's clock around the same calls adds a fixed ~30 us per timed region (first launch + waking up from the wait): 9 % of 20 steps of a
 * 16.6 us frame. b = NULL: one stream. */
int rz_time_span(rz_ctx *a, rz_ctx *b, uint32_t lead, uint32_t frames, double *span_ms);

/* Tuning knobs (bench sweeps / tests); 0 / -1 = automatic. Keys: "morph_split" (0,1,2,4,8 lanes
 * per vertex quad; without dense targets it only sets the wave step: 1 -> 256 vertices, >= 4 -> 64), "unroll" (0,4,8 morphs in flight per lane), "grid_cap" (total workgroups),
 * "geo_lds" (0/1: rest geometry transposed through LDS vs 4-byte loads), "nontemporal" (0/1,
 * morph-stream loads), "nt_store" (0/1, output stores), "fast" (-1 auto, 0 always run the
 * separate prep kernel, 1 one-launch frame when possible), "out_cap" (-1 auto, 0 off, else vertices a wave
 * parks in LDS before writing them out in one burst), "inst_loop" (-1 auto, 0 off, 2..8 / 10..64 poses
 * per workgroup in instanced morph-free frames, 9 = the register-resident form), "pose_prefetch" (-1 auto / 1: the first frame of a zero-copy pose — world matrices, or local rotations
 * [+ translations] of a single character whose hierarchy is solved inside the deform kernel — carries a helper workgroup that stages the
 * NEXT pose into device memory when the host has already written it — a per-frame loop whose host runs ahead of the GPU then never
 * pays the PCIe round trip; a staged pose is only ever taken by a frame of the same pose kind;
 * 0: off; rz_get_tuning("pose_staged") tells whether the current pose was staged that way),
 * "inst_subsets" (-1 auto / 1: a crowd workgroup stages
 * only the bones its vertex run names when that list is shorter than the skeleton — same bits, a fraction of the LDS and of the
 * per-workgroup front; 0: always the whole palette), "inst_block" (0 auto, 256 / 512 / 1024 threads per
 * workgroup of the instanced kernel; for crowds "fast" -1 / 1 = palettes formed inside the skin kernel, one launch per frame, 0 =
 * rz_prep_kernel in front), "inst_order" (1 default / 0: which workgroups of the instanced kernel an XCD gets — 1: every vertex run
 * of ITS pose groups, so an XCD's L2 pulls one eighth of the poses' matrices; 0: one vertex run of every pose group), "graph" (0/1: rz_deform_n replays hipGraphs of 16 captured frames instead of launching every kernel —
 * for launch-bound replay of small frames), "zero_copy" (-1 auto = on, 0: every pose is copied to the device; one character's
 * per-frame inputs are otherwise read by the frame's kernels straight from a pinned, device-mapped slot), "fuse_fk" (-1 auto, 0, 1:
 * a device-animated single character solves its bone hierarchy inside the deform kernel — one launch per frame — instead of
 * rz_fk_kernel [+ rz_prep_kernel] in front), "fuse_fk_plain" (-1 auto / 1: the fused frame of a PLAIN pose — no physics overrides, <= 512 bones, <= 256 morphs —
 * runs a kernel variant whose hierarchy solve is specialised for an uploaded / a sampled pose; 0: always the generic solve; same bits;
 * rz_get_tuning("effective_fk_kind") says which: 0 generic, 1 uploaded, 2 sampled),
 * "pose_pull" (-1 auto: the world matrices of a pose of more than 256 KB — a crowd's — are pulled out of their pinned ring slot by a
 * kernel on the upload stream, as the upper three rows of every affine matrix [a pose with any other bottom row travels whole]; local
 * rotations, a quarter of the bytes, stay with the copy engine, which does not disturb the frame they run under; 1: every such pose
 * is pulled; 0: the runtime copies every pose as it was handed over; rz_get_tuning("pose_pulled") / ("pose_rows") tell what the
 * last upload did),
 * "overlap" (-1 / 0 off, 1: crowds run their front kernels on the upload stream under
 * the previous frame's skin kernel — measured slower on this runtime, kept for experiments). There is no key that makes a frame
 * emit anything but the deformed mesh: ablation switches exist only in a tools-only build and "dbg" is rejected here.
 * rz_get_tuning also answers "effective_split" / "effective_unroll" / "effective_grid" / "effective_fast" / "effective_out_cap" /
 * "effective_inst_group" / "effective_inst_block" / "effective_fuse_fk" / "effective_overlap" / "pose_resident" /
 * "effective_subsets" / "effective_subset_bones" / "effective_inst_lds" / "effective_fk_kind" / "effective_variant" (the last template
 * argument of the single-mesh frame kernel: 0 everything compiled in, 3 without the fused consumers, 1 / 2 also with the specialised
 * hierarchy solve) and the counts
 * "verts" / "bones" / "morphs" / "instances". Unknown keys return RZ_ERR_INVALID.
 * NOT a pure getter for crowds: an "effective_*" key describes the frame the NEXT rz_deform will launch, and a crowd's plan depends on
 * the per-run bone lists of its launch shape — when the shape, the mesh or the skeleton changed since the last frame the call brings
 * them up to date first, exactly as the next frame would (stream drained, one small kernel, one readback, a captured graph dropped).
 * Poll these keys at setup time or after a frame, not between a shape change and the frame in a latency-critical loop. (An unknown
 * "effective_*" key is refused before any of that work.) */
int rz_set_tuning(rz_ctx *ctx, const char *key, int value);
int rz_get_tuning(rz_ctx *ctx, const char *key, int *value);

/* Setup-time search over launch shapes (like a GEMM library's "find" mode; no reference counterpart — WebGPU hides
 * the dispatch shape, engine.ts:2393-2402). Candidates: the built-in heuristic plan (entry 0) and every distinct plan among morph
 * split {1,2,4,8} x {1,2,4} workgroups per CU (instanced frames: poses per workgroup x workgroups per CU), timed with the
 * CURRENT mesh / morphs / pose on this GPU: `frames` frames per round (0 = 100, at most 1000) after ~0.25 s of untimed frames (clocks), 5 rounds dealt round-robin over
 * the candidates (so clock / thermal drift hits all of them alike), the MEDIAN round of each is its time.
 *   rz_autotune_measure  fills `table` (at most `cap` entries, *count = how many) and changes nothing;
 *   rz_autotune_apply    adopts one entry as the "morph_split" / "grid_cap" / "inst_loop" tuning values;
 *   rz_autotune          = measure + rz_autotune_pick + apply. The heuristic plan is KEPT unless a candidate is clearly faster — median
 *                        >= 2 % lower and its slowest round under the heuristic's fastest round: the landscape is flat near the
 *                        optimum and a search that follows noise returns a different plan on every run.
 * Several GPUs deforming shards of one mesh (one process per GPU) take the element-wise MAX of their tables' `ms` over the
 * ranks, then every rank picks from that one table with rz_autotune_pick and applies the same entry — one plan on all ranks,
 * judged by the slowest GPU, which is what the frame time is (bench.py does this over torch.distributed).
 * Every candidate is a plan the parity tests cover. The result belongs to the workload it was timed on: uploading
 * another mesh or other morph targets, or changing the instance count, returns the three keys to their heuristics, and
 * so does rz_set_tuning(key, 0 / -1). */
typedef struct rz_tune_entry {
    int morph_split, grid_cap, inst_loop;   /* the request (rz_set_tuning values; 0 / 0 / -1 = the heuristics) */
    int eff_split, eff_grid, eff_inst_group;/* what it resolves to on this context */
    int same_as;                            /* index of an earlier entry that resolves to the same launch (its time is shared), or -1 */
    float ms;                               /* median round, ms per frame */
    float ms_min, ms_max;                   /* fastest / slowest round */
} rz_tune_entry;
int rz_autotune_measure(rz_ctx *ctx, uint32_t frames, rz_tune_entry *table, int cap, int *count);
/* index of the entry to adopt under the stability rule above (entry 0 unless some entry's ms < 0.98 x entry 0's and, when the
 * spreads are filled in, its ms_max < entry 0's ms_min; ranks that MAX-reduce `ms` reduce ms_max by MAX and ms_min by MIN) */
int rz_autotune_pick(const rz_tune_entry *table, int count);
int rz_autotune_apply(rz_ctx *ctx, const rz_tune_entry *entry);
int rz_autotune(rz_ctx *ctx, uint32_t frames);

/* Device pointers of the output buffers ([I][Vpad][3] floats each) and the padded vertex count,
 * so a host framework can wrap them without a copy. */
int rz_output_ptrs(rz_ctx *ctx, void **pos, void **nrm, uint32_t *v_padded);

/* ---- multi-GPU: vertex shards + optional all-gather of deformed positions ----
 * One context per GPU / process. rz_comm_unique_id() is called by rank 0 and the 128 bytes are
 * broadcast by the host (IPC, torch.distributed, files ...); every rank then calls
 * rz_comm_init() with the same id. rz_allgather() runs ncclAllGather over xGMI on the
 * context's stream: every rank ends up with the full [v_total_padded][3] position (and, when
 * with_normals != 0, normal) arrays of instance 0. */
int rz_comm_unique_id(char id[128]);
/* Which RCCL this library is bound to (lazily, on the first multi-GPU call): the file it came from, ncclGetVersion(),
 * and whether it is a copy the process had ALREADY loaded (PyTorch ships its own librccl.so with the same soname): the
 * library looks for one with RTLD_NOLOAD first, so that a process never runs two RCCL runtimes side by side. */
int rz_rccl_info(char *path, size_t path_bytes, int *version, int *reused);
/* What the communicator itself says after rz_comm_init / rz_comm_init_all: ncclCommCount and ncclCommUserRank
 * (evidence that RCCL really spans `nranks` ranks — bench.py prints it per rank). */
int rz_comm_info(rz_ctx *ctx, int *comm_count, int *comm_user_rank);
int rz_comm_init(rz_ctx *ctx, int nranks, int rank, const char id[128], uint32_t v_total);
int rz_allgather(rz_ctx *ctx, int with_normals);
int rz_read_gathered(rz_ctx *ctx, uint32_t v0, uint32_t n, float *pos3, float *nrm3);

/* Single-process form (one Node process driving several GPUs, one context each): ctxs[r] is rank r and
 * must hold shard r of rz_shard_range(v_total, n, r). rz_comm_init_all = ncclCommInitAll; rz_allgather_all
 * issues every rank's ncclAllGather inside one ncclGroupStart/End. */
int rz_comm_init_all(rz_ctx **ctxs, int n, uint32_t v_total);
int rz_allgather_all(rz_ctx **ctxs, int n, int with_normals);

/* Peer-direct gather ("peer-direct stores replacing the all-gather"), single-process form.
 * The reference has one consumer of the deformed mesh — the rasteriser behind vs() (engine.ts:245-276) — so only ONE
 * GPU (`root`, an index into ctxs) needs the whole mesh. rz_gather_direct allocates the [n x chunk][3] position and
 * normal arrays on the root's GPU, enables peer access and re-points every context's output at its own slice of
 * them: from then on each rz_deform stores its shard straight into the root's memory over xGMI while it computes —
 * no collective, no second pass over the output, no RCCL. ctxs[r] must hold shard r of rz_shard_range(v_total, n, r);
 * contexts may share a GPU (then no peer mapping is involved). rz_read() on a contributor still returns its shard.
 * rz_gather_fence(root) makes the root's stream wait (hipStreamWaitEvent, nothing blocks on the host) for
 * everything the other contexts have enqueued so far, so a consumer enqueued on the root's stream sees the whole
 * frame; rz_read_gathered(root, ...) fences, synchronises and copies to the host. Uploading a new mesh to a
 * contributor returns that context to its private output buffers; uploading one to the root, or destroying the
 * root, returns all of them. */
int rz_gather_direct(rz_ctx **ctxs, int n, uint32_t v_total, int root);
int rz_gather_fence(rz_ctx *root);

#ifdef __cplusplus
}
#endif
#endif /* REZE_DEFORM_H */
