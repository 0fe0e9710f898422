// deform_kernels.h — parameter blocks and launchers shared by the kernel files (kernels/*.hip) and the host side (ctx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// rz_prep_kernel: palette = rows 0..2 of world * inverseBind (engine/src/engine.ts:926-928) and the
// ordered list of morphs with a non-zero weight.
struct RzPrepParams {
    const float *world;      // [I][B][16] column-major
    const float *inv_bind;   // [B][16]
    float4 *palette;         // [I][B][3]  row-major 3x4
    const float *morph_w;    // [I][M]
    uint32_t *act_idx;       // [I][Mpad]
    float *act_w;            // [I][Mpad]
    int *act_count;          // [I]
    int B;
    int M;
    int Mpad;
};

// rz_fk_kernel: forward kinematics + palette on the device (engine/src/model.ts:330-420, engine.ts:926-928).
// motion sampling inside rz_fk_kernel: frame-indexed VMD sampling on the device (the GPU twin of host/vmd-sampler.js; the reference has no
// counterpart — its loader drops positions and interpolation bytes, engine/src/vmd-loader.ts:129-140).
struct RzSampleParams {
    const float *frames;          // [I] fractional frame (30 fps) each instance is posed at; nullptr = no sampling
                                  // (the track record of every bone is word 2 of its bone record: RzFkParams::bone_rec)
    const float *key_frame;       // [K] ascending inside a track
    const float4 *key_rot;        // [K]
    const float *key_pos;         // [K][3]
    const uint4 *key_interp;      // [K] first 16 interpolation bytes of each key (nullptr = linear)

    const float *mkey_frame;      // [Km]
    const float *mkey_weight;     // [Km]
    const uint32_t *feed_off;     // [M + 1] per vertex morph: the tracks that feed it ...
    const uint4 *feed_range;      //   ... as (first key, end, bits(first frame), bits(last frame)) into mkey_*: own track first, then group-morph tracks ascending
    const float *feed_ratio;
    float *morph_w;               // [I][M] out
    int M;
    int frames_inline;            // 1: every instance is posed at `frame0` (one character: the frame rides in the kernel arguments)
    float frame0;
};

struct RzFkParams {
    const float4 *local_q;      // [I][B] local rotations (x,y,z,w)
    const float *local_t;       // [I][B][3] local translations (SkeletonRuntime.localTranslations) or nullptr = all zero
    // static data of the solve, ONE block per skeleton (kernels/fk.hip.h): 64-byte bone records
    //   w0: parent (-1 = root) | append parent (-1 = no append rotation) | bits(append ratio) | flags (bit 0: append-move —
    //       the append parent's local translation * ratio is appended too, model.ts:388-393)
    //   w1: bits of the parent-relative bind translation x y z | 0
    //   w2: the motion's track of the bone (first key, end, bits(first frame), bits(last frame)); zeros without a motion
    //   w3: ancestors for the radix-4 doubling rounds 0 and 1 (a1 | a2 << 16, a3, a4 | a8 << 16, a12; 0xffff = above the root)
    // followed by [sample.M] 32-byte vertex-morph records (first feed's key range | first feed, end of feeds, bits(ratio), 0)
    const uint4 *bone_rec;      // [B][4] (+ [M][2])
    const uint2 *anc_more;      // [n_rounds - 2][B] (a1 | a2 << 16, a3) for the doubling rounds beyond the second (hierarchies deeper than 16), or null
    const float *inv_bind;      // [B][16]
    float *world;               // [I][B][16] out
    float4 *palette;            // [I][B][3] out
    int B;
    int n_rounds;               // radix-4 doubling rounds = ceil(log4(depth of the hierarchy)); 0 = roots only
    // physics hand-off (engine.ts:2379-2381, physics.ts:715-751): world matrices that replace the solved ones AFTER the
    // hierarchy solve — children keep the matrices solved from the un-overridden parent, exactly like the reference's
    // in-place boneWorldMatrices.set(). Entries sorted by instance; ovr_off[i]..ovr_off[i+1] are instance i's.
    const int *ovr_off;         // [I + 1] or nullptr
    const int *ovr_bone;        // [n]
    const float *ovr_world;     // [n][16] column-major
    // PMX bone morphs (morph type 2: a weight-scaled translation + rotation offset on a bone's LOCAL pose; the reference's
    // loader skips the section, engine/src/pmx-loader.ts:489-497). Stored per bone; applied after the pose is staged /
    // sampled and before the local matrices are formed:  t += w * t_e ;  q = q * slerp(I, q_e, w)  for its entries in
    // ascending morph order (host twin: host/model.js posedLocals()).
    const uint32_t *bm_off;     // [B + 1] or nullptr = the model has no bone morphs
    const uint32_t *bm_morph;   // [n] morph index of each entry, ascending inside a bone
    const float4 *bm_rot;       // [n] x y z w
    const float4 *bm_tr;        // [n] x y z (w unused)
    const float *bm_w;          // [I][bm_M] morph weights of an uploaded (not sampled) pose
    int bm_M;
    RzSampleParams sample;      // sample.frames != nullptr: local_q / local_t are ignored, the pose is sampled in the kernel
    // FUSED frame of a zero-copy local pose (one character; RzDeformParams: pf_* / st_tag): `local_q` / `local_t` name the pinned slot.
    float4 *copy_q;             // device pose block: workgroup 0 leaves the pose there for the frames that replay it, or null
    float *copy_t;
    const float4 *st_local_q;   // where the previous frame's helper staged this pose if RzDeformParams::st_tag holds st_expect, or null
    const float *st_local_t;
    uint64_t st_expect;
};

// rz_deform_kernel: fused morph + 4-bone LBS (engine/src/engine.ts:253-272).
struct RzDeformParams {
    const float *geom;          // 6 planes of Vp floats: x y z nx ny nz
    const uint32_t *joints01;   // [Vp] j0 | j1 << 16
    const uint32_t *joints23;   // [Vp] j2 | j3 << 16
    const uint32_t *weights;    // [Vp] 4 x unorm8
    float4 *palette;            // [I][B][3]  (read by !FAST, written once per frame by FAST)
    const float *world;         // [I][B][16] (FAST)
    const float *inv_bind;      // [B][16]    (FAST)
    const float *dense;         // [M][3][Vp] planes (MODE 1)
    const uint32_t *act_idx;    // [I][Mpad]
    const float *act_w;         // [I][Mpad]
    const int *act_count;       // [I]
    const float *morph_w;       // [I][M]   (MODE 2)
    // Zero-copy first frame (single character): `world` / `morph_w` point into PINNED HOST memory (the staging slot the
    // host has just written) and workgroup 0 of the FAST kernel leaves a copy in device memory for the frames that replay
    // this pose. null = the pose is already resident and `world` / `morph_w` are device pointers.
    float *world_copy;          // [B][16] device destination, or null
    float *morph_w_copy;        // [M]     device destination (MODE 2), or null
    // Pose PREFETCH of zero-copy frames (FAST, one character). A frame whose pose still sits in its pinned slot
    // pays one PCIe round trip (2-3 us) that a 16 us shard frame cannot hide. So the frame of pose u also carries ONE helper
    // workgroup (blockIdx.x == 0; the workers shift by one) that looks at the ring slot the NEXT upload will use: if the host
    // has already written pose u + 1 there — it runs several frames ahead of the GPU in a per-frame loop — the helper copies
    // it into the device pose block that upload will name and tags it with the upload's sequence number. The frame of pose
    // u + 1 checks the tag (one scalar load): staged -> it reads HBM like a resident pose; not staged (the host was late, the
    // slot was not written yet, another pose kind came) -> it reads the pinned slot exactly as before. No event, no wait,
    // nothing on the host; a miss costs nothing but the helper's slot.
    const float *pf_src;            // pinned slot of the next upload (device-mapped), or null = no helper in this launch
    const uint64_t *pf_src_seq;     // its header: the sequence number of the pose it holds, written by the host AFTER the pose
    float *pf_dst;                  // device pose block the next upload will name ([world | weights], the slot's layout)
    uint64_t *pf_tag;               // device: sequence number of the pose staged in pf_dst
    uint64_t pf_expect;             // header value of the completely written next pose
    uint32_t pf_bytes;              // bytes to stage (multiple of 16)
    const uint64_t *st_tag;         // THIS frame's pose: staged in the device block when *st_tag == st_expect, else in the pinned slot
    uint64_t st_expect;
    const float *st_world;          // [B][16] staged copy
    const float *st_morph_w;        // [M]     staged copy (MODE 2; fused-hierarchy frames: every mode)
    // FUSED single-character frame (fk_on): every workgroup of the (!FAST) kernel first solves the bone hierarchy itself —
    // motion sampling included when the pose is sampled — straight into its LDS palette, and compacts the morph weights
    // into its LDS list: no rz_fk_kernel, no rz_prep_kernel, ONE launch per device-animated frame. Workgroup 0 also
    // leaves world matrices, palette and sampled weights in memory (rz_read_world / rz_read_palette).
    RzFkParams fk;
    int fk_on;
    int fk_kind;                // 0: the generic solve; 1 / 2: the kernel variant specialised for a PLAIN uploaded / sampled pose (kernels/fk.hip.h: fk_solve<FUSED, KIND>)
    const uint32_t *sp_ptr;     // [Vp+1]   (MODE 2) per-vertex CSR row pointers
    const float4 *sp_entries;   // [E]      (dx,dy,dz,bits(morph))
    float *out_pos;             // [I][Vp][3]
    float *out_nrm;             // [I][Vp][3]
    const float *edge;          // [Vp] per-vertex outline edge size, or null (hull epilogue off)
    float *out_hull;            // [I][Vp][3] worldPos + worldNormal * edge * 0.01   (engine.ts:458-461)
    uint32_t *aabb;             // [I][2 slots][6] order-preserving keys of min xyz / max xyz, or null
    int aabb_slot;              // slot this launch accumulates into (the other one is re-armed)
    uint32_t Vp;                // padded vertex count (multiple of 1024)
    uint32_t n_verts;           // real vertex count V
    uint32_t n_quads;           // ceil(V / 4)
    uint32_t quads_per_wave;    // contiguous run owned by each wave of the grid (multiple of 8)
    uint32_t out_cap;           // vertices buffered in LDS per wave before a 16-B/lane flush (0 = store directly)
    uint32_t sp_cap;            // MODE 2: CSR entries a wave stages in LDS at a time (multiple of 64 = whole 1 KiB bursts)
#ifdef RZ_ABLATE
    int dbg;                    // ablation switch — tools-only build (see RZ_DBG in deform_kernels.hip); absent from the product
    unsigned long long *tl;     // tools-only build, dbg = 100: per-wave timeline, 8 x u64 per wave (RZ_STAMP in deform_kernels.hip)
#endif
    int inst_order;             // instanced skin: 0 = an XCD takes one vertex run of every pose group, 1 = every vertex run of its pose groups
    // Instanced skin, BONE-SUBSET form: a vertex run only references a few of the skeleton's bones (PMX meshes are bone-local),
    // so its workgroups stage just those — rz_run_subsets_kernel lists, per vertex run, the sorted set of bones its vertices
    // name and rewrites the joints as slots of that list. null = the whole palette is staged and joints01 / joints23 are used.
    const uint32_t *rj01;       // [Vp] slot(j0) | slot(j1) << 16   (joints already clamped to B - 1)
    const uint32_t *rj23;       // [Vp] slot(j2) | slot(j3) << 16
    const uint16_t *sub_list;   // [runs][sub_stride] ascending bone indices of each run's subset
    const uint32_t *sub_count;  // [runs]
    int sub_stride;
    int sub_max;                // longest list of the launch shape: every list is padded to it with bone 0
    int dma;                    // FAST: stage raw matrices by LDS-DMA (else plain loads after the first morph phase)
    int B;
    int M;
    int Mpad;
};

// Active-morph list of a single-instance frame, passed BY VALUE in the kernel arguments (FAST path):
// compacted on the host by rz_set_pose, so the frame needs no prep kernel and no dependent load
// before the morph stream starts.
constexpr int kKargMorphs = 128;
constexpr int kKargPad = 8;          // the remainder loop may peek up to S-1 entries past `count`
struct RzMorphList {
    int count;
    uint32_t idx[kKargMorphs + kKargPad];
    float w[kKargMorphs + kKargPad];
};

// Compile-time variant selection of the single-mesh frame kernels (kernels/deform_dense.hip, kernels/deform_small.hip).
struct RzVariant {
    int mode;    // 0 none, 1 dense, 2 sparse
    int S;       // morph split 1,2,4,8 (mode 1 only)
    int U;       // 4 or 8
    bool nt;     // nontemporal morph loads
    bool nts;    // nontemporal output stores
    bool geo;    // rest geometry through LDS
    bool fast;   // fused palette + kernarg morph list (single instance)
};

// LDS the hierarchy solve needs behind the palette rows: per bone 48 B (local rotation | record | bind translation, then the
// second matrix buffer of the doubling rounds) + 12 B (local translation), 16-byte rounded
__host__ __device__ inline size_t rz_fk_scratch_bytes(int B) { return ((size_t)B * 60 + 15) & ~(size_t)15; }

hipError_t rz_launch_prep(const RzPrepParams &p, uint32_t instances, hipStream_t st);
hipError_t rz_launch_fk(const RzFkParams &p, uint32_t instances, hipStream_t st);
size_t rz_fk_lds_bytes(const RzFkParams &p);      // dynamic LDS of rz_fk_kernel (palette rows + solve scratch + the pose's morph weights)
hipError_t rz_launch_deform(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, uint32_t grid_x,
                            uint32_t instances, hipStream_t st);
size_t rz_deform_lds_bytes(const RzDeformParams &p, const RzVariant &v);
// the two single-mesh frame kernels behind rz_launch_deform (kernels/deform_dense.hip: v.mode == 1; kernels/deform_small.hip: modes 0 / 2)
hipError_t rz_launch_deform_dense(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st);
hipError_t rz_launch_deform_small(const RzDeformParams &p, const RzVariant &v, dim3 grid, size_t lds, hipStream_t st);
// instanced skin (no morphs): G poses per workgroup share one decode of each vertex
// register-resident form: 2048 vertices per workgroup decoded once, poses streamed through a 2-deep LDS palette ring
hipError_t rz_launch_skin_instances_reg(const RzDeformParams &p, int n_inst, int poses_per_wg, uint32_t grid_x, bool nts,
                                        hipStream_t st);
hipError_t rz_launch_skin_instances(const RzDeformParams &p, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                    int block, bool nts, size_t lds_bytes, hipStream_t st);
// Crowd frame with the hierarchy solved in the skin kernel's front (kernels/crowd.hip: rz_skin_instances_fk_kernel): per vertex run the
// closure of its named bones under "parent of", one 80-byte record per closure slot (plan.cpp: ensure_subfk)
struct RzSubFk {
    const uint32_t *count;      // [runs] closure slots of each run
    const uint4 *rec;           // [runs][stride][5]
    uint32_t stride;            // closure slots of the largest run
    uint32_t rounds;            // radix-4 doubling rounds the deepest closure needs (<= 3)
};
hipError_t rz_launch_skin_instances_fk(const RzDeformParams &p, const RzSubFk &f, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                       int block, bool nts, size_t lds_bytes, hipStream_t st);
size_t rz_skin_instances_fk_lds_bytes(int G, uint32_t closure, uint32_t named);
// LDS bytes of the instanced skin kernel: G poses of `bones` staged bones (the whole skeleton, or the largest run subset)
size_t rz_skin_instances_lds_bytes(int G, uint32_t bones, bool dma, bool subsets);
// per vertex run of `per` vertices: the ascending list of bones its vertices name + the joints rewritten as slots of it
hipError_t rz_launch_run_subsets(const uint32_t *j01, const uint32_t *j23, uint32_t v_lim, uint32_t per, uint32_t runs, uint32_t B,
                                 uint16_t *list, uint32_t *count, uint32_t *rj01, uint32_t *rj23, hipStream_t st);
// a crowd's per-frame pose pulled out of a pinned, device-mapped slot (kernels/front.hip: rz_pull_pose_kernel)
hipError_t rz_launch_pull_pose(const void *src, void *dst, uint32_t bones, size_t raw_bytes, hipStream_t st);
uint32_t rz_quads_per_tile(int S);
bool rz_has_all_variants();      // false in the product: only the variants a plan can select by default are compiled in
#ifdef RZ_ALL_VARIANTS
hipError_t rz_launch_gate(const uint32_t *flag, hipStream_t st);     // test hook (tools-only build): the stream waits until *flag != 0
#endif
hipError_t rz_launch_deinterleave(const float *src, int stride, int offset, uint32_t n, float *px, float *py,
                                  float *pz, hipStream_t st);
hipError_t rz_launch_pack_skinning(const uint16_t *joints4, const uint8_t *weights4, uint32_t n, uint32_t *j01,
                                   uint32_t *j23, uint32_t *wq, hipStream_t st);
