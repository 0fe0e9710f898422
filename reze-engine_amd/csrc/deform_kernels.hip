// deform_kernels.hip — hand-written CDNA4 (gfx950) kernels for the per-frame PMX morph + skin path.
//
// Replaces, in the reference (paths relative to the reference repo root):
//   rz_prep_kernel    the WGSL compute shader  engine/src/engine.ts:906-930
//                     (skinMatrices[b] = worldMatrices[b] * inverseBindMatrices[b]) — plus the
//                     active-morph compaction, which has no reference counterpart.
//   rz_deform_kernel  the skinning body of the WGSL vertex shader vs()  engine/src/engine.ts:253-272
//                     (and its copies :440-443, :700-703), run once per frame instead of once per
//                     draw pass, fused with vertex-morph accumulation (new capability; the reference
//                     skips PMX morphs, engine/src/pmx-loader.ts:450-553).
//
// Design (memory-bound gather-transform, no MFMA):
//   * static mesh is planar SoA in HBM: x[],y[],z[],nx[],ny[],nz[] float planes, joints as two
//     u32 planes (j0|j1<<16, j2|j3<<16), weights as one u32 plane, dense morph targets as
//     D[m][3][Vp] planes. A lane owns a QUAD of 4 consecutive vertices, so every stream is read
//     with one 16-byte load per lane = 1 KiB contiguous per wave-instruction.
//   * the per-instance bone palette (3x4 affine rows, 48 B/bone) and the active-morph list are
//     staged once per workgroup into LDS; bones are gathered from LDS with ds_read_b128.
//   * MORPH SPLIT S: S lanes of a wave cooperate on one quad; lane-slice s accumulates the active
//     morphs a = s, s+S, ... and the partial sums are combined with a __shfl_xor butterfly. S > 1
//     multiplies the number of waves in flight when the per-GPU shard is small (8-GPU strong
//     scaling); S = 1 is the pure streaming form for large shards.
//   * outputs are packed float3 arrays (the vertex-buffer layout a renderer binds).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "deform_kernels.h"

#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"   // 32-bit LDS pointers built from integers

namespace {

constexpr int kBlock = 256;

// Build-time switch (A/B builds: make flavor): bit log2(S) set = the kernel-argument morph list of the dense one-launch kernel with
// morph split S is read through SGPRs (readfirstlane) instead of the per-lane indexed load the compiler folds the selects into.
#ifndef RZ_PIN_ML
#define RZ_PIN_ML 0
#endif

// Ablation switches for profiling experiments exist only in the tools-only build (make ablate ->
// tools/ablate/libreze_deform_ablate.so, -DRZ_ABLATE). In the shipped library RZ_DBG is the constant 0, the branches
// fold away, and rz_set_tuning("dbg", ...) is rejected: no key can make rz_deform emit anything but the deformed mesh.
#ifdef RZ_ABLATE
#define RZ_DBG(p) ((p).dbg)
// Per-wave TIMELINE (tools-only build, rz_set_tuning dbg = 100; tools/timeline.py): a wave keeps up to seven readings of the
// chip-wide 100 MHz counter (s_memrealtime: 10 ns steps, the same clock on every XCD) in scalar registers and writes them out
// when it ends, with the XCC / CU / SIMD it ran on (16 x u64 per wave: 0..6 stamps, 7 where, 8..12 stamps inside the fused
// hierarchy solve). Slot 6 is taken after every store of the wave has been acknowledged.
#define RZ_TL_DECL unsigned long long tl_t[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tl_f[5] = {0ull, 0ull, 0ull, 0ull, 0ull}
#define RZ_STAMP(k) do { if (p.tl) tl_t[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RZ_TL_FLUSH(wave_index) do { if (p.tl) { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        tl_t[6] = __builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0) { \
            unsigned long long *o_ = p.tl + (size_t)(wave_index) * 16; \
            for (int k_ = 0; k_ < 7; ++k_) o_[k_] = tl_t[k_]; \
            for (int k_ = 0; k_ < 5; ++k_) o_[8 + k_] = tl_f[k_]; \
            o_[7] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) | \
                    ((unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) << 8); \
        } } } while (0)
#else
#define RZ_DBG(p) 0
#define RZ_TL_DECL do {} while (0)
#define RZ_STAMP(k) do {} while (0)
#define RZ_TL_FLUSH(wave_index) do {} while (0)
#endif

__device__ __forceinline__ float4 ld_stream(const float4 *p, bool nt)
{
    // morph planes are read exactly once per frame: optionally bypass-hint the load
    if (nt) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// ------------------------------------------------------------------------------------------------
// prep: palette rows + ordered compaction of the non-zero morph weights. One workgroup per instance.
// ------------------------------------------------------------------------------------------------
// Ordered compaction of the non-zero morph weights of one pose (whole workgroup of kBlock threads; `mw` may be LDS or global,
// `aidx` / `aw` likewise): entries keep ascending morph order = the oracle's accumulation order; the tail up to Mpad is
// zero-padded so unrolled readers may over-read harmlessly. Returns the number of active morphs (workgroup-uniform).
__device__ __forceinline__ int compact_active(const float *mw, const int M, const int Mpad, uint32_t *aidx, float *aw, int *wave_cnt)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int m0 = 0; m0 < M; m0 += kBlock) {
        const int m = m0 + tid;
        const float w = (m < M) ? mw[m] : 0.0f;
        const bool on = (w != 0.0f);
        const unsigned long long bal = __ballot(on);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) {
            const int c = wave_cnt[k];
            before += (k < wave) ? c : 0;
            total += c;
        }
        if (on) {
            aidx[base + before + rank] = (uint32_t)m;
            aw[base + before + rank] = w;
        }
        base += total;
        __syncthreads();
    }
    for (int k = base + tid; k < Mpad; k += kBlock) { aidx[k] = 0; aw[k] = 0.0f; }
    return base;
}

__global__ void __launch_bounds__(kBlock) rz_prep_kernel(RzPrepParams p)
{
    const int inst = blockIdx.x;
    const int tid = threadIdx.x;
    const float *world = p.world + (size_t)inst * p.B * 16;
    float4 *pal = p.palette + (size_t)inst * p.B * 3;

    for (int b = tid; b < p.B; b += kBlock) {
        const float4 *Wm = reinterpret_cast<const float4 *>(world + (size_t)b * 16);
        const float4 *Im = reinterpret_cast<const float4 *>(p.inv_bind + (size_t)b * 16);
        // column-major: a_k = column k of W (x,y,z = rows 0..2)
        float4 a0 = Wm[0], a1 = Wm[1], a2 = Wm[2], a3 = Wm[3];
        float r0[4], r1[4], r2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 bc = Im[c];
            // out[c*4+r] = ((a0[r]*b0 + a1[r]*b1) + a2[r]*b2) + a3[r]*b3   (engine.ts:928)
            r0[c] = fmaf(a3.x, bc.w, fmaf(a2.x, bc.z, fmaf(a1.x, bc.y, a0.x * bc.x)));
            r1[c] = fmaf(a3.y, bc.w, fmaf(a2.y, bc.z, fmaf(a1.y, bc.y, a0.y * bc.x)));
            r2[c] = fmaf(a3.z, bc.w, fmaf(a2.z, bc.z, fmaf(a1.z, bc.y, a0.z * bc.x)));
        }
        pal[b * 3 + 0] = make_float4(r0[0], r0[1], r0[2], r0[3]);
        pal[b * 3 + 1] = make_float4(r1[0], r1[1], r1[2], r1[3]);
        pal[b * 3 + 2] = make_float4(r2[0], r2[1], r2[2], r2[3]);
    }

    if (p.M > 0) {
        __shared__ int wave_cnt[kBlock / 64];
        const int n = compact_active(p.morph_w + (size_t)inst * p.M, p.M, p.Mpad, p.act_idx + (size_t)inst * p.Mpad, p.act_w + (size_t)inst * p.Mpad, wave_cnt);
        if (tid == 0) p.act_count[inst] = n;
    }
}

// ------------------------------------------------------------------------------------------------
// forward kinematics on the device: the reference's Model.computeWorldMatrices
// (engine/src/model.ts:330-420) for I poses at once, fused with the palette product (engine.ts:926-928).
// One workgroup per pose; bones are processed level by level (all bones of a level in parallel, one
// barrier per level), parents are read back from LDS as 3x4 affine rows. Per bone:
//   R = fromQuat(q)                                              math.ts:352-384
//   append rotation: R = fromQuat(slerp(I, +-q_append, |ratio|)) * R    model.ts:359-386 (uses the append
//                    parent's LOCAL rotation, so it adds no ordering dependency)
//   L = T(bind) * R ;  W = W_parent * L                           model.ts:398-414
// f32 throughout (the host computes in doubles with f32 stores): differences are ~1e-7 relative per level.
// Writes world [I][B][16] column-major and palette [I][B][3] rows of W * inverseBind.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_rows(float x, float y, float z, float w, float (&r)[9])
{
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    // row-major 3x3: r[row*3+col]; column-major source: out[0]=1-(yy+zz), out[1]=xy+wz, out[2]=xz-wy, out[4]=xy-wz ...
    r[0] = 1.0f - (yy + zz); r[1] = xy - wz;          r[2] = xz + wy;
    r[3] = xy + wz;          r[4] = 1.0f - (xx + zz); r[5] = yz - wx;
    r[6] = xz - wy;          r[7] = yz + wx;          r[8] = 1.0f - (xx + yy);
}

// ------------------------------------------------------------------------------------------------
// sample_bone / sample_morph — MMD motion sampling, run by rz_fk_kernel's staging pass for every (instance, bone) and
// (instance, vertex morph) when the pose comes from rz_set_pose_sampled:
//   rotation  slerp between the surrounding keys, parameter warped by the later key's R Bezier curve
//   position  per-axis lerp, each axis warped by its own X / Y / Z curve
//   morph     linear between the surrounding morph keys; a vertex morph sums its own track and the
//             group-morph tracks that feed it (ratio-scaled), own first, groups ascending
// Same arithmetic as host/vmd-sampler.js (doubles there, f32 here).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bezier_y(float x, float x1, float y1, float x2, float y2)
{
    if (x <= 0.0f) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    if (x1 == y1 && x2 == y2) return x;           // the default 20,20,107,107 curve is the identity
    float lo = 0.0f, hi = 1.0f, t = x;
    for (int i = 0; i < 24; ++i) {                // bisection-guarded Newton on x(t) = x
        const float s = 1.0f - t;
        const float fx = 3.0f * s * s * t * x1 + 3.0f * s * t * t * x2 + t * t * t - x;
        if (fabsf(fx) < 1e-7f) break;
        if (fx > 0.0f) hi = t; else lo = t;
        const float dfx = 3.0f * s * s * x1 + 6.0f * s * t * (x2 - x1) + 3.0f * t * t * (1.0f - x2);
        const float tn = dfx != 0.0f ? t - fx / dfx : 0.5f * (lo + hi);
        t = (tn > lo && tn < hi) ? tn : 0.5f * (lo + hi);
    }
    const float s = 1.0f - t;
    return 3.0f * s * s * t * y1 + 3.0f * s * t * t * y2 + t * t * t;
}

// Key spans. A track's keys are sorted by frame (duplicates allowed); the span of `frame` is (i0, i1 = i0 + 1) with i0 the
// LAST key whose frame is <= `frame` — what host/vmd-sampler.js: span() bisects for — clamped to the first / last key.
// The sampler does not bisect first: the track record carries the first and last key's frames, so the span is GUESSED by
// linear interpolation (baked motions have evenly spaced keys: the guess is right) and the keys of the guessed span are
// loaded speculatively together with their frames; only a wrong guess (uneven keys, duplicates) pays for a bisection of
// what the guess left. Chain of dependent loads per bone: record -> keys, instead of bone -> track -> offsets -> ends ->
// log2(n) probes -> keys.
struct KeyRange { uint32_t b, e; float f0, f1; };
__device__ __forceinline__ KeyRange key_range(const uint4 r) { return KeyRange{r.x, r.y, __uint_as_float(r.z), __uint_as_float(r.w)}; }

// 0 = clamped to key `i0` (before the first / after the last / single key); 1 = interior: g is the guessed first key of the span
__device__ __forceinline__ int span_guess(const KeyRange &k, float frame, uint32_t &g)
{
    const uint32_t n = k.e - k.b;
    if (n == 1 || frame <= k.f0) { g = k.b; return 0; }
    if (frame >= k.f1) { g = k.e - 1; return 0; }
    g = k.b + min((uint32_t)((frame - k.f0) / (k.f1 - k.f0) * (float)(n - 1)), n - 2);
    return 1;
}

// the guess missed: bisect [b, e) around it for the last key <= frame (kf[g] has been loaded as f_g)
__device__ __forceinline__ uint32_t span_bisect(const float *kf, const KeyRange &k, float frame, uint32_t g, float f_g)
{
    uint32_t lo = k.b, hi = k.e - 1;
    if (f_g <= frame) lo = g; else hi = g;                 // kf[lo] <= frame < kf[hi] holds on either side
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (kf[mid] <= frame) lo = mid; else hi = mid; }
    return lo;
}

// A track is sampled in two halves so that a thread can have SEVERAL tracks' loads in flight at once (its bones and its morph):
// *_issue() turns the track record into the guessed key span and REQUESTS the keys; *_finish() — called after every issue —
// waits for them, repairs a wrong guess and interpolates. Chain of dependent loads for any number of tracks: records -> keys.
struct BoneKeys {
    int mode;                   // 0 = the motion does not key the bone, 1 = clamped to key i0, 2 = interior span (i0, i0 + 1)
    uint32_t i0;
    KeyRange kr;
    float f_a, f_b;
    float4 a, b;
    float pa0, pa1, pa2, pb0, pb1, pb2;
    uint4 ip;
};

__device__ __forceinline__ BoneKeys bone_issue(const RzSampleParams &p, float frame, const uint4 rec)
{
    BoneKeys k;
    k.kr = key_range(rec);
    k.mode = 0; k.i0 = 0u; k.f_a = k.f_b = 0.0f;
    k.a = k.b = make_float4(0.f, 0.f, 0.f, 1.f);
    k.pa0 = k.pa1 = k.pa2 = k.pb0 = k.pb1 = k.pb2 = 0.0f;
    k.ip = make_uint4(0, 0, 0, 0);
    if (k.kr.e == k.kr.b) return k;
    if (!span_guess(k.kr, frame, k.i0)) {                   // clamped: the key itself
        const float *pa = p.key_pos + (size_t)k.i0 * 3;
        k.mode = 1; k.a = p.key_rot[k.i0]; k.pa0 = pa[0]; k.pa1 = pa[1]; k.pa2 = pa[2];
        return k;
    }
    // speculative: everything the guessed span needs, requested together
    const uint32_t i0 = k.i0;
    k.mode = 2;
    k.f_a = p.key_frame[i0]; k.f_b = p.key_frame[i0 + 1];
    k.a = p.key_rot[i0]; k.b = p.key_rot[i0 + 1];
    const float *pa = p.key_pos + (size_t)i0 * 3;
    k.pa0 = pa[0]; k.pa1 = pa[1]; k.pa2 = pa[2]; k.pb0 = pa[3]; k.pb1 = pa[4]; k.pb2 = pa[5];
    if (p.key_interp) k.ip = p.key_interp[i0 + 1];
    return k;
}

__device__ __forceinline__ void bone_finish(const RzSampleParams &p, float frame, BoneKeys &k, float4 &q, float &tx, float &ty, float &tz)
{
    q = make_float4(0.f, 0.f, 0.f, 1.f);
    tx = ty = tz = 0.f;
    if (k.mode == 0) return;
    if (k.mode == 1) { q = k.a; tx = k.pa0; ty = k.pa1; tz = k.pa2; return; }
    float f_a = k.f_a, f_b = k.f_b;
    float4 a = k.a, b = k.b;
    float pa0 = k.pa0, pa1 = k.pa1, pa2 = k.pa2, pb0 = k.pb0, pb1 = k.pb1, pb2 = k.pb2;
    uint4 ip = k.ip;
    if (!(f_a <= frame && frame < f_b)) {                   // the guess missed (uneven keys, duplicates): bisect what it left
        const uint32_t i0 = span_bisect(p.key_frame, k.kr, frame, k.i0, f_a);
        f_a = p.key_frame[i0]; f_b = p.key_frame[i0 + 1];
        a = p.key_rot[i0]; b = p.key_rot[i0 + 1];
        const float *pa = p.key_pos + (size_t)i0 * 3;
        pa0 = pa[0]; pa1 = pa[1]; pa2 = pa[2]; pb0 = pa[3]; pb1 = pa[4]; pb2 = pa[5];
        if (p.key_interp) ip = p.key_interp[i0 + 1];
    }
    const float x = (frame - f_a) / (f_b - f_a);
    float cx = x, cy = x, cz = x, cr = x;
    if (p.key_interp) {                                     // bytes [X_x1 Y_x1 Z_x1 R_x1 | X_y1 .. | X_x2 .. | X_y2 ..] of the LATER key
        auto byte = [](uint32_t w, int n) { return (float)((w >> (8 * n)) & 255u) * (1.0f / 127.0f); };
        cx = bezier_y(x, byte(ip.x, 0), byte(ip.y, 0), byte(ip.z, 0), byte(ip.w, 0));
        cy = bezier_y(x, byte(ip.x, 1), byte(ip.y, 1), byte(ip.z, 1), byte(ip.w, 1));
        cz = bezier_y(x, byte(ip.x, 2), byte(ip.y, 2), byte(ip.z, 2), byte(ip.w, 2));
        cr = bezier_y(x, byte(ip.x, 3), byte(ip.y, 3), byte(ip.z, 3), byte(ip.w, 3));
    }
    // Quat.slerp (math.ts:156-189)
    float c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0.0f) { c = -c; b.x = -b.x; b.y = -b.y; b.z = -b.z; b.w = -b.w; }
    if (c > 0.9995f) {
        q = make_float4(a.x + cr * (b.x - a.x), a.y + cr * (b.y - a.y), a.z + cr * (b.z - a.z), a.w + cr * (b.w - a.w));
        const float il = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        q.x *= il; q.y *= il; q.z *= il; q.w *= il;
    } else {
        const float th0 = acosf(c), sn = sinf(th0), th = th0 * cr;
        const float ka = sinf(th0 - th) / sn, kb = sinf(th) / sn;
        q = make_float4(ka * a.x + kb * b.x, ka * a.y + kb * b.y, ka * a.z + kb * b.z, ka * a.w + kb * b.w);
    }
    tx = pa0 + (pb0 - pa0) * cx; ty = pa1 + (pb1 - pa1) * cy; tz = pa2 + (pb2 - pa2) * cz;
}

__device__ __forceinline__ void sample_bone(const RzSampleParams &p, float frame, int bone, float4 &q, float &tx, float &ty, float &tz)
{
    BoneKeys k = bone_issue(p, frame, p.bone_range[bone]);
    bone_finish(p, frame, k, q, tx, ty, tz);
}

struct MorphKeys { int mode; uint32_t i0; KeyRange kr; float f_a, f_b, w_a, w_b; };     // mode as in BoneKeys

__device__ __forceinline__ MorphKeys morph_issue(const RzSampleParams &p, float frame, const uint4 rec)
{
    MorphKeys k;
    k.kr = key_range(rec);
    k.mode = 0; k.i0 = 0u; k.f_a = k.f_b = k.w_a = k.w_b = 0.0f;
    if (k.kr.e == k.kr.b) return k;
    if (!span_guess(k.kr, frame, k.i0)) { k.mode = 1; k.w_a = p.mkey_weight[k.i0]; return k; }
    k.mode = 2;
    k.f_a = p.mkey_frame[k.i0]; k.f_b = p.mkey_frame[k.i0 + 1]; k.w_a = p.mkey_weight[k.i0]; k.w_b = p.mkey_weight[k.i0 + 1];
    return k;
}

// the track's weight at `frame`; `keyed` = false when the track holds no key (it then contributes nothing at all)
__device__ __forceinline__ float morph_finish(const RzSampleParams &p, float frame, const MorphKeys &k, bool &keyed)
{
    keyed = k.mode != 0;
    if (k.mode == 0) return 0.0f;
    if (k.mode == 1) return k.w_a;
    float f_a = k.f_a, f_b = k.f_b, w_a = k.w_a, w_b = k.w_b;
    if (!(f_a <= frame && frame < f_b)) {
        const uint32_t i0 = span_bisect(p.mkey_frame, k.kr, frame, k.i0, f_a);
        f_a = p.mkey_frame[i0]; f_b = p.mkey_frame[i0 + 1]; w_a = p.mkey_weight[i0]; w_b = p.mkey_weight[i0 + 1];
    }
    return w_a + (w_b - w_a) * ((frame - f_a) / (f_b - f_a));
}

// feeds [f0, f1) of one vertex morph, accumulated in feed order on top of `w`
__device__ __forceinline__ float sample_feeds(const RzSampleParams &p, float frame, uint32_t f0, uint32_t f1, float w)
{
    for (uint32_t f = f0; f < f1; ++f) {
        const MorphKeys k = morph_issue(p, frame, p.feed_range[f]);
        bool keyed;
        const float wk = morph_finish(p, frame, k, keyed);
        if (keyed) w += wk * p.feed_ratio[f];
    }
    return w;
}

__device__ __forceinline__ float sample_morph(const RzSampleParams &p, float frame, int m)
{
    return sample_feeds(p, frame, p.feed_off[m], p.feed_off[m + 1], 0.0f);
}

// Quat.slerp(identity, a, t)  (math.ts:156-189): the append rotation (model.ts:367-386) and the bone-morph rotation use it
__device__ __forceinline__ float4 slerp_from_identity(float4 a, const float t)
{
    float c = a.w;
    if (c < 0.0f) { c = -c; a.x = -a.x; a.y = -a.y; a.z = -a.z; a.w = -a.w; }
    float sx, sy, sz, sw;
    if (c > 0.9995f) {
        sx = t * a.x; sy = t * a.y; sz = t * a.z; sw = 1.0f + t * (a.w - 1.0f);
        const float il = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz + sw * sw);
        sx *= il; sy *= il; sz *= il; sw *= il;
    } else {
        const float th0 = acosf(c), sn = sinf(th0), th = th0 * t;
        const float s0 = sinf(th0 - th) / sn, s1 = sinf(th) / sn;
        sx = s1 * a.x; sy = s1 * a.y; sz = s1 * a.z; sw = s0 + s1 * a.w;
    }
    return make_float4(sx, sy, sz, sw);
}

// W = P * L for 3x4 affine rows (bottom rows 0 0 0 1): the product a child's world matrix is made of (model.ts:405-414)
__device__ __forceinline__ void affine_mul(const float4 p0, const float4 p1, const float4 p2, const float4 l0, const float4 l1, const float4 l2,
                                           float4 &w0, float4 &w1, float4 &w2)
{
    const float P[12] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w };
    float W[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        W[i * 4 + 0] = P[i * 4] * l0.x + P[i * 4 + 1] * l1.x + P[i * 4 + 2] * l2.x;
        W[i * 4 + 1] = P[i * 4] * l0.y + P[i * 4 + 1] * l1.y + P[i * 4 + 2] * l2.y;
        W[i * 4 + 2] = P[i * 4] * l0.z + P[i * 4 + 1] * l1.z + P[i * 4 + 2] * l2.z;
        W[i * 4 + 3] = P[i * 4] * l0.w + P[i * 4 + 1] * l1.w + P[i * 4 + 2] * l2.w + P[i * 4 + 3];
    }
    w0 = make_float4(W[0], W[1], W[2], W[3]); w1 = make_float4(W[4], W[5], W[6], W[7]); w2 = make_float4(W[8], W[9], W[10], W[11]);
}

// The body of the hierarchy solve, shared by rz_fk_kernel (one workgroup per pose, results to global memory) and by the
// FUSED single-character frame, where every workgroup of rz_deform_kernel runs it as its prologue: `wl` is then the deform
// kernel's LDS palette (it ends up holding rows 0..2 of W * inverseBind), `scr` aliases its wave scratch, the sampled morph
// weights go to `lds_mw`, and only workgroup 0 (`to_global`) also leaves world matrices / palette / weights in memory.
//
// Shape of the solve (round 4): the topology comes as ONE 32-byte record per bone (two 16-byte loads instead of seven scalar
// arrays), and the parent chain is resolved by POINTER DOUBLING instead of level by level: every bone holds the product M of
// the local matrices of a run of its ancestors ending at itself and the index A of the bone above that run; a round does
// M[b] = M[A[b]] * M[b], A[b] = A[A[b]] for all bones at once (ping-pong buffers, one barrier), so ceil(log2(depth)) rounds
// — 4 for a 12-level tree — replace depth - 1 barrier-separated levels. The products are associated differently from the
// reference's parent-first recursion ((L0 L1)(L2 L3) instead of ((L0 L1) L2) L3): same f32 error class, ~1e-7 per product.
// LDS behind `scr`: rz_fk_scratch_bytes(B) = B x (48 + 8 + 12) bytes. Ends with a barrier.
template <bool FUSED>
__device__ __forceinline__ void fk_solve(const RzFkParams &p, const int inst, float4 *wl, unsigned char *scr, float *lds_mw, const bool to_global,
                                         const uint64_t st_tagv = 0ull, unsigned long long *fs = nullptr)
{
#ifdef RZ_ABLATE
#define RZ_FSTAMP(k) do { if (fs) fs[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RZ_FSTAMP(k) do { (void)fs; } while (0)
#endif
    // region X, 48 B per bone: local rotation | record word 0 | bind translation while the local matrices are formed, then the
    // second matrix buffer of the doubling rounds
    float4 *sq = reinterpret_cast<float4 *>(scr);                    // [B] local rotations of this pose
    uint4 *s_rec = reinterpret_cast<uint4 *>(sq + p.B);              // [B] (parent, append parent, bits(append ratio), flags)
    float4 *s_bind = reinterpret_cast<float4 *>(s_rec + p.B);        // [B] parent-relative bind translation
    float4 *m2 = reinterpret_cast<float4 *>(scr);                    // [B][3] aliases the three arrays above
    int *s_anc = reinterpret_cast<int *>(scr + (size_t)p.B * 48);    // [2][B] ping-pong ancestor indices
    float *s_lt = reinterpret_cast<float *>(s_anc + 2 * (size_t)p.B);   // [B][3] local translations of this pose
    const int tid = threadIdx.x;
    const float4 *lq = p.local_q + (size_t)inst * p.B;
    const float *glt = p.local_t ? p.local_t + (size_t)inst * p.B * 3 : nullptr;
    const bool sampled = p.sample.frames != nullptr || p.sample.frames_inline;      // rz_set_pose_sampled: the pose is evaluated right here
    const bool bone_morphs = p.bm_off != nullptr;
    const bool has_t = sampled || glt != nullptr || bone_morphs;
    const float *lt = has_t ? s_lt : nullptr;
    const float frame = sampled ? (p.sample.frames_inline ? p.sample.frame0 : p.sample.frames[inst]) : 0.0f;
    float *world = p.world + (size_t)inst * p.B * 16;
    float4 *pal = p.palette + (size_t)inst * p.B * 3;
    // this thread's inverse bind matrix (consumed after the rounds) is requested first, so its latency hides behind the
    // staging pass and the rounds instead of sitting in front of the output pass
    float4 pib0 = make_float4(0.f, 0.f, 0.f, 0.f), pib1 = pib0, pib2 = pib0, pib3 = pib0;
    if (tid < p.B) {
        const float4 *Im = reinterpret_cast<const float4 *>(p.inv_bind + (size_t)tid * 16);
        pib0 = Im[0]; pib1 = Im[1]; pib2 = Im[2]; pib3 = Im[3];
    }
    // Sampled pose, the common sizes (<= 512 bones: two per thread; <= 256 vertex morphs: one per thread): the track records of
    // the thread's bones AND the feed list of its morph are requested together, then all their keys, then the morph's keys —
    // three dependent round trips for the whole pose. Bone loop followed by morph loop (rounds 2-3) was five: records -> keys,
    // then feed offsets -> feed record -> keys ("pose staged" 4.2 / 6.3 us median / max after a wave's entry then, 4.2 / 5.7 now: NOTEBOOK.md R4.2, profiles/r4_timeline_sampled-demo.txt).
    const bool inter = sampled;
    int m_done = 0;                       // vertex morphs [0, m_done) have been sampled by the interleaved pass
    if (inter) {
        const int b0 = tid, b1 = tid + kBlock;
        const bool hb0 = b0 < p.B, hb1 = b1 < p.B, hm = tid < p.sample.M;
        uint4 ra0, ra1, rb0, rb1;
        uint4 tr0 = make_uint4(0, 0, 0, 0), tr1 = tr0;
        if (hb0) { ra0 = p.bone_rec[2 * b0]; ra1 = p.bone_rec[2 * b0 + 1]; tr0 = p.sample.bone_range[b0]; }
        if (hb1) { rb0 = p.bone_rec[2 * b1]; rb1 = p.bone_rec[2 * b1 + 1]; tr1 = p.sample.bone_range[b1]; }
        uint32_t f0 = 0u, f1 = 0u;
        if (hm) { f0 = p.sample.feed_off[tid]; f1 = p.sample.feed_off[tid + 1]; }
        BoneKeys k0 = bone_issue(p.sample, frame, tr0), k1 = bone_issue(p.sample, frame, tr1);
        uint4 fr = make_uint4(0, 0, 0, 0);
        float ratio0 = 0.0f;
        if (f1 > f0) { fr = p.sample.feed_range[f0]; ratio0 = p.sample.feed_ratio[f0]; }
        const MorphKeys mk = morph_issue(p.sample, frame, fr);
        auto park = [&](const int b, BoneKeys &k, const uint4 r0, const uint4 r1) {
            float4 q;
            float tx, ty, tz;
            bone_finish(p.sample, frame, k, q, tx, ty, tz);
            sq[b] = q; s_rec[b] = r0;
            s_bind[b] = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), 0.0f);
            s_lt[b * 3] = tx; s_lt[b * 3 + 1] = ty; s_lt[b * 3 + 2] = tz;
        };
        if (hb0) park(b0, k0, ra0, ra1);
        if (hb1) park(b1, k1, rb0, rb1);
        if (hm) {
            bool keyed;
            const float wk = morph_finish(p.sample, frame, mk, keyed);
            float w = 0.0f;
            if (keyed) w += wk * ratio0;
            if (f1 > f0 + 1u) w = sample_feeds(p.sample, frame, f0 + 1u, f1, w);      // group-morph tracks that feed it too
            if (FUSED || bone_morphs) lds_mw[tid] = w;
            if (to_global) p.sample.morph_w[(size_t)inst * p.sample.M + tid] = w;
        }
        m_done = min(p.sample.M, kBlock);
    }
    // one cooperative pass stages everything the later passes touch: the record loads are issued in front of the pose
    // (sampled: the track record, then its keys), so the static topology rides under the pose's own latency
    for (int i = inter ? tid + 2 * kBlock : tid; i < p.B; i += kBlock) {
        const uint4 r0 = p.bone_rec[2 * i], r1 = p.bone_rec[2 * i + 1];
        float4 q;
        float tx = 0.0f, ty = 0.0f, tz = 0.0f;
        if (sampled) {
            sample_bone(p.sample, frame, i, q, tx, ty, tz);
        } else if (FUSED && p.st_local_q) {
            // zero-copy pose that the previous frame's helper may have staged: the staged copy is asked for at once, the tag
            // (requested by the caller) decides afterwards; a miss re-reads the pinned slot and workgroup 0 keeps the pose
            q = p.st_local_q[i];
            if (glt) { tx = p.st_local_t[i * 3]; ty = p.st_local_t[i * 3 + 1]; tz = p.st_local_t[i * 3 + 2]; }
            if (st_tagv != p.st_expect) {
                q = lq[i];
                if (glt) { tx = glt[i * 3]; ty = glt[i * 3 + 1]; tz = glt[i * 3 + 2]; }
                if (to_global && p.copy_q) {
                    p.copy_q[i] = q;
                    if (glt) { p.copy_t[i * 3] = tx; p.copy_t[i * 3 + 1] = ty; p.copy_t[i * 3 + 2] = tz; }
                }
            }
        } else {
            q = lq[i];
            if (glt) { tx = glt[i * 3]; ty = glt[i * 3 + 1]; tz = glt[i * 3 + 2]; }
            if (FUSED && to_global && p.copy_q) {        // zero-copy first frame without a prefetch: keep the pose for the replays
                p.copy_q[i] = q;
                if (glt) { p.copy_t[i * 3] = tx; p.copy_t[i * 3 + 1] = ty; p.copy_t[i * 3 + 2] = tz; }
            }
        }
        sq[i] = q; s_rec[i] = r0;
        s_bind[i] = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), 0.0f);
        if (has_t) { s_lt[i * 3] = tx; s_lt[i * 3 + 1] = ty; s_lt[i * 3 + 2] = tz; }
    }
    if (sampled)        // vertex-morph weights of this pose: consumed by the prep / deform kernels that follow (FUSED: by this very workgroup)
        for (int m = m_done + tid; m < p.sample.M; m += kBlock) {      // (morphs beyond the interleaved pass)
            const float w = sample_morph(p.sample, frame, m);
            if (FUSED || bone_morphs) lds_mw[m] = w;
            if (to_global) p.sample.morph_w[(size_t)inst * p.sample.M + m] = w;
        }
    else if (bone_morphs && !FUSED)         // (FUSED: the caller has staged the uploaded weights already)
        for (int m = tid; m < p.bm_M; m += kBlock) lds_mw[m] = p.bm_w[(size_t)inst * p.bm_M + m];
    RZ_FSTAMP(0);             // pose staged (sampled), before the barrier
    __syncthreads();
    RZ_FSTAMP(1);
    if (bone_morphs) {
        // PMX bone morphs on the staged local pose: every bone folds its own entries, ascending morph index
        for (int b = tid; b < p.B; b += kBlock) {
            const uint32_t e0 = p.bm_off[b], e1 = p.bm_off[b + 1];
            if (e0 == e1) continue;
            float4 q = sq[b];
            float tx = s_lt[b * 3], ty = s_lt[b * 3 + 1], tz = s_lt[b * 3 + 2];
            for (uint32_t e = e0; e < e1; ++e) {
                const float w = lds_mw[p.bm_morph[e]];
                if (w == 0.0f) continue;
                const float4 t4 = p.bm_tr[e];
                tx += w * t4.x; ty += w * t4.y; tz += w * t4.z;
                const float4 r = slerp_from_identity(p.bm_rot[e], w);
                q = make_float4(q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y,          // Hamilton product q * r (math.ts Quat.multiply)
                                q.w * r.y - q.x * r.z + q.y * r.w + q.z * r.x,
                                q.w * r.z + q.x * r.y - q.y * r.x + q.z * r.w,
                                q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z);
            }
            sq[b] = q; s_lt[b * 3] = tx; s_lt[b * 3 + 1] = ty; s_lt[b * 3 + 2] = tz;
        }
        __syncthreads();
    }
    // Pass A, every bone in parallel: its LOCAL matrix L = T(bind + t) * R * T(add) (rows 0..2) into `wl`, its parent into
    // the first ancestor buffer. The quaternion / append / slerp math is off the rounds' critical path.
    for (int b = tid; b < p.B; b += kBlock) {
        const float4 q = sq[b];
        const uint4 rec = s_rec[b];
        const float4 bind = s_bind[b];
        float R[9];
        quat_to_rows(q.x, q.y, q.z, q.w, R);
        const int ap = (int)rec.y;
        const float ratio_raw = __uint_as_float(rec.z);
        float ax = 0.0f, ay = 0.0f, az = 0.0f;       // append-move: T(add) of L = T(bind) * R * T(add)
        if (ap >= 0) {
            const float ratio = fminf(1.0f, fmaxf(-1.0f, ratio_raw));
            if (fabsf(ratio) > 1e-6f) {
                if (lt && (rec.w & 1u)) {                // model.ts:388-393 uses the UNclamped ratio here
                    ax = lt[ap * 3] * ratio_raw; ay = lt[ap * 3 + 1] * ratio_raw; az = lt[ap * 3 + 2] * ratio_raw;
                }
                float4 a = sq[ap];
                const float t = fabsf(ratio);
                if (ratio < 0.0f) { a.x = -a.x; a.y = -a.y; a.z = -a.z; }
                const float4 sl = slerp_from_identity(a, t);
                float A[9], X[9];
                quat_to_rows(sl.x, sl.y, sl.z, sl.w, A);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) X[i * 3 + j] = A[i * 3] * R[j] + A[i * 3 + 1] * R[3 + j] + A[i * 3 + 2] * R[6 + j];
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = X[i];
            }
        }
        // translation column of L = T(bind + local) * R * T(add)  =  bind + local + R * add
        float tx = bind.x, ty = bind.y, tz = bind.z;
        if (lt) {
            tx += lt[b * 3]; ty += lt[b * 3 + 1]; tz += lt[b * 3 + 2];
            tx += R[0] * ax + R[1] * ay + R[2] * az;
            ty += R[3] * ax + R[4] * ay + R[5] * az;
            tz += R[6] * ax + R[7] * ay + R[8] * az;
        }
        wl[b * 3] = make_float4(R[0], R[1], R[2], tx);
        wl[b * 3 + 1] = make_float4(R[3], R[4], R[5], ty);
        wl[b * 3 + 2] = make_float4(R[6], R[7], R[8], tz);
        s_anc[b] = (int)rec.x;
    }
    RZ_FSTAMP(2);             // local matrices formed
    __syncthreads();          // (also: every read of region X is done, the rounds may write it)
    // Doubling rounds. Round k reads (M, A) from one buffer pair and writes the other: M'[b] = M[A[b]] * M[b], A'[b] = A[A[b]];
    // a bone whose run has reached its root (A < 0) is carried over unchanged. After ceil(log2(levels)) rounds every A is -1
    // and M is the world matrix (roots: W = L from the start).
    // A thread's first two bones (skeletons up to 512 bones: all of them) keep their matrix and their ancestor index in REGISTERS
    // across the rounds: a round then reads only the ancestor's matrix and the ancestor's ancestor — both addressed by a value the
    // thread already holds, so one LDS latency per round instead of two dependent ones — and writes its own for the others.
    float4 *src = wl, *dst = m2;
    int *asrc = s_anc, *adst = s_anc + p.B;
    constexpr int NBR = 2;
    float4 rm[NBR][3];
    int ra[NBR];
#pragma unroll
    for (int k = 0; k < NBR; ++k) {
        const int b = tid + k * kBlock;
        ra[k] = -1;
        rm[k][0] = rm[k][1] = rm[k][2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < p.B) { ra[k] = asrc[b]; rm[k][0] = src[b * 3]; rm[k][1] = src[b * 3 + 1]; rm[k][2] = src[b * 3 + 2]; }
    }
    for (int span = 1; span < p.n_levels; span <<= 1) {
#pragma unroll
        for (int k = 0; k < NBR; ++k) {
            const int b = tid + k * kBlock;
            if (b < p.B) {
                const int a = ra[k];
                if (a >= 0) {
                    ra[k] = asrc[a];
                    affine_mul(src[a * 3], src[a * 3 + 1], src[a * 3 + 2], rm[k][0], rm[k][1], rm[k][2], rm[k][0], rm[k][1], rm[k][2]);
                }
                dst[b * 3] = rm[k][0]; dst[b * 3 + 1] = rm[k][1]; dst[b * 3 + 2] = rm[k][2];
                adst[b] = ra[k];
            }
        }
        for (int b = tid + NBR * kBlock; b < p.B; b += kBlock) {       // bones beyond the register slots: through LDS, as before
            const int a = asrc[b];
            float4 w0 = src[b * 3], w1 = src[b * 3 + 1], w2 = src[b * 3 + 2];
            int an = -1;
            if (a >= 0) {
                an = asrc[a];
                affine_mul(src[a * 3], src[a * 3 + 1], src[a * 3 + 2], w0, w1, w2, w0, w1, w2);
            }
            dst[b * 3] = w0; dst[b * 3 + 1] = w1; dst[b * 3 + 2] = w2;
            adst[b] = an;
        }
        __syncthreads();
        float4 *t4 = src; src = dst; dst = t4;
        int *ti = asrc; asrc = adst; adst = ti;
    }
    RZ_FSTAMP(3);             // doubling rounds done
    if (p.ovr_off) {
        // physics-driven bones: the supplied world matrix replaces the solved one (rows 0..2 of the column-major 4x4)
        for (int k = p.ovr_off[inst] + tid; k < p.ovr_off[inst + 1]; k += kBlock) {
            const int b = p.ovr_bone[k];
            const float4 *m = reinterpret_cast<const float4 *>(p.ovr_world + (size_t)k * 16);
            const float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
            src[b * 3] = make_float4(c0.x, c1.x, c2.x, c3.x);
            src[b * 3 + 1] = make_float4(c0.y, c1.y, c2.y, c3.y);
            src[b * 3 + 2] = make_float4(c0.z, c1.z, c2.z, c3.z);
        }
        __syncthreads();
    }
    // all rounds done: one parallel pass writes the world matrices and the palette
    for (int b = tid; b < p.B; b += kBlock) {
        const float4 w0 = src[b * 3], w1 = src[b * 3 + 1], w2 = src[b * 3 + 2];
        const float W[12] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w };
        // world, column-major 4x4 (what queue.writeBuffer(worldMatrixBuffer) would have carried)
        if (to_global) {
            float4 *wo = reinterpret_cast<float4 *>(world + (size_t)b * 16);
            wo[0] = make_float4(W[0], W[4], W[8], 0.0f);
            wo[1] = make_float4(W[1], W[5], W[9], 0.0f);
            wo[2] = make_float4(W[2], W[6], W[10], 0.0f);
            wo[3] = make_float4(W[3], W[7], W[11], 1.0f);
        }
        // palette rows 0..2 of W * IB (IB general 4x4, column-major)
        const float4 *Im = reinterpret_cast<const float4 *>(p.inv_bind + (size_t)b * 16);
        const bool mine = b == tid;
        const float4 ibm[4] = { mine ? pib0 : Im[0], mine ? pib1 : Im[1], mine ? pib2 : Im[2], mine ? pib3 : Im[3] };
        float r[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 bc = ibm[c];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                r[i][c] = fmaf(W[i * 4 + 3], bc.w, fmaf(W[i * 4 + 2], bc.z, fmaf(W[i * 4 + 1], bc.y, W[i * 4] * bc.x)));
        }
        const float4 q0 = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]), q1 = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]),
                     q2 = make_float4(r[2][0], r[2][1], r[2][2], r[2][3]);
        if (to_global) { pal[b * 3] = q0; pal[b * 3 + 1] = q1; pal[b * 3 + 2] = q2; }
        if (FUSED) { wl[b * 3] = q0; wl[b * 3 + 1] = q1; wl[b * 3 + 2] = q2; }     // bone b's rows (in wl or in the other buffer) are only ever read by this thread in this pass
    }
    RZ_FSTAMP(4);             // palette rows written
    if (FUSED) __syncthreads();
#undef RZ_FSTAMP
}

__global__ void __launch_bounds__(kBlock) rz_fk_kernel(RzFkParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *scr = smem + (size_t)p.B * 48;
    fk_solve<false>(p, (int)blockIdx.x, reinterpret_cast<float4 *>(smem), scr, reinterpret_cast<float *>(scr + rz_fk_scratch_bytes(p.B)), true);
}


// ------------------------------------------------------------------------------------------------
// helpers for the skin phase
// ------------------------------------------------------------------------------------------------
struct Skinned { float px, py, pz, nx, ny, nz; };

// vs() lines engine.ts:255-272 for one vertex. `pal` = LDS palette (3 float4 rows per bone).
//   weights: w_i = (u8_i/255) / sum_k(u8_k/255)  (engine.ts:255-257)  ==  u8_i / isum  up to rounding;
//            isum == 0 takes the select((1,0,0,0)) branch (a sum of unorm8 values is either 0 or >= 1/255 > 1e-4).
//   blend:   because the map is linear, M = sum_i w_i * S[j_i] is formed once (12 FMA per bone) and applied
//            to the position and the normal, instead of transforming both by every bone (24 FMA per bone).
//            Rounding differs from the oracle's evaluation order by a few ulp (tolerance 1e-4).
__device__ __forceinline__ Skinned skin_vertex(const float4 *pal, float x, float y, float z, float nx, float ny,
                                               float nz, uint32_t j01, uint32_t j23, uint32_t wq, uint32_t bmax)
{
    const uint32_t b0 = wq & 255u, b1 = (wq >> 8) & 255u, b2 = (wq >> 16) & 255u, b3 = wq >> 24;
    const uint32_t isum = b0 + b1 + b2 + b3;
    const bool ok = isum != 0u;
    const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
    const float w[4] = { ok ? (float)b0 * inv : 1.0f, (float)b1 * inv, (float)b2 * inv, (float)b3 * inv };
    // joints are < B by construction (pmx-loader.ts:861-880); clamp so bad input cannot read past the palette
    const uint32_t j[4] = { min(j01 & 0xffffu, bmax), min(j01 >> 16, bmax), min(j23 & 0xffffu, bmax),
                            min(j23 >> 16, bmax) };
    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 r0 = pal[j[i] * 3 + 0], r1 = pal[j[i] * 3 + 1], r2 = pal[j[i] * 3 + 2];
        m0.x = fmaf(w[i], r0.x, m0.x); m0.y = fmaf(w[i], r0.y, m0.y); m0.z = fmaf(w[i], r0.z, m0.z); m0.w = fmaf(w[i], r0.w, m0.w);
        m1.x = fmaf(w[i], r1.x, m1.x); m1.y = fmaf(w[i], r1.y, m1.y); m1.z = fmaf(w[i], r1.z, m1.z); m1.w = fmaf(w[i], r1.w, m1.w);
        m2.x = fmaf(w[i], r2.x, m2.x); m2.y = fmaf(w[i], r2.y, m2.y); m2.z = fmaf(w[i], r2.z, m2.z); m2.w = fmaf(w[i], r2.w, m2.w);
    }
    Skinned o;
    o.px = fmaf(m0.z, z, fmaf(m0.y, y, fmaf(m0.x, x, m0.w)));
    o.py = fmaf(m1.z, z, fmaf(m1.y, y, fmaf(m1.x, x, m1.w)));
    o.pz = fmaf(m2.z, z, fmaf(m2.y, y, fmaf(m2.x, x, m2.w)));
    const float tx = fmaf(m0.z, nz, fmaf(m0.y, ny, m0.x * nx));
    const float ty = fmaf(m1.z, nz, fmaf(m1.y, ny, m1.x * nx));
    const float tz = fmaf(m2.z, nz, fmaf(m2.y, ny, m2.x * nx));
    const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
    // normalize() of a zero / non-finite vector is undefined in WGSL: the rest normal is returned (build-defined)
    const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
    const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
    o.nx = good ? tx * rl : nx;
    o.ny = good ? ty * rl : ny;
    o.nz = good ? tz * rl : nz;
    return o;
}

// ------------------------------------------------------------------------------------------------
// fused morph + skin.
//   S     morph-split: lanes per quad (1,2,4,8)
//   U     morphs in flight per lane-slice iteration (3*U 16-byte loads issued back to back)
//   MODE  0 = no morphs, 1 = dense planes, 2 = per-vertex sparse CSR (S must be 1)
//   NT    nontemporal loads on the morph stream (read once per frame: do not keep it in cache)
//   NTS   nontemporal stores of the outputs
//   GEO   1 = rest geometry is read with 16-byte loads by the quad owner and transposed through
//             the wave's LDS scratch; 0 = only the morphed position goes through LDS and the
//             vertex-per-lane phase reads normal/joints/weights with 4-byte loads
//   FAST  single-instance frame in ONE launch: the palette (world * inverseBind, engine.ts:926-928)
//         is formed by every workgroup itself — each thread requests its bone's two matrices before
//         anything else and does the 36 FMAs while the first group of morph loads is in flight, one
//         barrier before the first skin phase publishes it — and the active-morph list comes in the
//         kernel arguments (compacted on the host by rz_set_pose). No prep kernel, no launch boundary,
//         no load in front of the morph stream. !FAST reads the palette / list produced by
//         rz_prep_kernel (instanced frames, > 128 active morphs).
// grid = (tiles capped, instances); block = 256.
// dynamic LDS = palette | active-morph list (!FAST) | per-wave transpose scratch.
//
// Two phases per tile, both fully coalesced:
//   phase 1 (lane = quad of 4 vertices, slice s of S): stream the active morph planes with
//           16-byte loads, FMA into 12 partial sums, __shfl_xor butterfly across the S slices,
//           add to the rest position, park the quad in the wave's LDS scratch (ds_write_b128).
//   phase 2 (lane = one vertex): read its attributes back (conflict-free ds_read_b32), gather the
//           four bones' 3x4 rows from the LDS palette, LBS, normalize, store 12 B + 12 B per lane
//           (a wave writes 768 contiguous bytes per store instruction).
// ------------------------------------------------------------------------------------------------
template <bool NTS> __device__ __forceinline__ void st3(float *d, float a, float b, float c)
{
    if (NTS) {
        __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 1); __builtin_nontemporal_store(c, d + 2);
    } else {
        d[0] = a; d[1] = b; d[2] = c;
    }
}

// __launch_bounds__(256, 2): the persistent grid is two workgroups per CU (2 waves per SIMD), so the register
// allocator may use up to 256 VGPRs but not one more (a 257th would halve residency).
template <int S, int U, int MODE, bool NT, bool NTS, bool GEO, bool FAST>
__global__ void __launch_bounds__(kBlock, 2) rz_deform_kernel(const float *k_geom, const float *k_world, const float *k_inv_bind, const uint32_t k_bf,
                                                              const uint32_t k_Vp, const uint32_t k_nq, const uint32_t k_qpw, const uint32_t *k_j01,
                                                              const uint32_t *k_j23, const uint32_t *k_wq, const RzDeformParams p, const RzMorphList ml)
{
    // The leading arguments repeat what the FIRST loads of a wave need: the file is compiled with -amdgpu-kernarg-preload-count=16,
    // so the command processor hands the first 14 dwords over in SGPRs when the wave starts (16 user SGPRs less the kernel-argument
    // pointer: everything up to k_j23; k_wq follows by scalar load like `p`) — the matrices, the partition, the mesh planes, and
    // k_bf = bone count | helper-workgroup flag << 16 | may-be-staged flag << 17 | pose-in-pinned-memory flag << 18 | worker workgroups << 19. Everything else comes out of `p` by scalar loads, which take ~0.9 us to arrive (profiles/r4_timeline_c2.txt:
    // "entry -> prologue done"): a 3-17 us frame no longer waits for them before asking for its matrices and its mesh.
    constexpr int QPW = 64 / S;              // quads per wave
    constexpr int VW = 4 * QPW;              // vertices per wave per tile
    constexpr int NPL = GEO ? 9 : 3;         // scratch planes per wave
    constexpr int ROUNDS = (VW + 63) / 64;
    constexpr bool LDS_LIST = MODE == 2 || (!FAST && MODE == 1);   // MODE 2 keeps all M weights in LDS on both paths

    const int tid = threadIdx.x;
    const int inst = blockIdx.y;
    const int lane = tid & 63, wave = tid >> 6;
    const int kB = (int)(k_bf & 0xffffu);            // == p.B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *pal = reinterpret_cast<float4 *>(smem);                       // B*3 float4
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry

    // Zero-copy pose prefetch (see RzDeformParams): workgroup 0 of such a launch is the helper, the workers shift by one.
    const bool pf_on = (k_bf >> 16) & 1u;            // == p.pf_src != nullptr (only one-launch and fused-hierarchy frames ever carry one)
    if (pf_on && blockIdx.x == 0) {
        // seqlock read of the next upload's pinned slot: header == the expected sequence number -> the host has finished
        // writing that pose (it writes the header last); copy; header again; only then the tag. The ring protocol already
        // keeps the host from re-using the slot while this kernel runs, the second look is belt and braces.
        const uint64_t h1 = __builtin_nontemporal_load(p.pf_src_seq);
        if (h1 != p.pf_expect) return;                                   // workgroup-uniform
        const float4 *src = reinterpret_cast<const float4 *>(p.pf_src);
        float4 *dst = reinterpret_cast<float4 *>(p.pf_dst);
        const uint32_t n4 = p.pf_bytes / 16;
        // eight independent host loads in flight per thread (a 16.6 KB pose is one pass); indices past the end are clamped,
        // so the tail threads re-copy the last cell instead of branching
        const uint32_t last = n4 - 1;
        for (uint32_t i = tid; i < n4; i += kBlock * 8) {
            const uint32_t i0 = min(i, last), i1 = min(i + kBlock, last), i2 = min(i + 2 * kBlock, last), i3 = min(i + 3 * kBlock, last);
            const uint32_t i4 = min(i + 4 * kBlock, last), i5 = min(i + 5 * kBlock, last), i6 = min(i + 6 * kBlock, last), i7 = min(i + 7 * kBlock, last);
            const float4 a0 = src[i0], a1 = src[i1], a2 = src[i2], a3 = src[i3], a4 = src[i4], a5 = src[i5], a6 = src[i6], a7 = src[i7];
            dst[i0] = a0; dst[i1] = a1; dst[i2] = a2; dst[i3] = a3; dst[i4] = a4; dst[i5] = a5; dst[i6] = a6; dst[i7] = a7;
        }
        __syncthreads();
        if (tid == 0) {
            const uint64_t h2 = __builtin_nontemporal_load(p.pf_src_seq);
            if (h2 == p.pf_expect) *p.pf_tag = p.pf_expect;               // consumed by the NEXT kernel on this stream
        }
        return;
    }
    const uint32_t wid = blockIdx.x - (pf_on ? 1u : 0u);                 // worker index of this workgroup
    // THIS frame's pose: staged in device memory by the previous frame's helper, or still in its pinned slot. The answer is one
    // tag away, and waiting for it before asking for the matrices would put two memory latencies in a row in front of the
    // palette. So a frame that MAY find its pose staged (spec) asks for the tag and, at once, for the matrices of the staged
    // copy (the device pose block: valid memory whatever it holds); the tag is looked at when the palette is formed, and
    // only a miss then fetches the matrices from the pinned slot. Sparse weights are needed at once: that mode waits for the tag.
    const bool spec = FAST && ((k_bf >> 17) & 1u);                       // == p.st_tag != nullptr
    const float *world_in = k_world;                                     // == spec ? p.st_world : p.world (re-pointed at the pinned slot on a miss)
    const bool from_host = ((k_bf >> 18) & 1u) && !spec;                 // (p.world_copy != nullptr) the matrices are asked for over the host link up front

    // FAST: this thread's bone (tid < B covers the first 256 bones) — its world and inverse-bind matrices are
    // requested FIRST, as plain loads into registers, so they are the oldest entries of the vmcnt queue: the palette
    // math below only has to wait for them (a counted wait) while the morph loads issued after them stay in flight.
    float4 ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3;
    const bool early = FAST && tid < kB && RZ_DBG(p) != 3;      // dbg 3: ablation — no palette staging (output is garbage)
    // Zero-copy frame (world_copy != null: `world` is pinned HOST memory, a few microseconds away) with a dense morph stream:
    // vmcnt retires in order, so host loads at the head of the queue would hold back the first morph FMAs; there the world
    // matrices are requested BEHIND the first morph group instead, and the palette is formed after the last group.
    const bool late_world = FAST && MODE == 1 && from_host;
    bool world_pending = early && late_world;
    // (The loads are NOT predicated on `early`: threads past the last bone re-read bone B - 1. Inside a divergent branch the
    // compiler closed the block by shuffling the loaded registers, which made every wave WAIT for its matrices right here, in front
    // of the mesh loads that follow — a whole memory round trip at the head of every frame. FAST kernels only: a dead load otherwise.)
    const int eb = MODE != 1 ? min(tid, kB - 1) : tid;        // (dense kernels keep the predicated form: they have no register to spare)
    auto load_world = [&]() {
        const float4 *gw = reinterpret_cast<const float4 *>(world_in) + eb * 4;
        ew0 = gw[0]; ew1 = gw[1]; ew2 = gw[2]; ew3 = gw[3];
        world_pending = false;
    };
    if (FAST && (MODE != 1 || early)) {
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + eb * 4;
        if (!late_world) load_world();
        ei0 = gi[0]; ei1 = gi[1]; ei2 = gi[2]; ei3 = gi[3];
    }
    // Skeletons of 257..512 bones (the demo model has 349): without a dense stream there are registers to spare, so the thread's
    // SECOND bone is asked for up front as well — left to form_palette()'s late loop it was one more memory round trip in front
    // of the first skin phase of every workgroup (dense frames keep the late loop: it hides under their morph stream).
    constexpr bool EARLY2 = FAST && MODE != 1;
    float4 fw0, fw1, fw2, fw3, fi0, fi1, fi2, fi3;
    const bool early2 = EARLY2 && early && tid + kBlock < kB;
    const int eb2 = min(tid + kBlock, kB - 1);
    auto load_world2 = [&]() {
        const float4 *gw = reinterpret_cast<const float4 *>(world_in) + eb2 * 4;
        fw0 = gw[0]; fw1 = gw[1]; fw2 = gw[2]; fw3 = gw[3];
    };
    if constexpr (EARLY2) {                      // (unpredicated like the first bone's: with <= 256 bones every thread re-reads bone B - 1)
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + eb2 * 4;
        load_world2();
        fi0 = gi[0]; fi1 = gi[1]; fi2 = gi[2]; fi3 = gi[3];
    }
    const int s = lane / QPW;                // morph slice of this lane
    const int qi = lane % QPW;
    const size_t Vp = k_Vp;
    const size_t plane4 = Vp / 4;            // float4 per plane
    // persistent, evenly balanced partition: every wave of the grid owns one contiguous run of quads
    // (a multiple of 8 quads = 128 B per plane) and walks it QPW quads at a time; the last step is masked.
    // Without a dense stream the four waves of a workgroup take runs that lie a quarter of the mesh apart instead of next to
    // each other: work on such frames is uneven — the demo model's 60 expression morphs all sit on one 1 800-vertex face region,
    // 28 consecutive 64-vertex steps — and four neighbouring heavy steps on ONE CU share its LDS and its texture path
    // (2.7 us in the row walk with four face waves per CU: NOTEBOOK.md R4.1). Dense frames stream evenly: unchanged.
    // (the worker count rides in k_bf's upper bits — gridDim.x is a hidden kernel argument, i.e. one more scalar load; a grid too large
    // for the 13 bits keeps neighbouring runs)
    const uint32_t n_workers = k_bf >> 19;
    const uint32_t wave_global = (MODE != 1 && n_workers) ? (uint32_t)wave * n_workers + wid : wid * (kBlock / 64) + wave;
    const size_t q_begin = (size_t)wave_global * k_qpw;
    const size_t q_end = min((size_t)k_nq, q_begin + k_qpw);

    // Everything a step reads from the static mesh. Frames WITHOUT a dense morph stream (MODE 0 / 2: one character, small
    // crowds, sparse targets) are latency-bound — a 30 k-vertex frame is two or three dependent memory round trips and a
    // launch — so their steps ask for ALL of it at once: the quad's rest position, the skin phase's normal / joints / weights
    // (vertex per lane) and, for sparse targets, the bounds of the vertex's row; and the FIRST step asks right here,
    // in front of whatever the workgroup does first (the hierarchy solve, the staging of the morph weights, the palette), so
    // the mesh arrives under that prologue instead of behind it (NOTEBOOK.md R4.1: one round trip is ~1.1 us of a 4-7 us
    // frame). Dense frames keep the skin phase's loads behind the morph phase: there the kernel lives at 245 VGPRs.
    constexpr bool PRE = MODE != 1 && !GEO;
    // (Dense frames keep the skin phase's attribute loads behind the morph phase. Round 4 tried to bring them in by LDS-DMA at the
    // top of the step — 6 x 4 bytes per vertex straight into LDS, no VGPR held — so that the skin phase would not start with a
    // memory round trip: C5 123.2 -> 128.5 us, a 1/8 shard 16.52 -> 16.75 us, only C3 gained (7.4 -> 7.0): NOTEBOOK.md R4.3. Removed.)
    constexpr bool PRE_SP = PRE && MODE == 2 && ROUNDS == 1;      // (S = 1 steps are 4 rounds: their bounds are loaded round by round)
    float4 gx, gy, gz, gnx, gny, gnz;
    uint4 gj01, gj23, gw;
    float pnx[PRE ? ROUNDS : 1], pny[PRE ? ROUNDS : 1], pnz[PRE ? ROUNDS : 1];
    uint32_t pj01[PRE ? ROUNDS : 1], pj23[PRE ? ROUNDS : 1], pwq[PRE ? ROUNDS : 1];
    uint32_t sb0[1], sb1[1];                    // PRE_SP: row bounds of this lane's vertex
    auto issue = [&](const size_t qw) {
        const size_t q = qw + qi;
        if (s == 0 && q < q_end) {
            const float4 *G = reinterpret_cast<const float4 *>(k_geom) + q;
            gx = G[0]; gy = G[plane4]; gz = G[2 * plane4];
            if (GEO) {
                gnx = G[3 * plane4]; gny = G[4 * plane4]; gnz = G[5 * plane4];
                gj01 = reinterpret_cast<const uint4 *>(k_j01)[q];
                gj23 = reinterpret_cast<const uint4 *>(k_j23)[q];
                gw = reinterpret_cast<const uint4 *>(k_wq)[q];
            }
        }
        if constexpr (PRE) {
            const int v_live = (int)min((size_t)VW, (q_end - qw) * 4);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int vl = r * 64 + lane;
                if (vl < v_live) {
                    const size_t v = qw * 4 + vl;
                    pnx[r] = k_geom[3 * Vp + v]; pny[r] = k_geom[4 * Vp + v]; pnz[r] = k_geom[5 * Vp + v];
                    pj01[r] = k_j01[v]; pj23[r] = k_j23[v];
                }
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)         // (the weights plane's pointer is not among the preloaded arguments: asked for last)
                if (r * 64 + lane < v_live) pwq[r] = k_wq[qw * 4 + r * 64 + lane];
            if constexpr (PRE_SP) {
                sb0[0] = 0u; sb1[0] = 0u;
                if (lane < v_live) { sb0[0] = p.sp_ptr[qw * 4 + lane]; sb1[0] = p.sp_ptr[qw * 4 + lane + 1]; }
            }
        }
    };
    if (PRE && q_begin < q_end) issue(q_begin);

    // ---- from here on the kernel reads `p` (scalar loads of the kernel arguments: the loads above are in flight under them) ----
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(smem + (size_t)p.B * 48);   // Mpad   (LDS_LIST)
    float *s_w = reinterpret_cast<float *>(s_idx + (LDS_LIST ? p.Mpad : 0));
    float *scratch_all = s_w + (LDS_LIST ? p.Mpad : 0);                   // 16-B aligned: Mpad % 4 == 0
    const uint64_t st_tagv = spec ? *p.st_tag : 0ull;                   // requested here, compared later (workgroup-uniform)
    const bool staged_now = MODE == 2 && spec && st_tagv == p.st_expect;
    const float *morph_w_in = (staged_now && p.st_morph_w) ? p.st_morph_w : p.morph_w;
    int fused_count = 0;
    if (!FAST && p.fk_on) {
        // FUSED single-character frame: hierarchy solve (and motion sampling) as this workgroup's prologue, palette straight
        // into `pal`; the solve's scratch and the pose's morph weights alias the wave scratch, which nothing uses yet.
        __shared__ int fz_cnt[kBlock / 64];
        unsigned char *fscr = reinterpret_cast<unsigned char *>(scratch_all);
        float *lds_mw = reinterpret_cast<float *>(fscr + rz_fk_scratch_bytes(p.B));
        const bool sampled = p.fk.sample.frames != nullptr || p.fk.sample.frames_inline;
        // zero-copy local pose: staged in the device block by the previous frame's helper when the tag says so (requested here,
        // compared after the speculative loads of the staged copy have been issued), else still in its pinned slot; on a miss
        // workgroup 0 leaves the pose in the device block for the frames that replay it
        const bool fspec = p.st_tag != nullptr;
        const uint64_t ftag = fspec ? *p.st_tag : 0ull;
        if ((MODE != 0 || p.fk.bm_off) && !sampled) {
            const float *mw0 = fspec ? p.st_morph_w : p.morph_w;               // uploaded weights (staged copy, pinned slot or device block)
            for (int i = tid; i < p.M; i += kBlock) {
                float w = mw0[i];
                const bool miss = fspec && ftag != p.st_expect;
                if (miss) w = p.morph_w[i];
                lds_mw[i] = w;
                if (wid == 0 && p.morph_w_copy && (!fspec || miss)) p.morph_w_copy[i] = w;
            }
        }
#ifdef RZ_ABLATE
        fk_solve<true>(p.fk, 0, pal, fscr, lds_mw, wid == 0, ftag, p.tl ? tl_f : nullptr);
#else
        fk_solve<true>(p.fk, 0, pal, fscr, lds_mw, wid == 0, ftag);       // ends with a barrier: pal and lds_mw are complete
#endif
        if (MODE == 1) fused_count = compact_active(lds_mw, p.M, p.Mpad, s_idx, s_w, fz_cnt);
        if (MODE == 2)
            for (int i = tid; i < p.M; i += kBlock) s_w[i] = lds_mw[i];
        __syncthreads();
    } else if (!FAST) {
        const float4 *gpal = p.palette + (size_t)inst * p.B * 3;
        for (int i = tid; i < p.B * 3; i += kBlock) pal[i] = gpal[i];
        if (MODE == 1) {
            const uint32_t *gi = p.act_idx + (size_t)inst * p.Mpad;
            const float *gw = p.act_w + (size_t)inst * p.Mpad;
            for (int i = tid; i < p.Mpad; i += kBlock) { s_idx[i] = gi[i]; s_w[i] = gw[i]; }
        } else if (MODE == 2) {
            const float *gw = p.morph_w + (size_t)inst * p.M;
            for (int i = tid; i < p.M; i += kBlock) s_w[i] = gw[i];
        }
        __syncthreads();
    }

    // Sparse targets, one-launch frame: the pose's morph weights go to LDS. Up to 256 morphs (one per thread) the weight is only
    // REQUESTED here — it travels with the matrices and the first step's mesh loads — and is parked in LDS in front of the barrier
    // that publishes the palette (publish_weights): one wait and one barrier for everything the first skin phase needs, where
    // rounds 1-3 had a load -> LDS -> barrier sequence of their own in front of the palette (0.8 us of every such frame).
    const bool keep_w = FAST && MODE == 2 && wid == 0 && p.morph_w_copy != nullptr && !(staged_now && p.st_morph_w);     // zero-copy first frame, as for `world`
    bool w_pending = FAST && MODE == 2 && p.M <= kBlock;
    float w_early = 0.0f;
    if (w_pending && tid < p.M) w_early = morph_w_in[tid];
    auto publish_weights = [&]() {
        if (w_pending) {
            if (tid < p.M) { s_w[tid] = w_early; if (keep_w) p.morph_w_copy[tid] = w_early; }
            w_pending = false;
        }
    };
    if (FAST && MODE == 2 && !w_pending) {
        for (int i = tid; i < p.M; i += kBlock) {
            const float w = morph_w_in[i];
            s_w[i] = w;
            if (keep_w) p.morph_w_copy[i] = w;
        }
        __syncthreads();
    }

    RZ_STAMP(1);                 // prologue done (hierarchy solve / staged palette + morph list / sparse weights)
    const int count = (MODE == 1) ? (FAST ? ml.count : (p.fk_on ? fused_count : p.act_count[inst])) : 0;
    float *scr = scratch_all + (size_t)wave * NPL * VW;
    const uint32_t bmax = (uint32_t)(p.B - 1);
    float *opos = p.out_pos + (size_t)inst * Vp * 3;
    float *onrm = p.out_nrm + (size_t)inst * Vp * 3;
    bool need_palette = FAST && RZ_DBG(p) != 3;
    float bb[6] = { __builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };

    // palette rows 0..2 of world * inverseBind, out[c*4+r] = ((a0[r]*b0 + a1[r]*b1) + a2[r]*b2) + a3[r]*b3 (engine.ts:928)
    auto palette_rows = [&](int b, const float4 &a0, const float4 &a1, const float4 &a2, const float4 &a3, const float4 &b0,
                            const float4 &b1, const float4 &b2, const float4 &b3) {
        const float4 bc[4] = { b0, b1, b2, b3 };
        float r0[4], r1[4], r2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r0[c] = fmaf(a3.x, bc[c].w, fmaf(a2.x, bc[c].z, fmaf(a1.x, bc[c].y, a0.x * bc[c].x)));
            r1[c] = fmaf(a3.y, bc[c].w, fmaf(a2.y, bc[c].z, fmaf(a1.y, bc[c].y, a0.y * bc[c].x)));
            r2[c] = fmaf(a3.z, bc[c].w, fmaf(a2.z, bc[c].z, fmaf(a1.z, bc[c].y, a0.z * bc[c].x)));
        }
        const float4 q0 = make_float4(r0[0], r0[1], r0[2], r0[3]), q1 = make_float4(r1[0], r1[1], r1[2], r1[3]),
                     q2 = make_float4(r2[0], r2[1], r2[2], r2[3]);
        pal[b * 3 + 0] = q0; pal[b * 3 + 1] = q1; pal[b * 3 + 2] = q2;
        if (wid == 0 && p.palette) {      // keep the skinMatrixBuffer observable (rz_read_palette)
            float4 *gp = p.palette + (size_t)b * 3;
            gp[0] = q0; gp[1] = q1; gp[2] = q2;
        }
    };
    // executed once per wave, wherever the first step has its loads in flight; no barrier here
    // zero-copy first frame: `world` is pinned host memory; workgroup 0 leaves the matrices in device memory for the replays
    bool keep_world = FAST && wid == 0 && from_host;     // (a staged pose already sits where world_copy points)
    auto form_palette = [&]() {
        if (world_pending) load_world();          // no morph group ran in front of us
        if (spec && st_tagv != p.st_expect) {
            // miss: the previous frame's helper did not stage this pose (the host was not ahead): it is in its pinned slot
            world_in = p.world;
            keep_world = wid == 0 && p.world_copy != nullptr;
            if (early) load_world();
            if (early2) load_world2();
        }
        if (early) {
            palette_rows(tid, ew0, ew1, ew2, ew3, ei0, ei1, ei2, ei3);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + tid * 4; d[0] = ew0; d[1] = ew1; d[2] = ew2; d[3] = ew3; }
        }
        if (early2) {
            palette_rows(tid + kBlock, fw0, fw1, fw2, fw3, fi0, fi1, fi2, fi3);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + (tid + kBlock) * 4; d[0] = fw0; d[1] = fw1; d[2] = fw2; d[3] = fw3; }
        }
        for (int b = tid + (EARLY2 ? 2 : 1) * kBlock; b < p.B; b += kBlock) {      // bones beyond what was asked for up front: plain loads, late
            const float4 *gw = reinterpret_cast<const float4 *>(world_in) + b * 4;
            const float4 *gi = reinterpret_cast<const float4 *>(p.inv_bind) + b * 4;
            const float4 w0 = gw[0], w1 = gw[1], w2 = gw[2], w3 = gw[3];
            palette_rows(b, w0, w1, w2, w3, gi[0], gi[1], gi[2], gi[3]);
            if (keep_world) { float4 *d = reinterpret_cast<float4 *>(p.world_copy) + b * 4; d[0] = w0; d[1] = w1; d[2] = w2; d[3] = w3; }
        }
        need_palette = false;
    };
    bool need_sync = FAST && RZ_DBG(p) != 3;          // one workgroup barrier publishes the palette before the first phase 2

    // Write batching (p.out_cap > 0): deformed vertices are parked in a per-wave LDS buffer of out_cap vertices and
    // flushed as 16-byte-per-lane stores when it fills and at the end of the run. Interleaving 24 B of stores per
    // vertex with the read stream cost 10.7 us of a 130 us C5 frame (ablation dbg 5) although the bytes are only
    // 3 % of the traffic; batched, the HBM write bursts are long and rare.
    const uint32_t cap = p.out_cap;
    float *ob_pos = scratch_all + (size_t)(kBlock / 64) * NPL * VW + (size_t)wave * cap * 6;
    float *ob_nrm = ob_pos + (size_t)cap * 3;
    float *sp_all = scratch_all + (size_t)(kBlock / 64) * NPL * VW + (size_t)(kBlock / 64) * cap * 6;      // MODE 2: 4 x sp_cap staged CSR entries (16-byte aligned: every term is a multiple of 4 floats)
    uint32_t ob_fill = 0;               // vertices parked
    size_t ob_v0 = q_begin * 4;         // global vertex index of the first parked vertex
    auto flush_out = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t n4 = ob_fill * 3 / 4;                    // float4 per array (runs are multiples of 4 vertices)
        float4 *gp = reinterpret_cast<float4 *>(opos + ob_v0 * 3);
        float4 *gn = reinterpret_cast<float4 *>(onrm + ob_v0 * 3);
        const float4 *lp = reinterpret_cast<const float4 *>(ob_pos);
        const float4 *ln = reinterpret_cast<const float4 *>(ob_nrm);
        for (uint32_t i = lane; i < n4; i += 64) {
            if (NTS) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const float4 a = lp[i], b = ln[i];
                __builtin_nontemporal_store(f4v{a.x, a.y, a.z, a.w}, reinterpret_cast<f4v *>(gp + i));
                __builtin_nontemporal_store(f4v{b.x, b.y, b.z, b.w}, reinterpret_cast<f4v *>(gn + i));
            } else {
                gp[i] = lp[i]; gn[i] = ln[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
        ob_v0 += ob_fill;
        ob_fill = 0;
    };

    // ---- sparse morph targets (MODE 2): the step's piece of the vertex-ordered CSR goes through LDS ----
    // entry = (dx, dy, dz, bits(morph)), a vertex's entries ascending by morph, the entries of a step's vertices one contiguous
    // range [E0, E1). The wave copies that range into its LDS buffer by LDS-DMA — consecutive lanes, 16-byte entries: every
    // instruction is one 1 KiB burst, no register is held and ALL of them are in flight at once — and only then does each lane
    // (= one vertex) walk its own row, out of LDS: acc = fma(w, d, acc) over ascending entries — the order of the CPU oracle's
    // sparse accumulate, whatever the launch shape. Ranges larger than the buffer go through it in pieces; a row that straddles
    // two pieces keeps its running sum. A 64-vertex step asks for its first piece at the TOP of the step, as
    // soon as its row bounds are there: it lands while the palette is formed and the quad is parked.
    // Rounds 1-3 let every lane walk its row in global memory, 4 entries at a time: a load instruction then touches 64 different
    // cache lines and every four entries cost a memory round trip — the face of the demo model (60 expression morphs on the same
    // ~1 800 vertices, up to 60 entries per vertex) kept its waves 4.3 us in that loop (NOTEBOOK.md R4.1; now: profiles/r4_timeline_demo.txt).
    // LDS slots are XOR-swizzled (bits 0..3 with bits 4..7 of the entry's index in the piece): lanes read rows whose starts are
    // a row length apart, and with rows of 16 / 32 / 48 entries — or the demo shape's 20 — plain slots put a whole wave on the same
    // few banks. The DMA cannot scatter, so the swizzle is applied on the way IN: lane L of a burst fetches the entry whose slot L is
    // (an involution inside aligned 16-entry groups: the burst still reads the same 256-byte segments).
    float4 *sp_buf = reinterpret_cast<float4 *>(sp_all) + (size_t)wave * p.sp_cap;
    const uint32_t sp_cap = p.sp_cap;
    auto sp_slot = [](uint32_t i) { return i ^ ((i >> 4) & 15u); };
    // Four bursts share one LDS base (M0); the instruction offset — added to the global AND the LDS address — steps through
    // them. Rewriting M0 for every burst doubled the time a wave needs to issue a 20 KB piece (tools/dmabench: 2 155 vs 995 cycles).
    auto sp_stage = [&](const uint32_t c0, const uint32_t c1) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        for (uint32_t i = 0; c0 + i < c1; i += 256) {
            const lptr_t l = (lptr_t)(uint32_t)(uintptr_t)(sp_buf + i);
            const uint32_t e0 = c0 + sp_slot(i + (uint32_t)lane), e1 = c0 + sp_slot(i + 64 + (uint32_t)lane);
            const uint32_t e2 = c0 + sp_slot(i + 128 + (uint32_t)lane), e3 = c0 + sp_slot(i + 192 + (uint32_t)lane);
            if (e0 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e0), l, 16, 0, 0);
            if (e1 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e1 - 64), l, 16, 1024, 0);
            if (e2 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e2 - 128), l, 16, 2048, 0);
            if (e3 < c1) __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(p.sp_entries + e3 - 192), l, 16, 3072, 0);
        }
    };
    uint32_t spE0 = 0u, spE1 = 0u;          // PRE_SP: the step's entry range

    // One step = QPW quads. The body is instantiated twice: FIRST (the run's first step, which also forms the
    // palette from the early-loaded matrices) and the steady-state form, where those 32 registers are dead.
    auto step = [&](const size_t qw, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const size_t q = qw + qi;                                    // this lane's quad
        const bool live = q < q_end;
        float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ay = ax, az = ax;

        // rest geometry of the quad (slice 0 only), issued first so it overlaps the morph stream; without a dense stream: the
        // whole step's loads, and the run's first step has asked at the top of the kernel already
        if (!PRE || !FIRST) issue(qw);
        if constexpr (PRE_SP) {
            // the row bounds are the oldest loads in flight (the first step's were asked for at the top of the kernel): the first
            // piece of the step's entries is requested before anything else the step does
            const int n_live = (int)min((size_t)64, (q_end - qw) * 4);
            spE0 = __builtin_amdgcn_readfirstlane(sb0[0]);
            spE1 = __builtin_amdgcn_readlane(sb1[0], n_live - 1);
            if (spE0 < spE1) sp_stage(spE0, min(spE1, spE0 + sp_cap));
        }

        if (MODE == 1 && live) {
            const float4 *D = reinterpret_cast<const float4 *>(p.dense) + q;
            // slice s of S accumulates the active morphs a = s, s+S, s+2S, ... in ascending order.
            // The loop counter a0 is wave-uniform, so on the FAST path the list entries are fetched with
            // scalar loads straight from the kernel arguments and each lane picks its slice's entry with
            // v_cndmask — no vector-memory load sits in front of the morph stream.
            auto entry = [&](int base, uint32_t &m, float &w) {
                if (FAST) {
                    // (PIN: the compiler folds the select chain below into ONE per-lane indexed load ml.idx[base + s] from the
                    // kernel-argument segment, i.e. a vector-memory load in front of every group's morph loads; readfirstlane
                    // keeps the entries in SGPRs — scalar loads — and the select a v_cndmask. NOTEBOOK.md R4.10 / R5.2.)
                    constexpr bool PIN = (RZ_PIN_ML >> (S == 1 ? 0 : S == 2 ? 1 : S == 4 ? 2 : 3)) & 1;
                    if constexpr (PIN) {
                        auto sg = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                        m = sg((uint32_t)ml.idx[base]); w = __uint_as_float(sg(__float_as_uint(ml.w[base])));
#pragma unroll
                        for (int k = 1; k < S; ++k) {
                            const uint32_t mk = sg((uint32_t)ml.idx[base + k]);
                            const float wk = __uint_as_float(sg(__float_as_uint(ml.w[base + k])));
                            m = (s == k) ? mk : m; w = (s == k) ? wk : w;
                        }
                    } else {
                        m = (uint32_t)ml.idx[base]; w = ml.w[base];
#pragma unroll
                        for (int k = 1; k < S; ++k) {
                            const uint32_t mk = (uint32_t)ml.idx[base + k];
                            const float wk = ml.w[base + k];
                            m = (s == k) ? mk : m; w = (s == k) ? wk : w;
                        }
                    }
                } else {
                    m = s_idx[base + s]; w = s_w[base + s];
                }
            };
            int a0 = 0;
            // full groups of U morphs per slice: 3*U independent 16-byte loads in flight per lane
            for (; a0 + U * S <= count; a0 += U * S) {
                float4 dx[U], dy[U], dz[U];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t m;
                    entry(a0 + u * S, m, w[u]);
                    const float4 *d = D + (size_t)m * 3 * plane4;
                    dx[u] = ld_stream(d, NT);
                    dy[u] = ld_stream(d + plane4, NT);
                    dz[u] = ld_stream(d + 2 * plane4, NT);
                }
                if (FAST && FIRST && world_pending) load_world();     // zero-copy frame: behind the first group's loads
                // first group of the first step: the palette math overlaps the 3*U loads just issued. On a zero-copy frame the
                // matrices come over the host link (a few microseconds): there the palette waits until the LAST group has
                // issued its loads, so the whole morph stream of the step is in flight under that latency.
                if (FAST && FIRST && need_palette && (!from_host || a0 + 2 * U * S > count)) form_palette();
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ax.x = fmaf(w[u], dx[u].x, ax.x); ax.y = fmaf(w[u], dx[u].y, ax.y);
                    ax.z = fmaf(w[u], dx[u].z, ax.z); ax.w = fmaf(w[u], dx[u].w, ax.w);
                    ay.x = fmaf(w[u], dy[u].x, ay.x); ay.y = fmaf(w[u], dy[u].y, ay.y);
                    ay.z = fmaf(w[u], dy[u].z, ay.z); ay.w = fmaf(w[u], dy[u].w, ay.w);
                    az.x = fmaf(w[u], dz[u].x, az.x); az.y = fmaf(w[u], dz[u].y, az.y);
                    az.z = fmaf(w[u], dz[u].z, az.z); az.w = fmaf(w[u], dz[u].w, az.w);
                }
            }
            for (; a0 < count; a0 += S) {     // remainder, one morph per slice at a time (list is zero-padded)
                uint32_t m;
                float w;
                entry(a0, m, w);
                if (a0 + s < count) {
                    const float4 *d = D + (size_t)m * 3 * plane4;
                    float4 dx = ld_stream(d, NT), dy = ld_stream(d + plane4, NT), dz = ld_stream(d + 2 * plane4, NT);
                    ax.x = fmaf(w, dx.x, ax.x); ax.y = fmaf(w, dx.y, ax.y); ax.z = fmaf(w, dx.z, ax.z); ax.w = fmaf(w, dx.w, ax.w);
                    ay.x = fmaf(w, dy.x, ay.x); ay.y = fmaf(w, dy.y, ay.y); ay.z = fmaf(w, dy.z, ay.z); ay.w = fmaf(w, dy.w, ay.w);
                    az.x = fmaf(w, dz.x, az.x); az.y = fmaf(w, dz.y, az.y); az.z = fmaf(w, dz.z, az.z); az.w = fmaf(w, dz.w, az.w);
                }
            }
        }
        if (MODE == 1 && S > 1) {
            // combine the S partial sums of each quad: __shfl_xor butterfly over the slice bits of the
            // lane id (every lane takes part; lanes past the end of the run carry zeros)
#pragma unroll
            for (int off = QPW; off < 64; off <<= 1) {
                ax.x += __shfl_xor(ax.x, off); ax.y += __shfl_xor(ax.y, off);
                ax.z += __shfl_xor(ax.z, off); ax.w += __shfl_xor(ax.w, off);
                ay.x += __shfl_xor(ay.x, off); ay.y += __shfl_xor(ay.y, off);
                ay.z += __shfl_xor(ay.z, off); ay.w += __shfl_xor(ay.w, off);
                az.x += __shfl_xor(az.x, off); az.y += __shfl_xor(az.y, off);
                az.z += __shfl_xor(az.z, off); az.w += __shfl_xor(az.w, off);
            }
        }

        if (FIRST) RZ_STAMP(2);       // first step: morph phase done (its loads have landed)
        if (FAST && FIRST && need_palette) form_palette();            // no morph group ran (MODE 0/2, or nothing active)
        if (FAST && MODE == 2 && FIRST) publish_weights();
        if (FAST && FIRST && need_sync) { __syncthreads(); need_sync = false; }   // palette (and sparse weights) of every wave are in LDS
        if (FIRST) RZ_STAMP(3);       // first step: palette published

        // ---- park the quad in the wave's scratch: plane-major [NPL][VW] dwords ----
        if (s == 0 && live) {
            float4 *sc4 = reinterpret_cast<float4 *>(scr) + qi;
            sc4[0 * QPW] = make_float4(gx.x + ax.x, gx.y + ax.y, gx.z + ax.z, gx.w + ax.w);
            sc4[1 * QPW] = make_float4(gy.x + ay.x, gy.y + ay.y, gy.z + ay.z, gy.w + ay.w);
            sc4[2 * QPW] = make_float4(gz.x + az.x, gz.y + az.y, gz.z + az.z, gz.w + az.w);
            if (GEO) {
                sc4[3 * QPW] = gnx; sc4[4 * QPW] = gny; sc4[5 * QPW] = gnz;
                uint4 *su4 = reinterpret_cast<uint4 *>(scr) + qi;
                su4[6 * QPW] = gj01; su4[7 * QPW] = gj23; su4[8 * QPW] = gw;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const size_t vw0 = qw * 4;     // first vertex of this wave's step
        const int v_live = RZ_DBG(p) == 4 ? 0 : (int)min((size_t)VW, (q_end - qw) * 4);   // dbg 4: ablation — morph phase only

        float spx[ROUNDS], spy[ROUNDS], spz[ROUNDS];        // MODE 2: this lane's vertex's morph offset, per round
        if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int vl = r * 64 + lane;
                const int n_live = min(64, v_live - r * 64);        // lanes of this round that own a vertex (wave-uniform)
                spx[r] = 0.0f; spy[r] = 0.0f; spz[r] = 0.0f;
                if (n_live <= 0) continue;
                uint32_t b0, b1, E0, E1;
                if constexpr (PRE_SP) { b0 = sb0[0]; b1 = sb1[0]; E0 = spE0; E1 = spE1; }        // (the first piece is on its way)
                else {
                    b0 = 0u; b1 = 0u;
                    if (vl < v_live) { b0 = p.sp_ptr[vw0 + vl]; b1 = p.sp_ptr[vw0 + vl + 1]; }
                    E0 = __builtin_amdgcn_readfirstlane(b0);
                    E1 = __builtin_amdgcn_readlane(b1, n_live - 1);
                }
                for (uint32_t c0 = E0; c0 < E1; c0 += sp_cap) {             // wave-uniform; no iteration at all for a step without offsets
                    const uint32_t c1 = min(E1, c0 + sp_cap);
                    if (!PRE_SP || c0 != E0) sp_stage(c0, c1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t lo = max(b0, c0), hi = min(b1, c1);
                    constexpr int SU = 8;                                   // LDS reads in flight per lane (entry, then its weight)
                    for (uint32_t j = lo; j < hi; j += SU) {
                        float4 ent[SU];
                        float w[SU];
#pragma unroll
                        for (int u = 0; u < SU; ++u) ent[u] = sp_buf[sp_slot(min(j + u, hi - 1) - c0)];
#pragma unroll
                        for (int u = 0; u < SU; ++u) w[u] = s_w[__float_as_uint(ent[u].w)];
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const float wu = j + u < hi ? w[u] : 0.0f;       // (past the row's end the clamped index re-read its last entry)
                            spx[r] = fmaf(wu, ent[u].x, spx[r]); spy[r] = fmaf(wu, ent[u].y, spy[r]); spz[r] = fmaf(wu, ent[u].z, spz[r]);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();                        // every lane is done with this piece before the next one lands
                }
            }
        }

        // ---- phase 2: one vertex per lane ----
        auto skin_round = [&](const int r, float nx, float ny, float nz, uint32_t j01, uint32_t j23, uint32_t wq) {
            const int vl = r * 64 + lane;
            if (vl < v_live) {
                const size_t v = vw0 + vl;
                float x = scr[0 * VW + vl], y = scr[1 * VW + vl], z = scr[2 * VW + vl];
                if constexpr (MODE == 2) { x += spx[r]; y += spy[r]; z += spz[r]; }
                if constexpr (PRE) {
                    // (asked for at the top of the step)
                } else if (GEO) {
                    nx = scr[3 * VW + vl]; ny = scr[4 * VW + vl]; nz = scr[5 * VW + vl];
                    const uint32_t *su = reinterpret_cast<const uint32_t *>(scr);
                    j01 = su[6 * VW + vl]; j23 = su[7 * VW + vl]; wq = su[8 * VW + vl];
                } else {
                    nx = p.geom[3 * Vp + v]; ny = p.geom[4 * Vp + v]; nz = p.geom[5 * Vp + v];
                    j01 = p.joints01[v]; j23 = p.joints23[v]; wq = p.weights[v];
                }
                Skinned o = skin_vertex(pal, x, y, z, nx, ny, nz, j01, j23, wq, bmax);
                if (cap) {
                    const uint32_t li = (ob_fill + vl) * 3;
                    ob_pos[li] = o.px; ob_pos[li + 1] = o.py; ob_pos[li + 2] = o.pz;
                    ob_nrm[li] = o.nx; ob_nrm[li + 1] = o.ny; ob_nrm[li + 2] = o.nz;
                } else if (RZ_DBG(p) != 5 || o.px == 1234.5f) {   // dbg 5: ablation — skin phase without its output stream
                    st3<NTS>(opos + v * 3, o.px, o.py, o.pz);
                    st3<NTS>(onrm + v * 3, o.nx, o.ny, o.nz);
                }
                if (p.edge) {
                    // fused consumer: the outline pass's inverted hull, engine.ts:458-461
                    //   expandedPos = worldPos + worldNormal * edgeSize * 0.01
                    const float e = p.edge[v];
                    st3<NTS>(p.out_hull + ((size_t)inst * Vp + v) * 3, o.px + (o.nx * e) * 0.01f, o.py + (o.ny * e) * 0.01f,
                             o.pz + (o.nz * e) * 0.01f);
                }
                if (p.aabb && v < p.n_verts) {       // padding vertices of the last quad stay out of the box
                    bb[0] = fminf(bb[0], o.px); bb[1] = fminf(bb[1], o.py); bb[2] = fminf(bb[2], o.pz);
                    bb[3] = fmaxf(bb[3], o.px); bb[4] = fmaxf(bb[4], o.py); bb[5] = fmaxf(bb[5], o.pz);
                }
            }
        };
        if constexpr (PRE) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, pnx[r], pny[r], pnz[r], pj01[r], pj23[r], pwq[r]);
        } else if constexpr (MODE == 2) {      // (GEO form of the sparse kernel, tools-only build: spx[r] wants a constant index)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u);
        } else {
#pragma unroll 1
            for (int r = 0; r < ROUNDS; ++r) skin_round(r, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u);
        }
        __builtin_amdgcn_wave_barrier();
        if (FIRST) RZ_STAMP(4);       // first step: skin phase issued
        if (cap) {
            ob_fill += (uint32_t)v_live;
            if (ob_fill + VW > cap) flush_out();      // the next step might not fit
        }
    };
    {
        size_t qw = q_begin;
        if (qw < q_end) { step(qw, std::true_type{}); qw += QPW; }
        for (; qw < q_end; qw += QPW) step(qw, std::false_type{});
    }
    RZ_STAMP(5);                 // last step done
    if (cap && ob_fill) flush_out();
    if (FAST && need_palette) form_palette();    // a wave with an empty run still owes the workgroup its bones ...
    if (FAST && MODE == 2) publish_weights();    // ... its morph weights ...
    if (FAST && need_sync) __syncthreads();      // ... and its barrier
    if (p.aabb) {
        // fused per-frame bounding box: per-lane running min/max -> wave butterfly -> one atomic per wave and
        // component on order-preserving integer keys. The kernel also re-arms the OTHER slot for the next frame,
        // so no memset launch sits between frames.
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                bb[k] = fminf(bb[k], __shfl_xor(bb[k], off));
                bb[3 + k] = fmaxf(bb[3 + k], __shfl_xor(bb[3 + k], off));
            }
        }
        uint32_t *slot = p.aabb + ((size_t)inst * 2 + (p.aabb_slot & 1)) * 6;
        if (lane < 6 && q_begin < q_end) {
            const float sel = lane == 0 ? bb[0] : lane == 1 ? bb[1] : lane == 2 ? bb[2] : lane == 3 ? bb[3] : lane == 4 ? bb[4] : bb[5];
            const uint32_t bits = __float_as_uint(sel);
            const uint32_t key = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
            if (lane < 3) atomicMin(slot + lane, key); else atomicMax(slot + lane, key);
        }
        if (wid == 0 && tid < 6) {
            uint32_t *next = p.aabb + ((size_t)inst * 2 + ((p.aabb_slot + 1) & 1)) * 6;
            next[tid] = tid < 3 ? 0xffffffffu : 0u;
        }
    }
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kBlock / 64) + wave);
}

// ------------------------------------------------------------------------------------------------
// instanced skin (MODE 0, I > 1 — BASELINE config C4: many poses of one static mesh).
// A workgroup owns a run of vertices and a GROUP of G instances whose palettes it stages together in
// LDS (LDS-DMA from the prep kernel's palette). Each lane decodes a vertex ONCE — rest position/normal,
// the four joints as palette row offsets, the four weights as floats — and then loops over the G poses:
// 12 ds_read_b128 + blend + transform + 24 B store per pose. The static mesh is read I/G times instead
// of I times, and the per-pose body has no global load in front of it (the generic kernel was latency-
// bound here: 43 % of wave time in s_waitcnt, VALU 30 %, LDS 31 % — profiles/r1_sq_counters.txt).
// grid = (vertex runs, instance groups); block = 256; dynamic LDS = G * B * 48 bytes.
// ------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// BLOCK threads per workgroup (256: two workgroups per CU; 512 / 1024: one, with 8 / 16 waves sharing one staged palette
// group — half / a quarter of the palette traffic per CU). NB = how many influences the pose loop gathers: the host picks
// nothing here, every wave decides per vertex step from a ballot over its lanes' weights (wave-uniform, so no divergence):
// a step whose 64 vertices are all BDEF1 / BDEF2 (real PMX models cluster them by mesh part) reads 3 / 6 palette rows per
// pose instead of 12. Skipped terms are w = 0, i.e. fma(0, row, m) = m: the result bits do not depend on the path taken.
//
// SUB = bone-subset form. A vertex run names only a few of the skeleton's bones (PMX meshes are bone-local: the synthetic C4
// mesh's 3 750-vertex runs touch ~34 of 200), and rz_run_subsets_kernel has listed them per run and rewritten the joints as
// slots of that list. The workgroup stages ONLY those bones of its G poses: 8 x 34 matrices instead of 8 x 200 — the front of
// every workgroup (LDS-DMA + palette product, during which the CU stores nothing) shrinks from 2.9 us to 1.3 us (profiles/r3_c4_front.txt), and the
// group's LDS footprint from 102 KB to 30 KB. World matrices are staged behind the palette region, so the product needs no
// in-place rounds. The palette rows a vertex gathers hold the same bits wherever they sit in LDS: outputs do not change.
// (Tried and measured slower or without effect, then removed again — NOTEBOOK.md R3.1 / R3.4 / R3.9: forcing three workgroups per CU
// (80 VGPRs spill), a branch-free pose loop with clamped tail lanes, four poses unrolled, raised wave priority around the stores.)
template <int BLOCK, bool NTS, bool SUB>
__global__ void __launch_bounds__(BLOCK) rz_skin_instances_kernel(const uint32_t *k_sub_count, const uint16_t *k_sub_list, const float4 *k_src,
                                                                  const float *k_inv_bind, const int G, const int n_inst, const uint32_t verts_per_wg,
                                                                  const uint32_t k_grid, const uint32_t k_bf, const uint32_t k_Vp, const RzDeformParams p)
{
    // The leading arguments are preloaded into SGPRs at wave start (14 dwords; see rz_deform_kernel): what the FRONT of a workgroup
    // needs — the run's bone list, the matrices it stages (k_src = the poses' world matrices, or their finished palettes behind
    // rz_prep_kernel / rz_fk_kernel), the inverse bind matrices, the launch shape (k_grid = gridDim.x | gridDim.y << 16: the hidden
    // arguments would be one more scalar load) and k_bf = bone count | inst_order << 16 | dma << 17.
    const int kB = (int)(k_bf & 0xffffu);
    const bool k_order = (k_bf >> 16) & 1u, k_dma = (k_bf >> 17) & 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *pal = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RZ_TL_DECL;
    RZ_STAMP(0);                 // entry
    // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest). inst_order 0: x = vertex run — an XCD sees every
    // pose group and one eighth of the mesh; 1: consecutive workgroups take consecutive pose groups — an XCD sees one eighth of
    // the poses' matrices and the whole mesh.
    const uint32_t n_groups = k_grid >> 16, lin = blockIdx.x + (k_grid & 0xffffu) * blockIdx.y;
    const uint32_t wg_group = k_order ? lin % n_groups : blockIdx.y, wg_run = k_order ? lin / n_groups : blockIdx.x;
    const int inst0 = (int)wg_group * G;
    const int ng = min(G, n_inst - inst0);
    const int rows = kB * 3;                       // float4 per palette, in global memory and (finished) in LDS
    constexpr int rstride = 3;                      // float4 per bone of a finished palette
    // SUB: this run's bone list (workgroup-uniform: scalar loads)
    const int ns = SUB ? (int)k_sub_count[wg_run] : 0;
    const uint16_t *sub = SUB ? k_sub_list + (size_t)wg_run * kB : nullptr;      // (the lists' stride is the bone count)
    const int lrows = SUB ? ns * 3 : rows;          // float4 per pose of the finished LDS palettes
    float4 *stage = pal + (size_t)G * ns * 3;       // SUB, one-launch frame: staged world matrices sit behind the palette region
    if constexpr (SUB) {
        // prep-kernel / device-FK path (dma): the listed bones' finished rows, 3 float4 per bone, straight to their place.
        // one-launch frame: the listed bones' world matrices, 4 float4 per bone, into the staging region.
        // Either way element e of the linear LDS image is (pose g, slot s, cell k): a per-lane global address, a linear LDS one.
        const int epb = k_dma ? 3 : 4;
        const int n = RZ_DBG(p) == 8 ? 0 : ng * ns * epb;           // dbg 8 (tools-only build): neither staging nor product
        const float4 *src = k_src + (size_t)inst0 * kB * (k_dma ? 3 : 4);
        float4 *dst = k_dma ? pal : stage;
        for (int c = wave * 64; c < n; c += BLOCK) {
            const int e = c + lane;
            if (e < n) {
                const int gs = k_dma ? e / 3 : e >> 2, k = e - gs * epb;
                const int g = gs / ns, sl = gs - g * ns;
                const float4 *a = src + ((size_t)g * kB + sub[sl]) * epb + k;
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)a, (lptr_t)(uint32_t)(uintptr_t)(dst + c), 16, 0, 0);
            }
        }
    } else {
        // prep-kernel path (dma): the group's finished palettes are contiguous in global memory -> one linear LDS-DMA copy.
        // one-launch frame (!dma): the group's WORLD matrices are staged instead, whole (64-byte slots, the same linear
        // 16-byte LDS-DMA); the conversion pass below multiplies by the inverse bind matrix and re-packs the rows to the
        // same 48-byte stride. (Leaving them in the 64-byte slots made every fourth bone share its LDS banks: 52 % of the
        // skin loop's LDS cycles were bank conflicts, against 11 % at 48 bytes — profiles/r2_sq_counters_c4.txt.)
        const float4 *src = k_src + (size_t)inst0 * kB * (k_dma ? 3 : 4);
        const int n = (RZ_DBG(p) == 6 || RZ_DBG(p) == 7 || RZ_DBG(p) == 8) ? 0 : (k_dma ? ng * rows : ng * kB * 4);   // dbg 6 / 7 (tools-only build): no palette staging
        for (int c = wave * 64; c < n; c += BLOCK) {
            const int e = c + lane;
            if (e < n) {
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + e), (lptr_t)(uint32_t)(uintptr_t)(pal + c), 16, 0, 0);
            }
        }
    }
    // One-launch frame (the host only plans it for B <= BLOCK): which (bone, pose stripe) this thread converts. With
    // B <= BLOCK / 2 the spare threads take a second, third ... stripe of the group's poses (stripe s converts poses s,
    // s + stripes, ...): 200 bones on 512 threads = 2 stripes.
    // SUB: the same mapping over the run's ns listed bones (the host plans the form only for ns <= BLOCK).
    const int cvB = SUB ? max(ns, 1) : kB;
    const int stripes = !k_dma ? max(1, min(ng, BLOCK / cvB)) : 1;
    const int cv_b0 = tid % cvB;
    const int cv_g0 = tid / cvB;
    const bool cv_on = !k_dma && cv_g0 < stripes && (!SUB || ns > 0);
    float4 ib0 = {0, 0, 0, 0}, ib1 = ib0, ib2 = ib0, ib3 = ib0;
    if (cv_on) {                                    // requested first: lands while the staging copy is in flight
        const float4 *gi = reinterpret_cast<const float4 *>(k_inv_bind) + (SUB ? (int)sub[cv_b0] : cv_b0) * 4;
        ib0 = gi[0]; ib1 = gi[1]; ib2 = gi[2]; ib3 = gi[3];
    }
    const size_t Vp = k_Vp;
    const uint32_t v_begin = wg_run * verts_per_wg;
    const uint32_t v_end = min(p.n_quads * 4u, v_begin + verts_per_wg);
    const uint32_t bmax = (uint32_t)(p.B - 1);
    // software-pipelined vertex loop: the next vertex's nine attribute loads are issued before the current
    // vertex's pose loop, so their L2 latency hides behind the poses' LDS gathers + FMA
    // (Round 4 tried two ways of sharing the run's last, partial step (a 3 750-vertex run is 7 steps of 512 and one of 166) among
    // all waves — its vertices dealt out in quarter-wave pieces: 33.5 -> 36.3 us; its poses split between the waves that hold the
    // same piece: 33.0 -> 33.7 us. The waves of a workgroup end up to 5.7 us apart (profiles/r4_timeline_c4.txt), but not because
    // of that step: NOTEBOOK.md R4.5. Both removed.)
    auto vert_of = [&](const uint32_t vb) { return vb + (uint32_t)tid; };
    uint32_t v = vert_of(v_begin);
    float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0;
    uint32_t j01 = 0, j23 = 0, wq = 0;
    const uint32_t *jp01 = SUB ? p.rj01 : p.joints01, *jp23 = SUB ? p.rj23 : p.joints23;     // SUB: joints as slots of the run's list
    if (v < v_end) {
        x = p.geom[0 * Vp + v]; y = p.geom[1 * Vp + v]; z = p.geom[2 * Vp + v];
        nx = p.geom[3 * Vp + v]; ny = p.geom[4 * Vp + v]; nz = p.geom[5 * Vp + v];
        j01 = jp01[v]; j23 = jp23[v]; wq = p.weights[v];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // palettes / world matrices (and the first vertex) have landed
    RZ_STAMP(1);                 // staged matrices have landed
    __syncthreads();
    if constexpr (SUB) {
        if (!p.dma && RZ_DBG(p) != 8) {
            // palette rows of the listed bones: slot (g, s) = rows 0..2 of world * inverseBind (engine.ts:926-928), the same
            // packed FMA chain as below and as rz_prep_kernel — out of the staging region, into the palette region: no hazard,
            // one barrier. With ns ~ 34 and 512 threads every (pose, bone) pair has a thread of its own.
            const f2 px01 = {ib0.x, ib1.x}, px23 = {ib2.x, ib3.x}, py01 = {ib0.y, ib1.y}, py23 = {ib2.y, ib3.y};
            const f2 pz01 = {ib0.z, ib1.z}, pz23 = {ib2.z, ib3.z}, pw01 = {ib0.w, ib1.w}, pw23 = {ib2.w, ib3.w};
            auto row = [&](float e0, float e1, float e2, float e3, const f2 &bx, const f2 &by, const f2 &bz, const f2 &bw) {
                return pk_fma(f2{e3, e3}, bw, pk_fma(f2{e2, e2}, bz, pk_fma(f2{e1, e1}, by, f2{e0, e0} * bx)));
            };
            if (cv_on)
                for (int g = cv_g0; g < ng; g += stripes) {
                    const float4 *slot = stage + ((size_t)g * ns + cv_b0) * 4;
                    const float4 a0 = slot[0], a1 = slot[1], a2 = slot[2], a3 = slot[3];      // the world matrix's columns
                    const f2 r0 = row(a0.x, a1.x, a2.x, a3.x, px01, py01, pz01, pw01), r1 = row(a0.x, a1.x, a2.x, a3.x, px23, py23, pz23, pw23);
                    const f2 r2 = row(a0.y, a1.y, a2.y, a3.y, px01, py01, pz01, pw01), r3 = row(a0.y, a1.y, a2.y, a3.y, px23, py23, pz23, pw23);
                    const f2 r4 = row(a0.z, a1.z, a2.z, a3.z, px01, py01, pz01, pw01), r5 = row(a0.z, a1.z, a2.z, a3.z, px23, py23, pz23, pw23);
                    float4 *dst = pal + ((size_t)g * ns + cv_b0) * 3;
                    dst[0] = make_float4(r0.x, r0.y, r1.x, r1.y);
                    dst[1] = make_float4(r2.x, r2.y, r3.x, r3.y);
                    dst[2] = make_float4(r4.x, r4.y, r5.x, r5.y);
                }
            __syncthreads();
            // (the skinMatrixBuffer is not written here: a workgroup only holds its run's bones. rz_read_palette forms it on
            // demand with rz_prep_kernel — the same chain, the same bits.)
        }
    } else
    if (!p.dma && RZ_DBG(p) != 8) {                  // dbg 8 (tools-only build): neither staging nor conversion
        // in-place conversion: slot (g, b) = rows 0..2 of world * inverseBind (engine.ts:926-928). Packed math: the
        // inverse bind matrix is held as column PAIRS (c0,c1),(c2,c3) per k, so each result row is two v_pk_fma
        // chains (the same chain as rz_prep_kernel: ((a0*b0 + a1*b1) + a2*b2) + a3*b3); the next pose's cells are read
        // before the current product is formed.
        // Rounds of up to four poses per thread: read the staged world matrices (64-byte slots) and form the palette rows
        // in registers (engine.ts:926-928, packed math, the same FMA chain as rz_prep_kernel: ((a0*b0 + a1*b1) + a2*b2) +
        // a3*b3) -> barrier -> write them back at the 48-byte stride. Poses ascend, and the compact rows of pose g only
        // ever land on staged matrices of poses <= g, which every thread has read by then.
        const f2 px01 = {ib0.x, ib1.x}, px23 = {ib2.x, ib3.x}, py01 = {ib0.y, ib1.y}, py23 = {ib2.y, ib3.y};
        const f2 pz01 = {ib0.z, ib1.z}, pz23 = {ib2.z, ib3.z}, pw01 = {ib0.w, ib1.w}, pw23 = {ib2.w, ib3.w};
        auto row = [&](float e0, float e1, float e2, float e3, const f2 &bx, const f2 &by, const f2 &bz, const f2 &bw) {
            return pk_fma(f2{e3, e3}, bw, pk_fma(f2{e2, e2}, bz, pk_fma(f2{e1, e1}, by, f2{e0, e0} * bx)));
        };
        constexpr int PR = 4;                               // poses per thread per round
        for (int g_base = 0; g_base < ng; g_base += PR * stripes) {       // workgroup-uniform trip count
            f2 res[PR][6];
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                const int g = g_base + cv_g0 + i * stripes;
                if (cv_on && g < ng) {
                    const float4 *slot = pal + ((size_t)g * p.B + cv_b0) * 4;
                    const float4 a0 = slot[0], a1 = slot[1], a2 = slot[2], a3 = slot[3];      // the world matrix's columns
                    res[i][0] = row(a0.x, a1.x, a2.x, a3.x, px01, py01, pz01, pw01); res[i][1] = row(a0.x, a1.x, a2.x, a3.x, px23, py23, pz23, pw23);
                    res[i][2] = row(a0.y, a1.y, a2.y, a3.y, px01, py01, pz01, pw01); res[i][3] = row(a0.y, a1.y, a2.y, a3.y, px23, py23, pz23, pw23);
                    res[i][4] = row(a0.z, a1.z, a2.z, a3.z, px01, py01, pz01, pw01); res[i][5] = row(a0.z, a1.z, a2.z, a3.z, px23, py23, pz23, pw23);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                const int g = g_base + cv_g0 + i * stripes;
                if (cv_on && g < ng) {
                    float4 *dst = pal + ((size_t)g * p.B + cv_b0) * 3;
                    dst[0] = make_float4(res[i][0].x, res[i][0].y, res[i][1].x, res[i][1].y);
                    dst[1] = make_float4(res[i][2].x, res[i][2].y, res[i][3].x, res[i][3].y);
                    dst[2] = make_float4(res[i][4].x, res[i][4].y, res[i][5].x, res[i][5].y);
                }
            }
        }
        __syncthreads();
        if (p.palette) {
            // keep the skinMatrixBuffer observable (rz_read_palette): the vertex runs of a pose group each copy one
            // slice of the finished palettes out of LDS, coalesced
            const int n = ng * rows, per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
            const int lo = (int)wg_run * per, hi = min(n, lo + per);
            float4 *gp = p.palette + (size_t)inst0 * rows;
            for (int i = lo + tid; i < hi; i += BLOCK) gp[i] = pal[i];
        }
    }
    RZ_STAMP(2);                 // palettes formed and published: the front is over
    for (uint32_t vb = v_begin; vb < v_end; vb += BLOCK) {   // workgroup-uniform trip count (the ballots below need whole waves)
        if (vb == v_begin + BLOCK) RZ_STAMP(3);      // first vertex step done (8 poses written)
        const uint32_t vn = vert_of(vb + BLOCK);
        float xn = 0, yn = 0, zn = 0, nxn = 0, nyn = 0, nzn = 0;
        uint32_t j01n = 0, j23n = 0, wqn = 0;
        if (vn < v_end) {
            xn = p.geom[0 * Vp + vn]; yn = p.geom[1 * Vp + vn]; zn = p.geom[2 * Vp + vn];
            nxn = p.geom[3 * Vp + vn]; nyn = p.geom[4 * Vp + vn]; nzn = p.geom[5 * Vp + vn];
            j01n = jp01[vn]; j23n = jp23[vn]; wqn = p.weights[vn];
        }
        const bool live = v < v_end;
        // decode once per vertex (engine.ts:255-258)
        const uint32_t b0 = wq & 255u, b1 = (wq >> 8) & 255u, b2 = (wq >> 16) & 255u, b3 = wq >> 24;
        const uint32_t isum = b0 + b1 + b2 + b3;
        const bool ok = isum != 0u;
        const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
        const float w0 = ok ? (float)b0 * inv : 1.0f, w1 = (float)b1 * inv, w2 = (float)b2 * inv, w3 = (float)b3 * inv;
        const uint32_t jmax = SUB ? 0xffffu : bmax;     // SUB: slots are in range by construction
        const uint32_t o0 = min(j01 & 0xffffu, jmax) * rstride, o1 = min(j01 >> 16, jmax) * rstride,
                       o2 = min(j23 & 0xffffu, jmax) * rstride, o3 = min(j23 >> 16, jmax) * rstride;
        float *dp = p.out_pos + ((size_t)inst0 * Vp + v) * 3;
        float *dn = p.out_nrm + ((size_t)inst0 * Vp + v) * 3;
        // Packed-math form (v_pk_fma_f32 = two f32 FMAs per lane per instruction): palette rows are blended as
        // (xy),(zw) register pairs straight out of ds_read_b128, and position + normal are transformed together
        // as the pairs (x,nx),(y,ny),(z,nz), so one FMA chain yields (p_r, n_r) for row r. Every chain is spelled out
        // with explicit FMAs in the order of skin_vertex() above — bones ascending from w0 * row, then
        // fma(m.z, z, fma(m.y, y, fma(m.x, x, m.w))) — so a pose of a crowd has the SAME BITS as that pose run alone
        // through rz_deform_kernel (tests/test_gpu_round2.py checks it at full C4 size).
        const f2 vx = {x, nx}, vy = {y, ny}, vz = {z, nz};
        const f2 W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3};
#ifdef RZ_ABLATE
        if (p.dbg == 2 || p.dbg == 7) {          // dbg 2 / 7: ablation — the output stream without gathers / math
            if (live)
                for (int g = 0; g < ng; ++g) {
                    st3<NTS>(dp, x, y, z); st3<NTS>(dn, nx, ny, nz);
                    dp += Vp * 3; dn += Vp * 3;
                }
            x = xn; y = yn; z = zn; nx = nxn; ny = nyn; nz = nzn; j01 = j01n; j23 = j23n; wq = wqn;
            v = vn;
            continue;
        }
#endif
        auto pose_loop = [&](auto nb_tag) {
            constexpr int NB = decltype(nb_tag)::value;
            const float4 *pg = pal;
#pragma unroll 2
            for (int g = 0; g < ng; ++g) {
                f2 r[3][2];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 a = pg[o0 + k];
                    r[k][0] = W0 * f2{a.x, a.y};
                    r[k][1] = W0 * f2{a.z, a.w};
                    if (NB >= 2) {
                        const float4 c = pg[o1 + k];
                        r[k][0] = pk_fma(W1, f2{c.x, c.y}, r[k][0]);
                        r[k][1] = pk_fma(W1, f2{c.z, c.w}, r[k][1]);
                    }
                    if (NB >= 4) {
                        const float4 d = pg[o2 + k], e = pg[o3 + k];
                        r[k][0] = pk_fma(W3, f2{e.x, e.y}, pk_fma(W2, f2{d.x, d.y}, r[k][0]));
                        r[k][1] = pk_fma(W3, f2{e.z, e.w}, pk_fma(W2, f2{d.z, d.w}, r[k][1]));
                    }
                }
                // (p_r, t_r) = fma(m_r.z, (z,nz), fma(m_r.y, (y,ny), fma(m_r.x, (x,nx), (m_r.w, 0))))
                f2 q[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f2 m_xy = r[k][0], m_zw = r[k][1];
                    q[k] = pk_fma(f2{m_zw.x, m_zw.x}, vz, pk_fma(f2{m_xy.y, m_xy.y}, vy, pk_fma(f2{m_xy.x, m_xy.x}, vx, f2{m_zw.y, 0.0f})));
                }
                const float tx = q[0].y, ty = q[1].y, tz = q[2].y;
                const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
                const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
                if (live && (RZ_DBG(p) != 1 || l2 == 1234.5f)) {   // dbg 1 (tools-only build): compute without the output stream
                    st3<NTS>(dp, q[0].x, q[1].x, q[2].x);
                    st3<NTS>(dn, good ? tx * rl : nx, good ? ty * rl : ny, good ? tz * rl : nz);
                }
                pg += lrows;
                dp += Vp * 3;
                dn += Vp * 3;
            }
        };
        const bool any34 = __ballot((wq >> 16) != 0u) != 0ull;
        const bool any2 = __ballot(((wq >> 8) & 255u) != 0u) != 0ull;
        if (__ballot(live) == 0ull) {}                     // a wave past the end of the run (last step only)
        else if (any34) pose_loop(std::integral_constant<int, 4>{});
        else if (any2) pose_loop(std::integral_constant<int, 2>{});
        else pose_loop(std::integral_constant<int, 1>{});
        x = xn; y = yn; z = zn; nx = nxn; ny = nyn; nz = nzn; j01 = j01n; j23 = j23n; wq = wqn;
        v = vn;
    }
    RZ_STAMP(5);                 // last vertex step issued
    RZ_TL_FLUSH(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / 64) + wave);
}

#ifdef RZ_ALL_VARIANTS
// ------------------------------------------------------------------------------------------------
// instanced skin, register-resident form: a workgroup owns a run of KV*256 vertices and a RANGE of poses.
// Each lane loads and decodes its KV vertices ONCE into registers (the static mesh is read once per
// (run, pose range) instead of once per pose), then walks the poses: the palette of pose g+1 streams into
// the other half of a 2-deep LDS ring by LDS-DMA while pose g is skinned, and every pose is written as one
// contiguous KV*256*12-byte block per output array. LDS is only 2 palettes (19 KB at 200 bones).
// grid = (vertex runs, pose ranges); block = 256.
// ------------------------------------------------------------------------------------------------
template <int KV, bool NTS>
__global__ void __launch_bounds__(kBlock) rz_skin_instances_reg_kernel(const RzDeformParams p, int n_inst, int poses_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *ring = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = p.B * 3;
    const int inst0 = blockIdx.y * poses_per_wg;
    const int ng = min(poses_per_wg, n_inst - inst0);
    const size_t Vp = p.Vp;
    const uint32_t v_lim = p.n_quads * 4u;
    const uint32_t v0 = blockIdx.x * (KV * kBlock) + tid;
    const uint32_t bmax = (uint32_t)(p.B - 1);

    auto dma_palette = [&](int g) {
        const float4 *src = p.palette + (size_t)(inst0 + g) * rows;
        float4 *dst = ring + (size_t)(g & 1) * rows;
        for (int c = wave * 64; c < rows; c += kBlock) {
            const int e = c + lane;
            if (e < rows) {
                typedef const __attribute__((address_space(1))) void *gptr_t;
                typedef __attribute__((address_space(3))) void *lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(uintptr_t)(src + e), (lptr_t)(uint32_t)(uintptr_t)(dst + c), 16, 0, 0);
            }
        }
    };
    dma_palette(0);

    // ---- decode KV vertices per lane, once ----
    f2 vx[KV], vy[KV], vz[KV];
    float w0[KV], w1[KV], w2[KV], w3[KV];
    uint32_t j01[KV], j23[KV];
    {
        float x[KV], y[KV], z[KV], nx[KV], ny[KV], nz[KV];
        uint32_t wq[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t v = v0 + k * kBlock;
            const bool live = v < v_lim;
            const size_t vs = live ? v : 0;
            x[k] = p.geom[0 * Vp + vs]; y[k] = p.geom[1 * Vp + vs]; z[k] = p.geom[2 * Vp + vs];
            nx[k] = p.geom[3 * Vp + vs]; ny[k] = p.geom[4 * Vp + vs]; nz[k] = p.geom[5 * Vp + vs];
            j01[k] = p.joints01[vs]; j23[k] = p.joints23[vs]; wq[k] = p.weights[vs];
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t b0 = wq[k] & 255u, b1 = (wq[k] >> 8) & 255u, b2 = (wq[k] >> 16) & 255u, b3 = wq[k] >> 24;
            const uint32_t isum = b0 + b1 + b2 + b3;
            const bool ok = isum != 0u;
            const float inv = __builtin_amdgcn_rcpf((float)(ok ? isum : 1u));
            w0[k] = ok ? (float)b0 * inv : 1.0f; w1[k] = (float)b1 * inv; w2[k] = (float)b2 * inv; w3[k] = (float)b3 * inv;
            // joints -> clamped palette row offsets, two 16-bit fields per register (B*3 <= 65535 is checked on the host)
            const uint32_t o0 = min(j01[k] & 0xffffu, bmax) * 3u, o1 = min(j01[k] >> 16, bmax) * 3u;
            const uint32_t o2 = min(j23[k] & 0xffffu, bmax) * 3u, o3 = min(j23[k] >> 16, bmax) * 3u;
            j01[k] = o0 | (o1 << 16); j23[k] = o2 | (o3 << 16);
            vx[k] = f2{x[k], nx[k]}; vy[k] = f2{y[k], ny[k]}; vz[k] = f2{z[k], nz[k]};
        }
    }
    const f2 vw = {1.0f, 0.0f};

    for (int g = 0; g < ng; ++g) {
        // pose g's palette has landed (own DMA drained, barrier publishes everyone's part and also retires
        // every wave's reads of the other ring slot, which pose g+1 may now overwrite)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g + 1 < ng) dma_palette(g + 1);
        const float4 *pg = ring + (size_t)(g & 1) * rows;
        float *op = p.out_pos + (size_t)(inst0 + g) * Vp * 3;
        float *on = p.out_nrm + (size_t)(inst0 + g) * Vp * 3;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t v = v0 + k * kBlock;
            const uint32_t o0 = j01[k] & 0xffffu, o1 = j01[k] >> 16, o2 = j23[k] & 0xffffu, o3 = j23[k] >> 16;
            f2 r[3][2];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 a = pg[o0 + q], c = pg[o1 + q], d = pg[o2 + q], e = pg[o3 + q];
                const f2 axy = {a.x, a.y}, azw = {a.z, a.w}, cxy = {c.x, c.y}, czw = {c.z, c.w};
                const f2 dxy = {d.x, d.y}, dzw = {d.z, d.w}, exy = {e.x, e.y}, ezw = {e.z, e.w};
                r[q][0] = w3[k] * exy + (w2[k] * dxy + (w1[k] * cxy + w0[k] * axy));
                r[q][1] = w3[k] * ezw + (w2[k] * dzw + (w1[k] * czw + w0[k] * azw));
            }
            const f2 q0 = r[0][0].x * vx[k] + (r[0][0].y * vy[k] + (r[0][1].x * vz[k] + r[0][1].y * vw));
            const f2 q1 = r[1][0].x * vx[k] + (r[1][0].y * vy[k] + (r[1][1].x * vz[k] + r[1][1].y * vw));
            const f2 q2 = r[2][0].x * vx[k] + (r[2][0].y * vy[k] + (r[2][1].x * vz[k] + r[2][1].y * vw));
            const float tx = q0.y, ty = q1.y, tz = q2.y;
            const float l2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
            const bool good = (l2 > 0.0f) && (l2 < __builtin_inff());
            const float rl = __builtin_amdgcn_rsqf(good ? l2 : 1.0f);
            if (v < v_lim) {
                st3<NTS>(op + (size_t)v * 3, q0.x, q1.x, q2.x);
                st3<NTS>(on + (size_t)v * 3, good ? tx * rl : vx[k].y, good ? ty * rl : vy[k].y, good ? tz * rl : vz[k].y);
            }
            __builtin_amdgcn_sched_barrier(0);   // one vertex at a time: bounds the live palette rows (12 x float4)
        }
    }
}

// Test hook of the tools-only build (tests/conftest.py: rzv): a one-thread kernel that holds its stream until the host opens
// the gate (a word in pinned memory) — so a test can put frames BEHIND it, write the next pose, and only then let them run:
// the pose-prefetch helper then finds the next pose complete by construction, not because the host happened to be ahead.
// Gives up after two seconds of the 100 MHz counter: a test that dies with the gate closed must not take the GPU with it.
__global__ void rz_gate_kernel(const uint32_t *flag)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_nontemporal_load(flag) == 0u) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

#endif  // RZ_ALL_VARIANTS

// ------------------------------------------------------------------------------------------------
// upload-time re-layout kernels (one-off, not on the per-frame path)
// ------------------------------------------------------------------------------------------------
// packed [n][stride] floats -> planes; `stride` = 3 (packed xyz) or 8 (reference interleaved vertex)
__global__ void rz_deinterleave_kernel(const float *src, int stride, int offset, uint32_t n, float *px,
                                       float *py, float *pz)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float *s = src + (size_t)v * stride + offset;
    px[v] = s[0]; py[v] = s[1]; pz[v] = s[2];
}

__global__ void rz_pack_skinning_kernel(const uint16_t *joints4, const uint8_t *weights4, uint32_t n,
                                        uint32_t *j01, uint32_t *j23, uint32_t *wq)
{
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint2 j = reinterpret_cast<const uint2 *>(joints4)[v];
    j01[v] = j.x; j23[v] = j.y;
    wq[v] = reinterpret_cast<const uint32_t *>(weights4)[v];
}

// Plan-time pass of the bone-subset crowd frame (one-off per launch shape, not per frame): one workgroup per vertex run.
// Marks the bones the run's vertices name (all four joints of every vertex, clamped to B - 1 like the skin kernels do — a
// zero-weight influence still gathers its bone's rows, so it stays the SAME bone: fma(0, row, m) keeps m's bits only while the
// row is the one the full-palette form would have read), ranks them ascending, writes the list and the joints as slots of it.
__global__ void __launch_bounds__(kBlock) rz_run_subsets_kernel(const uint32_t *j01, const uint32_t *j23, uint32_t v_lim, uint32_t per,
                                                                uint32_t B, uint16_t *list, uint32_t *count, uint32_t *rj01, uint32_t *rj23)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t nw = (B + 31) / 32;
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem);            // [nw] bone bitmap
    uint32_t *before = bits + nw;                                    // [nw + 1] listed bones in front of each word
    uint16_t *slot = reinterpret_cast<uint16_t *>(before + nw + 1);  // [B]
    const uint32_t tid = threadIdx.x, run = blockIdx.x;
    const uint32_t v0 = run * per, v1 = min(v_lim, v0 + per), bmax = B - 1;
    for (uint32_t i = tid; i < nw; i += kBlock) bits[i] = 0;
    __syncthreads();
    for (uint32_t v = v0 + tid; v < v1; v += kBlock) {
        const uint32_t a = j01[v], b = j23[v];
        const uint32_t j[4] = { min(a & 0xffffu, bmax), min(a >> 16, bmax), min(b & 0xffffu, bmax), min(b >> 16, bmax) };
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicOr(&bits[j[k] >> 5], 1u << (j[k] & 31u));
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < nw; ++i) { before[i] = acc; acc += __popc(bits[i]); }
        before[nw] = acc;
        count[run] = acc;
    }
    __syncthreads();
    for (uint32_t b = tid; b < B; b += kBlock) {
        const uint32_t w = bits[b >> 5], m = 1u << (b & 31u);
        if (w & m) {
            const uint32_t sl = before[b >> 5] + __popc(w & (m - 1u));
            slot[b] = (uint16_t)sl;
            list[(size_t)run * B + sl] = (uint16_t)b;
        }
    }
    __syncthreads();
    for (uint32_t v = v0 + tid; v < v1; v += kBlock) {
        const uint32_t a = j01[v], b = j23[v];
        rj01[v] = (uint32_t)slot[min(a & 0xffffu, bmax)] | ((uint32_t)slot[min(a >> 16, bmax)] << 16);
        rj23[v] = (uint32_t)slot[min(b & 0xffffu, bmax)] | ((uint32_t)slot[min(b >> 16, bmax)] << 16);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by reze_deform.cpp)
// ------------------------------------------------------------------------------------------------
hipError_t rz_launch_prep(const RzPrepParams &p, uint32_t instances, hipStream_t st)
{
    hipLaunchKernelGGL(rz_prep_kernel, dim3(instances), dim3(kBlock), 0, st, p);
    return hipGetLastError();
}

size_t rz_fk_lds_bytes(const RzFkParams &p)
{
    size_t lds = (size_t)p.B * 48 + rz_fk_scratch_bytes(p.B);
    if (p.bm_off) lds += (size_t)std::max(std::max(p.bm_M, p.sample.M), 1) * 4;      // the pose's morph weights, for the bone morphs
    return lds;
}

hipError_t rz_launch_fk(const RzFkParams &p, uint32_t instances, hipStream_t st)
{
    const size_t lds = rz_fk_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;      // (the host checks first and says why: launch_fk in reze_deform.cpp)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rz_fk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rz_fk_kernel, dim3(instances), dim3(kBlock), lds, st, p);
    return hipGetLastError();
}

size_t rz_deform_lds_bytes(const RzDeformParams &p, const RzVariant &v)
{
    const size_t vw = 256 / v.S;   // vertices per wave per tile
    size_t scratch = (size_t)(kBlock / 64) * (v.geo ? 9 : 3) * vw * 4;
    const size_t list = (v.mode == 2 || (!v.fast && v.mode == 1)) ? (size_t)p.Mpad * 8 : 0;
    size_t work = scratch + (size_t)(kBlock / 64) * p.out_cap * 24 + (v.mode == 2 ? (size_t)(kBlock / 64) * p.sp_cap * 16 : 0);     // sparse: staged CSR pieces
    if (p.fk_on) work = std::max(work, rz_fk_scratch_bytes(p.B) + (size_t)std::max(p.M, 1) * 4 + 16);   // the fused solve's scratch aliases it
    return (size_t)p.B * 48 + list + work;
}

uint32_t rz_quads_per_tile(int S) { return (kBlock / 64) * (64 / S); }

template <int S, int U, int MODE, bool NT, bool NTS, bool GEO, bool FAST>
static hipError_t launch_one(const RzDeformParams &p, const RzMorphList &ml, dim3 grid, size_t lds, hipStream_t st)
{
    auto k = rz_deform_kernel<S, U, MODE, NT, NTS, GEO, FAST>;
    if (p.B > 0xffff) return hipErrorInvalidValue;      // k_bf carries the bone count in 16 bits (the 48 B per bone LDS palette keeps it far below today)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // the leading arguments (kernel-argument preload: see the kernel) repeat fields of `p`; k_world is the pose the kernel asks for
    // FIRST — the copy a helper may have staged when the frame looks for one, else p.world
    const float *k_world = (FAST && p.st_tag) ? p.st_world : p.world;
    const uint32_t workers = grid.x - (p.pf_src ? 1u : 0u);
    const uint32_t k_bf = (uint32_t)p.B | (p.pf_src ? 1u << 16 : 0u) | (p.st_tag ? 1u << 17 : 0u) | (p.world_copy ? 1u << 18 : 0u) | (workers < 8192u ? workers << 19 : 0u);
    hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, p.geom, k_world, p.inv_bind, k_bf, p.Vp, p.n_quads, p.quads_per_wave, p.joints01, p.joints23, p.weights, p, ml);
    return hipGetLastError();
}

// Which variants the library carries. The PRODUCT instantiates what a plan can select by default or through rz_autotune:
// rest geometry by 4-byte loads (GEO = false), nontemporal morph loads (NT = true, dense mode), 8 morphs in flight (U = 8) —
// 16 + 8 + 8 kernels instead of 160. The variants measured slower everywhere (GEO = true, NT = false, U = 4: profiles/r1_*sweep*)
// live in the tools-only build (-DRZ_ALL_VARIANTS, `make variants`), where the parity tests still cover every one of them;
// in the product rz_set_tuning refuses the keys that would select them.
#ifdef RZ_ALL_VARIANTS
constexpr bool kAllVariants = true;
#else
constexpr bool kAllVariants = false;
#endif

template <int S, int U, int MODE, bool NT, bool NTS>
static hipError_t launch_gf(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds,
                            hipStream_t st)
{
    if constexpr (kAllVariants) {
        if (v.geo) return v.fast ? launch_one<S, U, MODE, NT, NTS, true, true>(p, ml, grid, lds, st)
                                 : launch_one<S, U, MODE, NT, NTS, true, false>(p, ml, grid, lds, st);
    } else if (v.geo) return hipErrorInvalidValue;
    return v.fast ? launch_one<S, U, MODE, NT, NTS, false, true>(p, ml, grid, lds, st)
                  : launch_one<S, U, MODE, NT, NTS, false, false>(p, ml, grid, lds, st);
}

template <int S, int U, int MODE>
static hipError_t launch_nt(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds,
                            hipStream_t st)
{
    if constexpr (MODE == 1) {           // the nontemporal load hint only exists on the dense morph stream
        if (v.nt || !kAllVariants) {
            if (!v.nt) return hipErrorInvalidValue;
            return v.nts ? launch_gf<S, U, MODE, true, true>(p, ml, v, grid, lds, st)
                         : launch_gf<S, U, MODE, true, false>(p, ml, v, grid, lds, st);
        }
    }
    if constexpr (MODE != 1 || kAllVariants) {
        return v.nts ? launch_gf<S, U, MODE, false, true>(p, ml, v, grid, lds, st)
                     : launch_gf<S, U, MODE, false, false>(p, ml, v, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

template <int S>
static hipError_t launch_dense(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, dim3 grid, size_t lds,
                               hipStream_t st)
{
    if (v.U >= 8) return launch_nt<S, 8, 1>(p, ml, v, grid, lds, st);
    if constexpr (kAllVariants) return launch_nt<S, 4, 1>(p, ml, v, grid, lds, st);
    return hipErrorInvalidValue;
}

bool rz_has_all_variants() { return kAllVariants; }

hipError_t rz_launch_deform(const RzDeformParams &p, const RzMorphList &ml, const RzVariant &v, uint32_t grid_x,
                            uint32_t instances, hipStream_t st)
{
    const size_t lds = rz_deform_lds_bytes(p, v);
    dim3 grid(grid_x, instances);
    if (v.mode == 0) return v.S == 4 ? launch_nt<4, 1, 0>(p, ml, v, grid, lds, st) : launch_nt<1, 1, 0>(p, ml, v, grid, lds, st);
    if (v.mode == 2) return v.S == 4 ? launch_nt<4, 1, 2>(p, ml, v, grid, lds, st) : launch_nt<1, 1, 2>(p, ml, v, grid, lds, st);
    switch (v.S) {
    case 2: return launch_dense<2>(p, ml, v, grid, lds, st);
    case 4: return launch_dense<4>(p, ml, v, grid, lds, st);
    case 8: return launch_dense<8>(p, ml, v, grid, lds, st);
    default: return launch_dense<1>(p, ml, v, grid, lds, st);
    }
}

size_t rz_skin_instances_lds_bytes(int G, uint32_t bones, bool dma, bool subsets)
{
    // finished palettes are 48 B per (pose, bone). One-launch frame: the whole-palette form stages the world matrices in
    // the palette region's place (64-byte slots, re-packed in place); the subset form stages them behind it (48 + 64).
    const size_t per = dma ? 48 : (subsets ? 112 : 64);
    return (size_t)G * bones * per;
}

template <int BLOCK>
static hipError_t launch_skin_instances(const RzDeformParams &p, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                        bool nts, size_t lds, hipStream_t st)
{
    const bool sub = p.sub_list != nullptr;
    auto k = sub ? (nts ? rz_skin_instances_kernel<BLOCK, true, true> : rz_skin_instances_kernel<BLOCK, false, true>)
                 : (nts ? rz_skin_instances_kernel<BLOCK, true, false> : rz_skin_instances_kernel<BLOCK, false, false>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(grid_x, (n_inst + G - 1) / G);
    if (grid.x > 0xffffu || grid.y > 0xffffu || p.B > 0xffff || (sub && p.sub_stride != p.B)) return hipErrorInvalidValue;      // (k_grid packs both; the lists' stride is the bone count)
    // leading arguments = what the front of a workgroup needs, preloaded into SGPRs (see the kernel)
    const float4 *k_src = p.dma ? p.palette : reinterpret_cast<const float4 *>(p.world);
    const uint32_t k_grid = grid.x | (grid.y << 16), k_bf = (uint32_t)p.B | (p.inst_order ? 1u << 16 : 0u) | (p.dma ? 1u << 17 : 0u);
    hipLaunchKernelGGL(k, grid, dim3(BLOCK), lds, st, p.sub_count, p.sub_list, k_src, p.inv_bind, G, n_inst, verts_per_wg, k_grid, k_bf, p.Vp, p);
    return hipGetLastError();
}

hipError_t rz_launch_skin_instances(const RzDeformParams &p, int G, int n_inst, uint32_t verts_per_wg, uint32_t grid_x,
                                    int block, bool nts, size_t lds_bytes, hipStream_t st)
{
    if (block == 1024) return launch_skin_instances<1024>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    if (block == 512) return launch_skin_instances<512>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
    return launch_skin_instances<256>(p, G, n_inst, verts_per_wg, grid_x, nts, lds_bytes, st);
}

hipError_t rz_launch_run_subsets(const uint32_t *j01, const uint32_t *j23, uint32_t v_lim, uint32_t per, uint32_t runs, uint32_t B,
                                 uint16_t *list, uint32_t *count, uint32_t *rj01, uint32_t *rj23, hipStream_t st)
{
    if (runs == 0) return hipSuccess;
    const uint32_t nw = (B + 31) / 32;
    const size_t lds = (size_t)(2 * nw + 1) * 4 + (size_t)B * 2;
    hipLaunchKernelGGL(rz_run_subsets_kernel, dim3(runs), dim3(kBlock), lds, st, j01, j23, v_lim, per, B, list, count, rj01, rj23);
    return hipGetLastError();
}

hipError_t rz_launch_skin_instances_reg(const RzDeformParams &p, int n_inst, int poses_per_wg, uint32_t grid_x, bool nts,
                                        hipStream_t st)
{
#ifdef RZ_ALL_VARIANTS
    constexpr int KV = 8;
    const size_t lds = (size_t)2 * p.B * 48;
    auto k = nts ? rz_skin_instances_reg_kernel<KV, true> : rz_skin_instances_reg_kernel<KV, false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(grid_x, (n_inst + poses_per_wg - 1) / poses_per_wg);
    hipLaunchKernelGGL(k, grid, dim3(kBlock), lds, st, p, n_inst, poses_per_wg);
    return hipGetLastError();
#else
    (void)p; (void)n_inst; (void)poses_per_wg; (void)grid_x; (void)nts; (void)st;
    return hipErrorInvalidValue;       // the register-resident crowd kernel (measured slower, inst_loop = 9) is a tools-only variant
#endif
}

#ifdef RZ_ALL_VARIANTS
hipError_t rz_launch_gate(const uint32_t *flag, hipStream_t st)
{
    hipLaunchKernelGGL(rz_gate_kernel, dim3(1), dim3(1), 0, st, flag);
    return hipGetLastError();
}
#endif

hipError_t rz_launch_deinterleave(const float *src, int stride, int offset, uint32_t n, float *px, float *py,
                                  float *pz, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rz_deinterleave_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, stride, offset, n,
                       px, py, pz);
    return hipGetLastError();
}

hipError_t rz_launch_pack_skinning(const uint16_t *joints4, const uint8_t *weights4, uint32_t n, uint32_t *j01,
                                   uint32_t *j23, uint32_t *wq, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rz_pack_skinning_kernel, dim3((n + 255) / 256), dim3(256), 0, st, joints4, weights4, n,
                       j01, j23, wq);
    return hipGetLastError();
}
