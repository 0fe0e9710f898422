// kernels/fk.hip.h — the bone hierarchy on the device: MMD motion sampling and Model.computeWorldMatrices (engine/src/model.ts:330-420)
// as device functions, shared by rz_fk_kernel (front.hip: one workgroup per pose) and by the deform kernels' fused single-character
// frame, where every workgroup runs the solve as its prologue (deform_dense.hip / deform_small.hip).
#pragma once
#include "common.hip.h"

namespace {

// Every function of this header is inlined into several kernels (rz_fk_kernel, the deform kernels' fused prologue, the crowd kernel's
// front), and frames that differ only in WHICH kernel solved the hierarchy are held to the same bits (tests/test_gpu_round5.py). The
// compiler's choice of which multiply to fuse into which add depends on the code around it, so contraction is switched off here and
// every FMA is spelled out (hipcc's default, -ffp-contract=fast-honor-pragmas, restored at the end of the file).
#pragma clang fp contract(off)

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
// bone_issue / bone_finish / sample_morph — MMD motion sampling, run by the hierarchy solve's staging pass for every (instance, bone) and
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

// UNPRED: no load inside a divergent branch — a bone without a track or with a clamped frame asks for key 0 / its clamp key twice
// instead, so a thread with two bones and a morph has all their keys in flight at once (the compiler ends a divergent block of loads
// with a wait for them). Slower on the generic kernels (NOTEBOOK R5.6), 0.4-2.1 % faster on the specialised sampled solve, which is
// the only caller that asks for it (R5.10).
template <bool UNPRED = false>
__device__ __forceinline__ BoneKeys bone_issue(const RzSampleParams &p, float frame, const uint4 rec)
{
    BoneKeys k;
    k.kr = key_range(rec);
    if constexpr (UNPRED) {
        uint32_t g = 0u;
        int mode = 0;
        if (k.kr.e != k.kr.b) mode = span_guess(k.kr, frame, g) ? 2 : 1;
        k.mode = mode; k.i0 = g;
        const uint32_t i0 = g, i1 = mode == 2 ? g + 1u : g;
        k.f_a = p.key_frame[i0]; k.f_b = p.key_frame[i1];
        k.a = p.key_rot[i0]; k.b = p.key_rot[i1];
        const float *pa = p.key_pos + (size_t)i0 * 3, *pb = p.key_pos + (size_t)i1 * 3;
        k.pa0 = pa[0]; k.pa1 = pa[1]; k.pa2 = pa[2]; k.pb0 = pb[0]; k.pb1 = pb[1]; k.pb2 = pb[2];
        k.ip = make_uint4(0, 0, 0, 0);
        if (p.key_interp) k.ip = p.key_interp[i1];
        return k;
    }
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
        W[i * 4 + 0] = fmaf(P[i * 4 + 2], l2.x, fmaf(P[i * 4 + 1], l1.x, P[i * 4] * l0.x));
        W[i * 4 + 1] = fmaf(P[i * 4 + 2], l2.y, fmaf(P[i * 4 + 1], l1.y, P[i * 4] * l0.y));
        W[i * 4 + 2] = fmaf(P[i * 4 + 2], l2.z, fmaf(P[i * 4 + 1], l1.z, P[i * 4] * l0.z));
        W[i * 4 + 3] = fmaf(P[i * 4 + 2], l2.w, fmaf(P[i * 4 + 1], l1.w, fmaf(P[i * 4], l0.w, P[i * 4 + 3])));
    }
    w0 = make_float4(W[0], W[1], W[2], W[3]); w1 = make_float4(W[4], W[5], W[6], W[7]); w2 = make_float4(W[8], W[9], W[10], W[11]);
}

// ---- static data of the solve -------------------------------------------------------------------------------------------------
// ONE block of device memory per skeleton (rebuilt when the topology or the motion changes: upload.cpp rebuild_fk_static):
//   [B] bone records of 64 bytes, four uint4:
//       w0  parent (-1 = root) | append parent (-1 = none) | bits(append ratio) | flags (bit 0: append-move, model.ts:388-393)
//       w1  bits of the parent-relative bind translation x y z | 0
//       w2  the motion's track of this bone: (first key, one past the last key, bits(first key's frame), bits(last key's frame));
//           first == end = the motion does not key it. One load gives the sampler everything it needs to guess the key span.
//       w3  ancestors for the first two radix-4 doubling rounds: round 0 (a1 | a2 << 16, a3), round 1 (a4 | a8 << 16, a12),
//           a_k = the bone k levels up, 0xffff = above the root
//   [M] vertex-morph records of 32 bytes behind them (with a motion): the key range of the morph's FIRST feed, then
//       (first feed, one past the last feed, bits(ratio of the first feed), 0) — one round trip to the keys instead of
//       offsets -> feed record -> keys.
// The leading kernel arguments carry its address (and FkAux) so that a wave asks for its records before it reads anything else.
struct FkEarly {           // the records of this thread's two bones (tid, tid + 256) and of its vertex morph
    uint4 a0, a1, a2, a3, b0, b1, b2, b3, m0, m1;
};
constexpr uint32_t kNoAnc = 0xffffu;

// FkAux, one preloaded 64-bit kernel argument: bits(frame of a single sampled character) | vertex morphs of the motion << 32 | doubling rounds << 48
__device__ __forceinline__ FkEarly fk_issue_static(const uint4 *rec, const int B, const int M, const int tid)
{
    // (unpredicated: threads past the last bone / morph re-read the last record — a dead load is cheaper than the wait the compiler
    // puts behind a divergent block of loads, NOTEBOOK.md R4.9)
    FkEarly e;
    const uint4 *ra = rec + (size_t)min(tid, B - 1) * 4, *rb = rec + (size_t)min(tid + kBlock, B - 1) * 4;
    e.a0 = ra[0]; e.a1 = ra[1]; e.a2 = ra[2]; e.a3 = ra[3];
    e.b0 = rb[0]; e.b1 = rb[1]; e.b2 = rb[2]; e.b3 = rb[3];
    e.m0 = make_uint4(0, 0, 0, 0); e.m1 = e.m0;
    if (M > 0) {
        const uint4 *rm = rec + (size_t)B * 4 + (size_t)min(tid, M - 1) * 2;
        e.m0 = rm[0]; e.m1 = rm[1];
    }
    return e;
}

// L = T(bind + t) * R(q) * T(add) of one bone, rows 0..2 (model.ts:355-414): `apq` = its append parent's LOCAL rotation (used when
// the record names one), `apt` = that parent's local translation (append-move)
__device__ __forceinline__ void fk_local_matrix(const float4 q, const uint4 rec, const float bx, const float by, const float bz, const bool has_t,
                                                const float ltx, const float lty, const float ltz, float4 a, const float apx, const float apy,
                                                const float apz, float4 &l0, float4 &l1, float4 &l2)
{
    float R[9];
    quat_to_rows(q.x, q.y, q.z, q.w, R);
    const int ap = (int)rec.y;
    const float ratio_raw = __uint_as_float(rec.z);
    float ax = 0.0f, ay = 0.0f, az = 0.0f;       // append-move: T(add) of L = T(bind) * R * T(add)
    if (ap >= 0) {
        const float ratio = fminf(1.0f, fmaxf(-1.0f, ratio_raw));
        if (fabsf(ratio) > 1e-6f) {
            if (has_t && (rec.w & 1u)) {             // model.ts:388-393 uses the UNclamped ratio here
                ax = apx * ratio_raw; ay = apy * ratio_raw; az = apz * ratio_raw;
            }
            const float t = fabsf(ratio);
            if (ratio < 0.0f) { a.x = -a.x; a.y = -a.y; a.z = -a.z; }
            const float4 sl = slerp_from_identity(a, t);
            float A[9], X[9];
            quat_to_rows(sl.x, sl.y, sl.z, sl.w, A);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) X[i * 3 + j] = fmaf(A[i * 3 + 2], R[6 + j], fmaf(A[i * 3 + 1], R[3 + j], A[i * 3] * R[j]));
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = X[i];
        }
    }
    // translation column of L = T(bind + local) * R * T(add)  =  bind + local + R * add
    float tx = bx, ty = by, tz = bz;
    if (has_t) {
        tx += ltx; ty += lty; tz += ltz;
        tx += fmaf(R[2], az, fmaf(R[1], ay, R[0] * ax));
        ty += fmaf(R[5], az, fmaf(R[4], ay, R[3] * ax));
        tz += fmaf(R[8], az, fmaf(R[7], ay, R[6] * ax));
    }
    l0 = make_float4(R[0], R[1], R[2], tx);
    l1 = make_float4(R[3], R[4], R[5], ty);
    l2 = make_float4(R[6], R[7], R[8], tz);
}

// One radix-4 doubling round for one bone held in registers: M <- M[a3] * (M[a2] * (M[a1] * M)), a_k = the bone k * 4^round levels
// up (kNoAnc = above the root: the run has reached it). `src` holds every bone's product after the previous round, 3 float4 per bone.
__device__ __forceinline__ void fk_round(const float4 *src, const uint32_t a1, const uint32_t a2, const uint32_t a3, float4 &m0, float4 &m1, float4 &m2)
{
    if (a1 == kNoAnc) return;
    // all three ancestors' rows are asked for at once (the addresses are known from the start): one LDS latency per round
    const uint32_t c2 = a2 == kNoAnc ? a1 : a2, c3 = a3 == kNoAnc ? a1 : a3;
    const float4 p0 = src[a1 * 3], p1 = src[a1 * 3 + 1], p2 = src[a1 * 3 + 2];
    const float4 q0 = src[c2 * 3], q1 = src[c2 * 3 + 1], q2 = src[c2 * 3 + 2];
    const float4 r0 = src[c3 * 3], r1 = src[c3 * 3 + 1], r2 = src[c3 * 3 + 2];
    affine_mul(p0, p1, p2, m0, m1, m2, m0, m1, m2);
    if (a2 != kNoAnc) affine_mul(q0, q1, q2, m0, m1, m2, m0, m1, m2);
    if (a3 != kNoAnc) affine_mul(r0, r1, r2, m0, m1, m2, m0, m1, m2);
}

// palette rows 0..2 of W * IB for one bone (IB general 4x4, column-major; engine.ts:926-928): ((w0*b0 + w1*b1) + w2*b2) + w3*b3
__device__ __forceinline__ void fk_palette_rows(const float4 w0, const float4 w1, const float4 w2, const float4 i0, const float4 i1, const float4 i2,
                                                const float4 i3, float4 &q0, float4 &q1, float4 &q2)
{
    const float W[12] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w };
    const float4 ibm[4] = { i0, i1, i2, i3 };
    float r[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 bc = ibm[c];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            r[i][c] = fmaf(W[i * 4 + 3], bc.w, fmaf(W[i * 4 + 2], bc.z, fmaf(W[i * 4 + 1], bc.y, W[i * 4] * bc.x)));
    }
    q0 = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]); q1 = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]);
    q2 = make_float4(r[2][0], r[2][1], r[2][2], r[2][3]);
}

// The body of the hierarchy solve, shared by rz_fk_kernel (one workgroup per pose, results to global memory) and by the
// FUSED single-character frame, where every workgroup of the deform kernels runs it as its prologue: `wl` is then the deform
// kernel's LDS palette (it ends up holding rows 0..2 of W * inverseBind), `scr` aliases its wave scratch, the sampled morph
// weights go to `lds_mw`, and only workgroup 0 (`to_global`) also leaves world matrices / palette / weights in memory.
//
// Shape of the solve: the static data of a thread's bones and morph arrive as FkEarly — requested by the caller before it reads its
// kernel arguments (round 5; rounds 1-4 asked for them 0.9 us later, behind the scalar loads of `p`) — then the pose (sampled: the
// keys of the guessed spans; uploaded: rotations / translations), then the parent chains are resolved by RADIX-4 POINTER DOUBLING
// over ancestor tables computed at upload time: every bone holds the product M of the local matrices of a run of its ancestors
// ending at itself; round r multiplies in the runs ending 4^r, 2 * 4^r and 3 * 4^r levels up, so ceil(log4(depth)) rounds — 2 for
// a 12-level tree (4 radix-2 rounds of ~0.38 us each in round 4, six barrier-separated levels before that) — each ONE LDS latency,
// three products and a barrier. The products are associated differently from the reference's parent-first recursion: same f32
// error class, ~1e-7 per product (tests/test_gpu_round4.py: test_pointer_doubling_hierarchy_solve, against float64).
// LDS behind `scr`: rz_fk_scratch_bytes(B) = B x (48 + 12) bytes. Ends with a barrier.
// KIND specialises the body at compile time for the two common single-character poses (the deform kernels' fused frame, round 5): the
// generic form carries every feature behind workgroup-uniform branches — 47 KB of code in front of a 10 KB deform kernel, of which a frame
// executes every instruction once. KIND 1 = an uploaded pose, KIND 2 = a sampled pose, both PLAIN: no physics overrides,
// at most 512 bones (two per thread), at most 256 vertex morphs (one per thread) — the host picks the variant when all of that holds
// (RzDeformParams::fk_kind) and the generic form (KIND 0) otherwise. Same device functions, same bits.
template <bool FUSED, int KIND = 0>
__device__ __forceinline__ void fk_solve(const RzFkParams &p, const FkEarly &early, const int inst, float4 *wl, unsigned char *scr, float *lds_mw,
                                         const bool to_global, const uint64_t st_tagv = 0ull, unsigned long long *fs = nullptr)
{
    constexpr bool PLAIN = KIND != 0;
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
    float *s_lt = reinterpret_cast<float *>(scr + (size_t)p.B * 48);    // [B][3] local translations of this pose
    const int tid = threadIdx.x;
    const float4 *lq = p.local_q + (size_t)inst * p.B;
    const float *glt = p.local_t ? p.local_t + (size_t)inst * p.B * 3 : nullptr;
    const bool sampled = KIND == 2 || (KIND == 0 && (p.sample.frames != nullptr || p.sample.frames_inline));      // rz_set_pose_sampled: the pose is evaluated right here
    const bool bone_morphs = p.bm_off != nullptr;
    const bool has_t = sampled || glt != nullptr || bone_morphs;
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
    const int b0 = tid, b1 = tid + kBlock;
    const bool hb0 = b0 < p.B, hb1 = b1 < p.B;
    auto park = [&](const int b, const float4 q, const float tx, const float ty, const float tz, const uint4 r0, const uint4 r1) {
        sq[b] = q; s_rec[b] = r0;
        s_bind[b] = make_float4(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), 0.0f);
        if (has_t) { s_lt[b * 3] = tx; s_lt[b * 3 + 1] = ty; s_lt[b * 3 + 2] = tz; }
    };
    // an uploaded pose's bone: from the copy the previous frame's helper may have staged (FUSED zero-copy frame: the staged copy is
    // asked for at once, the tag — requested by the caller — decides afterwards; a miss re-reads the pinned slot and workgroup 0 keeps
    // the pose for the replays), else from where the pose lies
    auto uploaded = [&](const int i, float4 &q, float &tx, float &ty, float &tz) {
        tx = ty = tz = 0.0f;
        if (FUSED && p.st_local_q) {
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
    };
    // Staging pass. Sampled pose, the common sizes (<= 512 bones: two per thread; <= 256 vertex morphs: one per thread): the keys of
    // the thread's two bones AND of its morph's first feed are requested together — with the records already here, ONE more round
    // trip for the whole pose (rounds 2-3: five, round 4: three).
    int m_done = 0;                       // vertex morphs [0, m_done) have been sampled by the interleaved pass
    if (sampled) {
        const bool hm = tid < p.sample.M;
        BoneKeys k0 = bone_issue<KIND == 2>(p.sample, frame, hb0 ? early.a2 : make_uint4(0, 0, 0, 0)), k1 = bone_issue<KIND == 2>(p.sample, frame, hb1 ? early.b2 : make_uint4(0, 0, 0, 0));
        const uint32_t f0 = hm ? early.m1.x : 0u, f1 = hm ? early.m1.y : 0u;
        const float ratio0 = __uint_as_float(early.m1.z);
        const MorphKeys mk = morph_issue(p.sample, frame, f1 > f0 ? early.m0 : make_uint4(0, 0, 0, 0));
        float4 q;
        float tx, ty, tz;
        if (hb0) { bone_finish(p.sample, frame, k0, q, tx, ty, tz); park(b0, q, tx, ty, tz, early.a0, early.a1); }
        if (hb1) { bone_finish(p.sample, frame, k1, q, tx, ty, tz); park(b1, q, tx, ty, tz, early.b0, early.b1); }
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
    } else {
        float4 q;
        float tx, ty, tz;
        if (hb0) { uploaded(b0, q, tx, ty, tz); park(b0, q, tx, ty, tz, early.a0, early.a1); }
        if (hb1) { uploaded(b1, q, tx, ty, tz); park(b1, q, tx, ty, tz, early.b0, early.b1); }
    }
    if constexpr (!PLAIN)
    for (int i = tid + 2 * kBlock; i < p.B; i += kBlock) {      // skeletons beyond 512 bones: the rest, record by record
        const uint4 r0 = p.bone_rec[4 * i], r1 = p.bone_rec[4 * i + 1];
        float4 q;
        float tx, ty, tz;
        if (sampled) {
            BoneKeys k = bone_issue(p.sample, frame, p.bone_rec[4 * i + 2]);
            bone_finish(p.sample, frame, k, q, tx, ty, tz);
        } else uploaded(i, q, tx, ty, tz);
        park(i, q, tx, ty, tz, r0, r1);
    }
    if (sampled && !PLAIN)        // vertex-morph weights of this pose: consumed by the prep / deform kernels that follow (FUSED: by this very workgroup)
        for (int m = m_done + tid; m < p.sample.M; m += kBlock) {      // (morphs beyond the interleaved pass)
            const float w = sample_morph(p.sample, frame, m);
            if (FUSED || bone_morphs) lds_mw[m] = w;
            if (to_global) p.sample.morph_w[(size_t)inst * p.sample.M + m] = w;
        }
    else if (bone_morphs && !FUSED)         // (FUSED: the caller has staged the uploaded weights already)
        for (int m = tid; m < p.bm_M; m += kBlock) lds_mw[m] = p.bm_w[(size_t)inst * p.bm_M + m];
    RZ_FSTAMP(0);             // pose staged, before the barrier
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
    // Pass A, every bone in parallel: its LOCAL matrix L = T(bind + t) * R * T(add) (rows 0..2). The quaternion / append / slerp math
    // is off the rounds' critical path. A thread's first two bones (skeletons up to 512 bones: all of them) keep their matrix in
    // REGISTERS from here to the palette: a round reads only its ancestors' rows and writes its own for the others.
    constexpr int NBR = 2;
    float4 rm[NBR][3];
    auto local_of = [&](const int b, float4 &l0, float4 &l1, float4 &l2) {
        const uint4 rec = s_rec[b];
        const float4 bind = s_bind[b];
        const int ap = (int)rec.y;
        const float4 a = ap >= 0 ? sq[ap] : make_float4(0.f, 0.f, 0.f, 1.f);
        float apx = 0.0f, apy = 0.0f, apz = 0.0f, ltx = 0.0f, lty = 0.0f, ltz = 0.0f;
        if (has_t) {
            ltx = s_lt[b * 3]; lty = s_lt[b * 3 + 1]; ltz = s_lt[b * 3 + 2];
            if (ap >= 0 && (rec.w & 1u)) { apx = s_lt[ap * 3]; apy = s_lt[ap * 3 + 1]; apz = s_lt[ap * 3 + 2]; }
        }
        fk_local_matrix(sq[b], rec, bind.x, bind.y, bind.z, has_t, ltx, lty, ltz, a, apx, apy, apz, l0, l1, l2);
    };
#pragma unroll
    for (int k = 0; k < NBR; ++k) {
        const int b = tid + k * kBlock;
        rm[k][0] = rm[k][1] = rm[k][2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < p.B) {
            local_of(b, rm[k][0], rm[k][1], rm[k][2]);
            wl[b * 3] = rm[k][0]; wl[b * 3 + 1] = rm[k][1]; wl[b * 3 + 2] = rm[k][2];
        }
    }
    if constexpr (!PLAIN)
    for (int b = tid + NBR * kBlock; b < p.B; b += kBlock) {
        float4 l0, l1, l2;
        local_of(b, l0, l1, l2);
        wl[b * 3] = l0; wl[b * 3 + 1] = l1; wl[b * 3 + 2] = l2;
    }
    RZ_FSTAMP(2);             // local matrices formed
    __syncthreads();          // (also: every read of region X is done, the rounds may write it)
    // Doubling rounds (ping-pong between `wl` and region X). Rounds 0 and 1 take their ancestors from the bone record's w3; deeper
    // hierarchies (> 16 levels) read the further rounds' tables from memory (p.anc_more: [rounds - 2][B] x (a1 | a2 << 16, a3)).
    float4 *src = wl, *dst = m2;
    for (int r = 0; r < p.n_rounds; ++r) {
#pragma unroll
        for (int k = 0; k < NBR; ++k) {
            const int b = tid + k * kBlock;
            if (b < p.B) {
                const uint4 w3 = k == 0 ? early.a3 : early.b3;
                uint32_t lo = r == 0 ? w3.x : w3.z, hi = r == 0 ? w3.y : w3.w;
                if (r >= 2) { const uint2 am = p.anc_more[(size_t)(r - 2) * p.B + b]; lo = am.x; hi = am.y; }
                fk_round(src, lo & 0xffffu, lo >> 16, hi & 0xffffu, rm[k][0], rm[k][1], rm[k][2]);
                dst[b * 3] = rm[k][0]; dst[b * 3 + 1] = rm[k][1]; dst[b * 3 + 2] = rm[k][2];
            }
        }
        if constexpr (!PLAIN)
        for (int b = tid + NBR * kBlock; b < p.B; b += kBlock) {       // bones beyond the register slots: through LDS
            uint32_t lo, hi;
            if (r < 2) { const uint4 w3 = p.bone_rec[4 * b + 3]; lo = r == 0 ? w3.x : w3.z; hi = r == 0 ? w3.y : w3.w; }
            else { const uint2 am = p.anc_more[(size_t)(r - 2) * p.B + b]; lo = am.x; hi = am.y; }
            float4 w0 = src[b * 3], w1 = src[b * 3 + 1], w2 = src[b * 3 + 2];
            fk_round(src, lo & 0xffffu, lo >> 16, hi & 0xffffu, w0, w1, w2);
            dst[b * 3] = w0; dst[b * 3 + 1] = w1; dst[b * 3 + 2] = w2;
        }
        __syncthreads();
        float4 *t4 = src; src = dst; dst = t4;
    }
    RZ_FSTAMP(3);             // doubling rounds done
    const bool overrides = !PLAIN && p.ovr_off != nullptr;
    if (overrides) {
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
    auto emit = [&](const int b, const float4 w0, const float4 w1, const float4 w2, const bool mine) {
        // world, column-major 4x4 (what queue.writeBuffer(worldMatrixBuffer) would have carried)
        if (to_global) {
            float4 *wo = reinterpret_cast<float4 *>(world + (size_t)b * 16);
            wo[0] = make_float4(w0.x, w1.x, w2.x, 0.0f);
            wo[1] = make_float4(w0.y, w1.y, w2.y, 0.0f);
            wo[2] = make_float4(w0.z, w1.z, w2.z, 0.0f);
            wo[3] = make_float4(w0.w, w1.w, w2.w, 1.0f);
        }
        const float4 *Im = reinterpret_cast<const float4 *>(p.inv_bind + (size_t)b * 16);
        float4 q0, q1, q2;
        fk_palette_rows(w0, w1, w2, mine ? pib0 : Im[0], mine ? pib1 : Im[1], mine ? pib2 : Im[2], mine ? pib3 : Im[3], q0, q1, q2);
        if (to_global) { pal[b * 3] = q0; pal[b * 3 + 1] = q1; pal[b * 3 + 2] = q2; }
        if (FUSED) { wl[b * 3] = q0; wl[b * 3 + 1] = q1; wl[b * 3 + 2] = q2; }     // bone b's rows (in wl or in the other buffer) are only ever read by this thread in this pass
    };
    if (!overrides) {                   // (overrides land in LDS: the registers are only good without them; workgroup-uniform)
#pragma unroll
        for (int k = 0; k < NBR; ++k)
            if (tid + k * kBlock < p.B) emit(tid + k * kBlock, rm[k][0], rm[k][1], rm[k][2], k == 0);
        if constexpr (!PLAIN)
            for (int b = tid + NBR * kBlock; b < p.B; b += kBlock) emit(b, src[b * 3], src[b * 3 + 1], src[b * 3 + 2], false);
    } else {
        for (int b = tid; b < p.B; b += kBlock) emit(b, src[b * 3], src[b * 3 + 1], src[b * 3 + 2], b == tid);
    }
    RZ_FSTAMP(4);             // palette rows written
    if (FUSED) __syncthreads();
#undef RZ_FSTAMP
}

#pragma clang fp contract(fast)

}  // namespace
