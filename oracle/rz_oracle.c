/*
 * rz_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic of the reference's per-frame deformation path, used
 * only as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under reze-engine_amd/ may include, link or call this file.
 *
 * What it restates (citations are into /root/reference/):
 *   rzo_palette      engine/src/engine.ts:906-930   skinMatrices[b] = worldMatrices[b] * inverseBindMatrices[b]
 *   rzo_unorm8       engine/src/engine.ts:354-355   vertex format "unorm8x4": u8 / 255
 *   rzo_skin         engine/src/engine.ts:245-276   vs(): weight renormalise (:255-258), LBS position
 *                                                   (:260-266,:270), LBS normal + normalize (:267-268,:272)
 *   rzo_morph_*      NO REFERENCE IMPLEMENTATION.   The reference skips the PMX morph section
 *                                                   (engine/src/pmx-loader.ts:450-553). Semantics are
 *                                                   build-defined from the PMX layout that skipMorphs()
 *                                                   documents (type 1 = vertexIndex + vec3 offset,
 *                                                   pmx-loader.ts:483-488; type 0 = morphIndex + ratio,
 *                                                   :479-482): p~ = p + SUM_m w_m * delta_m[v], bind
 *                                                   space, before skinning, normals not morphed.
 *
 * PARITY STATUS. The reference has no tests and no golden vectors for this path, its skinning is
 * WGSL inside a vertex shader (engine.ts:245-276) whose outputs are never written to a buffer, and
 * the image has no WebGPU / WGSL executor and no tsc. What the oracle is held to instead:
 *   - THE REFERENCE'S OWN SHADER TEXT, INTERPRETED. tools/wgsl_eval.py parses the bodies of `@vertex fn vs`
 *     (engine.ts:245-276) and of the skin-matrix compute shader's `fn main` (:919-928) out of the reference checkout and
 *     evaluates them statement by statement (binary32 per operation; matrix products summed column by column, left to
 *     right — the association WGSL leaves to the implementation, fixed below as this oracle's canonical one). On the
 *     reference's 349-bone model under three reference-produced poses, rzo_palette and rzo_skin agree with that
 *     interpretation BIT FOR BIT: every palette element, the 256-vertex slices and every 7th vertex of the model
 *     (4 121 vertices, 234 bones, all three influence types) — tests/test_oracle.py, tests/golden/ref_wgsl.npz,
 *     tools/ref_wgsl_run.py. The formula is therefore the shader's text, not this file's re-typing of it. (Still not a
 *     GPU executing the shader: a real driver may contract to FMA or reassociate; that is what the 1e-4 tolerance is for.)
 *   - rzo_palette (row a6) ALSO pinned to reference EXECUTION: engine.ts:926-928 is the column-major product
 *     world * inverseBind, which the reference's own Mat4.multiply (math.ts:303-320) computes; tools/ref_erased_run.py
 *     runs that method on the same model and poses: |diff| <= 2^-22 * SUM_k |a_k * b_k| per element (doubles + one f32
 *     store there, four f32 roundings here).
 *   - rzo_skin (rows a1-a3) ALSO checked against vs()'s formula evaluated in the reference run with math.ts' Mat4 / Vec3
 *     primitives in doubles (every M_i * vec4 is a Mat4.multiply; for BDEF1 vertices the position is Mat4.multiply alone):
 *     2e-7 relative on the slices, 3.2e-7 on the wide sample (bar in the tests: 1e-6). On top of that: analytic
 *     known-answer tests (identity pose => rest mesh, single-bone rigid motion about a pivot, hand-computed 2-bone blend,
 *     zero-weight and zero-normal branches) and bit-exact three-way agreement between this file, the NumPy twin
 *     (oracle/rz_oracle_np.py) and the JS Math.fround twin (oracle/js/skin_f32.js).
 *   - the inputs they consume (world matrices, inverse bind, joints, weights: rows a9-a12) ARE pinned
 *     to the reference's own code: tools/ref_erased_run.py runs math.ts/model.ts/pmx-loader.ts/
 *     vmd-loader.ts with their TypeScript types erased on the reference's assets and stores numeric
 *     fixtures in tests/golden/, which the host side reproduces bit for bit (tests/test_host_js.py).
 *   - rzo_morph_*: PARITY UNPINNED — the reference has no morph implementation at all.
 *
 * Rounding model (the canonical evaluation the GPU kernel is compared with, tolerance 1e-4):
 * every operation is a single IEEE-754 binary32 operation, no FMA contraction (build with
 * -ffp-contract=off), evaluation order
 *      m * v = ((m0*x + m1*y) + m2*z) + m3*w        (columns left to right)
 * bone accumulation i = 0..3 ascending starting from 0, morph accumulation m ascending starting
 * from 0 with zero-weight morphs skipped.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <pthread.h>
#include <stdlib.h>

#define RZO_API __attribute__((visibility("default")))

/* engine.ts:926-928 — column-major mat4 product, out[c*4+r] = sum_k a[k*4+r] * b[c*4+k]. */
RZO_API void rzo_palette(const float *world, const float *inv_bind, int n_bones, float *skin)
{
    for (int b = 0; b < n_bones; ++b) {
        const float *a = world + (size_t)b * 16;
        const float *m = inv_bind + (size_t)b * 16;
        float *o = skin + (size_t)b * 16;
        for (int c = 0; c < 4; ++c) {
            float b0 = m[c * 4 + 0], b1 = m[c * 4 + 1], b2 = m[c * 4 + 2], b3 = m[c * 4 + 3];
            for (int r = 0; r < 4; ++r) {
                float t = a[r] * b0;
                t = t + a[4 + r] * b1;
                t = t + a[8 + r] * b2;
                t = t + a[12 + r] * b3;
                o[c * 4 + r] = t;
            }
        }
    }
}

/* WebGPU "unorm8x4" vertex fetch (engine.ts:354-355): float(v) / 255.0 */
static inline float rzo_unorm8(uint8_t v) { return (float)v / 255.0f; }

/*
 * Dense morph accumulation (build-defined, see header). deltas is [M][V][3] (morph-major, the
 * order PMX stores vertex-morph offsets once expanded), weights is [M].
 * out_pos[v] = pos[v] + d, d = ((0 + w_0*D_0[v]) + w_1*D_1[v]) + ... over morphs with w != 0.
 */
RZO_API void rzo_morph_dense(int n_verts, int n_morphs, const float *deltas, const float *weights,
                             const float *pos3, float *out_pos3)
{
    for (int v = 0; v < n_verts; ++v) {
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        for (int m = 0; m < n_morphs; ++m) {
            float w = weights[m];
            if (w == 0.0f) continue;
            const float *d = deltas + ((size_t)m * n_verts + v) * 3;
            dx = dx + w * d[0];
            dy = dy + w * d[1];
            dz = dz + w * d[2];
        }
        out_pos3[(size_t)v * 3 + 0] = pos3[(size_t)v * 3 + 0] + dx;
        out_pos3[(size_t)v * 3 + 1] = pos3[(size_t)v * 3 + 1] + dy;
        out_pos3[(size_t)v * 3 + 2] = pos3[(size_t)v * 3 + 2] + dz;
    }
}

/*
 * Sparse morph accumulation: the PMX on-disk form (pmx-loader.ts:483-488) — morph m owns entries
 * [morph_off[m], morph_off[m+1]) of (vert_idx, delta xyz). Accumulation order per vertex is
 * ascending m, then file order within a morph (a vertex listed twice in one morph adds twice).
 */
RZO_API void rzo_morph_sparse(int n_verts, int n_morphs, const uint32_t *morph_off,
                              const uint32_t *vert_idx, const float *delta3, const float *weights,
                              const float *pos3, float *out_pos3)
{
    float *acc = (float *)calloc((size_t)n_verts * 3 + 1, sizeof(float));
    for (int m = 0; m < n_morphs; ++m) {
        float w = weights[m];
        if (w == 0.0f) continue;
        for (uint32_t e = morph_off[m]; e < morph_off[m + 1]; ++e) {
            uint32_t v = vert_idx[e];
            if (v >= (uint32_t)n_verts) continue;
            acc[(size_t)v * 3 + 0] = acc[(size_t)v * 3 + 0] + w * delta3[(size_t)e * 3 + 0];
            acc[(size_t)v * 3 + 1] = acc[(size_t)v * 3 + 1] + w * delta3[(size_t)e * 3 + 1];
            acc[(size_t)v * 3 + 2] = acc[(size_t)v * 3 + 2] + w * delta3[(size_t)e * 3 + 2];
        }
    }
    for (size_t i = 0; i < (size_t)n_verts * 3; ++i) out_pos3[i] = pos3[i] + acc[i];
    free(acc);
}

/*
 * vs() of engine.ts:245-276 for vertices [v0, v1).
 *   :255  weightSum = w.x + w.y + w.z + w.w        (left to right)
 *   :256  inv = select(1, 1/weightSum, weightSum > 1e-4)
 *   :257  nw  = select((1,0,0,0), w * inv, weightSum > 1e-4)
 *   :264  skinnedPos += (m * pos4) * w
 *   :266  skinnedNrm += (mat3(m) * normal) * w
 *   :272  normal = normalize(skinnedNrm)
 * normalize() of a zero vector is undefined in WGSL; build-defined here: a zero / non-finite
 * length returns the rest normal unchanged.
 */
static void rzo_skin_range(size_t v0, size_t v1, const float *pos3, const float *nrm3,
                           const uint16_t *joints4, const uint8_t *weights4, const float *skin,
                           float *out_pos3, float *out_nrm3)
{
    for (size_t v = v0; v < v1; ++v) {
        float px = pos3[v * 3], py = pos3[v * 3 + 1], pz = pos3[v * 3 + 2];
        float nx = nrm3[v * 3], ny = nrm3[v * 3 + 1], nz = nrm3[v * 3 + 2];
        float w[4];
        for (int i = 0; i < 4; ++i) w[i] = rzo_unorm8(weights4[v * 4 + i]);
        float sum = ((w[0] + w[1]) + w[2]) + w[3];
        if (sum > 0.0001f) {
            float inv = 1.0f / sum;
            for (int i = 0; i < 4; ++i) w[i] = w[i] * inv;
        } else {
            w[0] = 1.0f; w[1] = 0.0f; w[2] = 0.0f; w[3] = 0.0f;
        }
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, tx = 0.0f, ty = 0.0f, tz = 0.0f;
        for (int i = 0; i < 4; ++i) {
            const float *m = skin + (size_t)joints4[v * 4 + i] * 16;
            float ax = ((m[0] * px + m[4] * py) + m[8] * pz) + m[12] * 1.0f;
            float ay = ((m[1] * px + m[5] * py) + m[9] * pz) + m[13] * 1.0f;
            float az = ((m[2] * px + m[6] * py) + m[10] * pz) + m[14] * 1.0f;
            sx = sx + ax * w[i];
            sy = sy + ay * w[i];
            sz = sz + az * w[i];
            float bx = (m[0] * nx + m[4] * ny) + m[8] * nz;
            float by = (m[1] * nx + m[5] * ny) + m[9] * nz;
            float bz = (m[2] * nx + m[6] * ny) + m[10] * nz;
            tx = tx + bx * w[i];
            ty = ty + by * w[i];
            tz = tz + bz * w[i];
        }
        out_pos3[v * 3] = sx; out_pos3[v * 3 + 1] = sy; out_pos3[v * 3 + 2] = sz;
        float len = sqrtf((tx * tx + ty * ty) + tz * tz);
        if (len > 0.0f && isfinite(len)) {
            out_nrm3[v * 3] = tx / len; out_nrm3[v * 3 + 1] = ty / len; out_nrm3[v * 3 + 2] = tz / len;
        } else {
            out_nrm3[v * 3] = nx; out_nrm3[v * 3 + 1] = ny; out_nrm3[v * 3 + 2] = nz;
        }
    }
}

RZO_API void rzo_skin(int n_verts, const float *pos3, const float *nrm3, const uint16_t *joints4,
                      const uint8_t *weights4, const float *skin, float *out_pos3, float *out_nrm3)
{
    rzo_skin_range(0, (size_t)n_verts, pos3, nrm3, joints4, weights4, skin, out_pos3, out_nrm3);
}

/*
 * Outline pass's inverted hull, engine.ts:458-461:  expandedPos = worldPos + worldNormal * edgeSize * 0.01
 * (left to right: (N * edgeSize) * 0.01, then + P), with a per-vertex edge size.
 */
RZO_API void rzo_hull(int n_verts, const float *pos3, const float *nrm3, const float *edge, float *out3)
{
    for (size_t v = 0; v < (size_t)n_verts; ++v)
        for (int k = 0; k < 3; ++k) out3[v * 3 + k] = pos3[v * 3 + k] + (nrm3[v * 3 + k] * edge[v]) * 0.01f;
}

/* ---- threaded whole-frame driver: used for full-size parity runs and as the C fallback of the
 * cpu_baseline leg when Node is unavailable (labelled "C stand-in" by bench.py). ---- */
typedef struct {
    size_t v0, v1;
    int n_verts, n_morphs;
    const float *pos3, *nrm3, *deltas, *mw, *skin;
    const uint16_t *j4;
    const uint8_t *w4;
    float *opos, *onrm;
} rzo_job;

static void *rzo_worker(void *arg)
{
    rzo_job *j = (rzo_job *)arg;
    size_t n = j->v1 - j->v0;
    if (n == 0) return NULL;
    const float *p = j->pos3;
    float *tmp = NULL;
    if (j->n_morphs > 0) {
        /* morph this range into a scratch buffer addressed with the global vertex index */
        tmp = (float *)malloc(n * 3 * sizeof(float));
        for (size_t v = j->v0; v < j->v1; ++v) {
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            for (int m = 0; m < j->n_morphs; ++m) {
                float w = j->mw[m];
                if (w == 0.0f) continue;
                const float *d = j->deltas + ((size_t)m * j->n_verts + v) * 3;
                dx = dx + w * d[0];
                dy = dy + w * d[1];
                dz = dz + w * d[2];
            }
            tmp[(v - j->v0) * 3 + 0] = j->pos3[v * 3 + 0] + dx;
            tmp[(v - j->v0) * 3 + 1] = j->pos3[v * 3 + 1] + dy;
            tmp[(v - j->v0) * 3 + 2] = j->pos3[v * 3 + 2] + dz;
        }
        p = tmp - j->v0 * 3;
    }
    rzo_skin_range(j->v0, j->v1, p, j->nrm3, j->j4, j->w4, j->skin, j->opos, j->onrm);
    free(tmp);
    return NULL;
}

/* One whole frame: palette -> (dense morph) -> skin, split over n_threads contiguous ranges. */
RZO_API void rzo_deform(int n_verts, int n_bones, int n_morphs, const float *pos3, const float *nrm3,
                        const uint16_t *joints4, const uint8_t *weights4, const float *world,
                        const float *inv_bind, const float *deltas, const float *morph_w,
                        float *out_pos3, float *out_nrm3, int n_threads)
{
    float *skin = (float *)malloc((size_t)n_bones * 16 * sizeof(float));
    rzo_palette(world, inv_bind, n_bones, skin);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    rzo_job *jobs = (rzo_job *)calloc((size_t)n_threads, sizeof(rzo_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    size_t chunk = ((size_t)n_verts + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        size_t a = (size_t)t * chunk, b = a + chunk;
        if (a > (size_t)n_verts) a = n_verts;
        if (b > (size_t)n_verts) b = n_verts;
        rzo_job j = { a, b, n_verts, deltas ? n_morphs : 0, pos3, nrm3, deltas, morph_w, skin,
                      joints4, weights4, out_pos3, out_nrm3 };
        jobs[t] = j;
        if (n_threads == 1) rzo_worker(&jobs[t]);
        else pthread_create(&tid[t], NULL, rzo_worker, &jobs[t]);
    }
    if (n_threads > 1) for (int t = 0; t < n_threads; ++t) pthread_join(tid[t], NULL);
    free(tid); free(jobs); free(skin);
}
