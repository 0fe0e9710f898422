"""Deterministic synthetic PMX-shaped workloads for tests and bench.py.

There is no network and the reference's model files must not be copied, so every large input is
generated: a mesh with the real demo model's bounding box and skinning mix, a random bone tree
posed by forward kinematics, and dense (or 2 %-sparse) vertex-morph targets.

Shapes follow the typed arrays the reference hands to the GPU (engine/src/model.ts:42-50):
positions/normals Float32 [V,3], joints Uint16 [V,4], weights Uint8 [V,4] summing to 255
(engine/src/pmx-loader.ts:136-179), inverse bind Float32 [B,16] column-major translation-only
(engine/src/pmx-loader.ts:791-824), world matrices Float32 [B,16] (engine/src/model.ts:330-420).
"""
import numpy as np

SEED = 0x5EED
BBOX_LO = np.array([-8.0, 0.0, -3.0], dtype=np.float32)
BBOX_HI = np.array([8.0, 22.0, 4.0], dtype=np.float32)


def _f32(x):
    return np.float32(x)


def quat_to_mat_f32(q):
    """Column-major rotation from a quaternion, doubles in / f32 stores (engine/src/math.ts:352-384)."""
    x, y, z, w = (float(v) for v in q)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    m = np.zeros(16, dtype=np.float32)
    m[0] = 1 - (yy + zz); m[1] = xy + wz; m[2] = xz - wy
    m[4] = xy - wz; m[5] = 1 - (xx + zz); m[6] = yz + wx
    m[8] = xz + wy; m[9] = yz - wx; m[10] = 1 - (xx + yy)
    m[15] = 1
    return m


def mat_mul_f32(a, b):
    """out = a * b, column-major, double arithmetic left-to-right, f32 store (math.ts:303-320)."""
    a64 = a.astype(np.float64)
    b64 = b.astype(np.float64)
    out = np.empty(16, dtype=np.float32)
    for c in range(4):
        b0, b1, b2, b3 = b64[c * 4: c * 4 + 4]
        for r in range(4):
            out[c * 4 + r] = ((a64[r] * b0 + a64[4 + r] * b1) + a64[8 + r] * b2) + a64[12 + r] * b3
    return out


def fk_world(parents, bind_translation, local_quats):
    """Parent-first FK without append transforms: L = T(bind) * R(q); W = W_parent * L
    (engine/src/model.ts:398-414). Returns float32 [B,16]."""
    B = len(parents)
    world = np.zeros((B, 16), dtype=np.float32)
    done = np.zeros(B, dtype=bool)

    def solve(i):
        if done[i]:
            return
        t = np.zeros(16, dtype=np.float32)
        t[0] = t[5] = t[10] = t[15] = 1
        t[12:15] = bind_translation[i]
        local = mat_mul_f32(t, quat_to_mat_f32(local_quats[i]))
        p = int(parents[i])
        if p >= 0:
            solve(p)
            world[i] = mat_mul_f32(world[p], local)
        else:
            world[i] = local
        done[i] = True

    for i in range(B):
        solve(i)
    return world


def inverse_bind_translation_only(parents, bind_translation):
    """IB = T(-bindWorld), bindWorld = sum of parent-relative offsets (pmx-loader.ts:791-824)."""
    B = len(parents)
    acc = np.zeros((B, 3), dtype=np.float32)
    for i in range(B):        # parents precede children in the synthetic tree
        p = int(parents[i])
        if p >= 0:
            # Mat4.multiply of two pure translations: f32(f64(parent) + f64(local))
            acc[i] = (acc[p].astype(np.float64) + bind_translation[i].astype(np.float64)).astype(np.float32)
        else:
            acc[i] = bind_translation[i]
    ib = np.zeros((B, 16), dtype=np.float32)
    ib[:, 0] = ib[:, 5] = ib[:, 10] = ib[:, 15] = 1
    ib[:, 12:15] = -acc
    return ib


def make_skeleton(n_bones, rng, max_depth=12, max_angle=0.5):
    parents = np.full(n_bones, -1, dtype=np.int32)
    depth = np.zeros(n_bones, dtype=np.int32)
    for i in range(1, n_bones):
        for _ in range(64):
            p = int(rng.integers(0, i))
            if depth[p] < max_depth - 1:
                break
        else:
            p = 0
        parents[i] = p
        depth[i] = depth[p] + 1
    bind = rng.uniform(-1.0, 1.0, size=(n_bones, 3)).astype(np.float32)
    axis = rng.normal(size=(n_bones, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(-max_angle, max_angle, size=n_bones)
    quats = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)
    quats = quats.astype(np.float32)   # localRotations is a Float32Array (model.ts:55)
    return parents, bind, quats


def make_skinning(n_verts, n_bones, rng):
    """40 % BDEF1 / 52 % BDEF2 / 8 % BDEF4 with bone locality, u8 weights summing to 255 using the
    loader's rounding rules (pmx-loader.ts:136-179)."""
    kind = rng.random(n_verts)
    centre = (np.arange(n_verts, dtype=np.int64) * n_bones) // max(n_verts, 1)
    j = np.clip(centre[:, None] + rng.integers(-4, 5, size=(n_verts, 4)), 0, n_bones - 1)
    joints = np.zeros((n_verts, 4), dtype=np.uint16)
    weights = np.zeros((n_verts, 4), dtype=np.uint8)
    b1 = kind < 0.40
    b2 = (kind >= 0.40) & (kind < 0.92)
    b4 = kind >= 0.92
    joints[b1, 0] = j[b1, 0]
    weights[b1, 0] = 255
    w0 = np.clip(np.floor(rng.random(n_verts).astype(np.float32) * 255 + 0.5), 0, 255).astype(np.int64)
    joints[b2, 0] = j[b2, 0]
    joints[b2, 1] = j[b2, 1]
    weights[b2, 0] = w0[b2]
    weights[b2, 1] = 255 - w0[b2]
    wf = rng.random((n_verts, 4)).astype(np.float32)
    w8 = np.floor(wf * 255 + 0.5).astype(np.int64)          # Math.round for non-negative values
    s = w8.sum(axis=1)
    s[s == 0] = 1
    scale = 255.0 / s
    q = np.clip(np.floor(w8[:, :3] * scale[:, None] + 0.5), 0, 255).astype(np.int64)
    last = np.clip(255 - q.sum(axis=1), 0, 255)
    w4 = np.concatenate([q, last[:, None]], axis=1)
    # the loader's final safety pass guarantees an exact 255 sum; emulate by fixing the largest
    diff = 255 - w4.sum(axis=1)
    big = np.argmax(w4, axis=1)
    w4[np.arange(n_verts), big] += diff
    joints[b4] = j[b4]
    weights[b4] = w4[b4].astype(np.uint8)
    return joints, weights


def make_mesh(n_verts, n_bones, seed=SEED):
    """Returns dict(pos, nrm, joints, weights, parents, bind, quats, inv_bind, world)."""
    rng = np.random.default_rng(seed)
    pos = (BBOX_LO + rng.random((n_verts, 3), dtype=np.float32) * (BBOX_HI - BBOX_LO)).astype(np.float32)
    n = rng.standard_normal((n_verts, 3), dtype=np.float32)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)
    nrm = n.astype(np.float32)
    joints, weights = make_skinning(n_verts, n_bones, rng)
    parents, bind, quats = make_skeleton(n_bones, rng)
    inv_bind = inverse_bind_translation_only(parents, bind)
    world = fk_world(parents, bind, quats)
    return dict(pos=pos, nrm=nrm, joints=joints, weights=weights, parents=parents, bind=bind,
                quats=quats, inv_bind=inv_bind, world=world)


def make_pose(parents, bind, n_bones, seed):
    """A different random pose of the same skeleton (per-instance poses of config C4)."""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=(n_bones, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(-0.5, 0.5, size=n_bones)
    quats = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)
    return fk_world(parents, bind, quats.astype(np.float32))


def make_morphs_dense(n_verts, n_morphs, seed=SEED + 1):
    """deltas [M,V,3] uniform [-0.05,0.05], weights [M] uniform [0,1]."""
    rng = np.random.default_rng(seed)
    deltas = rng.random((n_morphs, n_verts, 3), dtype=np.float32)
    deltas -= np.float32(0.5)
    deltas *= np.float32(0.1)
    w = rng.random(n_morphs, dtype=np.float32)
    return deltas, w


def make_morphs_sparse(n_verts, n_morphs, density=0.02, seed=SEED + 2, region=None):
    """PMX on-disk form: morph_off [M+1], vert_idx [E] (unique, ascending within a morph),
    delta3 [E,3]; weights [M]. density ~ 607/28842 of the demo model. `region` = (first, count) makes every
    morph start inside the same vertex range — the demo model's shape, where all 60 vertex morphs are facial
    expressions over the same ~600 vertices, i.e. a few vertices carry dozens of entries each."""
    rng = np.random.default_rng(seed)
    per = max(1, int(round(n_verts * density)))
    offs = [0]
    idx = []
    for _ in range(n_morphs):
        k = int(min(n_verts, max(1, rng.integers(per // 2, per * 3 // 2 + 1))))
        if region is not None:
            start = int(region[0] + rng.integers(0, max(1, region[1] // 4)))
        else:
            start = int(rng.integers(0, max(1, n_verts - k)))
        # vertex morphs touch a locality (a face region): a run with random holes
        cand = np.arange(start, min(n_verts, start + 2 * k))
        pick = np.sort(rng.choice(cand, size=min(k, len(cand)), replace=False))
        idx.append(pick.astype(np.uint32))
        offs.append(offs[-1] + len(pick))
    vert_idx = np.concatenate(idx) if idx else np.zeros(0, dtype=np.uint32)
    delta3 = ((rng.random((len(vert_idx), 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1))
    w = rng.random(n_morphs, dtype=np.float32)
    return np.array(offs, dtype=np.uint32), vert_idx, delta3.astype(np.float32), w


def sparse_to_dense(n_verts, morph_off, vert_idx, delta3):
    M = len(morph_off) - 1
    d = np.zeros((M, n_verts, 3), dtype=np.float32)
    for m in range(M):
        lo, hi = int(morph_off[m]), int(morph_off[m + 1])
        np.add.at(d[m], vert_idx[lo:hi].astype(np.int64), delta3[lo:hi])
    return d


# ---- range generators: any rank can produce ITS shard of a large mesh without building the whole mesh ----
# Per-vertex data is generated in fixed blocks of RANGE_BLOCK vertices, every block from its own seed
# (SeedSequence [seed, block]), so [begin, begin + count) of an n_total-vertex mesh is the same numbers no matter
# which ranges the other ranks ask for; the skeleton (and its pose) only depends on (n_bones, seed). bench.py uses these
# at every N, so the N = 1 and N = 8 runs deform the same mesh while an 8-rank node never holds 8 copies of it.
RANGE_BLOCK = 16384


def make_mesh_range(n_total, n_bones, begin, count, seed=SEED):
    """Shard [begin, begin + count) of the n_total-vertex block-seeded synthetic mesh (same distributions as make_mesh).
    Returns dict(pos, nrm, joints, weights, parents, bind, quats, inv_bind, world); per-vertex arrays have `count` rows."""
    pos, nrm, joints, weights = [], [], [], []
    for blk in range(begin // RANGE_BLOCK, (begin + count + RANGE_BLOCK - 1) // RANGE_BLOCK if count else begin // RANGE_BLOCK):
        v0 = blk * RANGE_BLOCK
        n = min(RANGE_BLOCK, n_total - v0)
        rng = np.random.default_rng([seed, 1, blk])
        p = (BBOX_LO + rng.random((n, 3), dtype=np.float32) * (BBOX_HI - BBOX_LO)).astype(np.float32)
        nn = rng.standard_normal((n, 3), dtype=np.float32)
        nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-12)
        j, w = make_skinning(n, n_bones, rng)
        # bone locality follows the GLOBAL vertex index: re-centre the block's joints around floor(v * B / V)
        centre_local = (np.arange(n, dtype=np.int64) * n_bones) // max(n, 1)
        centre_global = ((v0 + np.arange(n, dtype=np.int64)) * n_bones) // max(n_total, 1)
        used = w > 0
        j = np.where(used, np.clip(j.astype(np.int64) - centre_local[:, None] + centre_global[:, None], 0, n_bones - 1), 0).astype(np.uint16)
        lo, hi = max(begin, v0) - v0, min(begin + count, v0 + n) - v0
        pos.append(p[lo:hi]); nrm.append(nn[lo:hi].astype(np.float32)); joints.append(j[lo:hi]); weights.append(w[lo:hi])
    cat = lambda xs, shape, dt: np.ascontiguousarray(np.concatenate(xs)) if xs else np.zeros(shape, dtype=dt)  # noqa: E731
    rng = np.random.default_rng([seed, 0])
    parents, bind, quats = make_skeleton(n_bones, rng)
    return dict(pos=cat(pos, (0, 3), np.float32), nrm=cat(nrm, (0, 3), np.float32), joints=cat(joints, (0, 4), np.uint16),
                weights=cat(weights, (0, 4), np.uint8), parents=parents, bind=bind, quats=quats,
                inv_bind=inverse_bind_translation_only(parents, bind), world=fk_world(parents, bind, quats))


def make_morphs_dense_range(n_total, n_morphs, begin, count, seed=SEED + 1):
    """deltas [M, count, 3] of vertices [begin, begin + count) (uniform [-0.05, 0.05], block-seeded) and weights [M]."""
    parts = []
    for blk in range(begin // RANGE_BLOCK, (begin + count + RANGE_BLOCK - 1) // RANGE_BLOCK if count else begin // RANGE_BLOCK):
        v0 = blk * RANGE_BLOCK
        n = min(RANGE_BLOCK, n_total - v0)
        rng = np.random.default_rng([seed, 1, blk])
        d = rng.random((n_morphs, n, 3), dtype=np.float32)
        d -= np.float32(0.5)
        d *= np.float32(0.1)
        lo, hi = max(begin, v0) - v0, min(begin + count, v0 + n) - v0
        parts.append(d[:, lo:hi])
    deltas = np.ascontiguousarray(np.concatenate(parts, axis=1)) if parts else np.zeros((n_morphs, 0, 3), dtype=np.float32)
    w = np.random.default_rng([seed, 0]).random(n_morphs, dtype=np.float32)
    return deltas, w


def make_morphs_demo_shape(n_verts, n_morphs=60, total=36397, largest=1718, region=None, seed=SEED + 3):
    """Sparse vertex morphs with the DEMO MODEL's statistics (60 vertex morphs, 36 397 offsets in all, the largest
    morph 1 718, mean 607 per morph) and its shape: every morph is a facial expression, so all of them sit on the same
    face region (`region` = (first vertex, count); default 1 800 vertices in the upper part of the mesh) — a few vertices
    carry dozens of entries each. Same return as make_morphs_sparse: morph_off [M+1], vert_idx [E], delta3 [E,3], weights [M].
    The PMX on-disk layout this mirrors: engine/src/pmx-loader.ts:483-488 (vertex index + 3 floats per offset)."""
    rng = np.random.default_rng(seed)
    if region is None:
        region = (int(n_verts * 0.62), min(1800, n_verts))
    first, count = int(region[0]), int(min(region[1], n_verts - region[0]))
    largest = min(largest, count)
    # sizes: one morph of `largest`, the rest log-normal, scaled and nudged to sum to `total` exactly
    raw = np.exp(rng.normal(0.0, 0.9, size=n_morphs - 1))
    sizes = np.maximum(4, np.floor(raw / raw.sum() * (total - largest))).astype(np.int64)
    sizes = np.minimum(sizes, largest)
    sizes = np.concatenate([[largest], sizes])
    k = 1
    while sizes.sum() != total and total <= n_morphs * largest:
        d = 1 if sizes.sum() < total else -1
        if 4 <= sizes[k] + d <= largest:
            sizes[k] += d
        k = k + 1 if k + 1 < n_morphs else 1
    order = rng.permutation(n_morphs)
    sizes = sizes[order]
    offs = np.zeros(n_morphs + 1, dtype=np.uint32)
    idx = []
    for m in range(n_morphs):
        pick = np.sort(rng.choice(count, size=int(min(sizes[m], count)), replace=False)) + first
        idx.append(pick.astype(np.uint32))
        offs[m + 1] = offs[m] + len(pick)
    vert_idx = np.concatenate(idx)
    delta3 = ((rng.random((len(vert_idx), 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).astype(np.float32)
    w = rng.random(n_morphs, dtype=np.float32)
    return offs, vert_idx, delta3, w
