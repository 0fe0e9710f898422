"""NumPy float32 twin of oracle/rz_oracle.c — CPU ORACLE, test infrastructure only.

Independent second restatement of the same reference lines, used to cross-check the C oracle
bit-for-bit (tests/test_oracle.py). Every ufunc call below is one IEEE binary32 operation per
element, in the same order as rz_oracle.c, so agreement must be exact, not approximate.

Reference lines restated (paths relative to /root/reference/):
  palette   engine/src/engine.ts:926-928   skinMatrices[b] = world[b] * inverseBind[b] (column-major)
  skin      engine/src/engine.ts:253-272   vs(): renormalised unorm8 weights, 4-bone LBS, normalize
  morph     no reference implementation (engine/src/pmx-loader.ts:450-553 skips the section);
            build-defined: p~ = p + sum_m w_m * delta_m[v], ascending m, zero weights skipped.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

F = np.float32


def palette(world, inv_bind):
    """world, inv_bind: [B,16] float32 column-major -> skin [B,16]."""
    a = np.ascontiguousarray(world, dtype=F).reshape(-1, 4, 4)      # a[b, k, r] = A[k*4+r]
    m = np.ascontiguousarray(inv_bind, dtype=F).reshape(-1, 4, 4)   # m[b, c, k] = B[c*4+k]
    out = np.empty_like(a)
    for c in range(4):
        t = a[:, 0, :] * m[:, c, 0:1]
        t = t + a[:, 1, :] * m[:, c, 1:2]
        t = t + a[:, 2, :] * m[:, c, 2:3]
        t = t + a[:, 3, :] * m[:, c, 3:4]
        out[:, c, :] = t
    return out.reshape(-1, 16)


def morph_dense(deltas, weights, pos):
    """deltas [M,V,3], weights [M], pos [V,3] -> morphed pos [V,3]."""
    acc = np.zeros_like(pos, dtype=F)
    for m in range(deltas.shape[0]):
        w = F(weights[m])
        if w == 0:
            continue
        acc = acc + w * deltas[m].astype(F, copy=False)
    return pos.astype(F, copy=False) + acc


def morph_sparse(n_verts, morph_off, vert_idx, delta3, weights, pos):
    """CSR-by-morph entries (PMX file order). Duplicate vertices inside one morph add in file order."""
    acc = np.zeros((n_verts, 3), dtype=F)
    for m in range(len(morph_off) - 1):
        w = F(weights[m])
        if w == 0:
            continue
        lo, hi = int(morph_off[m]), int(morph_off[m + 1])
        for e in range(lo, hi):          # sequential on purpose: duplicates must add in order
            v = int(vert_idx[e])
            if v >= n_verts:
                continue
            acc[v] = acc[v] + w * delta3[e].astype(F)
    return pos.astype(F, copy=False) + acc


def skin(pos, nrm, joints4, weights4, skin_mats):
    """pos, nrm [V,3] f32; joints4 [V,4] u16; weights4 [V,4] u8; skin_mats [B,16] -> (pos', nrm')."""
    pos = np.ascontiguousarray(pos, dtype=F)
    nrm = np.ascontiguousarray(nrm, dtype=F)
    S = np.ascontiguousarray(skin_mats, dtype=F)
    w = weights4.astype(F) / F(255.0)
    s = ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]
    ok = s > F(0.0001)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = F(1.0) / s
    wn = w * inv[:, None]
    wn[~ok] = np.array([1, 0, 0, 0], dtype=F)
    px, py, pz = pos[:, 0], pos[:, 1], pos[:, 2]
    nx, ny, nz = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    sp = [np.zeros(len(pos), dtype=F) for _ in range(3)]
    sn = [np.zeros(len(pos), dtype=F) for _ in range(3)]
    one = F(1.0)
    for i in range(4):
        m = S[joints4[:, i].astype(np.int64)]
        wi = wn[:, i]
        for r in range(3):
            a = ((m[:, r] * px + m[:, 4 + r] * py) + m[:, 8 + r] * pz) + m[:, 12 + r] * one
            sp[r] = sp[r] + a * wi
            b = (m[:, r] * nx + m[:, 4 + r] * ny) + m[:, 8 + r] * nz
            sn[r] = sn[r] + b * wi
    ln = np.sqrt((sn[0] * sn[0] + sn[1] * sn[1]) + sn[2] * sn[2])
    good = (ln > 0) & np.isfinite(ln)
    with np.errstate(divide="ignore", invalid="ignore"):
        on = np.stack([sn[0] / ln, sn[1] / ln, sn[2] / ln], axis=1)
    on[~good] = nrm[~good]
    return np.stack(sp, axis=1), on.astype(F)


def hull(pos, nrm, edge):
    """engine.ts:458-461: worldPos + (worldNormal * edgeSize) * 0.01, f32 per op."""
    return pos.astype(F) + (nrm.astype(F) * edge.astype(F)[:, None]) * F(0.01)


def deform(pos, nrm, joints4, weights4, world, inv_bind, deltas=None, morph_w=None):
    """Whole frame: palette -> dense morph (optional) -> skin."""
    S = palette(world, inv_bind)
    p = pos if deltas is None else morph_dense(deltas, morph_w, pos)
    return skin(p, nrm, joints4, weights4, S)
