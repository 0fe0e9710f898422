"""The latitude WGSL leaves an implementation when it evaluates the reference's vs() (engine/src/engine.ts:253-272).

The oracle fixes ONE evaluation: every operation rounded to binary32, m * v summed column by column left to right, no
fused multiply-add, normalize(v) = v / sqrt(dot(v, v)). A conforming driver may differ: WGSL lets it contract a * b + c
into one FMA, re-associate the four-term sums of a matrix-vector product, and implement normalize() through
inverseSqrt; and this build's kernels blend the four palette matrices first and transform once (a re-association across
the bone sum). This module evaluates vs() under each of those models — float64 arithmetic rounded to binary32 after every
operation the model rounds at — so the tests can state how far the legal results spread around the oracle, and that the
GPU sits inside the same envelope. Test infrastructure only."""
import numpy as np

F = np.float32


def _r(x):
    """round to binary32, carry on in float64 (so a product of two rounded values is exact)"""
    return np.asarray(x, dtype=np.float64).astype(F).astype(np.float64)


def _weights(weights4, divide):
    w = _r(weights4.astype(np.float64) / 255.0)
    s = _r(_r(_r(w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3])
    ok = s > 1e-4
    with np.errstate(divide="ignore", invalid="ignore"):
        wn = _r(w / s[:, None]) if divide else _r(w * _r(1.0 / s)[:, None])
    wn[~ok] = np.array([1.0, 0.0, 0.0, 0.0])
    return wn


def _dot4(a, b, assoc, fma):
    """sum_k a[k] * b[k] for k = 0..3 (arrays of shape [4, V]) under an association and with / without contraction"""
    if fma:             # contraction: one rounding per a * b + c; the first product is rounded on its own
        if assoc == "left":
            t = _r(a[0] * b[0])
            for k in (1, 2, 3):
                t = _r(a[k] * b[k] + t)
            return t
        if assoc == "right":
            t = _r(a[3] * b[3])
            for k in (2, 1, 0):
                t = _r(a[k] * b[k] + t)
            return t
        lo = _r(a[1] * b[1] + _r(a[0] * b[0]))
        hi = _r(a[3] * b[3] + _r(a[2] * b[2]))
        return _r(lo + hi)
    p = [_r(a[k] * b[k]) for k in range(4)]
    if assoc == "left":
        return _r(_r(_r(p[0] + p[1]) + p[2]) + p[3])
    if assoc == "right":
        return _r(p[0] + _r(p[1] + _r(p[2] + p[3])))
    return _r(_r(p[0] + p[1]) + _r(p[2] + p[3]))


def vs(pos, nrm, joints4, weights4, skin_mats, assoc="left", fma=False, rsqrt=False, divide=False, blend_first=False):
    """vs() under one evaluation model; returns (pos', nrm') as float64 arrays holding binary32 values.
    assoc: 'left' (the oracle's), 'right', 'pair'; fma: contract multiply-adds; rsqrt: normalize = v * inverseSqrt(dot);
    divide: weights / sum instead of weights * (1 / sum); blend_first: M = sum_i w_i S[j_i], then M * v (this build's kernels)."""
    S = np.asarray(skin_mats, dtype=np.float64).reshape(-1, 4, 4)          # S[b, col, row]
    V = len(pos)
    wn = _weights(weights4, divide)
    p4 = np.stack([pos[:, 0], pos[:, 1], pos[:, 2], np.ones(V)]).astype(np.float64)
    n3 = nrm.astype(np.float64)

    def acc(t, w, x):           # t + w * x
        return _r(w * x + t) if fma else _r(t + _r(w * x))

    sp = np.zeros((3, V))
    sn = np.zeros((3, V))
    if blend_first:
        M = np.zeros((V, 4, 4))
        for i in range(4):
            m = S[joints4[:, i].astype(np.int64)]
            for c in range(4):
                for r in range(3):
                    M[:, c, r] = acc(M[:, c, r], wn[:, i], m[:, c, r])
        for r in range(3):
            cols = np.stack([M[:, k, r] for k in range(4)])
            sp[r] = _dot4(cols, p4, assoc, fma)
            z = np.stack([cols[0], cols[1], cols[2], np.zeros(V)])
            sn[r] = _dot4(z, np.stack([n3[:, 0], n3[:, 1], n3[:, 2], np.zeros(V)]), assoc, fma)
    else:
        for i in range(4):
            m = S[joints4[:, i].astype(np.int64)]
            for r in range(3):
                cols = np.stack([m[:, k, r] for k in range(4)])
                t = _dot4(cols, p4, assoc, fma)
                sp[r] = acc(sp[r], wn[:, i], t)
                # mat3x3 * normal: a three-term sum; 'pair' and 'left' coincide, 'right' differs
                q = [cols[k] * n3[:, k] for k in range(3)]
                if fma:
                    tn = _r(q[2] + _r(q[1] + _r(q[0]))) if assoc != "right" else _r(q[0] + _r(q[1] + _r(q[2])))
                else:
                    q = [_r(x) for x in q]
                    tn = _r(_r(q[0] + q[1]) + q[2]) if assoc != "right" else _r(q[0] + _r(q[1] + q[2]))
                sn[r] = acc(sn[r], wn[:, i], tn)
    d = _r(_r(_r(sn[0] * sn[0]) + _r(sn[1] * sn[1])) + _r(sn[2] * sn[2]))
    with np.errstate(divide="ignore", invalid="ignore"):
        if rsqrt:
            inv = _r(1.0 / np.sqrt(d))
            out_n = np.stack([_r(sn[k] * inv) for k in range(3)], axis=1)
        else:
            ln = _r(np.sqrt(d))
            out_n = np.stack([_r(sn[k] / ln) for k in range(3)], axis=1)
    return sp.T.copy(), out_n


MODELS = {
    "oracle order (left, no fma)": dict(),
    "fma-contracted, left": dict(fma=True),
    "pairwise association": dict(assoc="pair"),
    "right-to-left association": dict(assoc="right"),
    "fma-contracted, pairwise": dict(assoc="pair", fma=True),
    "normalize through inverseSqrt": dict(rsqrt=True),
    "weights / sum": dict(divide=True),
    "blended matrix first (this build's kernels), fma": dict(blend_first=True, fma=True, rsqrt=True, divide=True),
    "everything at once": dict(assoc="right", fma=True, rsqrt=True, divide=True),
}


def distances(p, n, p_ref, n_ref):
    """the parity metric of SURVEY 8c: positions relative to max(|P_ref|, 1), normals absolute (unit vectors)"""
    ep = np.linalg.norm(np.asarray(p, np.float64) - p_ref, axis=1) / np.maximum(np.linalg.norm(p_ref, axis=1), 1.0)
    en = np.linalg.norm(np.asarray(n, np.float64) - n_ref, axis=1)
    return float(ep.max()), float(en.max())
