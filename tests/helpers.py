"""Shared parity metric and small utilities for the tests."""
import numpy as np

POS_TOL = 1e-4   # per vertex: |Pg - Pr|_2 <= POS_TOL * max(|Pr|_2, 1)   (north_star: 1e-4 relative fp32)
NRM_TOL = 1e-4   # per vertex: |Ng - Nr|_2 <= NRM_TOL                      (unit vectors)


def parity_errors(pos_g, nrm_g, pos_r, nrm_r):
    pos_g = np.asarray(pos_g, dtype=np.float64)
    pos_r = np.asarray(pos_r, dtype=np.float64)
    nrm_g = np.asarray(nrm_g, dtype=np.float64)
    nrm_r = np.asarray(nrm_r, dtype=np.float64)
    ep = np.linalg.norm(pos_g - pos_r, axis=1) / np.maximum(np.linalg.norm(pos_r, axis=1), 1.0)
    en = np.linalg.norm(nrm_g - nrm_r, axis=1)
    return ep, en


def assert_parity(pos_g, nrm_g, pos_r, nrm_r, what=""):
    assert np.isfinite(pos_g).all() and np.isfinite(nrm_g).all(), "NaN/Inf in GPU output " + what
    ep, en = parity_errors(pos_g, nrm_g, pos_r, nrm_r)
    assert ep.max() <= POS_TOL, "%s position error max %.3e (p99.9 %.3e)" % (what, ep.max(), np.percentile(ep, 99.9))
    assert en.max() <= NRM_TOL, "%s normal error max %.3e (p99.9 %.3e)" % (what, en.max(), np.percentile(en, 99.9))
    return float(ep.max()), float(en.max())


def fk_reference(parents, bind, quats, trans=None, append_parent=None, append_ratio=None, append_move=None):
    """Model.computeWorldMatrices with this build's local translations (engine/src/model.ts:330-420 restated in float64):
    R = fromQuat(q); with an append parent and |clamp(ratio)| > 1e-6: R = fromQuat(slerp(I, +-q_ap, |ratio|)) * R and,
    if the bone also appends movement, add = t_ap * ratio (unclamped, :388-393);
    L = T(bind + t) * R * T(add); W = W_parent * L. Returns float64 [B,16] column-major. Test infrastructure."""
    import numpy as np
    B = len(parents)

    def rot(q):
        x, y, z, w = (float(v) for v in q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def slerp_from_identity(q, t):            # Quat.slerp(identity, q, t), math.ts:156-189
        q = np.array(q, dtype=np.float64)
        c = q[3]
        if c < 0:
            q, c = -q, -c
        if c > 0.9995:
            r = np.array([t * q[0], t * q[1], t * q[2], 1 + t * (q[3] - 1)])
            return r / np.linalg.norm(r)
        th0 = np.arccos(c)
        s0, s1 = np.sin(th0 - th0 * t) / np.sin(th0), np.sin(th0 * t) / np.sin(th0)
        return np.array([s1 * q[0], s1 * q[1], s1 * q[2], s0 + s1 * q[3]])

    world = np.zeros((B, 4, 4))
    done = np.zeros(B, dtype=bool)

    def solve(i):
        if done[i]:
            return
        R = rot(quats[i])
        add = np.zeros(3)
        ap = -1 if append_parent is None else int(append_parent[i])
        if ap >= 0:
            raw = 1.0 if append_ratio is None else float(append_ratio[i])
            ratio = max(-1.0, min(1.0, raw))
            if abs(ratio) > 1e-6:
                qa = np.array(quats[ap], dtype=np.float64)
                if ratio < 0:
                    qa[:3] = -qa[:3]
                R = rot(slerp_from_identity(qa, abs(ratio))) @ R
                if trans is not None and append_move is not None and append_move[i]:
                    add = np.asarray(trans[ap], dtype=np.float64) * raw
        L = np.eye(4)
        L[:3, :3] = R
        t = np.asarray(bind[i], dtype=np.float64) + (0 if trans is None else np.asarray(trans[i], dtype=np.float64))
        L[:3, 3] = t + R @ add
        p = int(parents[i])
        if p >= 0:
            solve(p)
            world[i] = world[p] @ L
        else:
            world[i] = L
        done[i] = True

    for i in range(B):
        solve(i)
    return np.transpose(world, (0, 2, 1)).reshape(B, 16)
