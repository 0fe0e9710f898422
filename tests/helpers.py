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


def assert_hull(hull_g, hull_r, what=""):
    """Outline hull P + N * edge * 0.01 (engine.ts:458-461): same per-vertex bar as positions —
    |Hg - Hr|_2 <= POS_TOL * max(|Hr|_2, 1). (edge <= 1.5, so the hull error is the position error + 0.015 x the normal error.)"""
    hull_g = np.asarray(hull_g, dtype=np.float64)
    hull_r = np.asarray(hull_r, dtype=np.float64)
    assert np.isfinite(hull_g).all(), "NaN/Inf in the hull " + what
    e = np.linalg.norm(hull_g - hull_r, axis=1) / np.maximum(np.linalg.norm(hull_r, axis=1), 1.0)
    assert e.max() <= POS_TOL, "%s hull error max %.3e" % (what, e.max())
    return float(e.max())


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


def bone_morph_reference(quats, trans, morph, bone, t3, q4, weights):
    """PMX bone morphs (type 2) folded into a local pose, float64 (host/model.js posedLocals(), fk_solve's bone-morph pass):
    entries in order (ascending morph index), weight w = weights[morph]: t[bone] += w * t_e ;
    q[bone] = q[bone] * slerp(identity, q_e, w) (Hamilton product, math.ts:77-85; slerp math.ts:156-189). Test infrastructure."""
    import numpy as np
    q = np.array(quats, dtype=np.float64).copy()
    t = np.zeros((len(q), 3)) if trans is None else np.array(trans, dtype=np.float64).copy()
    for k in range(len(morph)):
        w = float(weights[int(morph[k])])
        if w == 0.0:
            continue
        b = int(bone[k])
        t[b] += w * np.asarray(t3[k], dtype=np.float64)
        e = np.array(q4[k], dtype=np.float64)
        c = e[3]
        if c < 0:
            e, c = -e, -c
        if c > 0.9995:
            s = np.array([w * e[0], w * e[1], w * e[2], 1 + w * (e[3] - 1)])
            s /= np.linalg.norm(s)
        else:
            th0 = np.arccos(c)
            s0, s1 = np.sin(th0 - th0 * w) / np.sin(th0), np.sin(th0 * w) / np.sin(th0)
            s = np.array([s1 * e[0], s1 * e[1], s1 * e[2], s0 + s1 * e[3]])
        x, y, z, ww = q[b]
        q[b] = [ww * s[0] + x * s[3] + y * s[2] - z * s[1], ww * s[1] - x * s[2] + y * s[3] + z * s[0],
                ww * s[2] + x * s[1] - y * s[0] + z * s[3], ww * s[3] - x * s[0] - y * s[1] - z * s[2]]
    return q, t


def bezier_reference(x, x1, y1, x2, y2):
    """host/vmd-sampler.js bezier() in float64: y(x) of the cubic (0,0) (x1,y1) (x2,y2) (1,1)."""
    if x <= 0:
        return 0.0
    if x >= 1:
        return 1.0
    if x1 == y1 and x2 == y2:
        return x
    lo, hi, t = 0.0, 1.0, x
    for _ in range(64):
        s = 1 - t
        fx = 3 * s * s * t * x1 + 3 * s * t * t * x2 + t ** 3 - x
        if abs(fx) < 1e-12:
            break
        if fx > 0:
            hi = t
        else:
            lo = t
        d = 3 * s * s * x1 + 6 * s * t * (x2 - x1) + 3 * t * t * (1 - x2)
        tn = t - fx / d if d != 0 else (lo + hi) / 2
        t = tn if lo < tn < hi else (lo + hi) / 2
    s = 1 - t
    return 3 * s * s * t * y1 + 3 * s * t * t * y2 + t ** 3


def sample_reference(anim, frame, n_bones, n_morphs):
    """MMD motion sampling in float64 (the arithmetic of host/vmd-sampler.js): returns (quats [B,4], trans [B,3],
    morph weights [M]) at `frame` for a flattened motion dict with the rz_animation field names."""
    import numpy as np

    def span(kf, b, e, f):
        lo, hi = b, e - 1
        if f <= kf[lo]:
            return lo, lo, 0.0
        if f >= kf[hi]:
            return hi, hi, 0.0
        while hi - lo > 1:
            mid = (lo + hi) >> 1
            if kf[mid] <= f:
                lo = mid
            else:
                hi = mid
        return lo, hi, (f - kf[lo]) / (kf[hi] - kf[lo])

    q = np.tile(np.array([0.0, 0, 0, 1]), (n_bones, 1))
    t = np.zeros((n_bones, 3))
    kf = np.asarray(anim["key_frame"], dtype=np.float64)
    rot = np.asarray(anim["key_rot"], dtype=np.float64).reshape(-1, 4)
    pos = np.asarray(anim["key_pos"], dtype=np.float64).reshape(-1, 3)
    ip = None if anim.get("key_interp") is None else np.asarray(anim["key_interp"], dtype=np.float64).reshape(-1, 16) / 127.0
    for tr, b in enumerate(anim["track_bone"]):
        lo, hi = int(anim["key_off"][tr]), int(anim["key_off"][tr + 1])
        if b < 0 or b >= n_bones or hi == lo:
            continue
        i0, i1, x = span(kf, lo, hi, frame)
        if i0 == i1:
            q[b], t[b] = rot[i0], pos[i0]
            continue
        c = [x] * 4 if ip is None else [bezier_reference(x, ip[i1][k], ip[i1][k + 4], ip[i1][k + 8], ip[i1][k + 12]) for k in range(4)]
        a, bq = rot[i0].copy(), rot[i1].copy()
        d = float(a @ bq)
        if d < 0:
            d, bq = -d, -bq
        if d > 0.9995:
            r = a + c[3] * (bq - a)
            r /= np.linalg.norm(r)
        else:
            th0 = np.arccos(d)
            r = (np.sin(th0 - th0 * c[3]) * a + np.sin(th0 * c[3]) * bq) / np.sin(th0)
        q[b] = r
        t[b] = pos[i0] + (pos[i1] - pos[i0]) * np.array(c[:3])
    w = np.zeros(n_morphs)
    if n_morphs and anim.get("mkey_off") is not None:
        mkf = np.asarray(anim["mkey_frame"], dtype=np.float64)
        mw = np.asarray(anim["mkey_weight"], dtype=np.float64)
        for m in range(n_morphs):
            for f in range(int(anim["feed_off"][m]), int(anim["feed_off"][m + 1])):
                tr = int(anim["feed_track"][f])
                lo, hi = int(anim["mkey_off"][tr]), int(anim["mkey_off"][tr + 1])
                if hi == lo:
                    continue
                i0, i1, x = span(mkf, lo, hi, frame)
                w[m] += (mw[i0] + (mw[i1] - mw[i0]) * x) * float(anim["feed_ratio"][f])
    return q, t, w
