"""Writers for synthetic PMX 2.0 / VMD byte streams (tests only; no reference asset is involved)."""
import struct

import numpy as np


def _text(s):
    b = s.encode("utf-16le")
    return struct.pack("<i", len(b)) + b


def write_pmx(V=5000, B=40, n_vertex_morphs=6, seed=5, max_depth=None):
    """A PMX with V vertices (BDEF1/2/4 mix), a B-bone tree (one append-rotate bone), `n_vertex_morphs`
    sparse vertex morphs named v0.., plus 'blink' (vertex), 'twist' (a BONE morph, PMX type 2: bone 1 — the append parent of
    bone B/2 — bone 3 — whose translation the append-move bone follows — and bone 7) and 'grp' (group of v0 x0.5 + blink x1.0
    + twist x0.5)."""
    rng = np.random.default_rng(seed)
    out = bytearray(b"PMX ") + struct.pack("<f", 2.0) + bytes([8, 0, 0, 4, 1, 1, 2, 2, 1])
    out += _text("synthetic") + _text("") + _text("") + _text("")
    pos = (rng.random((V, 3), dtype=np.float32) * np.float32(16) - np.float32(8)).astype(np.float32)
    nrm = rng.standard_normal((V, 3), dtype=np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    kinds = rng.choice([0, 1, 2], size=V, p=[0.4, 0.5, 0.1])
    out += struct.pack("<i", V)
    for v in range(V):
        out += pos[v].tobytes() + nrm[v].astype(np.float32).tobytes() + struct.pack("<2f", 0.5, 0.5)
        c = (v * B) // V
        js = np.clip(c + rng.integers(-3, 4, size=4), 0, B - 1)
        out += bytes([int(kinds[v])])
        if kinds[v] == 0:
            out += struct.pack("<h", int(js[0]))
        elif kinds[v] == 1:
            out += struct.pack("<hhf", int(js[0]), int(js[1]), float(rng.random()))
        else:
            out += struct.pack("<4h", *[int(x) for x in js]) + rng.random(4).astype(np.float32).tobytes()
        out += struct.pack("<f", 1.0)
    tri = rng.integers(0, V, size=300).astype(np.int32)
    out += struct.pack("<i", len(tri)) + tri.tobytes()
    out += struct.pack("<i", 0)                                           # textures
    out += struct.pack("<i", 1) + _text("body") + _text("") + struct.pack("<11f", *([0.5] * 11)) + bytes([0x10])
    out += struct.pack("<5f", 0, 0, 0, 1, 1.25) + struct.pack("<bb", -1, -1) + bytes([0, 1, 0]) + _text("") + struct.pack("<i", len(tri))
    bpos = np.cumsum(rng.uniform(-1, 1, size=(B, 3)), axis=0).astype(np.float32)
    out += struct.pack("<i", B)
    depth = []
    for b in range(B):
        parent = -1 if b == 0 else int(rng.integers(max(0, b - 4), b))
        if max_depth is not None and b > 0:        # a humanoid-like tree (the demo model is ~15 levels deep), not a chain
            while depth[parent] >= max_depth - 1:
                parent = int(rng.integers(0, b))
        depth.append(0 if parent < 0 else depth[parent] + 1)
        # one append-rotate bone, and one that appends its append parent's rotation AND translation (ratio beyond 1,
        # so the clamp on the rotation ratio and the unclamped move ratio both show)
        flags = 0x0100 if b == B // 2 else (0x0300 if b == B // 2 + 2 and B > 8 else 0)
        out += _text("bone%d" % b) + _text("") + bpos[b].tobytes() + struct.pack("<h", parent) + struct.pack("<i", 0)
        out += struct.pack("<H", flags) + struct.pack("<3f", 0, 1, 0)
        if flags == 0x0100:
            out += struct.pack("<hf", 1, 0.5)
        elif flags == 0x0300:
            out += struct.pack("<hf", 3, 1.5)
    names = ["v%d" % i for i in range(n_vertex_morphs)] + ["blink"]
    out += struct.pack("<i", len(names) + 2)
    for n in names:
        k = int(rng.integers(V // 50, V // 10))
        start = int(rng.integers(0, V - k))
        idx = np.sort(rng.choice(np.arange(start, start + k), size=k // 2, replace=False)).astype(np.int32)
        d = ((rng.random((len(idx), 3), dtype=np.float32) - np.float32(0.5)) * np.float32(0.4)).astype(np.float32)
        out += _text(n) + _text("") + bytes([1, 1]) + struct.pack("<i", len(idx))
        for i in range(len(idx)):
            out += struct.pack("<i", int(idx[i])) + d[i].tobytes()
    twist = [(1, (0.0, 0.0, 0.0), (0.0, 0.0, 0.38268343, 0.92387953)), (3, (0.4, -0.3, 0.2), (0.25881905, 0.0, 0.0, 0.96592583)),
             (7, (0.0, 0.5, 0.0), (0.0, -0.5, 0.0, 0.8660254))]
    twist = [e for e in twist if e[0] < B]
    out += _text("twist") + _text("") + bytes([1, 2]) + struct.pack("<i", len(twist))
    for b, t, q in twist:
        out += struct.pack("<h", b) + struct.pack("<3f", *t) + struct.pack("<4f", *q)
    out += _text("grp") + _text("") + bytes([1, 0]) + struct.pack("<i", 3) + struct.pack("<hf", 0, 0.5) + struct.pack("<hf", len(names) - 1, 1.0)
    out += struct.pack("<hf", len(names), 0.5)
    out += struct.pack("<i", 0) + struct.pack("<i", 0) + struct.pack("<i", 0)      # display frames, rigid bodies, joints
    return bytes(out)


def write_vmd(bone_keys, morph_keys=()):
    """bone_keys: [(name, frame, (x,y,z,w)[, (px,py,pz)[, 64 interpolation bytes]])]; morph_keys: [(name, frame, weight)]."""
    def name15(s):
        b = s.encode("shift-jis")
        return b + b"\0" * (15 - len(b))
    out = bytearray(b"Vocaloid Motion Data 0002" + b"\0" * 5) + bytearray(b"model" + b"\0" * 15)
    out += struct.pack("<I", len(bone_keys))
    for key in bone_keys:
        n, f, q = key[0], key[1], key[2]
        pos = key[3] if len(key) > 3 else (0, 0, 0)
        interp = key[4] if len(key) > 4 else bytes([20] * 8 + [107] * 8) + bytes(48)
        out += name15(n) + struct.pack("<I", f) + struct.pack("<3f", *pos) + struct.pack("<4f", *q) + interp
    out += struct.pack("<I", len(morph_keys))
    for n, f, w in morph_keys:
        out += name15(n) + struct.pack("<If", f, w)
    return bytes(out)
