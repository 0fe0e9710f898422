"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (reze-engine_amd/) never does; it fails loudly when its HIP library is missing.

`c` is a ctypes binding of oracle/librz_oracle.so (built from rz_oracle.c by oracle/Makefile);
`np_twin` is the independent NumPy restatement used to cross-check it.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import rz_oracle_np as np_twin  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile rz_oracle.c -> librz_oracle.so (gcc, -ffp-contract=off)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librz_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        L.rzo_palette.argtypes = [fp, fp, ctypes.c_int, fp]
        L.rzo_skin.argtypes = [ctypes.c_int, fp, fp, ctypes.POINTER(ctypes.c_uint16),
                               ctypes.POINTER(ctypes.c_uint8), fp, fp, fp]
        L.rzo_morph_dense.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp]
        L.rzo_morph_sparse.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.POINTER(ctypes.c_uint32), fp, fp, fp, fp]
        L.rzo_deform.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp,
                                 ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint8),
                                 fp, fp, fp, fp, fp, fp, ctypes.c_int]
        L.rzo_hull.argtypes = [ctypes.c_int, fp, fp, fp, fp]
        L.rzo_hull.restype = None
        for f in (L.rzo_palette, L.rzo_skin, L.rzo_morph_dense, L.rzo_morph_sparse, L.rzo_deform):
            f.restype = None
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def palette(world, inv_bind):
    w, wp = _f(world)
    ib, ibp = _f(inv_bind)
    n = w.size // 16
    out = np.empty((n, 16), dtype=np.float32)
    lib().rzo_palette(wp, ibp, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def skin(pos, nrm, joints4, weights4, skin_mats):
    p, pp = _f(pos)
    n, npn = _f(nrm)
    s, sp = _f(skin_mats)
    j = np.ascontiguousarray(joints4, dtype=np.uint16)
    w = np.ascontiguousarray(weights4, dtype=np.uint8)
    V = p.size // 3
    op = np.empty((V, 3), dtype=np.float32)
    on = np.empty((V, 3), dtype=np.float32)
    lib().rzo_skin(V, pp, npn, j.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                   w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), sp,
                   op.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                   on.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return op, on


def morph_dense(deltas, weights, pos):
    d, dp = _f(deltas)
    w, wp = _f(weights)
    p, pp = _f(pos)
    V = p.size // 3
    M = w.size
    assert d.size == M * V * 3
    out = np.empty((V, 3), dtype=np.float32)
    lib().rzo_morph_dense(V, M, dp, wp, pp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def morph_sparse(n_verts, morph_off, vert_idx, delta3, weights, pos):
    mo = np.ascontiguousarray(morph_off, dtype=np.uint32)
    vi = np.ascontiguousarray(vert_idx, dtype=np.uint32)
    d, dp = _f(delta3)
    w, wp = _f(weights)
    p, pp = _f(pos)
    out = np.empty((n_verts, 3), dtype=np.float32)
    u32 = ctypes.POINTER(ctypes.c_uint32)
    lib().rzo_morph_sparse(n_verts, len(mo) - 1, mo.ctypes.data_as(u32), vi.ctypes.data_as(u32),
                           dp, wp, pp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def hull(pos, nrm, edge):
    """Outline hull of engine.ts:458-461 from deformed positions / normals and a per-vertex edge size."""
    p, pp = _f(pos)
    n, npn = _f(nrm)
    e, ep = _f(edge)
    out = np.empty_like(p).reshape(-1, 3)
    lib().rzo_hull(p.size // 3, pp, npn, ep, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def deform(pos, nrm, joints4, weights4, world, inv_bind, deltas=None, morph_w=None, threads=1):
    """Whole frame on the CPU: palette -> dense morph (optional) -> skin."""
    p, pp = _f(pos)
    n, npn = _f(nrm)
    wd, wdp = _f(world)
    ib, ibp = _f(inv_bind)
    j = np.ascontiguousarray(joints4, dtype=np.uint16)
    w = np.ascontiguousarray(weights4, dtype=np.uint8)
    V = p.size // 3
    B = wd.size // 16
    fp = ctypes.POINTER(ctypes.c_float)
    if deltas is not None:
        d, dp = _f(deltas)
        mw, mwp = _f(morph_w)
        M = mw.size
        assert d.size == M * V * 3
    else:
        dp = ctypes.cast(None, fp)
        mwp = ctypes.cast(None, fp)
        M = 0
    op = np.empty((V, 3), dtype=np.float32)
    on = np.empty((V, 3), dtype=np.float32)
    lib().rzo_deform(V, B, M, pp, npn, j.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                     w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), wdp, ibp, dp, mwp,
                     op.ctypes.data_as(fp), on.ctypes.data_as(fp), int(threads))
    return op, on
