"""Per-wave timeline of ONE frame (tools-only build: make -C reze-engine_amd/csrc ablate; rz_set_tuning dbg = 100).
Every wave stamps the chip-wide 100 MHz counter (10 ns steps) at: 0 entry, 1 prologue done, 2 first step's morph phase done,
3 palette published, 4 first step's skin phase issued, 5 last step done, 6 all stores acknowledged (crowd kernel: 0 entry,
1 staged matrices landed, 2 palettes published, 3 first vertex step done, 5 last step issued, 6 stores acknowledged).
usage: python tools/archive/timeline.py <config> [key=value ...]     config: c2 c3 demo sparse2 shard c5 c4 | sampled-c2 sampled-demo local-c2 local-c4 sampled-c4"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = rz.capi.load(os.path.join(ROOT, "tools", "ablate", "libreze_deform_ablate.so"))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
tune = dict(kv.split("=") for kv in sys.argv[2:])
anim = None
if "-" in cfg:
    anim, cfg = cfg.split("-")
shapes = {"c2": (30000, 200, 0, 1, None), "c3": (30000, 200, 64, 1, "dense"), "demo": (28842, 349, 60, 1, "demo"), "sparse2": (28842, 349, 60, 1, "sparse2"),
          "shard": (125184, 256, 64, 1, "dense"), "c5": (1000000, 256, 64, 1, "dense"), "c4": (30000, 200, 0, 256, None)}
V, B, M, I, kind = shapes[cfg]
ctx = rz.DeformContext(0, lib=L)
mesh = synth.make_mesh_range(max(V, 30000), B, 0, V) if kind == "dense" else synth.make_mesh(V, B)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
mw = None
if kind == "dense":
    d, mw = synth.make_morphs_dense_range(max(V, 30000), M, 0, V); ctx.upload_morphs_dense(d); del d
elif kind == "demo":
    off, vi, d3, mw = synth.make_morphs_demo_shape(V, M); ctx.upload_morphs_sparse(off, vi, d3)
elif kind == "sparse2":
    off, vi, d3, mw = synth.make_morphs_sparse(V, M, density=0.02); ctx.upload_morphs_sparse(off, vi, d3)
else:
    ctx.upload_morphs_dense(None)
ctx.set_instances(I)
world = mesh["world"]
if I > 1:
    world = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
for k, v in tune.items():
    ctx.set_tuning(**{k: int(v)})
if anim:
    ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    rng = np.random.default_rng(1)
    if anim == "sampled":
        nk = 8
        kq = rng.normal(size=(B, nk, 4)).astype(np.float32); kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        extra = {}
        if M:
            extra = dict(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 70.0], np.float32), M), mkey_weight=np.repeat(mw, 2),
                         feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
        ctx.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2,
                             np.tile(np.array([20] * 8 + [107] * 8, np.uint8), B * nk), **extra)
        ctx.set_pose_sampled((13.5 + 0.37 * np.arange(I)).astype(np.float32) % 70.0)
    else:
        q = rng.normal(size=(I, B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=2, keepdims=True)
        ctx.set_pose_local(q if I > 1 else q[0], None, mw)
else:
    ctx.set_pose(world, mw)
for _ in range(5):
    ctx.deform_n(200); ctx.sync()
n = ctypes.c_uint32(0)
L.rz_debug_timeline_arm.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
L.rz_debug_timeline_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32]
assert L.rz_debug_timeline_arm(ctx._h, ctypes.byref(n)) == 0, L.rz_last_error()
ctx.deform_n(50); ctx.sync()
ref = ctx.time_frames(500)
buf = np.zeros((n.value, 16), np.uint64)
assert L.rz_debug_timeline_read(ctx._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value) == 0, L.rz_last_error()
t = buf[buf[:, 0] != 0]
if len(t) == 0:
    sys.exit("no wave stamped anything")
t0 = int(t[:, 0].min())
us = lambda x: (np.asarray(x, np.int64) - t0) * 0.01
print("%s%s %s: %d waves stamped of %d slots; frame by events %.2f us, kernel %.2f us (with stamps compiled in); kernel %s" % (
    (anim + "-") if anim else "", cfg, " ".join("%s=%s" % kv for kv in tune.items()), len(t), n.value, ref["frame_ms"] * 1e3, ref["deform_kernel_ms"] * 1e3, ctx.kernel_name() if hasattr(ctx, "kernel_name") else ""))
names = ["entry", "prologue done", "first morph phase done", "palette published", "first skin phase issued", "last step done", "stores acknowledged"]
print("  stamp                        first     p10      p50      p90      last   (us after the first wave's entry)")
for k in range(7):
    col = t[:, k]
    col = col[col != 0]
    if len(col) == 0:
        continue
    u = us(col)
    print("  %d %-24s %7.2f %8.2f %8.2f %8.2f %8.2f" % (k, names[k], u.min(), np.percentile(u, 10), np.percentile(u, 50), np.percentile(u, 90), u.max()))
dur = (t[:, 6].astype(np.int64) - t[:, 0].astype(np.int64)) * 0.01
print("  wave lifetime (entry -> stores acknowledged): min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (dur.min(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max()))
seg = []
prev = 0
for k in (1, 2, 3, 4, 5, 6):
    if (t[:, k] != 0).all():
        d = (t[:, k].astype(np.int64) - t[:, prev].astype(np.int64)) * 0.01
        seg.append("%d->%d %.2f / %.2f" % (prev, k, np.percentile(d, 50), d.max()))
        prev = k
print("  segments, median / max over waves (us): " + " | ".join(seg))
xcc = (t[:, 7] & 0xff).astype(int)
print("  per XCD: " + " | ".join("x%d n=%d in %.2f..%.2f out ..%.2f" % (x, (xcc == x).sum(), us(t[xcc == x, 0]).min(), us(t[xcc == x, 0]).max(), us(t[xcc == x, 6]).max()) for x in sorted(set(xcc))))
# where an XCD's lag comes from: median of every stamp per XCD, and the median wave's time between the first step and the last
for k in (1, 2, 3, 5):
    if (t[:, k] != 0).all():
        print("  per XCD, median of stamp %d (%s): " % (k, names[k]) + " ".join("x%d %.2f" % (x, np.median(us(t[xcc == x, k]))) for x in sorted(set(xcc))))
if (t[:, 8] != 0).any():
    f = t[t[:, 8] != 0]
    fn = ["pose staged", "barrier passed", "local matrices formed", "doubling rounds done", "palette rows written"]
    print("  inside the hierarchy solve (us after the wave's own entry), median / max: " + " | ".join(
        "%s %.2f / %.2f" % (fn[k], np.median((f[:, 8 + k].astype(np.int64) - f[:, 0].astype(np.int64)) * 0.01), ((f[:, 8 + k].astype(np.int64) - f[:, 0].astype(np.int64)) * 0.01).max()) for k in range(5)))
wpw = 8 if cfg == "c4" else 4
nwg = n.value // wpw
full = buf[: nwg * wpw].reshape(nwg, wpw, 16)
okwg = (full[:, :, 0] != 0).all(axis=1)
if okwg.any():
    e5 = (full[okwg][:, :, 5].astype(np.int64) - t0) * 0.01
    print("  'last step done' inside a workgroup: spread max - min, median over workgroups %.2f us (max %.2f); across workgroups the slowest wave ends %.2f .. %.2f us" % (
        np.median(e5.max(axis=1) - e5.min(axis=1)), (e5.max(axis=1) - e5.min(axis=1)).max(), e5.max(axis=1).min(), e5.max(axis=1).max()))
late = np.argsort(t[:, 6])[-3:]
for w in late:
    print("  late wave: " + " ".join("%.2f" % v for v in us(t[w, :7])) + "  xcc %d" % (int(t[w, 7]) & 0xff))
