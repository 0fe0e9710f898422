#!/usr/bin/env python3
"""Drive the C ABI of the SANITIZED host library (make -C reze-engine_amd/csrc asan: the host sources under AddressSanitizer +
UndefinedBehaviorSanitizer, device code untouched). Re-executes itself with the sanitizer runtime preloaded.
  python tools/asan_run.py cpu            host paths that need no GPU: argument validation with a NULL context on every export,
                                          shard arithmetic, the launch-shape pick rule, create without a device
  python tools/asan_run.py gpu [seeds]    on the GPU box: the misuse script of tests/test_gpu_misuse.py, `seeds` walks (default 2)
                                          of the ABI state-machine fuzz and one sharded walk of tests/test_gpu_fuzz.py, a zero-copy
                                          ring soak (80 uploads of each pose kind, forks, graph replay, peer-direct gather)
Any sanitizer report aborts the process (halt_on_error, -fno-sanitize-recover): exit code 0 = clean."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "asan", "libreze_deform_asan.so")
if os.environ.get("REZE_ASAN_CHILD") != "1":
    # (incremental: a no-op when the sanitized build is up to date with the sources; exit code 77 = the toolchain for it is not
    # here — hipcc for the device objects, g++ with libasan, the ROCm headers — which the CPU test suite reports as a skip)
    import shutil
    if not (shutil.which("g++") and shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc")):
        print("ASAN-SKIP: no g++ / make / hipcc on this machine")
        sys.exit(77)
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()     # gcc's runtime (see the Makefile)
    if not os.path.isabs(rt) or not os.path.exists(rt):
        print("ASAN-SKIP: gcc has no libasan.so here")
        sys.exit(77)
    mk = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "reze-engine_amd", "csrc"), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if mk.returncode != 0:
        print("ASAN-SKIP: `make asan` failed:\n" + mk.stdout.decode()[-2000:])
        sys.exit(77)
    env = dict(os.environ, REZE_ASAN_CHILD="1", LD_PRELOAD=rt + (":" + os.environ["LD_PRELOAD"] if os.environ.get("LD_PRELOAD") else ""),
               # leaks: the interpreter's own; shadow gap: the ROCm runtime maps memory there
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:protect_shadow_gap=0:detect_odr_violation=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
rz.capi.LIB_PATH = LIB                      # every DeformContext / capi.load() of this process binds the sanitized build
L = rz.capi.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
if mode == "cpu":
    n = 0
    for name in rz.capi.SYMBOLS:
        if name in ("rz_last_error", "rz_abi_version", "rz_device_count", "rz_device_numa_node", "rz_create", "rz_shard_range", "rz_instance_range", "rz_gather_chunk", "rz_comm_unique_id", "rz_comm_init_all",
                    "rz_allgather_all", "rz_gather_direct", "rz_destroy", "rz_rccl_info", "rz_autotune_pick"):
            continue
        f = getattr(L, name)
        args = [None] + [0 if t in (ctypes.c_uint32, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t) else None for t in f.argtypes[1:]]
        rc = f(*args)
        assert rc < 0 and L.rz_last_error(), (name, rc)
        n += 1
    assert L.rz_destroy(None) == 0 or True
    for v_total in (0, 1, 255, 256, 257, 1023, 30000, 1000000, 1000001, (1 << 32) - 1):
        for nr in (1, 2, 3, 7, 8, 64):
            spans = [rz.shard_range(v_total, nr, r) for r in range(nr)]
            assert sum(c for _, c in spans) == v_total and all(b + c <= v_total for b, c in spans), (v_total, nr, spans)
            if spans[0][1] > (1 << 32) - 256:      # the rounded-up chunk would not fit 32 bits: refused, not wrapped
                assert L.rz_gather_chunk(v_total, nr, ctypes.byref(ctypes.c_uint32())) < 0
                continue
            ch = rz.capi.gather_chunk(v_total, nr)
            assert ch % 256 == 0 and ch >= spans[0][1] and all(b == min(v_total, r * ch) for r, (b, _) in enumerate(spans)), (v_total, nr, ch, spans)
    for inst in (0, 1, 7, 8, 100, 256, 65535):       # crowds shard along the instance axis: contiguous, tiling, ceil(I / N) each
        for nr in (1, 2, 3, 8, 64):
            spans = [rz.capi.instance_range(inst, nr, r) for r in range(nr)]
            per = -(-inst // nr)
            assert sum(c for _, c in spans) == inst and all(b == min(inst, r * per) and c <= per for r, (b, c) in enumerate(spans)), (inst, nr, spans)
    b, c = ctypes.c_uint32(), ctypes.c_uint32()
    assert L.rz_instance_range(10, 0, 0, ctypes.byref(b), ctypes.byref(c)) < 0 and L.rz_instance_range(10, 2, 2, ctypes.byref(b), ctypes.byref(c)) < 0 and L.rz_instance_range(10, 2, 0, None, None) < 0
    assert L.rz_shard_range(10, 0, 0, ctypes.byref(b), ctypes.byref(c)) < 0 and L.rz_shard_range(10, 2, 2, ctypes.byref(b), ctypes.byref(c)) < 0
    assert L.rz_shard_range(10, 2, 0, None, None) < 0
    T = rz.capi.RzTuneEntry * 4
    t = T()
    for i, (ms, lo, hi, same) in enumerate(((10.0, 9.9, 10.2, -1), (9.5, 9.4, 9.8, -1), (9.9, 9.8, 10.0, -1), (1.0, 1.0, 1.0, 0))):
        t[i].ms, t[i].ms_min, t[i].ms_max, t[i].same_as = ms, lo, hi, same
    assert L.rz_autotune_pick(t, 4) == 1 and L.rz_autotune_pick(None, 0) == 0 and L.rz_autotune_pick(t, 1) == 0
    h = ctypes.c_void_p()
    rc = L.rz_create(0, ctypes.byref(h))
    if rc == 0:
        L.rz_destroy(h)
    else:
        assert L.rz_last_error()
    assert L.rz_create(0, None) < 0 and L.rz_create(-5, ctypes.byref(h)) < 0
    cnt = ctypes.c_int(-1)
    L.rz_device_count(ctypes.byref(cnt)); L.rz_device_count(None)
    print("ASAN-CPU-OK: %d exports refused a NULL context; shard arithmetic, pick rule, create-without-device clean" % n)
    sys.exit(0)

# ---- gpu ----
import test_gpu_misuse, test_gpu_fuzz
import oracle
oracle.build()
exec(test_gpu_misuse.CHILD % ROOT)          # prints MISUSE-OK or exits non-zero
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import types
rzv = types.SimpleNamespace(DeformContext=lambda device=0: rz.DeformContext(device), capi=rz.capi)       # the sanitized build stands in for both libraries
for seed in range(1, 1 + seeds):
    test_gpu_fuzz.test_random_walk_over_the_abi_state_machine(rz, rzv, oracle, seed)
    print("fuzz walk seed %d clean" % seed, flush=True)
test_gpu_fuzz.test_random_walk_over_sharded_contexts(rz, oracle, 101)
print("sharded walk clean", flush=True)
# zero-copy ring soak: every pose kind past the ring's reuse distance, two frames in flight, graph replay
V, B, M = 20000, 120, 9
mesh = synth.make_mesh(V, B, seed=5)
c = rz.DeformContext(0)
c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
off, vi, d3, mw = synth.make_morphs_sparse(V, M, seed=6); c.upload_morphs_sparse(off, vi, d3)
c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
rng = np.random.default_rng(7)
q = rng.normal(size=(B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
lt = (rng.random((B, 3), dtype=np.float32) - 0.5) * 0.1
f = c.fork()
for k in range(80):
    for ctx in (c, f):
        ctx.set_pose(mesh["world"], mw); ctx.deform()
        ctx.set_pose_local(q, mw); ctx.deform()
        ctx.set_pose_local(q, mw, lt); ctx.deform()
        if k % 16 == 0:
            ctx.set_tuning(zero_copy=0); ctx.set_pose(mesh["world"], mw); ctx.deform(); ctx.set_tuning(zero_copy=-1)
c.sync(); f.sync()
c.set_tuning(graph=1); c.set_pose(mesh["world"], mw); c.deform_n(64); c.sync(); c.set_tuning(graph=0)
pg, ng = c.read()
assert np.isfinite(pg).all() and np.isfinite(ng).all()
f.close(); c.close()
print("ASAN-GPU-OK: misuse script, %d + 1 fuzz walks, ring / fork / graph soak clean" % seeds)
