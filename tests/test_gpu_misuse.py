"""Misuse of the C ABI must come back as a negative rz_status with a message — never a crash. Runs in a child process so
that a segfault shows up as a test failure instead of taking pytest down."""
import subprocess
import sys
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys
sys.path.insert(0, %r)
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
L = rz.capi.load()
c = rz.DeformContext(0)
h = c._h
N = None
bad = []
def expect_fail(name, rc):
    if rc >= 0:
        bad.append(name + " returned %%d" %% rc)
    elif not L.rz_last_error():
        bad.append(name + " left no message")
fp = ctypes.POINTER(ctypes.c_float)
# null context everywhere
for name in rz.capi.SYMBOLS:
    if name in ("rz_last_error", "rz_abi_version", "rz_device_count", "rz_device_numa_node", "rz_create", "rz_shard_range", "rz_instance_range", "rz_gather_chunk", "rz_comm_unique_id",
                "rz_comm_init_all", "rz_allgather_all", "rz_gather_direct", "rz_destroy", "rz_rccl_info", "rz_autotune_pick"):      # rz_destroy(NULL) is a no-op, like free; rz_autotune_pick takes a table and returns an index
        continue
    f = getattr(L, name)
    args = [None] + [0 if t in (ctypes.c_uint32, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t) else None for t in f.argtypes[1:]]
    expect_fail(name + "(NULL ctx)", f(*args))
# null / inconsistent data on a live context
expect_fail("upload_mesh(NULL)", L.rz_upload_mesh(h, 10, N, N, N))
expect_fail("upload_mesh_soa(NULL)", L.rz_upload_mesh_soa(h, 10, N, N, N, N))
expect_fail("upload_mesh(V=0)", L.rz_upload_mesh(h, 0, N, N, N))
expect_fail("upload_skeleton(NULL)", L.rz_upload_skeleton(h, 4, N))
expect_fail("upload_skeleton(B=0)", L.rz_upload_skeleton(h, 0, N))
expect_fail("morphs before mesh", L.rz_upload_morphs_dense(h, 2, N))
expect_fail("set_pose before skeleton", L.rz_set_pose(h, N, N))
expect_fail("deform before anything", L.rz_deform(h))
expect_fail("time_frames before anything", L.rz_time_frames(h, 3, N))
expect_fail("autotune before anything", L.rz_autotune(h, 3))
expect_fail("read before anything", L.rz_read(h, 0, 0, 1, N, N))
expect_fail("create(NULL out)", L.rz_create(0, N))
expect_fail("create(device 99)", L.rz_create(99, ctypes.byref(ctypes.c_void_p())))
expect_fail("shard_range(NULL)", L.rz_shard_range(100, 2, 0, N, N))
expect_fail("instance_range(NULL)", L.rz_instance_range(100, 2, 0, N, N))
expect_fail("instance_range(rank >= n)", L.rz_instance_range(100, 2, 2, ctypes.byref(ctypes.c_uint32()), ctypes.byref(ctypes.c_uint32())))
expect_fail("commit_pose without a mapping", L.rz_commit_pose(h))
expect_fail("map_pose before a skeleton", L.rz_map_pose(h, 0, ctypes.byref(fp()), N))
expect_fail("time_span before anything", L.rz_time_span(h, N, 0, 3, ctypes.byref(ctypes.c_double())))
expect_fail("shard_range(rank >= n)", L.rz_shard_range(100, 2, 5, ctypes.byref(ctypes.c_uint32()), ctypes.byref(ctypes.c_uint32())))
expect_fail("gather_direct(NULL list)", L.rz_gather_direct(N, 2, 100, 0))
expect_fail("comm_init_all(NULL list)", L.rz_comm_init_all(N, 2, 100))
expect_fail("allgather_all(NULL list)", L.rz_allgather_all(N, 2, 0))
expect_fail("comm_unique_id(NULL)", L.rz_comm_unique_id(N))
m = synth.make_mesh(300, 6, seed=1)
c.upload_mesh(m["pos"], m["nrm"], m["joints"], m["weights"]); c.upload_skeleton(m["inv_bind"])
expect_fail("set_pose(NULL world)", L.rz_set_pose(h, N, N))
expect_fail("map_pose(NULL out)", L.rz_map_pose(h, 0, N, N))
expect_fail("map_pose(layout 5)", L.rz_map_pose(h, 5, ctypes.byref(fp()), N))
expect_fail("map_pose(rows for one character)", L.rz_map_pose(h, 1, ctypes.byref(fp()), N))
expect_fail("time_span(0 frames)", L.rz_time_span(h, N, 2, 0, ctypes.byref(ctypes.c_double())))
expect_fail("time_span(NULL out)", L.rz_time_span(h, N, 0, 3, N))
expect_fail("time_span(ctx, ctx)", L.rz_time_span(h, h, 0, 3, ctypes.byref(ctypes.c_double())))
expect_fail("dense morphs NULL deltas", L.rz_upload_morphs_dense(h, 3, N))
expect_fail("sparse morphs NULL offsets", L.rz_upload_morphs_sparse(h, 3, N, N, N))
off = (ctypes.c_uint32 * 3)(0, 5, 2)
expect_fail("sparse morphs decreasing offsets", L.rz_upload_morphs_sparse(h, 2, off, N, N))
expect_fail("set_instances(0)", L.rz_set_instances(h, 0))
expect_fail("set_instances(70000)", L.rz_set_instances(h, 70000))
expect_fail("set_pose_local without topology", L.rz_set_pose_local(h, m["quats"].ctypes.data_as(fp), N, N))
expect_fail("set_pose_sampled without motion", L.rz_set_pose_sampled(h, m["quats"].ctypes.data_as(fp)))
expect_fail("upload_animation(NULL)", L.rz_upload_animation(h, N))
expect_fail("topology NULL parents", L.rz_upload_skeleton_topology(h, 6, N, N, N, N, N))
expect_fail("topology wrong B", L.rz_upload_skeleton_topology(h, 5, m["parents"].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), m["bind"].ctypes.data_as(fp), N, N, N))
cyc = np.array([1, 2, 0, 0, 0, 0], dtype=np.int32)
expect_fail("topology with a cycle", L.rz_upload_skeleton_topology(h, 6, cyc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), m["bind"].ctypes.data_as(fp), N, N, N))
expect_fail("edge scale wrong V", L.rz_upload_edge_scale(h, 7, m["quats"].ctypes.data_as(fp)))
expect_fail("read_palette before pose", L.rz_read_palette(h, 0, N))
expect_fail("tuning unknown key", L.rz_set_tuning(h, b"nonsense", 1))
expect_fail("ablation key absent from the product", L.rz_set_tuning(h, b"dbg", 3))
expect_fail("inst_block 300", L.rz_set_tuning(h, b"inst_block", 300))
expect_fail("overlap 7", L.rz_set_tuning(h, b"overlap", 7))
expect_fail("override_world without topology", L.rz_override_world(h, 1, N, (ctypes.c_uint32 * 1)(0), m["world"].ctypes.data_as(fp)))
u1 = (ctypes.c_uint32 * 1)(0)
expect_fail("bone morphs without topology", L.rz_upload_bone_morphs(h, 1, u1, u1, m["bind"].ctypes.data_as(fp), m["quats"].ctypes.data_as(fp)))
if L.rz_upload_bone_morphs(h, 0, N, N, N, N) != 0: bad.append("rz_upload_bone_morphs(n = 0) clears and is always legal")
fk = ctypes.c_void_p()
expect_fail("fork NULL out", L.rz_fork(h, N))
expect_fail("deform_pair with itself", L.rz_deform_pair(h, h, 2))
expect_fail("deform_pair NULL partner", L.rz_deform_pair(h, N, 2))
if L.rz_rccl_info(N, 0, N, N) not in (0, -6): bad.append("rccl_info(NULLs) must be OK or UNSUPPORTED")
expect_fail("tuning NULL key", L.rz_set_tuning(h, N, 1))
expect_fail("get_tuning NULL out", L.rz_get_tuning(h, b"bones", N))
c.set_pose(m["world"]); c.deform()
expect_fail("read out of range", L.rz_read(h, 0, 290, 20, N, N))
expect_fail("read instance out of range", L.rz_read(h, 3, 0, 1, N, N))
expect_fail("read_hull without edge scale", L.rz_read_hull(h, 0, 0, 1, N))
expect_fail("read_aabb while off", L.rz_read_aabb(h, 0, N))
expect_fail("read_gathered without gather", L.rz_read_gathered(h, 0, 1, N, N))
expect_fail("gather_fence on a non-root", L.rz_gather_fence(h))
expect_fail("allgather without comm", L.rz_allgather(h, 0))
expect_fail("comm_init bad rank", L.rz_comm_init(h, 2, 7, ctypes.create_string_buffer(128), 300))
expect_fail("time_frames NULL out", L.rz_time_frames(h, 3, N))
expect_fail("autotune_measure NULL table", L.rz_autotune_measure(h, 3, N, 0, N))
expect_fail("autotune_apply NULL entry", L.rz_autotune_apply(h, N))
bad_entry = rz.capi.RzTuneEntry(); bad_entry.morph_split = 3
expect_fail("autotune_apply bad split", L.rz_autotune_apply(h, ctypes.byref(bad_entry)))
if L.rz_autotune_pick(N, 0) != 0: bad.append("rz_autotune_pick(NULL) must answer entry 0")
expect_fail("comm_info without comm", L.rz_comm_info(h, N, N))
expect_fail("inst_subsets 5", L.rz_set_tuning(h, b"inst_subsets", 5))
# and the context still works afterwards
c.deform(); p, n = c.read()
assert np.isfinite(p).all()
c.close()
expect_fail("use after destroy is caught for NULL only", L.rz_deform(None))
print("MISUSE-OK" if not bad else "MISUSE-BAD: " + "; ".join(bad))
'''


@pytest.mark.gpu
def test_c_abi_misuse_returns_errors_never_crashes():
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, "child died with %d\n%s\n%s" % (p.returncode, out[-2000:], p.stderr.decode()[-3000:])
    assert "MISUSE-OK" in out, out[-3000:]


@pytest.mark.gpu
def test_napi_addon_misuse_with_a_live_context():
    """tests/js/addon_misuse.js --live: every export called with a live context followed by junk (wrong types, wrong
    lengths, huge ranges): exceptions or values, the context survives and still deforms."""
    import json
    import shutil
    if shutil.which("node") is None:
        pytest.skip("node is not installed on this box")
    p = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "addon_misuse.js"), "--live"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0, "node died with %d\n%s" % (p.returncode, p.stderr.decode()[-3000:])
    r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert r["alive"] is True and r["live"] >= 400


@pytest.mark.gpu
def test_contexts_give_all_device_memory_back():
    """Forty create / use-everything / destroy cycles: the free device memory reported by the runtime must come back to
    where it started (a leak of even one 1 MB buffer per cycle would show as 40 MB)."""
    child = r'''
import sys
sys.path.insert(0, %r)
import ctypes
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
V, B, M, I = 60000, 120, 12, 6
mesh = synth.make_mesh(V, B, seed=3)
deltas, mw = synth.make_morphs_dense(V, M, seed=4)
sp = synth.make_morphs_sparse(V, M, seed=5)[:3]
q = np.tile(np.array([0, 0, 0, 1], np.float32), (I, B, 1))
def cycle(k):
    cs = []
    for r in range(2):
        b, n, _ = rz.shard.shard_of(V, 2, r)
        shard, d = rz.shard.cut_mesh(mesh, deltas, b, n)
        c = rz.DeformContext(0)
        c.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"]); c.upload_skeleton(mesh["inv_bind"])
        c.upload_morphs_dense(d)
        c.upload_edge_scale(np.ones(n, np.float32)); c.enable_aabb(True)
        c.set_pose(mesh["world"], mw); c.deform(); c.autotune(2)
        cs.append(c)
    rz.capi.gather_direct(cs, V, root=k %% 2)
    for c in cs:
        c.set_pose(mesh["world"], mw); c.deform()
    cs[k %% 2].read_gathered()
    big = rz.DeformContext(0)
    big.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); big.upload_skeleton(mesh["inv_bind"])
    big.upload_morphs_sparse(*sp)
    big.set_instances(I)
    big.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    big.set_pose_local(q, None, np.zeros((I, B, 3), np.float32)); big.deform()
    big.upload_animation(np.arange(B), np.arange(B + 1) * 2, np.tile([0.0, 9.0], B), np.tile([0, 0, 0, 1], (2 * B, 1)), np.zeros((2 * B, 3)))
    big.set_pose_sampled(np.arange(I, dtype=np.float32)); big.deform(); big.read(instance=I - 1)
    big.upload_morphs_dense(None); big.set_pose(np.stack([mesh["world"]] * I)); big.deform_n(3); big.time_frames(2)
    for c in cs + [big]:
        c.close()
cycle(0); cycle(1)
base = free_bytes()
for k in range(40):
    cycle(k)
leak = base - free_bytes()
print("LEAK-BYTES", leak)
'''
    p = subprocess.run([sys.executable, "-c", child % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, "child died with %d\n%s" % (p.returncode, p.stderr.decode()[-3000:])
    leak = int(out.strip().splitlines()[-1].split()[-1])
    assert leak < (8 << 20), "device memory not returned: %d bytes after 40 cycles" % leak
