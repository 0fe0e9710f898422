"""Round 5 on the GPU: crowd frames of device-animated poses in ONE launch (the hierarchy solved in the skin kernel's front over the
closure of each vertex run's bones), crowd poses through the zero-copy ring, the radix-4 hierarchy solve, sparse staging next to the
LDS limit. Everything goes through the C ABI (ctypes); the oracle and the float64 restatements are the checkers."""
import numpy as np
import pytest

from helpers import fk_reference

pytestmark = pytest.mark.gpu

synth = None


@pytest.fixture(autouse=True)
def _synth(rz):
    global synth
    synth = rz.synth


def _crowd(rz, V, B, I, seed, append=True, depth_chain=14):
    rng = np.random.default_rng(seed)
    mesh = synth.make_mesh(V, B, seed=seed)
    # a skeleton whose parents come in any order, with one long chain and append bones
    order = rng.permutation(B)
    parents = np.full(B, -1, np.int32)
    for k in range(1, depth_chain):
        parents[order[k]] = order[k - 1]
    for k in range(depth_chain, B):
        parents[order[k]] = order[int(rng.integers(0, k))] if rng.random() < 0.95 else -1
    bind = (rng.random((B, 3), dtype=np.float32) - 0.5).astype(np.float32)
    ap = (np.where(rng.random(B) < 0.2, rng.integers(0, B, size=B), -1) if append else np.full(B, -1)).astype(np.int32)
    ratio = (rng.random(B, dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    mv = (rng.random(B) < 0.5).astype(np.uint8)
    q = rng.normal(size=(I, B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    lt = ((rng.random((I, B, 3), dtype=np.float32) - 0.5) * 0.3).astype(np.float32)
    inv_bind = mesh["inv_bind"]
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(inv_bind)
    c.upload_skeleton_topology(parents, bind, ap, ratio, mv)
    c.set_instances(I)
    return c, mesh, dict(parents=parents, bind=bind, ap=ap, ratio=ratio, mv=mv, q=q, lt=lt)


def _all_instances(c, I):
    return [c.read(i) for i in range(I)]


@pytest.mark.parametrize("V,B,I,with_t", [(30000, 200, 24, True), (30000, 200, 19, False), (9000, 349, 9, True), (4097, 40, 3, True)])
def test_crowd_hierarchy_solved_in_the_skin_kernel(rz, oracle, V, B, I, with_t):
    """A device-animated crowd (rz_set_pose_local with I > 1) runs as ONE launch: every workgroup of the crowd kernel solves the closure of
    its vertex run's bones in its front (kernels/crowd.hip: rz_skin_instances_fk_kernel). A bone's world matrix depends on its own chain
    only and both forms run the same device functions, so the frame must equal the two-launch frame (rz_fk_kernel + skin kernel,
    "fuse_fk" = 0) BIT FOR BIT on every instance; world matrices and palettes are formed on demand afterwards and equal too; and the
    whole thing sits within the parity bar of the float64 restatement of Model.computeWorldMatrices + the oracle's skin."""
    c, mesh, s = _crowd(rz, V, B, I, seed=V + B + I)
    lt = s["lt"] if with_t else None
    outs, worlds, pals = {}, {}, {}
    for fuse in (-1, 0):
        c.set_tuning(fuse_fk=fuse)
        c.set_pose_local(s["q"], None, lt)
        assert c.get_tuning("effective_fuse_fk") == (1 if fuse else 0) and c.get_tuning("effective_subsets") == 1
        assert c.get_tuning("effective_prep") == (0 if fuse else 1)          # no front kernel at all in the fused form
        c.deform()
        outs[fuse] = _all_instances(c, I)
        c.deform_n(3)                                                           # replays
        for a, b in zip(outs[fuse], _all_instances(c, I)):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        worlds[fuse] = [c.read_world(i) for i in (0, I - 1)]                    # fused: rz_fk_kernel runs on demand here
        pals[fuse] = [c.read_palette(i) for i in (0, I - 1)]
        c.deform()                                                              # ... and the next frame is the one-launch form again
        assert np.array_equal(c.read(I - 1)[0], outs[fuse][I - 1][0])
    if fuse == 0:
        assert c.get_tuning("effective_closure_bones") == 0
    for i in range(I):
        assert np.array_equal(outs[-1][i][0], outs[0][i][0]) and np.array_equal(outs[-1][i][1], outs[0][i][1]), "instance %d: fused front vs rz_fk_kernel" % i
    for k in range(2):
        assert np.array_equal(worlds[-1][k], worlds[0][k]) and np.array_equal(pals[-1][k], pals[0][k])
    for i in (0, I // 2, I - 1):
        ref = fk_reference(s["parents"], s["bind"], s["q"][i], None if lt is None else lt[i], s["ap"], s["ratio"], s["mv"])
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], ref.reshape(B, 16).astype(np.float32), mesh["inv_bind"])
        ep = np.linalg.norm(outs[-1][i][0] - pr, axis=1) / np.maximum(np.linalg.norm(pr, axis=1), 1.0)
        en = np.linalg.norm(outs[-1][i][1] - nr, axis=1)
        assert ep.max() <= 2e-4 and en.max() <= 2e-4, (i, ep.max(), en.max())      # (a 14-deep f32 chain against float64)
    c.close()


def test_crowd_with_sampled_motion_solved_in_the_skin_kernel(rz):
    """The same with the motion sampled on the device (rz_set_pose_sampled, one frame number per instance): the front of the crowd
    kernel samples the tracks of its closure bones itself. Fused == rz_fk_kernel + skin kernel, bit for bit; a new motion rebuilds the
    closure records (they carry the tracks)."""
    V, B, I = 20000, 120, 21
    c, mesh, s = _crowd(rz, V, B, I, seed=77)
    rng = np.random.default_rng(5)
    for motion in range(2):
        nk = 6 + motion
        kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
        kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        keyed = rng.random(B) < 0.8                                            # a fifth of the bones has no track at all
        tb = np.nonzero(keyed)[0].astype(np.int32)
        n = len(tb)
        c.upload_animation(tb, np.arange(n + 1) * nk, np.tile(np.cumsum(rng.integers(1, 9, size=nk)).astype(np.float32), n), kq[tb].reshape(-1, 4),
                           ((rng.random((n * nk, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32), rng.integers(1, 127, size=(n * nk, 16)).astype(np.uint8))
        frames = (rng.random(I) * 50.0).astype(np.float32)
        outs = {}
        for fuse in (-1, 0):
            c.set_tuning(fuse_fk=fuse)
            c.set_pose_sampled(frames)
            assert c.get_tuning("effective_fuse_fk") == (1 if fuse else 0)
            c.deform()
            outs[fuse] = _all_instances(c, I)
        for i in range(I):
            assert np.array_equal(outs[-1][i][0], outs[0][i][0]) and np.array_equal(outs[-1][i][1], outs[0][i][1]), "motion %d instance %d" % (motion, i)
        assert np.isfinite(outs[-1][0][0]).all()
    c.close()


def test_crowd_falls_back_to_the_fk_kernel_when_something_acts_on_the_solved_pose(rz):
    """Physics overrides replace world matrices between the solve and the palette, bone morphs fold weights into the local pose: both stay
    with rz_fk_kernel (the crowd kernel's front has neither), and so does a skeleton deeper than the records' three doubling rounds."""
    c, mesh, s = _crowd(rz, 12000, 64, 10, seed=3, depth_chain=8)
    c.set_pose_local(s["q"], None, s["lt"])
    assert c.get_tuning("effective_fuse_fk") == 1
    c.deform()
    base = c.read(4)
    w = c.read_world(4)
    c.override_world(np.array([5], np.uint32), w[5:6].copy(), instances=np.array([4], np.uint32))     # the solved matrix itself: nothing may change
    assert c.get_tuning("effective_fuse_fk") == 0
    c.deform()
    assert np.array_equal(c.read(4)[0], base[0])
    c.override_world(np.zeros(0, np.uint32), np.zeros((0, 16), np.float32))
    assert c.get_tuning("effective_fuse_fk") == 1
    c.close()
    c, mesh, s = _crowd(rz, 12000, 90, 6, seed=4, depth_chain=70)               # 70 levels: four rounds
    c.set_pose_local(s["q"], None, s["lt"])
    assert c.get_tuning("effective_fuse_fk") == 0
    c.deform()
    assert np.isfinite(c.read(5)[0]).all()
    c.close()


def _world_crowd(rz, V, B, I, seed, M=0):
    mesh = synth.make_mesh(V, B, seed=seed)
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=seed * 1000 + i) for i in range(I)]).astype(np.float32)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    mws = None
    if M:
        deltas, _ = synth.make_morphs_dense(V, M, seed=seed + 9)
        c.upload_morphs_dense(deltas)
        mws = np.random.default_rng(seed).random((I, M), dtype=np.float32)
        mws[:, ::3] = 0.0
    c.set_instances(I)
    return c, mesh, worlds, mws


@pytest.mark.parametrize("V,B,I,M,tune", [(6000, 201, 23, 0, {}), (30000, 200, 64, 0, {}), (5000, 201, 23, 5, {}), (6000, 201, 23, 0, dict(fast=0, overlap=1)),
                                          (6000, 130, 33, 0, dict(inst_subsets=0))])
def test_crowd_pose_pulled_as_three_rows_equals_the_copied_pose(rz, V, B, I, M, tune):
    """A crowd's world matrices (more than 256 KB) come down by rz_pull_pose_kernel on the upload stream, as the upper three rows of every
    matrix (pose.cpp: pack_rows_avx512; the reference uploads all sixteen floats, engine.ts:2383-2389). The device pose block must hold
    exactly what rz_set_pose was handed — read back through rz_read_world — and the frame must equal the frame of the same pose copied by
    hipMemcpyAsync ("pose_pull" = 0) BIT FOR BIT on every instance; bone counts that are no multiple of a wave's 64-bone step, morph
    weights behind the matrices, the overlapped-front protocol and the whole-palette crowd form included. A pose with a matrix that is
    not affine travels as it is (still pulled)."""
    c, mesh, worlds, mws = _world_crowd(rz, V, B, I, seed=B + I + M, M=M)
    c.set_tuning(**tune)
    assert c.get_tuning("pose_pull") == -1
    packs = "avx512f" in open("/proc/cpuinfo").read()          # (a host without AVX-512 pulls the pose as it is: pose.cpp can_pack_rows)
    outs = {}
    for pull in (-1, 0):
        c.set_tuning(pose_pull=pull)
        c.set_pose(worlds, mws)
        assert c.get_tuning("pose_pulled") == (1 if pull else 0) and c.get_tuning("pose_rows") == (1 if pull and packs else 0)
        c.deform()
        outs[pull] = _all_instances(c, I)
        for i in (0, I // 2, I - 1):
            assert np.array_equal(c.read_world(i).reshape(B, 16), worlds[i].reshape(B, 16)), "world matrices of instance %d (pose_pull = %d)" % (i, pull)
    for i in range(I):
        assert np.array_equal(outs[-1][i][0], outs[0][i][0]) and np.array_equal(outs[-1][i][1], outs[0][i][1]), "instance %d: pulled vs copied" % i
    assert np.isfinite(outs[-1][I - 1][0]).all() and np.abs(outs[-1][I - 1][0]).max() > 0
    # one projective bottom row somewhere: the pose is not packed, the frame does not change (rows 0..2 are all a vertex sees)
    odd = worlds.copy().reshape(I, B, 16)
    odd[I - 1, B - 1, 15] = 2.0
    odd[0, 0, 3] = 0.25
    c.set_tuning(pose_pull=-1)
    c.set_pose(odd, mws)
    assert c.get_tuning("pose_pulled") == 1 and c.get_tuning("pose_rows") == 0
    c.deform()
    for i in (0, I - 1):
        assert np.array_equal(c.read_world(i).reshape(B, 16), odd[i])
        got = c.read(i)
        assert np.array_equal(got[0], outs[0][i][0]) and np.array_equal(got[1], outs[0][i][1])
    c.close()


@pytest.mark.parametrize("with_t", [True, False])
def test_crowd_local_pose_pulled_equals_the_copied_pose(rz, with_t):
    """Local rotations (+ translations, whose byte count is no multiple of 16 here) of a big crowd can take the same pull, as they are ("pose_pull" = 1)."""
    V, B, I = 3000, 67, 271                      # 18 157 bones: 290 KB of rotations, 218 KB of translations
    c, mesh, s = _crowd(rz, V, B, I, seed=12, depth_chain=9)
    lt = s["lt"] if with_t else None
    outs = {}
    for pull in (1, 0, -1):                      # automatic mode leaves local poses with the copy engine (they are shorter than their frame)
        c.set_tuning(pose_pull=pull)
        c.set_pose_local(s["q"], None, lt)
        assert c.get_tuning("pose_pulled") == (1 if pull == 1 else 0) and c.get_tuning("pose_rows") == 0
        c.deform()
        outs[pull] = _all_instances(c, I)
    for i in range(I):
        for other in (0, -1):
            assert np.array_equal(outs[1][i][0], outs[other][i][0]) and np.array_equal(outs[1][i][1], outs[other][i][1]), "instance %d: pulled vs copied" % i
    assert np.isfinite(outs[1][I - 1][0]).all()
    c.close()


def test_crowd_pose_ring_never_serves_a_stale_or_torn_pose(rz):
    """Per-frame loop of a host-animated crowd with nothing waiting in between: every frame's upload goes through the pinned ring and the
    pull (or copy) on the upload stream while earlier frames are still running (eight ring slots, eight device pose blocks whose reuse
    one event per four uploads guards). Three different poses
    cycled for 60 frames; after every frame the outputs of three instances are read back and must be the bits of that pose's frame run
    alone — a pull that started too early (torn), a frame that started before its pull ended, or a ring slot overwritten under a pull
    would show. World matrices and local rotations, on a context and its fork alternating (two frames in flight)."""
    V, B, I = 8000, 200, 40
    c, mesh, worlds, _ = _world_crowd(rz, V, B, I, seed=5)
    poses = [worlds, worlds[::-1].copy(), np.roll(worlds, 7, axis=0).copy()]
    picks = (0, 17, I - 1)
    iso = []
    for p in poses:
        c.set_pose(p)
        c.deform()
        c.sync()
        iso.append([c.read(i) for i in picks])
    assert not np.array_equal(iso[0][0][0], iso[1][0][0])
    f = c.fork()
    ctxs = (c, f)
    last = {}
    for k in range(60):
        x = ctxs[k & 1]
        x.set_pose(poses[k % 3])
        x.deform()
        last[k & 1] = k % 3
        if k % 7 == 6 or k >= 56:
            for y in (0, 1):
                for n, i in enumerate(picks):
                    got = ctxs[y].read(i)
                    assert np.array_equal(got[0], iso[last[y]][n][0]) and np.array_equal(got[1], iso[last[y]][n][1]), "frame %d context %d instance %d" % (k, y, i)
    f.close()
    # the same context now animated by local rotations through the same ring
    c.close()
    c, mesh, s = _crowd(rz, 6000, 120, 160, seed=8, depth_chain=10)
    qs = [s["q"], s["q"][::-1].copy(), np.roll(s["q"], 5, axis=0).copy()]
    iso = []
    for q in qs:
        c.set_pose_local(q)
        c.deform()
        iso.append([c.read(i) for i in picks])
    for k in range(90):
        if k == 45:
            c.set_tuning(pose_pull=1)           # the second half pulled instead of copied
        c.set_pose_local(qs[k % 3])
        assert c.get_tuning("pose_pulled") == (1 if k >= 45 else 0)
        c.deform()
        if k % 5 == 4:
            for n, i in enumerate(picks):
                got = c.read(i)
                assert np.array_equal(got[0], iso[k % 3][n][0]) and np.array_equal(got[1], iso[k % 3][n][1]), "local frame %d instance %d" % (k, i)
    c.close()


def test_numa_node_query_and_binding(rz):
    """rz_device_numa_node reads the NUMA node of the device's PCI function out of sysfs (-1 when the system does not say); binding to it
    (capi.bind_to_device_node: what bench.py does for every rank) must leave the process on cores of exactly that node — checked in a
    child process, so this one keeps its affinity. A device that does not exist is refused, and the refusal leaves no stale HIP error
    behind for the next launch check (the N-API misuse script found that one)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    node = rz.capi.device_numa_node(0)
    assert isinstance(node, int) and node >= -1
    with pytest.raises(rz.capi.RzError):
        rz.capi.device_numa_node(99)
    c = rz.DeformContext(0)                                     # a launch right after the refused query: no stale error may surface
    mesh = synth.make_mesh(2000, 8)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
    c.set_pose(mesh["world"]); c.deform()
    assert np.isfinite(c.read()[0]).all()
    c.close()
    code = ("import json, os, sys; sys.path.insert(0, %r); import reze_engine_amd as rz; "
            "b = rz.capi.bind_to_device_node(0); print(json.dumps({'b': b, 'cpus': sorted(os.sched_getaffinity(0))}))" % root)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-800:]
    r = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][-1])
    if node < 0 or r["b"] is None:
        assert r["b"] is None                                   # one node / unknown: nothing bound
        return
    assert r["b"]["gpu_node"] == node
    want = set()
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        lo, _, hi = part.partition("-")
        want.update(range(int(lo), int(hi or lo) + 1))
    assert set(r["cpus"]) and set(r["cpus"]) <= want


@pytest.mark.parametrize("morphs", ["none", "sparse"])
def test_fused_frame_of_a_plain_pose_runs_the_specialised_solve_with_the_same_bits(rz, morphs):
    """The fused single-character frame of a PLAIN pose (no physics overrides, <= 512 bones, <= 256 morphs) runs a kernel variant
    whose hierarchy solve is specialised at compile time for an uploaded / a sampled pose (fk_solve<true, KIND>: a third / two thirds of
    the generic kernel's code, no scalar spills). Same device functions: the frame must equal the generic kernel's ("fuse_fk_plain" = 0)
    bit for bit, for local rotations, rotations + translations and a sampled motion; a pose the variants do not cover (an override set) goes back to
    the generic kernel by itself."""
    V, B = 20000, 300
    mesh = synth.make_mesh(V, B, seed=31)
    rng = np.random.default_rng(9)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
    mw = None
    if morphs == "sparse":
        off, vi, d3, mw = synth.make_morphs_sparse(V, 24, density=0.03)
        c.upload_morphs_sparse(off, vi, d3)
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    q = rng.normal(size=(B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    lt = ((rng.random((B, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    nk = 7
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32); kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    c.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.cumsum(rng.integers(1, 9, size=nk)).astype(np.float32), B), kq.reshape(-1, 4),
                       ((rng.random((B * nk, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32), rng.integers(1, 127, size=(B * nk, 16)).astype(np.uint8))
    poses = {"local": lambda: c.set_pose_local(q, mw), "local+t": lambda: c.set_pose_local(q, mw, lt), "sampled": lambda: c.set_pose_sampled(np.array([11.3], np.float32))}
    for name, put in poses.items():
        outs = {}
        for plain in (-1, 0):
            c.set_tuning(fuse_fk_plain=plain, fuse_fk=1)
            put()
            assert c.get_tuning("effective_fuse_fk") == 1
            assert c.get_tuning("effective_fk_kind") == ((2 if name == "sampled" else 1) if plain else 0), name
            c.deform()
            outs[plain] = (c.read(), c.read_world(0), c.read_palette(0))
            c.deform_n(3)
            assert np.array_equal(c.read()[0], outs[plain][0][0])
        assert np.array_equal(outs[-1][0][0], outs[0][0][0]) and np.array_equal(outs[-1][0][1], outs[0][0][1]), name
        assert np.array_equal(outs[-1][1], outs[0][1]) and np.array_equal(outs[-1][2], outs[0][2]), name
        assert np.isfinite(outs[-1][0][0]).all()
    # something the variants do not cover
    c.set_tuning(fuse_fk_plain=-1)
    c.set_pose_local(q, mw)
    w = c.read_world(0)
    c.override_world(np.array([3], np.uint32), w[3:4].copy())
    assert c.get_tuning("effective_fk_kind") == 0
    c.override_world(np.zeros(0, np.uint32), np.zeros((0, 16), np.float32))
    assert c.get_tuning("effective_fk_kind") == 1
    c.close()
