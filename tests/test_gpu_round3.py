"""GPU tests added in round 3: the bone-subset form of the crowd kernel (rz_skin_instances_kernel<..., SUB>), the
launch-shape search's stability rules, and the regression tests of the round-2 review."""
import numpy as np
import pytest

from helpers import assert_parity
from reze_engine_amd import synth

pytestmark = pytest.mark.gpu


def _crowd(rz, mesh, worlds, **tuning):
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(len(worlds))
    c.set_tuning(**tuning)
    c.set_pose(worlds)
    return c


def _poses(mesh, B, I, seed=4000):
    return np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=seed + i) for i in range(I)])


@pytest.mark.parametrize("V,B,I,tuning", [
    (30000, 200, 24, {}),                                    # the C4 shape, three pose groups
    (30000, 200, 24, {"inst_block": 256}),                   # two workgroups per CU
    (30000, 200, 40, {"inst_loop": 16}),                     # 16 poses per workgroup, last group partial (40 = 16 + 16 + 8)
    (30000, 200, 24, {"fast": 0}),                           # rz_prep_kernel in front: finished rows staged (48-byte DMA)
    (4097, 33, 5, {"grid_cap": 64}),                         # ragged: last run short, last group partial, odd bone count
    (1023, 40, 3, {}),                                       # fewer vertices than one workgroup step
    (20000, 1500, 6, {"grid_cap": 256}),                     # skeleton far larger than the workgroup: only subsets fit one launch
])
def test_bone_subset_crowd_equals_whole_palette_crowd(rz, oracle, V, B, I, tuning):
    """rz_skin_instances_kernel<.., SUB> (engine.ts:253-272 per instance): a workgroup stages only the bones its vertex run
    names. Same rows, same FMA chains => every instance is BIT-IDENTICAL to the whole-palette form, and oracle-green."""
    mesh = synth.make_mesh(V, B)
    worlds = _poses(mesh, B, I)
    c = _crowd(rz, mesh, worlds, **tuning)
    c.deform()
    assert c.get_tuning("effective_inst_group") >= 2
    assert c.get_tuning("effective_subsets") == 1, "the synthetic mesh is bone-local: the subset form must be planned"
    nb = c.get_tuning("effective_subset_bones")
    assert 0 < nb < B
    sub = [c.read(instance=k) for k in range(I)]
    pal = [c.read_palette(k) for k in (0, I - 1)]
    c.set_tuning(inst_subsets=0)
    c.deform()
    assert c.get_tuning("effective_subsets") == 0
    for k in range(I):
        pg, ng = c.read(instance=k)
        assert np.array_equal(pg, sub[k][0]) and np.array_equal(ng, sub[k][1]), "instance %d: subset vs whole palette" % k
    for a, k in zip(pal, (0, I - 1)):
        S = oracle.palette(worlds[k], mesh["inv_bind"]).reshape(-1, 4, 4)
        np.testing.assert_allclose(a, np.transpose(S, (0, 2, 1))[:, :3, :].reshape(-1, 12), rtol=1e-6, atol=1e-6)
        if c.get_tuning("effective_inst_group") >= 2 and B <= 512:
            assert np.array_equal(a, c.read_palette(k)), "palette formed on demand vs written by the frame"
    for k in sorted({0, I // 2, I - 1}):
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[k], mesh["inv_bind"], threads=4)
        assert_parity(sub[k][0], sub[k][1], pr, nr, "subset crowd instance %d" % k)
    c.close()


def test_bone_subsets_follow_the_launch_shape_mesh_and_skeleton(rz, oracle):
    """The run lists are derived data: a new shape (grid_cap / inst_loop / instance count), a new mesh or a new skeleton
    must rebuild them — a stale list would gather the wrong bones. Also: joints beyond the skeleton are clamped the same
    way in both forms, zero-weight influences keep their bone, and a mesh that is NOT bone-local falls back."""
    V, B, I = 12000, 96, 12
    mesh = synth.make_mesh(V, B)
    worlds = _poses(mesh, B, I)
    c = _crowd(rz, mesh, worlds)

    def check(what):
        c.deform()
        got = [c.read(instance=k) for k in (0, I - 1)]
        sub_on = c.get_tuning("effective_subsets")
        c.set_tuning(inst_subsets=0)
        c.deform()
        for (pg, ng), k in zip(got, (0, I - 1)):
            p0, n0 = c.read(instance=k)
            assert np.array_equal(pg, p0) and np.array_equal(ng, n0), "%s: instance %d" % (what, k)
        c.set_tuning(inst_subsets=-1)
        return sub_on

    assert check("first shape") == 1
    for cap in (64, 512, 1024):
        c.set_tuning(grid_cap=cap)
        assert check("grid_cap %d" % cap) == 1
    c.set_tuning(grid_cap=0, inst_loop=4)
    assert check("inst_loop 4") == 1
    c.set_tuning(inst_loop=-1)
    c.set_instances(7)
    c.set_pose(worlds[:7])
    I = 7
    assert check("7 instances") == 1
    # a new mesh on the same context: other joints, other size
    mesh2 = synth.make_mesh(9001, B, seed=77)
    mesh2["joints"][::5] = 60000                    # out-of-range joints: clamped to B - 1 by both forms
    mesh2["joints"][1::7, 1:] = B - 1               # zero-weight influences naming a far bone ...
    mesh2["weights"][1::7] = np.array([255, 0, 0, 0], np.uint8)
    c.upload_mesh(mesh2["pos"], mesh2["nrm"], mesh2["joints"], mesh2["weights"])
    c.set_pose(worlds[:7])
    assert check("new mesh") == 1
    pg, ng = c.read(instance=3)
    jc = np.minimum(mesh2["joints"], B - 1).astype(np.uint16)
    pr, nr = oracle.deform(mesh2["pos"], mesh2["nrm"], jc, mesh2["weights"], worlds[3], mesh["inv_bind"], threads=4)
    assert_parity(pg, ng, pr, nr, "new mesh, clamped joints")
    # a new skeleton (fewer bones): joints are clamped to the NEW bone count
    B2 = 40
    m3 = synth.make_mesh(9001, B2, seed=5)
    c.upload_skeleton(m3["inv_bind"])
    w3 = _poses(m3, B2, 7, seed=9)
    c.set_pose(w3)
    check("new skeleton")
    pg, ng = c.read(instance=6)
    jc = np.minimum(mesh2["joints"], B2 - 1).astype(np.uint16)
    pr, nr = oracle.deform(mesh2["pos"], mesh2["nrm"], jc, mesh2["weights"], w3[6], m3["inv_bind"], threads=4)
    assert_parity(pg, ng, pr, nr, "new skeleton, clamped joints")
    # joints scattered over the whole skeleton: every run names every bone -> no gain -> the whole palette is staged
    rng = np.random.default_rng(3)
    m4 = synth.make_mesh(9001, B2, seed=6)
    m4["joints"] = rng.integers(0, B2, size=(9001, 4)).astype(np.uint16)
    c.upload_mesh(m4["pos"], m4["nrm"], m4["joints"], m4["weights"])
    c.set_pose(w3)
    c.deform()
    assert c.get_tuning("effective_subsets") == 0 and c.get_tuning("effective_inst_group") >= 2
    pg, ng = c.read(instance=2)
    pr, nr = oracle.deform(m4["pos"], m4["nrm"], m4["joints"], m4["weights"], w3[2], m3["inv_bind"], threads=4)
    assert_parity(pg, ng, pr, nr, "scattered joints (fallback)")
    c.close()


def test_bone_subset_crowd_behind_the_device_hierarchy_solve(rz, oracle):
    """Device-solved crowds (rz_set_pose_local -> rz_fk_kernel writes the palettes): the skin kernel stages the listed bones'
    finished rows. Bit-identical to the whole-palette form; rz_read_palette still serves what rz_fk_kernel wrote."""
    V, B, I = 16000, 120, 20
    mesh = synth.make_mesh(V, B)
    rng = np.random.default_rng(11)
    q = rng.normal(size=(I, B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    c.set_instances(I)
    c.set_pose_local(q)
    c.deform()
    assert c.get_tuning("effective_subsets") == 1
    sub = [c.read(instance=k) for k in range(I)]
    pal = c.read_palette(I - 1)
    c.set_tuning(inst_subsets=0)
    c.deform()
    for k in range(I):
        pg, ng = c.read(instance=k)
        assert np.array_equal(pg, sub[k][0]) and np.array_equal(ng, sub[k][1])
    assert np.array_equal(pal, c.read_palette(I - 1))
    world = synth.fk_world(mesh["parents"], mesh["bind"], q[I - 1])
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world, mesh["inv_bind"], threads=4)
    assert_parity(sub[I - 1][0], sub[I - 1][1], pr, nr, "device-solved subset crowd")
    c.close()


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_gpu_sits_inside_the_envelope_of_legal_wgsl_evaluations(rz, oracle, pose):
    """vs() (engine.ts:253-272) under every evaluation WGSL allows a driver (FMA contraction, re-associated matrix-vector sums,
    normalize through inverseSqrt — tests/wgsl_latitude.py; their spread around the oracle is bounded in
    tests/test_oracle.py::test_envelope_of_the_evaluations_wgsl_allows): the MI355X kernel's result on the wide sample of the
    real model is within 1e-5 of EACH of them — ten times tighter than the 1e-4 parity bar, and the same order of magnitude
    as the distance between two legal evaluations (3e-7)."""
    import os
    import wgsl_latitude as wl
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_c1_pose0.npz"))
    v = g["wide_vertices"]
    pos, nrm = np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6])
    S = oracle.palette(g["world_" + pose], g["inv_bind"])
    c = rz.DeformContext(0)
    c.upload_mesh_interleaved(v, g["wide_joints"], g["wide_weights"])
    c.upload_skeleton(g["inv_bind"])
    c.set_pose(g["world_" + pose])
    c.deform()
    pg, ng = c.read()
    c.close()
    worst = {}
    for name, kw in wl.MODELS.items():
        p, n = wl.vs(pos, nrm, g["wide_joints"], g["wide_weights"], S, **kw)
        ep, en = wl.distances(pg, ng, p, n)
        assert ep <= 1e-5 and en <= 1e-5, "GPU vs '%s': %.3e / %.3e" % (name, ep, en)
        worst[name] = (ep, en)
    print("GPU distance to each legal evaluation (%s): %s" % (pose, {k: "%.2e / %.2e" % v for k, v in worst.items()}))


@pytest.mark.parametrize("morphs", ["dense", "sparse", "none"])
def test_pose_prefetch_stages_the_next_pose_and_never_a_wrong_one(rz, rzv, oracle, morphs):
    """Zero-copy frames (one character, rz_set_pose: the per-frame writeBuffer of engine.ts:2383-2389): the frame of pose u
    carries a helper workgroup that stages pose u + 1 into device memory when the host has ALREADY written it into its
    pinned slot, so that frame u + 1 need not read over the host link. A hit needs the host to run ahead of the GPU, a miss
    falls back to the pinned slot; either way every frame must hold exactly the bits of its own pose run in isolation.
      * forced hits, by construction: on the tools-only build the frame of pose a is queued BEHIND A GATE (rz_debug_gate: a
        kernel that holds the stream until the host opens it), pose b is uploaded, the gate opens -> the helper of frame a finds
        pose b complete whatever the host's and the GPU's relative speed -> pose_staged == 1 (no race decides the assertion);
      * forced misses: a sync after every frame (the host is never ahead) -> pose_staged == 0;
      * 300 frames in a free-running loop with replays, other pose kinds and consumerless uploads mixed in;
      * pose_prefetch = 0 gives the same bits."""
    V, B = 200000, 120
    mesh = synth.make_mesh(V, B, seed=21)
    rng = np.random.default_rng(22)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    M, deltas = 0, None
    if morphs == "dense":
        M = 24
        deltas, _ = synth.make_morphs_dense(V, M, seed=23)
        c.upload_morphs_dense(deltas)
    elif morphs == "sparse":
        M = 20
        off, vi, d3, _ = synth.make_morphs_sparse(V, M, seed=23)
        c.upload_morphs_sparse(off, vi, d3)
        deltas = synth.sparse_to_dense(V, off, vi, d3)
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    P = 6
    worlds = [synth.make_pose(mesh["parents"], mesh["bind"], B, seed=700 + k) for k in range(P)]
    mws = [(rng.random(M).astype(np.float32) * (rng.random(M) < 0.8)).astype(np.float32) if M else None for _ in range(P)]
    quats = rng.normal(size=(B, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    c.set_tuning(pose_prefetch=0)
    iso = []
    for k in range(P):
        c.set_pose(worlds[k], mws[k]); c.deform(); iso.append(c.read())
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[2], mesh["inv_bind"], deltas, mws[2])
    assert_parity(iso[2][0], iso[2][1], pr, nr, "isolated pose (%s)" % morphs)
    c.set_tuning(pose_prefetch=-1)

    def same(k, what):
        p, n = c.read()
        assert np.array_equal(p, iso[k][0]) and np.array_equal(n, iso[k][1]), "%s: pose %d (%s)" % (what, k, morphs)

    # forced hits, deterministically: the frame of pose a waits behind a gate while the host writes pose b into its slot
    g = rzv.DeformContext(0)
    g.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    g.upload_skeleton(mesh["inv_bind"])
    if morphs == "dense":
        g.upload_morphs_dense(deltas)
    elif morphs == "sparse":
        g.upload_morphs_sparse(off, vi, d3)
    hits = 0
    for a, b in ((0, 1), (3, 4), (5, 2)):
        g.set_pose(worlds[a], mws[a])
        g.deform(); g.sync()
        try:
            g._chk(rzv.lib.rz_debug_gate(g._h, 1))
            g.set_pose(worlds[a], mws[a]); g.deform()   # frame of pose a, held by the gate: its helper will look at the slot pose b lands in
            g.set_pose(worlds[b], mws[b])               # complete before that frame's kernel can start
        finally:
            g._chk(rzv.lib.rz_debug_gate(g._h, 0))
        staged_before = g.get_tuning("pose_staged")     # (synchronises) the helper of frame a has run
        g.deform()
        p, n = g.read()
        assert np.array_equal(p, iso[b][0]) and np.array_equal(n, iso[b][1]), "staged pose %d (%s)" % (b, morphs)
        hits += staged_before
    assert hits == 3, "the helper workgroup must stage the next pose in every gated round (%d / 3)" % hits
    g.close()
    # forced misses: the host is never ahead
    for k in (1, 4, 0):
        c.sync()
        c.set_pose(worlds[k], mws[k])
        assert c.get_tuning("pose_staged") == 0
        c.deform(); c.sync()
        same(k, "pinned-slot pose")
    # free-running loop: whatever mixture of hits and misses the timing produces, the bits are the pose's own
    checks = 0
    for f in range(300):
        k = int(rng.integers(0, P))
        r = rng.random()
        if r < 0.08:
            c.set_pose(worlds[(k + 1) % P], mws[(k + 1) % P])         # an upload no frame consumes
        if r > 0.9:
            c.set_pose_local(quats, mws[k]); c.deform()                # another pose kind in between (its sequence numbers carry the kind: never mistaken for a world pose)
        c.set_pose(worlds[k], mws[k])
        c.deform()
        if rng.random() < 0.3:
            c.deform_n(int(rng.integers(1, 4)))                        # replays of the resident pose
        if f % 7 == 0:
            same(k, "free-running frame %d" % f); checks += 1
    assert checks > 30
    # a new skeleton / morph set invalidates whatever was staged ahead (sequence epochs)
    c.set_pose(worlds[0], mws[0]); c.deform_n(4000); c.set_pose(worlds[0], mws[0]); c.deform(); c.set_pose(worlds[1], mws[1])
    c.upload_morphs_dense(None)
    w2 = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=999)
    c.set_pose(w2)
    assert c.get_tuning("pose_staged") == 0
    c.deform()
    pg, ng = c.read()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], w2, mesh["inv_bind"])
    assert_parity(pg, ng, pr, nr, "after the morph set was dropped")
    c.close()


def test_largest_skeleton_the_lds_palette_holds(rz, oracle):
    """Maximum sizes: 3 242 bones is what rz_upload_skeleton admits (48 B per bone + 8 KB of work area in 160 KB of LDS); a frame
    of such a skeleton must RUN — single character (with and without dense morphs) and as a crowd (bone subsets are what makes a
    crowd of it fit at all) — and one bone more must be refused at upload, not at the first frame."""
    B, V = 3242, 40000
    mesh = synth.make_mesh(V, B, seed=31)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_pose(mesh["world"])
    c.deform()
    pg, ng = c.read()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], threads=4)
    assert_parity(pg, ng, pr, nr, "3242 bones")
    deltas, mw = synth.make_morphs_dense(V, 6, seed=32)
    c.upload_morphs_dense(deltas)
    c.set_pose(mesh["world"], mw)
    c.deform()
    pg, ng = c.read()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw, threads=4)
    assert_parity(pg, ng, pr, nr, "3242 bones + dense morphs")
    c.upload_morphs_dense(None)
    worlds = _poses(mesh, B, 3, seed=33)
    c.set_instances(3)
    c.set_pose(worlds)
    c.deform()
    for k in range(3):
        pg, ng = c.read(instance=k)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[k], mesh["inv_bind"], threads=4)
        assert_parity(pg, ng, pr, nr, "3242-bone crowd, instance %d" % k)
    with pytest.raises(rz.capi.RzError):
        c.upload_skeleton(np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (3243, 1)))
    c.close()


def test_sixteen_million_vertices_index_arithmetic(rz, oracle):
    """Maximum sizes: a mesh beyond 2^24 vertices (ragged count) — byte offsets of the planes, of the morph targets and of the
    packed outputs pass 2^32; checked against the oracle on the head, the tail and a strided sample, with and without dense
    morphs, plus the identity-pose property over the WHOLE mesh (size-independent)."""
    V, B = (1 << 24) + 43, 64
    rng = np.random.default_rng(41)
    small = synth.make_mesh(4096, B, seed=42)
    pos = rng.random((V, 3), dtype=np.float32) * 10 - 5
    nrm = rng.standard_normal((V, 3), dtype=np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    reps = V // 4096 + 1
    joints = np.tile(small["joints"], (reps, 1))[:V]
    weights = np.tile(small["weights"], (reps, 1))[:V]
    c = rz.DeformContext(0)
    c.upload_mesh(pos, nrm, joints, weights)
    c.upload_skeleton(small["inv_bind"])
    pick = np.unique(np.concatenate([np.arange(0, 3000), np.arange(V - 3000, V), np.arange(0, V, 9973)]))

    def check(deltas, mw, what):
        c.set_pose(small["world"], mw)
        c.deform()
        pg, ng = c.read()
        d = None if deltas is None else np.ascontiguousarray(deltas[:, pick])
        pr, nr = oracle.deform(pos[pick], nrm[pick], joints[pick], weights[pick], small["world"], small["inv_bind"], d, mw, threads=8)
        assert_parity(pg[pick], ng[pick], pr, nr, what)
        assert np.isfinite(pg).all() and np.isfinite(ng).all()
        return pg

    check(None, None, "16 M vertices")
    q = np.zeros((B, 4), np.float32)
    q[:, 3] = 1
    c.set_pose(synth.fk_world(small["parents"], small["bind"], q))
    c.deform()
    pg, ng = c.read()
    assert np.abs(pg - pos).max() <= 2e-5 and np.abs(ng - nrm).max() <= 1e-6, "identity pose over the whole 16 M-vertex mesh"
    deltas = (rng.random((2, V, 3), dtype=np.float32) - 0.5) * 0.1
    c.upload_morphs_dense(deltas)
    check(deltas, np.array([0.75, 0.5], np.float32), "16 M vertices + 2 dense morphs")
    c.close()


def test_thousands_of_instances_of_a_tiny_mesh(rz, oracle):
    """Maximum sizes on the instance axis: 4 099 poses (a prime: the last pose group is partial) of a 257-vertex mesh — more pose
    groups than workgroup slots, runs shorter than a workgroup; both crowd forms, oracle parity on a sample, bit-identity between
    the forms on every instance."""
    V, B, I = 257, 10, 4099
    mesh = synth.make_mesh(V, B, seed=51)
    rng = np.random.default_rng(52)
    base = _poses(mesh, B, 16, seed=53)
    worlds = base[rng.integers(0, 16, size=I)]
    c = _crowd(rz, mesh, worlds)
    c.deform()
    first = c.get_tuning("effective_subsets")
    a = [c.read(instance=k) for k in range(0, I, 97)] + [c.read(instance=I - 1)]
    c.set_tuning(inst_subsets=0 if first else 1)
    c.deform()
    b = [c.read(instance=k) for k in range(0, I, 97)] + [c.read(instance=I - 1)]
    for (pa, na), (pb, nb) in zip(a, b):
        assert np.array_equal(pa, pb) and np.array_equal(na, nb)
    for k in (0, 1234, I - 1):
        pg, ng = c.read(instance=k)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[k], mesh["inv_bind"])
        assert_parity(pg, ng, pr, nr, "instance %d of %d" % (k, I))
    c.close()
