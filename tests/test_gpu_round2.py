"""GPU tests added in round 2: BASELINE config 4 at FULL size, real model vertices under a reference-produced pose (pinned
to reference execution), the physics hand-off for device-solved poses, and regression tests for the round-1 review
(graph replay keys, instance-count changes, duplicate VMD keys, ablation keys absent from the product)."""
import os

import numpy as np
import pytest

from helpers import assert_hull, assert_parity, bone_morph_reference, fk_reference, sample_reference
from reze_engine_amd import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_c1_pose0.npz")


def _identity_world(mesh, B):
    q = np.zeros((B, 4), dtype=np.float32)
    q[:, 3] = 1
    return synth.fk_world(mesh["parents"], mesh["bind"], q)


def test_c4_full_size_256x30k_200b(rz, oracle):
    """BASELINE config 4 at full size: 256 instances x 30 000 verts / 200 bones, per-instance palette in LDS
    (rz_skin_instances_kernel), with the built-in plan AND the plan rz_autotune picks. vs() restatement engine.ts:253-272:
      * oracle parity on instances 0, 127, 255 and one random instance;
      * instance k is BIT-IDENTICAL to the same pose run alone as a single-instance frame (one-launch and prep-kernel
        forms): the crowd kernel evaluates the same FMA chains in the same order as rz_deform_kernel;
      * every instance under the identity pose is the rest mesh."""
    V, B, I = 30000, 200, 256
    mesh = synth.make_mesh(V, B)
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    rnd = int(np.random.default_rng(2).integers(1, I - 1))
    picks = sorted({0, 127, 255, rnd})
    singles = {}
    for k in picks:                                        # the same poses as single-instance frames, both forms
        for fast in (1, 0):
            c.set_instances(1)
            c.set_tuning(fast=fast)
            c.set_pose(worlds[k])
            c.deform()
            singles[(k, fast)] = c.read()
    assert np.array_equal(singles[(0, 1)][0], singles[(0, 0)][0])
    c.set_tuning(fast=-1)
    c.set_instances(I)
    c.set_pose(worlds)
    plans = []
    for tuned in (False, True):
        if tuned:
            c.autotune(20)
        c.deform()
        assert c.get_tuning("effective_inst_group") >= 2, "C4 must run the instanced (palette-group-in-LDS) kernel"
        plans.append((c.get_tuning("effective_inst_group"), c.get_tuning("effective_grid")))
        for k in picks:
            pg, ng = c.read(instance=k)
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[k], mesh["inv_bind"], threads=4)
            assert_parity(pg, ng, pr, nr, "C4 full size instance %d (autotuned=%s)" % (k, tuned))
            for fast in (1, 0):
                ps, ns = singles[(k, fast)]
                assert np.array_equal(pg, ps) and np.array_equal(ng, ns), "instance %d differs from its single-instance frame (fast=%d)" % (k, fast)
        # palettes of the crowd are observable too (engine.ts:926-928)
        S = oracle.palette(worlds[rnd], mesh["inv_bind"]).reshape(-1, 4, 4)
        np.testing.assert_allclose(c.read_palette(rnd), np.transpose(S, (0, 2, 1))[:, :3, :].reshape(-1, 12), rtol=1e-6, atol=1e-6)
    # which XCD runs which workgroup (inst_order) is a pure scheduling choice: the other order gives the same bits
    assert c.get_tuning("inst_order") == 1
    ref = {k: c.read(instance=k) for k in picks}
    pal_ref = c.read_palette(rnd)
    c.set_tuning(inst_order=0)
    c.deform()
    for k in picks:
        pg, ng = c.read(instance=k)
        assert np.array_equal(pg, ref[k][0]) and np.array_equal(ng, ref[k][1]), "instance %d: inst_order 0 vs 1" % k
    assert np.array_equal(pal_ref, c.read_palette(rnd))
    c.set_tuning(inst_order=1)
    # round 3: by default a workgroup stages only the bones its vertex run names (bone-subset form); staging the whole
    # palette instead is the same arithmetic on the same rows, so the bits (and the observable palettes) are identical
    assert c.get_tuning("effective_subsets") == 1 and 0 < c.get_tuning("effective_subset_bones") < B
    c.set_tuning(inst_subsets=0)
    c.deform()
    assert c.get_tuning("effective_subsets") == 0
    for k in picks:
        pg, ng = c.read(instance=k)
        assert np.array_equal(pg, ref[k][0]) and np.array_equal(ng, ref[k][1]), "instance %d: whole palette vs bone subsets" % k
    assert np.array_equal(pal_ref, c.read_palette(rnd))
    c.set_tuning(inst_subsets=-1)
    # identity pose in EVERY instance == rest mesh, all 256 read back
    ident = np.tile(_identity_world(mesh, B)[None], (I, 1, 1))
    c.set_pose(ident)
    c.deform()
    for k in range(I):
        pg, ng = c.read(instance=k)
        assert np.abs(pg - mesh["pos"]).max() <= 2e-5 and np.abs(ng - mesh["nrm"]).max() <= 1e-6, "identity pose, instance %d" % k
    c.close()


@pytest.mark.parametrize("pose", ["pose0", "tween150"])
def test_real_model_vertices_under_the_reference_pose(rz, oracle, pose):
    """Real vertices of the demo model (256-vertex slices of its vertex / joints / weights buffers as the reference's
    loader produced them) deformed on the GPU under world matrices the reference's own Model.evaluatePose produced.
    Checked against (a) the oracle and (b) DIRECTLY against reference execution: the palette the reference's Mat4.multiply
    computed (math.ts:303-320) and the slice skinned with the reference's Mat4 / Vec3 primitives composed as vs()
    (engine.ts:255-272) — tests/golden/ref_c1_pose0.npz, tools/ref_erased_run.py. Tolerance 1e-4 (north_star)."""
    g = np.load(GOLD)
    v = g["slice_vertices"]
    pos, nrm = np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6])
    c = rz.DeformContext(0)
    for upload in ("soa", "interleaved"):
        if upload == "soa":
            c.upload_mesh(pos, nrm, g["slice_joints"], g["slice_weights"])
        else:
            c.upload_mesh_interleaved(v, g["slice_joints"], g["slice_weights"])      # the reference's own 8-float layout
        c.upload_skeleton(g["inv_bind"])
        for fast in (1, 0):
            c.set_tuning(fast=fast)
            c.set_pose(g["world_" + pose])
            c.deform()
            pg, ng = c.read()
            pr, nr = oracle.deform(pos, nrm, g["slice_joints"], g["slice_weights"], g["world_" + pose], g["inv_bind"])
            assert_parity(pg, ng, pr, nr, "real slice vs oracle (%s, fast=%d)" % (upload, fast))
            ref = g["skinned_" + pose]
            assert_parity(pg, ng, ref[:, :3], ref[:, 3:], "real slice vs REFERENCE EXECUTION (%s, fast=%d)" % (upload, fast))
            pal = c.read_palette()                          # rows 0..2, row-major 3x4
            ref_pal = np.transpose(g["palette_" + pose].reshape(-1, 4, 4), (0, 2, 1))[:, :3, :].reshape(-1, 12)
            scale = np.maximum(1.0, np.abs(ref_pal).max(axis=1, keepdims=True))
            assert (np.abs(pal - ref_pal) <= 1e-5 * scale).all(), "palette vs the reference's Mat4.multiply: %g" % np.abs(pal - ref_pal).max()
    c.close()


def test_override_world_is_the_physics_hand_off(rz, oracle):
    """rz_override_world mirrors engine.ts:2379-2381 for device-solved poses: the supplied world matrices replace the
    solved ones of the listed bones after the hierarchy solve; children keep the matrices solved from the un-overridden
    parent. Equals the host path fed with the same in-place edit; last entry wins; n = 0 clears; per-instance entries."""
    V, B, I = 5000, 60, 3
    mesh = synth.make_mesh(V, B, seed=91)
    rng = np.random.default_rng(92)
    q = rng.normal(size=(I, B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(I)
    with pytest.raises(Exception):
        c.override_world([1], np.eye(4, dtype=np.float32).reshape(1, 16))           # no topology yet
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    c.set_pose_local(q)
    c.deform()
    solved = np.stack([c.read_world(i) for i in range(I)])
    bones = np.array([5, 17, 17, 40, 3], dtype=np.uint32)
    insts = np.array([0, 0, 0, 2, 1], dtype=np.uint32)
    mats = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=300 + k)[int(b)] for k, b in enumerate(bones)])
    c.override_world(bones, mats, insts)
    for _ in range(2):                                      # persists across frames until replaced
        c.deform()
    want = solved.copy()
    for k in range(len(bones)):                             # in order: the later entry for (0, 17) wins
        want[insts[k], bones[k]] = mats[k]
    for i in range(I):
        got = c.read_world(i)
        assert np.array_equal(got, want[i]), "world matrices of instance %d" % i
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], want[i], mesh["inv_bind"])
        pg, ng = c.read(instance=i)
        assert_parity(pg, ng, pr, nr, "override, instance %d" % i)
    # the host path given the same in-place edit produces the same frame bit for bit
    dev = [c.read(instance=i) for i in range(I)]
    c.set_pose(want)
    c.deform()
    for i in range(I):
        ph, nh = c.read(instance=i)
        assert np.array_equal(ph, dev[i][0]) and np.array_equal(nh, dev[i][1])
    # invalid entries are rejected, the context keeps working
    with pytest.raises(Exception):
        c.override_world([B], mats[:1], [0])
    with pytest.raises(Exception):
        c.override_world([1], mats[:1], [I])
    bad = mats[:1].copy(); bad[0, 3] = np.nan
    with pytest.raises(Exception):
        c.override_world([1], bad, [0])
    c.set_pose_local(q)
    c.override_world([], None)                              # clear
    c.deform()
    for i in range(I):
        assert np.array_equal(c.read_world(i), solved[i])
    # an instance-count change drops the overrides (they name members of the old crowd)
    c.override_world(bones, mats, insts)
    c.set_instances(1)
    c.set_pose_local(q[0])
    c.deform()
    assert np.array_equal(c.read_world(0), solved[0])
    c.close()


def test_graph_replay_keys_cover_slot_parity_and_reallocated_buffers(rz, oracle):
    """Round-1 review: (1) aabb on + graph on + deform_n(32) twice — the second call replays two graphs with no plain frame
    behind them, the bounding box read back must be the last frame's; (2) an animation (and a topology) re-uploaded
    between two replays frees and re-allocates buffers the captured FK launches point at: the graph must be rebuilt."""
    V, B = 7000, 40
    mesh = synth.make_mesh(V, B, seed=31)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.enable_aabb(True)
    c.set_tuning(graph=1)
    c.set_pose(mesh["world"])
    for n in (32, 32, 33, 32, 47, 32):
        c.deform_n(n)
        pg, _ = c.read()
        box = c.read_aabb()
        assert np.isfinite(box).all(), "re-armed (empty) slot read back after deform_n(%d)" % n
        assert np.array_equal(box[:3], pg.min(axis=0)) and np.array_equal(box[3:], pg.max(axis=0))
    c.deform()                                               # and a plain frame after a replay accumulates into a clean slot
    box = c.read_aabb()
    assert np.array_equal(box[:3], pg.min(axis=0)) and np.array_equal(box[3:], pg.max(axis=0))
    c.enable_aabb(False)
    # (2) sampled poses: replay, re-upload a DIFFERENT motion (buffers freed + re-allocated), replay again
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    rng = np.random.default_rng(8)
    nk = 5

    def motion(seed):
        r = np.random.default_rng(seed)
        kq = r.normal(size=(B, nk, 4)).astype(np.float32)
        kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        return dict(track_bone=np.arange(B), key_off=np.arange(B + 1) * nk, key_frame=np.tile(np.arange(nk) * 10.0, B),
                    key_rot=kq, key_pos=(r.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.3, key_interp=None)
    outs = []
    for seed in (100, 101, 100):
        a = motion(seed)
        # churn the allocator so the re-uploaded tracks do not land on the old addresses by luck
        junk = [rz.DeformContext(0) for _ in range(2)]
        c.upload_animation(a["track_bone"], a["key_off"], a["key_frame"], a["key_rot"], a["key_pos"])
        for j in junk:
            j.close()
        c.set_pose_sampled(np.array([17.25], np.float32))
        c.deform_n(40)
        q, t, _ = sample_reference(a, 17.25, B, 0)
        ref = fk_reference(mesh["parents"], mesh["bind"], q, t)
        got = c.read_world(0)
        assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), "motion %d under graph replay" % seed
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], got, mesh["inv_bind"])
        pg, ng = c.read()
        assert_parity(pg, ng, pr, nr, "sampled pose under graph replay (motion %d)" % seed)
        outs.append(pg)
    assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])           # frees the fk_* arrays a captured graph would name
    c.set_pose_local(np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)))
    c.deform_n(40)
    np.testing.assert_allclose(c.read()[0], mesh["pos"], rtol=1e-6, atol=2e-5)
    c.close()


def test_crowd_back_to_one_instance_keeps_dense_morph_weights(rz, oracle):
    """Round-1 review: set_instances(I > 1) -> set_pose -> set_instances(1) -> deform without a new pose. The host-side
    active-morph list is only maintained for one instance, so the frame must compact instance 0's weights on the device."""
    V, B, M = 4000, 30, 7
    mesh = synth.make_mesh(V, B, seed=51)
    deltas, mw = synth.make_morphs_dense(V, M, seed=52)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.upload_morphs_dense(deltas)
    c.set_instances(3)
    worlds = np.stack([mesh["world"]] * 3)
    mws = np.stack([mw, mw * 0.5, mw * 0.0]).astype(np.float32)
    c.set_pose(worlds, mws)
    c.deform()
    c.set_instances(1)
    c.deform()
    pg, ng = c.read()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    assert_parity(pg, ng, pr, nr, "instance 0 after the crowd shrank to one")
    p0, _ = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
    assert np.abs(pr - p0).max() > 1e-3                      # the morphs matter in this frame
    c.close()


def test_device_sampler_accepts_duplicate_key_frames(rz):
    """Round-1 review: real VMD files carry duplicate-frame keys; host/vmd-sampler.js keeps them (stable sort) and its span
    search never divides by a zero span. The device sampler takes the same tracks and lands on the same keys."""
    B = 12
    mesh = synth.make_mesh(500, B, seed=61)
    rng = np.random.default_rng(62)
    frames_per = np.array([0, 10, 10, 20, 20, 20, 35], dtype=np.float32)      # duplicates inside, not at the ends only
    nk = len(frames_per)
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    a = dict(track_bone=np.arange(B), key_off=np.arange(B + 1) * nk, key_frame=np.tile(frames_per, B), key_rot=kq,
             key_pos=(rng.random((B, nk, 3), dtype=np.float32) - 0.5), key_interp=None)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    c.upload_animation(a["track_bone"], a["key_off"], a["key_frame"], a["key_rot"], a["key_pos"])
    for f in (0.0, 5.0, 10.0, 12.5, 20.0, 27.0, 35.0, 50.0):
        c.set_pose_sampled(np.array([f], np.float32))
        c.deform()
        q, t, _ = sample_reference(a, f, B, 0)
        ref = fk_reference(mesh["parents"], mesh["bind"], q, t)
        got = c.read_world(0)
        assert np.isfinite(got).all() and np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), "frame %g" % f
    bad = a["key_frame"].copy(); bad[3] = 5.0                  # descending is still an error
    with pytest.raises(Exception):
        c.upload_animation(a["track_bone"], a["key_off"], bad, a["key_rot"], a["key_pos"])
    c.close()


def test_ablation_key_is_not_part_of_the_product(rz):
    """The ablation switches ("dbg": kernels skip work, output is garbage) exist only in the tools-only build. The shipped
    library rejects the key and keeps producing the deformed mesh."""
    mesh = synth.make_mesh(2000, 16, seed=5)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_pose(mesh["world"])
    c.deform()
    before = c.read()
    for v in (1, 2, 3, 4, 5):
        with pytest.raises(rz.capi.RzError):
            c.set_tuning(dbg=v)
    c.deform()
    after = c.read()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    c.close()


@pytest.mark.parametrize("morphs", ["none", "dense", "sparse", "dense>128"])
def test_zero_copy_pose_ring_never_serves_a_stale_or_torn_pose(rz, oracle, morphs):
    """One character: rz_set_pose* enqueues NO copy — the pose sits in a slot of a pinned, device-mapped ring, the first
    frame's kernels read it from there (workgroup 0 of the one-launch kernel leaves it in device memory for replays), a slot
    is reused eight uploads later and only one event per four uploads guards the reuse. Hammer it: 600 frames cycling through
    five poses of three kinds (world matrices, local rotations, sampled), replays in between, uploads that no frame consumes,
    plans that need a resident pose (prep kernel) mixed in — every checked frame bit-identical to that pose in isolation."""
    V, B = 12000, 150
    mesh = synth.make_mesh(V, B, seed=15)
    rng = np.random.default_rng(16)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    M = {"none": 0, "dense": 12, "sparse": 20, "dense>128": 140}[morphs]
    deltas = None
    if morphs.startswith("dense"):
        deltas, _ = synth.make_morphs_dense(V, M, seed=17)
        c.upload_morphs_dense(deltas)
    elif morphs == "sparse":
        off, vi, d3, _ = synth.make_morphs_sparse(V, M, seed=17)
        c.upload_morphs_sparse(off, vi, d3)
        deltas = synth.sparse_to_dense(V, off, vi, d3)
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    nk = 4
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    extra = {}
    if M:
        extra = dict(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 30.0], np.float32), M), mkey_weight=rng.random(2 * M).astype(np.float32),
                     feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
    c.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.3, **extra)
    P = 5
    worlds = [synth.make_pose(mesh["parents"], mesh["bind"], B, seed=500 + k) for k in range(P)]
    quats = rng.normal(size=(P, B, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=2, keepdims=True)
    mws = [(rng.random(M).astype(np.float32) * (rng.random(M) < 0.7)).astype(np.float32) if M else None for _ in range(P)]
    frames = [np.array([rng.random() * 30], np.float32) for _ in range(P)]

    def put(kind, k):
        if kind == "w":
            c.set_pose(worlds[k], mws[k])
        elif kind == "l":
            c.set_pose_local(quats[k], mws[k])
        else:
            c.set_pose_sampled(frames[k])
    iso = {}
    for kind in "wls":
        for k in range(P):
            put(kind, k); c.deform(); iso[(kind, k)] = c.read()
    assert c.get_tuning("zero_copy") == -1
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[3], mesh["inv_bind"], deltas, mws[3])
    assert_parity(iso[("w", 3)][0], iso[("w", 3)][1], pr, nr, "isolated zero-copy world pose (%s)" % morphs)
    # ... and the same frames when every pose is copied to the device instead: identical bits
    c.set_tuning(zero_copy=0)
    for kind, k in (("w", 1), ("l", 2), ("s", 4)):
        put(kind, k); c.deform(); p2, n2 = c.read()
        assert np.array_equal(p2, iso[(kind, k)][0]) and np.array_equal(n2, iso[(kind, k)][1]), "zero_copy=0 differs (%s%d)" % (kind, k)
    c.set_tuning(zero_copy=-1)
    checks = 0
    for f in range(600):
        kind = "wls"[int(rng.integers(0, 3))]
        k = int(rng.integers(0, P))
        r = rng.random()
        if r < 0.15:                                       # an upload that is overwritten before any frame consumes it
            put("wl"[int(rng.integers(0, 2))], (k + 1) % P)
        if r > 0.9:                                        # a plan that wants the pose resident (prep kernel) for this frame only
            c.set_tuning(fast=0)
        put(kind, k)
        if kind == "w" and f % 50 == 0:                    # readable before any frame has consumed it
            assert np.array_equal(c.read_world(0), worlds[k])
        c.deform()
        if r > 0.9:
            c.set_tuning(fast=-1)
        if 0.4 < r < 0.5:
            c.deform_n(int(rng.integers(1, 5)))            # replays of the resident pose
        if f % 5 == 0 or f > 590:
            got = c.read()
            assert np.array_equal(got[0], iso[(kind, k)][0]) and np.array_equal(got[1], iso[(kind, k)][1]), "frame %d (%s%d, %s)" % (f, kind, k, morphs)
            checks += 1
    assert checks > 100
    c.close()


def test_growing_the_crowd_needs_a_new_pose(rz, oracle):
    """A single-character pose may still sit in its pinned slot (which holds exactly one instance): a crowd larger than
    the one the resident pose was set for must be given a pose before it can be deformed; shrinking keeps the pose."""
    V, B = 3000, 20
    mesh = synth.make_mesh(V, B, seed=7)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(4)                      # allocate for four up front
    c.set_instances(1)
    c.set_pose(mesh["world"])
    c.set_instances(4)                      # no frame has consumed the pose yet, and it is one instance wide
    with pytest.raises(rz.capi.RzError):
        c.deform()
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=70 + i) for i in range(4)])
    c.set_pose(worlds)
    c.deform()
    c.set_instances(2)                      # shrinking keeps instances 0..1 of the resident pose
    c.deform()
    for i in range(2):
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"])
        pg, ng = c.read(instance=i)
        assert_parity(pg, ng, pr, nr, "after shrinking the crowd, instance %d" % i)
    c.close()


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_wide_sample_of_the_real_model_against_reference_execution(rz, oracle, pose):
    """1 031 real vertices (every 28th of the demo model: all body parts, 166 bones, the model's own 40 / 53 / 7 % influence mix)
    under three reference-produced poses, deformed on the GPU and compared DIRECTLY with vs() as the reference run evaluated
    it with math.ts primitives (tests/golden, tools/ref_erased_run.py), then replicated as a small crowd: the instanced kernel
    on real skinning data (its wave-uniform influence skipping takes all three paths here) must give the same bits."""
    g = np.load(GOLD)
    v = g["wide_vertices"]
    pos, nrm = np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6])
    ref = g["skinnedwide_" + pose].astype(np.float64)
    c = rz.DeformContext(0)
    c.upload_mesh_interleaved(v, g["wide_joints"], g["wide_weights"])
    c.upload_skeleton(g["inv_bind"])
    c.set_pose(g["world_" + pose])
    c.deform()
    pg, ng = c.read()
    assert_parity(pg, ng, ref[:, :3], ref[:, 3:], "wide real sample vs REFERENCE EXECUTION (%s)" % pose)
    # ... and with the reference's own vs() / skin-matrix shader TEXT, interpreted (tests/golden/ref_wgsl.npz, tools/ref_wgsl_run.py)
    wg = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_wgsl.npz"))
    assert_parity(pg, ng, wg["wide_" + pose][:, :3], wg["wide_" + pose][:, 3:], "wide real sample vs the reference's WGSL text (%s)" % pose)
    ref_pal = np.transpose(wg["palette_" + pose].reshape(-1, 4, 4), (0, 2, 1))[:, :3, :].reshape(-1, 12)
    assert (np.abs(c.read_palette() - ref_pal) <= 1e-5 * np.maximum(1.0, np.abs(ref_pal).max(axis=1, keepdims=True))).all()
    pr, nr = oracle.deform(pos, nrm, g["wide_joints"], g["wide_weights"], g["world_" + pose], g["inv_bind"])
    assert_parity(pg, ng, pr, nr, "wide real sample vs oracle (%s)" % pose)
    # the fused outline hull against the outline pass's vs() text, interpreted (engine.ts:431-463)
    from helpers import assert_hull
    c.upload_edge_scale(wg["hull_edge"])
    c.deform()
    assert_hull(c.read_hull(), wg["hull_wide_" + pose], "hull vs the reference's outline shader text (%s)" % pose)
    c.upload_edge_scale(None)
    # the same vertices sorted by influence count (real models cluster them by mesh part), as a crowd of 5 poses
    order = np.argsort((g["wide_weights"] > 0).sum(axis=1), kind="stable")
    c.upload_mesh_interleaved(v[order], g["wide_joints"][order], g["wide_weights"][order])
    c.upload_skeleton(g["inv_bind"])
    c.set_pose(g["world_" + pose]); c.deform()
    single = c.read()
    assert np.array_equal(single[0], pg[order]) and np.array_equal(single[1], ng[order])
    c.set_instances(5)
    worlds = np.stack([g["world_pose0"], g["world_" + pose], g["world_tween150"], g["world_" + pose], g["world_tween500"]])
    for fast in (-1, 0):
        c.set_tuning(fast=fast)
        c.set_pose(worlds); c.deform()
        assert c.get_tuning("effective_inst_group") >= 2
        for k in (1, 3):
            pk, nk = c.read(instance=k)
            assert np.array_equal(pk, single[0]) and np.array_equal(nk, single[1]), "crowd instance %d (fast=%d) vs the pose alone" % (k, fast)
    c.close()


@pytest.mark.parametrize("morphs", ["none", "dense", "sparse"])
def test_fused_hierarchy_solve_is_the_same_frame(rz, oracle, morphs):
    """One launch per device-animated frame: with a single character whose pose is sampled on the device (default), or given as
    local rotations ("fuse_fk" = 1), every workgroup of the deform kernel solves the bone hierarchy itself (sampling, append
    rotate / move, overrides, level loop, palette) and compacts the morph weights — no rz_fk_kernel, no rz_prep_kernel. The
    frame must have the same bits as the three-kernel frame (fuse_fk = 0), the world matrices must match the float64
    restatement, the mesh the oracle. 300 bones (more than one bone per thread), append bones, translations, overrides."""
    V, B = 9000, 300
    mesh = synth.make_mesh(V, B, seed=33)
    rng = np.random.default_rng(34)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    M, deltas = 0, None
    if morphs == "dense":
        M = 150                                               # more than the 128 the kernel-argument list holds
        deltas, _ = synth.make_morphs_dense(V, M, seed=35)
        c.upload_morphs_dense(deltas)
    elif morphs == "sparse":
        M = 25
        off, vi, d3, _ = synth.make_morphs_sparse(V, M, seed=35)
        c.upload_morphs_sparse(off, vi, d3)
        deltas = synth.sparse_to_dense(V, off, vi, d3)
    ap = np.full(B, -1, dtype=np.int32)
    ratio = np.ones(B, dtype=np.float32)
    move = np.zeros(B, dtype=np.uint8)
    for k, b in enumerate(rng.choice(B, size=20, replace=False)):
        ap[b] = int(rng.integers(0, B)); ratio[b] = [0.5, -0.75, 1.5, -2.0, 1.0][k % 5]; move[b] = k % 2
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"], ap, ratio, move)
    nk = 5
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    anim = dict(track_bone=np.arange(B), key_off=np.arange(B + 1) * nk, key_frame=np.tile(np.arange(nk) * 8.0, B), key_rot=kq,
                key_pos=(rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.4, key_interp=rng.integers(0, 128, size=(B * nk, 16)).astype(np.uint8))
    if M:
        anim.update(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 32.0], np.float32), M),
                    mkey_weight=(rng.random(2 * M) * (rng.random(2 * M) < 0.8)).astype(np.float32), feed_off=np.arange(M + 1), feed_track=np.arange(M),
                    feed_ratio=np.ones(M, np.float32))
    c.upload_animation(anim["track_bone"], anim["key_off"], anim["key_frame"], anim["key_rot"], anim["key_pos"], anim["key_interp"],
                       anim.get("mkey_off"), anim.get("mkey_frame"), anim.get("mkey_weight"), anim.get("feed_off"), anim.get("feed_track"), anim.get("feed_ratio"))
    q = rng.normal(size=(B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = (rng.random((B, 3), dtype=np.float32) - 0.5)
    mw = (rng.random(M) * (rng.random(M) < 0.7)).astype(np.float32) if M else None
    ovr_b = np.array([7, 123], dtype=np.uint32)
    ovr_m = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=900 + k)[int(b)] for k, b in enumerate(ovr_b)])

    def frame(kind, fuse, overrides):
        c.set_tuning(fuse_fk=fuse)
        c.override_world(ovr_b if overrides else [], ovr_m if overrides else None)
        if kind == "sampled":
            c.set_pose_sampled(np.array([13.37], np.float32))
        else:
            c.set_pose_local(q, mw, t)
        want = 1 if (fuse == 1 or (fuse == -1 and (kind == "sampled" or morphs != "dense"))) else 0
        assert c.get_tuning("effective_fuse_fk") == want
        c.deform()
        if want:
            assert c.time_frames(3)["prep_kernel_ms"] < 1e-4          # no front kernel at all
        out = (c.read(), c.read_world(0), c.read_palette(0))
        c.deform_n(3)                                                 # replays of the fused frame
        again = c.read()
        assert np.array_equal(out[0][0], again[0]) and np.array_equal(out[0][1], again[1])
        return out
    c.set_tuning(fuse_fk=-1)                                           # automatic: sampled always, local unless dense morphs stream
    c.set_pose_local(q, mw, t)
    assert c.get_tuning("effective_fuse_fk") == (0 if morphs == "dense" else 1)
    for kind in ("sampled", "local"):
        for overrides in (False, True):
            ref3 = frame(kind, 0, overrides)                          # rz_fk_kernel (+ rz_prep_kernel) + deform kernel
            fused = frame(kind, 1 if kind == "local" else -1, overrides)
            assert np.array_equal(ref3[1], fused[1]), "world matrices (%s, overrides=%s)" % (kind, overrides)
            assert np.array_equal(ref3[2], fused[2]), "palette (%s, overrides=%s)" % (kind, overrides)
            assert np.array_equal(ref3[0][0], fused[0][0]) and np.array_equal(ref3[0][1], fused[0][1]), "mesh (%s, overrides=%s)" % (kind, overrides)
            if kind == "sampled":
                qs, ts, ws = sample_reference(anim, 13.37, B, M)
            else:
                qs, ts, ws = q, t, (np.zeros(0) if mw is None else mw)
            world = fk_reference(mesh["parents"], mesh["bind"], qs, ts, ap, ratio, move)
            if overrides:
                world[ovr_b] = ovr_m
            assert np.abs(fused[1] - world).max() <= 1e-4 * max(1.0, np.abs(world).max())
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], fused[1], mesh["inv_bind"], deltas, np.asarray(ws, np.float32) if M else None)
            assert_parity(fused[0][0], fused[0][1], pr, nr, "fused frame (%s, %s, overrides=%s)" % (kind, morphs, overrides))
    c.close()


@pytest.mark.parametrize("morphs", ["bone-only", "dense", "sparse"])
def test_bone_morphs_fold_into_device_solved_poses(rz, oracle, morphs):
    """PMX bone morphs (type 2; SURVEY §8f rank 3 — no reference counterpart, pmx-loader.ts:489-497 only skips them): entry
    (morph, bone, t, q) adds w * t to the bone's local translation and right-multiplies its rotation by slerp(I, q, w), w = the
    pose's weight of that morph, entries of a bone in ascending morph order, before append rotation and the hierarchy solve.
    Device-solved poses only: local rotations (three-kernel and fused frame), sampled poses (weights from morph tracks, sampled
    in the same kernel), one character and a crowd. World matrices vs the float64 restatement, mesh vs the oracle."""
    V, B = 6000, 120
    mesh = synth.make_mesh(V, B, seed=71)
    rng = np.random.default_rng(72)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    deltas = None
    if morphs == "dense":
        M = 20
        deltas, _ = synth.make_morphs_dense(V, M, seed=73)
        deltas[M - 6:] = 0                                    # the last six morphs are bone morphs: no vertex deltas
        c.upload_morphs_dense(deltas)
    elif morphs == "sparse":
        M = 20
        off, vi, d3, _ = synth.make_morphs_sparse(V, M, seed=73)
        c.upload_morphs_sparse(off, vi, d3)
        deltas = synth.sparse_to_dense(V, off, vi, d3)
    else:                                                     # a model whose only morphs are bone morphs (the reference's 武器.pmx)
        M = 6
        c.upload_morphs_sparse(np.zeros(M + 1, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 3), np.float32))
        deltas = np.zeros((M, V, 3), np.float32)
    ap = np.full(B, -1, dtype=np.int32)
    ratio = np.ones(B, dtype=np.float32)
    move = np.zeros(B, dtype=np.uint8)
    for k, b in enumerate(rng.choice(B, size=12, replace=False)):
        ap[b] = int(rng.integers(0, B)); ratio[b] = [0.5, -0.75, 1.5, 1.0][k % 4]; move[b] = k % 2
    with pytest.raises(rz.RzError):                           # needs the topology (device-solved poses)
        c.upload_bone_morphs([0], [0], np.zeros(3), [0, 0, 0, 1])
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"], ap, ratio, move)
    # 16 entries over the six bone morphs; some bones are moved by several morphs, two entries share (morph, bone);
    # the append parents are among the morphed bones so their children must follow the morphed rotation
    n = 16
    bm_m = np.sort(rng.integers(M - 6, M, size=n)).astype(np.uint32)
    targets = np.concatenate([ap[ap >= 0][:4], rng.integers(0, B, size=6)])
    bm_b = targets[rng.integers(0, len(targets), size=n)].astype(np.uint32)
    bm_b[1], bm_m[1] = bm_b[0], bm_m[0]
    bm_t = ((rng.random((n, 3)) - 0.5) * 0.8).astype(np.float32)
    bm_q = rng.normal(size=(n, 4)).astype(np.float32)
    bm_q /= np.linalg.norm(bm_q, axis=1, keepdims=True)
    bm_q[2] = [0.0, 0.0, 0.01, 1.0]; bm_q[2] /= np.linalg.norm(bm_q[2])       # the near-identity (lerp) branch of slerp
    perm = rng.permutation(n)                                  # the ABI takes entries in any order and folds equal (bone, morph) in the order given
    bm_m, bm_b, bm_t, bm_q = bm_m[perm], bm_b[perm], bm_t[perm], bm_q[perm]
    for bad in (dict(morph=[M]), dict(bone=[B]), dict(t=[np.nan, 0, 0]), dict(q=[0, np.inf, 0, 1])):
        with pytest.raises(rz.RzError):
            c.upload_bone_morphs(bad.get("morph", [0]), bad.get("bone", [0]), bad.get("t", [0, 0, 0]), bad.get("q", [0, 0, 0, 1]))
    order = np.argsort(bm_m, kind="stable")                    # ascending morph; ties in given order — what the restatement folds
    c.upload_bone_morphs(bm_m, bm_b, bm_t, bm_q)
    nk = 4
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    anim = dict(track_bone=np.arange(B), key_off=np.arange(B + 1) * nk, key_frame=np.tile(np.arange(nk) * 10.0, B), key_rot=kq,
                key_pos=(rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.4, key_interp=rng.integers(0, 128, size=(B * nk, 16)).astype(np.uint8))
    # morph tracks: one per morph + one "group" track feeding the last bone morph at ratio 0.5 (group morphs feed bone morphs too)
    mkw = (rng.random(2 * (M + 1)) * (rng.random(2 * (M + 1)) < 0.85)).astype(np.float32)
    feed_off = np.concatenate([np.arange(M), [M + 1]]).astype(np.uint32)
    anim.update(mkey_off=np.arange(M + 2) * 2, mkey_frame=np.tile(np.array([0.0, 30.0], np.float32), M + 1), mkey_weight=mkw,
                feed_off=feed_off, feed_track=np.arange(M + 1), feed_ratio=np.concatenate([np.ones(M), [0.5]]).astype(np.float32))
    c.upload_animation(anim["track_bone"], anim["key_off"], anim["key_frame"], anim["key_rot"], anim["key_pos"], anim["key_interp"],
                       anim["mkey_off"], anim["mkey_frame"], anim["mkey_weight"], anim["feed_off"], anim["feed_track"], anim["feed_ratio"])

    def expect(q, t, w):
        q2, t2 = bone_morph_reference(q, t, bm_m[order], bm_b[order], bm_t[order], bm_q[order], w)
        return fk_reference(mesh["parents"], mesh["bind"], q2, t2, ap, ratio, move)

    def check(got_world, got_mesh, q, t, w, what):
        world = expect(q, t, w)
        assert np.abs(got_world - world).max() <= 1e-4 * max(1.0, np.abs(world).max()), what
        plain = fk_reference(mesh["parents"], mesh["bind"], q, t, ap, ratio, move)
        assert np.abs(world - plain).max() > 1e-2, "the morphs of this case must move something"
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], got_world, mesh["inv_bind"], deltas, np.asarray(w, np.float32))
        assert_parity(got_mesh[0], got_mesh[1], pr, nr, what)

    q = rng.normal(size=(B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = (rng.random((B, 3), dtype=np.float32) - 0.5)
    mw = rng.random(M).astype(np.float32)
    mw[M - 5] = 0.0                                            # a bone morph at rest is skipped, not slerped by 0
    worlds = {}
    for fuse in (0, 1):
        c.set_tuning(fuse_fk=fuse)
        for with_t in (True, False):                           # without uploaded translations the morphs' translations still apply
            c.set_pose_local(q, mw, t if with_t else None)
            assert c.get_tuning("effective_fuse_fk") == fuse
            c.deform()
            got = (c.read_world(0), c.read())
            check(got[0], got[1], q, t if with_t else np.zeros((B, 3)), mw, "local pose, fuse_fk=%d, translations=%s" % (fuse, with_t))
            worlds[(fuse, with_t)] = got
        c.set_pose_sampled(np.array([17.25], np.float32))
        c.deform()
        got = (c.read_world(0), c.read())
        qs, ts, ws = sample_reference(anim, 17.25, B, M)
        check(got[0], got[1], qs, ts, ws, "sampled pose, fuse_fk=%d" % fuse)
        worlds[(fuse, "sampled")] = got
    for key in (True, False, "sampled"):                       # one launch or three: the same bits
        assert np.array_equal(worlds[(0, key)][0], worlds[(1, key)][0]) and np.array_equal(worlds[(0, key)][1][0], worlds[(1, key)][1][0])
    # all weights zero = the plain solve, bit for bit
    c.set_tuning(fuse_fk=-1)
    c.set_pose_local(q, np.zeros(M, np.float32), t)
    c.deform()
    w0 = c.read_world(0)
    c.upload_bone_morphs([], [], [], [])
    c.set_pose_local(q, np.zeros(M, np.float32), t)
    c.deform()
    assert np.array_equal(w0, c.read_world(0))
    # a crowd: every instance folds its own weights
    c.upload_bone_morphs(bm_m, bm_b, bm_t, bm_q)
    I = 3
    c.set_instances(I)
    qI = rng.normal(size=(I, B, 4)).astype(np.float32)
    qI /= np.linalg.norm(qI, axis=2, keepdims=True)
    mwI = rng.random((I, M)).astype(np.float32)
    c.set_pose_local(qI, mwI, None)
    c.deform()
    for i in range(I):
        check(c.read_world(i), c.read(instance=i), qI[i], np.zeros((B, 3)), mwI[i], "crowd instance %d, local" % i)
    frames = np.array([3.5, 17.25, 29.0], np.float32)
    c.set_pose_sampled(frames)
    c.deform()
    for i in range(I):
        qs, ts, ws = sample_reference(anim, float(frames[i]), B, M)
        check(c.read_world(i), c.read(instance=i), qs, ts, ws, "crowd instance %d, sampled" % i)
    # a new morph set drops the entries (they name morphs of the old set)
    c.set_instances(1)
    c.upload_morphs_sparse(np.zeros(3, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 3), np.float32))
    c.set_pose_local(q, np.ones(2, np.float32), t)
    c.deform()
    plain = fk_reference(mesh["parents"], mesh["bind"], q, t, ap, ratio, move)
    assert np.abs(c.read_world(0) - plain).max() <= 1e-4 * max(1.0, np.abs(plain).max())
    c.close()


@pytest.mark.parametrize("fuse", [0, 1])
def test_real_bone_morph_on_the_device_against_reference_execution(rz, oracle, fuse):
    """The reference's 武器.pmx bone morph through the device path: rz_upload_bone_morphs + rz_set_pose_local (base rotations, the
    morph's weight riding with the morph weights), hierarchy solved on the GPU. World matrices and the skinned quarter of the mesh
    against what the REFERENCE's own quaternion, hierarchy and matrix code produced (tests/golden/ref_bone_morph.npz)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_bone_morph.npz"))
    v = g["vertices"]
    B = len(g["parents"])
    c = rz.DeformContext(0)
    c.upload_mesh(v[:, 0:3], v[:, 3:6], g["joints"], g["weights"])
    c.upload_skeleton(g["inv_bind"])
    c.upload_morphs_sparse(np.zeros(2, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 3), np.float32))     # one morph, a bone morph: no vertex deltas
    c.upload_skeleton_topology(g["parents"], g["bind"].astype(np.float32))
    c.upload_bone_morphs(g["entry_morph"], g["entry_bone"], g["entry_translation"], g["entry_rotation"])
    c.set_tuning(fuse_fk=fuse)
    for k, w in enumerate(g["morph_weights"]):
        c.set_pose_local(g["base_rotations"], np.array([w], np.float32), None)
        assert c.get_tuning("effective_fuse_fk") == fuse
        c.deform()
        world = c.read_world(0)
        assert world.shape == (B, 16) and np.abs(world - g["world"][k]).max() <= 1e-5 * max(1.0, np.abs(g["world"][k]).max()), (k, np.abs(world - g["world"][k]).max())
        pg, ng = c.read()
        assert_parity(pg, ng, g["skinned"][k][:, 0:3], g["skinned"][k][:, 3:6], "device bone morph vs reference execution, weight %g" % w)
    c.close()


def test_fork_keeps_two_frames_in_flight_on_shared_static_data(rz, oracle):
    """rz_fork: a second context that BORROWS the lender's static buffers (mesh, skeleton, topology, dense morph targets, motion,
    bone morphs, edge scale) and owns its streams, pose slots and outputs — the WebGPU queue's 'encode frame f + 1 while frame f
    runs' for this ABI. Different poses alternate between the two contexts (world-matrix, local and sampled poses; the hull and
    bounding-box consumers on); every frame must match the oracle and be bit-identical to the same pose run on the lender alone.
    Static uploads are refused on both sides while the fork lives, the lender cannot be destroyed before its fork, a fork
    cannot be forked; after the fork is gone the lender accepts new static data again."""
    V, B, M = 20000, 96, 12
    mesh = synth.make_mesh(V, B, seed=81)
    deltas, _ = synth.make_morphs_dense(V, M, seed=82)
    rng = np.random.default_rng(83)
    a = rz.DeformContext(0)
    with pytest.raises(rz.RzError):
        a.fork()                                               # nothing to share yet
    a.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    a.upload_skeleton(mesh["inv_bind"])
    a.upload_morphs_dense(deltas)
    a.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    edge = rng.random(V).astype(np.float32)
    a.upload_edge_scale(edge)
    a.enable_aabb(True)
    nk = 3
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    anim = dict(track_bone=np.arange(B), key_off=np.arange(B + 1) * nk, key_frame=np.tile(np.arange(nk) * 10.0, B), key_rot=kq,
                key_pos=(rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.3, key_interp=None,
                mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 20.0], np.float32), M), mkey_weight=rng.random(2 * M).astype(np.float32),
                feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
    a.upload_animation(anim["track_bone"], anim["key_off"], anim["key_frame"], anim["key_rot"], anim["key_pos"], None,
                       anim["mkey_off"], anim["mkey_frame"], anim["mkey_weight"], anim["feed_off"], anim["feed_track"], anim["feed_ratio"])
    bm = (np.array([M - 1, M - 2], np.uint32), np.array([5, 40], np.uint32), (rng.random((2, 3)) - 0.5).astype(np.float32), kq[:2, 0])
    a.upload_bone_morphs(*bm)
    world0 = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=84)
    a.set_pose(world0, rng.random(M).astype(np.float32))
    a.deform()
    a.autotune(0)                                              # the fork inherits the tuned plan
    b = a.fork()
    assert b.get_tuning("effective_split") == a.get_tuning("effective_split") and b.get_tuning("effective_grid") == a.get_tuning("effective_grid")
    with pytest.raises(rz.RzError):
        b.fork()
    for ctx in (a, b):                                         # static data is frozen on both sides
        with pytest.raises(rz.RzError):
            ctx.upload_morphs_dense(deltas)
        with pytest.raises(rz.RzError):
            ctx.upload_skeleton(mesh["inv_bind"])
        with pytest.raises(rz.RzError):
            ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
        with pytest.raises(rz.RzError):
            ctx.upload_bone_morphs(*bm)
        with pytest.raises(rz.RzError):
            ctx.upload_edge_scale(edge)
    assert a._L.rz_destroy(a._h) != 0                          # the lender outlives its forks

    def pose(kind, seed):
        r = np.random.default_rng(seed)
        w = (r.random(M) * (r.random(M) < 0.7)).astype(np.float32)
        if kind == "world":
            return ("world", synth.make_pose(mesh["parents"], mesh["bind"], B, seed=seed), w)
        if kind == "local":
            q = r.normal(size=(B, 4)).astype(np.float32)
            return ("local", q / np.linalg.norm(q, axis=1, keepdims=True), w, (r.random((B, 3), dtype=np.float32) - 0.5))
        return ("sampled", np.array([r.random() * 20], np.float32))

    def apply(ctx, ps):
        if ps[0] == "world":
            ctx.set_pose(ps[1], ps[2])
        elif ps[0] == "local":
            ctx.set_pose_local(ps[1], ps[2], ps[3])
        else:
            ctx.set_pose_sampled(ps[1])

    def result(ctx):
        return ctx.read(), ctx.read_hull(), ctx.read_aabb(0)

    kinds = ["world", "local", "sampled", "world", "sampled", "local", "world", "world"]
    poses = [pose(k, 900 + i) for i, k in enumerate(kinds)]
    # frames alternate between the contexts WITHOUT a sync in between: two frames in flight
    got = []
    for i in range(0, len(poses), 2):
        apply(a, poses[i]); a.deform()
        apply(b, poses[i + 1]); b.deform()
        got.append(result(a)); got.append(result(b))
    # the same poses on the lender alone
    for i, ps in enumerate(poses):
        apply(a, ps); a.deform()
        ref = result(a)
        assert np.array_equal(ref[0][0], got[i][0][0]) and np.array_equal(ref[0][1], got[i][0][1]), "frame %d (%s): fork vs lender" % (i, ps[0])
        assert np.array_equal(ref[1], got[i][1]) and np.array_equal(ref[2], got[i][2])
        if ps[0] == "world":
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], ps[1], mesh["inv_bind"], deltas, ps[2])
            assert_parity(got[i][0][0], got[i][0][1], pr, nr, "forked frame %d" % i)
            assert_hull(got[i][1], oracle.hull(pr, nr, edge), "forked frame %d hull" % i)
    # a replay that alternates inside the library
    apply(a, poses[0]); apply(b, poses[3])
    a.deform_pair(b, 41)
    ra, rb = a.read(), b.read()
    assert np.array_equal(ra[0], got[0][0][0]) and np.array_equal(rb[0], got[3][0][0])
    with pytest.raises(rz.RzError):
        a.deform_pair(a, 2)
    # instance counts are per context
    b.set_instances(3)
    b.set_pose(np.stack([poses[0][1], poses[3][1], poses[6][1]]), None)
    b.deform()
    a.deform()
    assert np.array_equal(a.read()[0], got[0][0][0])
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], poses[6][1], mesh["inv_bind"], deltas, np.zeros(M, np.float32))
    pg, ng = b.read(instance=2)
    assert_parity(pg, ng, pr, nr, "crowd on the fork")
    b.close()
    a.upload_morphs_dense(deltas[:4])                          # the lender owns its static data again
    a.set_pose(world0, np.ones(4, np.float32))
    a.deform()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world0, mesh["inv_bind"], deltas[:4], np.ones(4, np.float32))
    assert_parity(*a.read(), pr, nr, "lender after its fork is gone")
    a.close()
