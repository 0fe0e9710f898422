"""Round 6 on the GPU: caller-written poses (rz_map_pose / rz_commit_pose), the event-timed K-step span (rz_time_span), crowds sharded
along the instance axis, the readers after a crowd frame that solved its hierarchy in LDS only, the launch-shape heuristics at the shard
sizes of C5, and the kernel-variant selection. Everything goes through the C ABI (ctypes); the oracle and the float64 restatements are
the checkers."""
import numpy as np
import pytest

from helpers import assert_hull, assert_parity, fk_reference, sample_reference
from test_gpu_round5 import _all_instances, _crowd, _world_crowd

pytestmark = pytest.mark.gpu

synth = None


@pytest.fixture(autouse=True)
def _synth(rz):
    global synth
    synth = rz.synth
    import test_gpu_round5
    test_gpu_round5.synth = rz.synth


def _rows(worlds):
    """[.., B, 16] column-major 4 x 4 -> [.., B, 12]: the four columns' x y z (RZ_POSE_ROWS12)."""
    w = np.asarray(worlds, np.float32)
    return np.ascontiguousarray(w.reshape(w.shape[:-1] + (4, 4))[..., :3].reshape(w.shape[:-1] + (12,)))


def test_readers_after_a_lds_only_crowd_frame_follow_the_current_pose(rz):
    """Advisor (round 5, medium): a crowd frame that solves its hierarchy in the skin kernel's front leaves neither world matrices nor
    palettes in memory and marks the context (`fk_stale`); rz_read_world / rz_read_palette then run rz_fk_kernel on demand. The mark
    spoke of THAT pose: after rz_set_pose with world matrices the readers ran the hierarchy solve on a pose block that now held world
    matrices and overwrote them. Now: the uploaded matrices come back as they went in, the frame is the frame of those matrices, and a
    skeleton replaced by a larger one leaves no record of the old one behind."""
    V, B, I = 12000, 64, 10
    c, mesh, s = _crowd(rz, V, B, I, seed=3, depth_chain=8)
    c.set_pose_local(s["q"], None, s["lt"])
    assert c.get_tuning("effective_fuse_fk") == 1
    c.deform()                                                  # hierarchy solved in LDS only
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=900 + i) for i in range(I)]).astype(np.float32)
    c.set_pose(worlds)
    c.deform()
    base = _all_instances(c, I)
    for i in (0, I - 1):
        assert np.array_equal(c.read_world(i).reshape(B, 16), worlds[i].reshape(B, 16)), "instance %d: rz_read_world after a subfk frame + rz_set_pose" % i
    pal = c.read_palette(I - 1)
    c.deform()                                                  # the resident pose was not damaged by the readers
    for a, b in zip(base, _all_instances(c, I)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c2 = rz.DeformContext(0)                                    # the same pose on a context that never saw a device-animated frame
    c2.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c2.upload_skeleton(mesh["inv_bind"]); c2.set_instances(I)
    c2.set_pose(worlds); c2.deform()
    assert np.array_equal(c2.read(I - 1)[0], base[I - 1][0]) and np.array_equal(c2.read_palette(I - 1), pal)
    c2.close()
    # one character again, device-animated, then host-animated: the readers follow
    c.set_pose_local(s["q"], None, s["lt"]); c.deform()
    c.set_instances(1)
    c.set_pose(worlds[3]); c.deform()
    assert np.array_equal(c.read_world(0).reshape(B, 16), worlds[3].reshape(B, 16))
    # a larger skeleton: the old topology's records are gone with it (they were sized for 64 bones)
    c.set_instances(I)
    c.set_pose_local(s["q"], None, s["lt"]); c.deform()
    mesh2 = synth.make_mesh(V, 3 * B, seed=5)
    c.upload_mesh(mesh2["pos"], mesh2["nrm"], mesh2["joints"], mesh2["weights"]); c.upload_skeleton(mesh2["inv_bind"])
    w2 = np.stack([synth.make_pose(mesh2["parents"], mesh2["bind"], 3 * B, seed=70 + i) for i in range(I)]).astype(np.float32)
    c.set_pose(w2); c.deform()
    assert np.array_equal(c.read_world(I - 1).reshape(3 * B, 16), w2[I - 1].reshape(3 * B, 16))
    with pytest.raises(rz.RzError):
        c.set_pose_local(np.zeros((I, 3 * B, 4), np.float32))   # no topology for this skeleton
    c.close()


def test_fork_of_a_device_animated_crowd_takes_the_one_launch_frame(rz):
    """Advisor (round 5, low): rz_fork did not copy the host-side mirrors of the hierarchy's records, so a fork never took the one-launch
    closure frame and a context and its fork alternated two different frame shapes. Same plan, same closure, same bits."""
    V, B, I = 16000, 120, 24
    c, mesh, s = _crowd(rz, V, B, I, seed=21, depth_chain=11)
    c.set_pose_local(s["q"], None, s["lt"])
    c.deform()
    f = c.fork()
    f.set_pose_local(s["q"], None, s["lt"])
    for k in ("effective_fuse_fk", "effective_closure_bones", "effective_subsets", "effective_grid", "effective_inst_group"):
        assert f.get_tuning(k) == c.get_tuning(k), k
    assert c.get_tuning("effective_closure_bones") > 0 and f.kernel_name() == c.kernel_name() and "rz_skin_instances_fk_kernel" in c.kernel_name()
    f.deform()
    for a, b in zip(_all_instances(c, I), _all_instances(f, I)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    f.close()
    c.close()


@pytest.mark.parametrize("V,B,I,M", [(6000, 201, 23, 0), (30000, 200, 64, 0), (5000, 201, 23, 5)])
def test_mapped_crowd_pose_equals_the_pose_handed_over(rz, V, B, I, M):
    """rz_map_pose / rz_commit_pose (ABI 7): the caller writes its matrices straight into the pinned ring slot the pull kernel reads — as
    48-byte rows or as whole 4 x 4 matrices — and the frame must be the frame of rz_set_pose with the same matrices BIT FOR BIT on every
    instance; the device block holds what a set_pose would have put there (rz_read_world, bottom row 0 0 0 1 written back); morph
    weights ride behind the matrices, unwritten ones are zero."""
    c, mesh, worlds, mws = _world_crowd(rz, V, B, I, seed=B + I + M, M=M)
    c.set_pose(worlds, mws)
    c.deform()
    ref = _all_instances(c, I)
    for layout in (rz.capi.POSE_ROWS12, rz.capi.POSE_WORLD16):
        c.set_pose(worlds[::-1].copy(), mws)                    # something else resident in between
        c.deform()
        mats, mw = c.map_pose(layout)
        assert mats.shape == (I, B, 12 if layout == rz.capi.POSE_ROWS12 else 16) and (mw is None) == (M == 0)
        if M:
            assert not mw.any()                                 # handed out zero-filled
            mw[:] = mws
        mats[:] = _rows(worlds) if layout == rz.capi.POSE_ROWS12 else worlds.reshape(I, B, 16)
        c.commit_pose()
        assert c.get_tuning("pose_pulled") == 1 and c.get_tuning("pose_rows") == (1 if layout == rz.capi.POSE_ROWS12 else 0)
        c.deform()
        got = _all_instances(c, I)
        for i in range(I):
            assert np.array_equal(got[i][0], ref[i][0]) and np.array_equal(got[i][1], ref[i][1]), "layout %d instance %d" % (layout, i)
        for i in (0, I - 1):
            assert np.array_equal(c.read_world(i).reshape(B, 16), worlds[i].reshape(B, 16))
    # a commit that has to be refused (rows mapped, then the pull switched off: nothing on the host could expand them) changes nothing:
    # the resident pose is still the one frames deform
    mats, _ = c.map_pose(rz.capi.POSE_ROWS12)
    mats[:] = _rows(worlds[::-1])
    c.set_tuning(pose_pull=0)
    with pytest.raises(rz.RzError) as e:
        c.commit_pose()
    assert e.value.code == -6
    c.set_tuning(pose_pull=-1)
    c.deform()
    assert np.array_equal(c.read(I - 1)[0], ref[I - 1][0]) and np.array_equal(c.read(0)[1], ref[0][1])
    with pytest.raises(rz.RzError):
        c.set_tuning(overlap=1) or c.map_pose()                 # the opt-in overlapped-front protocol hands its poses over with rz_set_pose
    c.close()


def test_mapped_pose_of_one_character_is_read_in_place(rz, oracle):
    """One character (<= 256 KB): the mapped memory is the zero-copy slot the frame's own kernel reads; same bits as rz_set_pose, with
    dense morphs (the host-compacted active list is built from the weights the caller wrote into the slot), through the prefetch
    protocol (the header is written by the commit), and against the oracle."""
    V, B, M = 30000, 200, 12
    mesh = synth.make_mesh(V, B, seed=4)
    deltas, mw0 = synth.make_morphs_dense(V, M, seed=6)
    mw0[::4] = 0.0
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(deltas)
    poses = [synth.make_pose(mesh["parents"], mesh["bind"], B, seed=50 + k).astype(np.float32) for k in range(5)]
    ref = []
    for w in poses:
        c.set_pose(w, mw0); c.deform(); ref.append(c.read())
    with pytest.raises(rz.RzError) as e:
        c.map_pose(rz.capi.POSE_ROWS12)
    assert e.value.code == -6 and "WORLD16" in str(e.value)
    for rounds in range(3):                                     # back to back, nothing waiting in between: the helper workgroup of frame k may stage pose k + 1
        for k, w in enumerate(poses):
            mats, mw = c.map_pose()
            mats[0] = w.reshape(B, 16)
            mw[0] = mw0
            c.commit_pose()
            c.deform()
            if rounds == 2 or k == 4:
                got = c.read()
                assert np.array_equal(got[0], ref[k][0]) and np.array_equal(got[1], ref[k][1]), (rounds, k)
    assert c.get_tuning("pose_resident") in (0, 1)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], poses[4], mesh["inv_bind"], deltas, mw0)
    assert_parity(*c.read(), pr, nr, "mapped pose of one character")
    # misuse: commit without a mapping; a pose handed over whole cancels a mapping; two maps in a row — the second one counts
    with pytest.raises(rz.RzError):
        c.commit_pose()
    c.map_pose()
    c.set_pose(poses[0], mw0)
    with pytest.raises(rz.RzError):
        c.commit_pose()
    c.deform()
    assert np.array_equal(c.read()[0], ref[0][0])
    c.map_pose()
    mats, mw = c.map_pose()
    mats[0] = poses[1].reshape(B, 16); mw[0] = mw0
    c.commit_pose(); c.deform()
    assert np.array_equal(c.read()[0], ref[1][0])
    # the crowd changed between map and commit: refused, the resident pose stays
    c.map_pose()
    c.set_instances(3)
    with pytest.raises(rz.RzError):
        c.commit_pose()
    c.close()


def test_mapped_crowd_ring_never_serves_a_stale_or_torn_pose(rz):
    """The stale / torn test of the crowd pose ring (test_gpu_round5), for caller-written poses: 80 frames with nothing waiting in between,
    three poses cycled, a context and its fork mapping alternately (two frames in flight, each its own ring) — and a caller that commits
    LATE: it maps, lets the GPU run on (a replay of the resident pose), writes, then commits. A slot handed out while a pull of its previous
    tenant was still reading it, or a frame that started before its pull ended, would show as wrong bits."""
    V, B, I = 8000, 200, 40
    c, mesh, worlds, _ = _world_crowd(rz, V, B, I, seed=5)
    poses = [worlds, worlds[::-1].copy(), np.roll(worlds, 7, axis=0).copy()]
    rows = [_rows(p) for p in poses]
    picks = (0, 17, I - 1)
    iso = []
    for p in poses:
        c.set_pose(p); c.deform(); c.sync()
        iso.append([c.read(i) for i in picks])
    f = c.fork()
    ctxs = (c, f)
    last = {}
    for k in range(80):
        x = ctxs[k & 1]
        mats, _ = x.map_pose(rz.capi.POSE_ROWS12)
        if k % 5 == 3 and (k & 1) in last:
            x.deform_n(2)                                       # the late committer: frames of the RESIDENT pose run between map and commit
        mats[:] = rows[k % 3]
        x.commit_pose()
        x.deform()
        last[k & 1] = k % 3
        if k % 7 == 6 or k >= 76:
            for y in (0, 1):
                for n, i in enumerate(picks):
                    got = ctxs[y].read(i)
                    assert np.array_equal(got[0], iso[last[y]][n][0]) and np.array_equal(got[1], iso[last[y]][n][1]), "frame %d context %d instance %d" % (k, y, i)
    # mapped and handed-over poses mixed on one ring
    for k in range(24):
        if k % 3 == 1:
            c.set_pose(poses[k % 3])
        else:
            mats, _ = c.map_pose(rz.capi.POSE_ROWS12 if k % 2 else rz.capi.POSE_WORLD16)
            mats[:] = rows[k % 3] if k % 2 else poses[k % 3].reshape(I, B, 16)
            c.commit_pose()
        c.deform()
        if k % 4 == 3:
            for n, i in enumerate(picks):
                assert np.array_equal(c.read(i)[0], iso[k % 3][n][0]), "mixed frame %d instance %d" % (k, i)
    f.close()
    c.close()


def test_time_span_is_the_event_time_of_exactly_k_frames(rz):
    """rz_time_span (what bench.py's `ms_per_step` is): K frames of the resident pose between two events on the stream, `lead` untimed
    frames in front of the opening event. It must be about K x the event-timed frame of rz_time_frames (not K - 1, not K + the lead
    frames), grow linearly in K, come out no longer WITH lead-in frames than without (they take the first launch out of the span),
    leave the outputs those of the pose, and with a fork alternate the frames over two streams (never slower than 1.15 x one stream,
    never faster than half)."""
    V, B, M = 125184, 256, 64
    mesh = synth.make_mesh(V, B)
    deltas, mw = synth.make_morphs_dense(V, M)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(deltas)
    c.set_pose(mesh["world"], mw)
    c.deform(); ref = c.read()
    for _ in range(20):
        c.deform_n(200)
    c.sync()
    frame_ms = min(c.time_frames(200)["frame_ms"] for _ in range(3))
    s20 = min(c.time_span(20) for _ in range(7))
    l20 = min(c.time_span(20, lead=2) for _ in range(7))
    l200 = min(c.time_span(200, lead=2) for _ in range(5))
    assert 0.9 * 20 * frame_ms <= l20 <= 1.08 * 20 * frame_ms, (l20, frame_ms)
    assert l20 <= s20 * 1.04 and s20 <= 1.25 * 20 * frame_ms + 0.01        # (observed: 16.5 against 16.9-17.2 us per frame), (l20, s20, frame_ms)
    assert 0.95 * 200 * frame_ms <= l200 <= 1.05 * 200 * frame_ms, (l200, frame_ms)
    assert 0.9 * 180 * frame_ms <= l200 - l20 <= 1.1 * 180 * frame_ms
    f = c.fork()
    f.set_pose(mesh["world"], mw)
    p200 = min(c.time_span(200, f, lead=3) for _ in range(5))
    assert 0.5 * l200 <= p200 <= 1.15 * l200, (p200, l200)
    for x in (c, f):
        got = x.read()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    with pytest.raises(rz.RzError):
        c.time_span(0)
    print("time_span: frame %.2f us; 20 frames from idle %.2f us per frame, behind 2 lead-in frames %.2f; 200 frames %.2f" % (frame_ms * 1e3, s20 / 20 * 1e3, l20 / 20 * 1e3, l200 / 200 * 1e3))
    f.close()
    c.close()


def test_crowd_sharded_along_the_instance_axis(rz, oracle):
    """SURVEY 8e, last sentence (BASELINE config 4 over N GPUs): every rank holds the whole mesh and poses its own contiguous range of the
    crowd's instances — no exchange. Three ranks' contexts (sharing this GPU) against one context posing the whole crowd: instance k of
    the crowd equals instance k - begin of its rank BIT FOR BIT, whatever launch shape the smaller crowd gets; the ranges tile the
    crowd; the oracle agrees."""
    V, B, I, N = 30000, 200, 50, 3
    c, mesh, worlds, _ = _world_crowd(rz, V, B, I, seed=9)
    c.set_pose(worlds); c.deform()
    whole = _all_instances(c, I)
    c.close()
    covered = 0
    for r in range(N):
        b, n = rz.shard.instances_of(I, N, r)
        assert b == covered and n == (17 if r < 2 else 16)
        covered += n
        x = rz.DeformContext(0)
        x.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); x.upload_skeleton(mesh["inv_bind"])
        x.set_instances(n)
        x.set_pose(worlds[b:b + n]); x.deform()
        for k in range(n):
            got = x.read(k)
            assert np.array_equal(got[0], whole[b + k][0]) and np.array_equal(got[1], whole[b + k][1]), "rank %d instance %d" % (r, b + k)
        with pytest.raises(rz.RzError):                         # a crowd takes no part in a gather: nothing to exchange
            x.comm_init(1, 0, rz.capi.comm_unique_id(), V)
        x.close()
    assert covered == I
    assert rz.shard.instances_of(3, 8, 7) == (3, 0) and rz.shard.instances_of(256, 8, 7) == (224, 32)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[I - 1], mesh["inv_bind"])
    assert_parity(whole[I - 1][0], whole[I - 1][1], pr, nr, "instance %d" % (I - 1))


def test_one_launch_sampled_crowd_against_the_float64_sampler(rz, oracle):
    """Review item 5b: the crowd frame that samples its motion AND solves its hierarchy in the skin kernel's front, against the float64
    restatement of the sampler + Model.computeWorldMatrices + the oracle's skin on three instances (round 5 compared it with the
    two-launch frame only — a self-comparison)."""
    V, B, I = 20000, 120, 21
    c, mesh, s = _crowd(rz, V, B, I, seed=77)
    rng = np.random.default_rng(5)
    nk = 6
    keyed = rng.random(B) < 0.8
    tb = np.nonzero(keyed)[0].astype(np.int32)
    n = len(tb)
    kq = rng.normal(size=(n, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    anim = dict(track_bone=tb, key_off=(np.arange(n + 1) * nk).astype(np.uint32), key_frame=np.tile(np.cumsum(rng.integers(1, 9, size=nk)).astype(np.float32), n),
                key_rot=kq.reshape(-1, 4), key_pos=((rng.random((n * nk, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32),
                key_interp=rng.integers(1, 127, size=(n * nk, 16)).astype(np.uint8))
    c.upload_animation(anim["track_bone"], anim["key_off"], anim["key_frame"], anim["key_rot"], anim["key_pos"], anim["key_interp"])
    frames = (rng.random(I) * 50.0).astype(np.float32)
    frames[0], frames[I - 1] = -2.0, 400.0                      # before the first key, past the last
    c.set_pose_sampled(frames)
    assert c.get_tuning("effective_fuse_fk") == 1 and c.get_tuning("effective_closure_bones") > 0 and "rz_skin_instances_fk_kernel" in c.kernel_name()
    c.deform()
    worst = 0.0
    for i in (0, I // 2, I - 1):
        q, t, _ = sample_reference(anim, float(frames[i]), B, 0)
        ref = fk_reference(s["parents"], s["bind"], q, t, s["ap"], s["ratio"], s["mv"])
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], ref.reshape(B, 16).astype(np.float32), mesh["inv_bind"])
        pg, ng = c.read(i)
        ep = np.linalg.norm(pg - pr, axis=1) / np.maximum(np.linalg.norm(pr, axis=1), 1.0)
        en = np.linalg.norm(ng - nr, axis=1)
        worst = max(worst, ep.max(), en.max())
        assert ep.max() <= 2e-4 and en.max() <= 2e-4, (i, ep.max(), en.max())      # (a 14-deep f32 chain + f32 sampling against float64: twice the skin's bar)
        got = c.read_world(i)                                    # solved on demand by rz_fk_kernel: the same functions, against float64
        assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    print("one-launch sampled crowd vs float64: worst error %.3e" % worst)
    c.close()


def test_the_launcher_picks_the_variant_a_frame_needs(rz, oracle):
    """Review item 5c: a context with neither an edge scale nor the bounding box launches kernel variant 3 (compiled without the fused
    consumers), one with either launches variant 0 — dense and morph-free frames alike; every variant against the oracle, the outputs
    of the two variants bit-identical, the hull and the box right."""
    for V, B, M in ((40000, 200, 16), (30000, 200, 0)):
        mesh = synth.make_mesh(V, B, seed=13)
        deltas, mw = synth.make_morphs_dense(V, M, seed=14) if M else (None, None)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
        c = rz.DeformContext(0)
        c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
        if M:
            c.upload_morphs_dense(deltas)
        c.set_pose(mesh["world"], mw)
        assert c.get_tuning("effective_variant") == 3 and c.kernel_name().endswith(", 3>")
        c.deform()
        plain = c.read()
        assert_parity(plain[0], plain[1], pr, nr, "variant 3, M = %d" % M)
        edge = (np.random.default_rng(2).random(V, dtype=np.float32) * 1.5).astype(np.float32)
        for what in ("edge", "aabb", "both"):
            c.upload_edge_scale(edge if what in ("edge", "both") else None)
            c.enable_aabb(what in ("aabb", "both"))
            c.set_pose(mesh["world"], mw)
            assert c.get_tuning("effective_variant") == 0 and c.kernel_name().endswith(", 0>"), what
            c.deform()
            got = c.read()
            assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1]), what
            if what != "aabb":
                assert_hull(c.read_hull(), pr + nr * edge[:, None].astype(np.float64) * 0.01, what)
            if what != "edge":
                box = c.read_aabb()
                assert np.array_equal(box[:3], plain[0].min(axis=0)) and np.array_equal(box[3:], plain[0].max(axis=0))
        c.upload_edge_scale(None)
        c.enable_aabb(False)
        c.set_pose(mesh["world"], mw)
        assert c.get_tuning("effective_variant") == 3
        with pytest.raises(rz.RzError):
            c.get_tuning("effective_nonsense")                  # refused before any work
        c.close()


@pytest.mark.parametrize("verts,split,grid", [(1000000, 2, 489), (875008, 2, 428), (797440, 4, 480), (625152, 4, 489), (530432, 4, 461), (500224, 2, 489), (400128, 2, 391), (375040, 2, 733), (333568, 4, 435), (313856, 4, 491), (281600, 4, 440), (250112, 4, 489), (156416, 4, 611), (125184, 4, 489), (93952, 8, 734), (30000, 8, 235)])
def test_heuristic_plan_at_the_shard_sizes_of_c5(rz, verts, split, grid):
    """Review item 2: what make_plan picks at the shard sizes of N = 1, 2, 4, 8 (and between them, and C3) — the morph split the round-6
    sweeps found best there (profiles/r6_plan_sweep.txt: at 250 112 vertices S = 2 left every wave with one long step, 37.4 us against
    34.7 at S = 4), whole wave steps where that costs at most one eighth of the workgroups (875 k: 428 workgroups of 4 steps) and not
    where it would leave half the CUs with one workgroup (282 k: 440, not 367 — profiles/r6_fresh_plans.txt), one step per wave at
    S = 8; S = 4 up to ~365 k vertices (the N = 3 shard: three whole steps per wave instead of S = 2's 1.5); one step per wave on as many
    workgroups as that takes where 1.5 would be left (375 k at S = 2, 156 k at S = 4) — profiles/r6_fresh_plans_mid.txt; and S = 4 on the same
    grid wherever S = 2 would leave a ragged run (2.25 / 2.5 / 3.25 steps: 530 k, 625 k, 797 k — profiles/r6_fresh_plans_hi.txt). The plan is a
    pure function of the sizes: the morph data may be anything."""
    B, M = 256, 64
    mesh = synth.make_mesh_range(verts, B, 0, min(verts, 4096))
    c = rz.DeformContext(0)
    n = min(verts, 4096)
    pos = np.zeros((verts, 3), np.float32); nrm = np.zeros((verts, 3), np.float32); nrm[:, 1] = 1
    j = np.zeros((verts, 4), np.uint16); w = np.zeros((verts, 4), np.uint8); w[:, 0] = 255
    pos[:n], nrm[:n], j[:n], w[:n] = mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]
    c.upload_mesh(pos, nrm, j, w); c.upload_skeleton(mesh["inv_bind"])
    c.upload_morphs_dense(np.zeros((M, verts, 3), np.float32))
    c.set_pose(mesh["world"], np.ones(M, np.float32))
    assert (c.get_tuning("effective_split"), c.get_tuning("effective_grid")) == (split, grid), (verts, c.get_tuning("effective_split"), c.get_tuning("effective_grid"))
    c.deform()
    assert np.isfinite(c.read(0, 0, 64)[0]).all()
    c.close()


@pytest.mark.parametrize("verts,split,grid", [(156416, 4, 611), (156419, 4, 612), (313856, 4, 491), (313859, 4, 491), (375040, 2, 733), (375043, 2, 733), (530432, 4, 461)])
def test_launch_shapes_of_the_second_pass_against_the_oracle(rz, oracle, verts, split, grid):
    """The launch shapes round 6's second pass introduced (NOTEBOOK R6.9) — one step per wave on more workgroups than fit at once (S = 4:
    611, S = 2: 733), S = 4 runs of 2.5 and 4.5 steps on the persistent grid — against the oracle over the WHOLE mesh, dense morphs
    streaming (12 targets keep the upload small; the plan depends on the vertex count alone; vertex counts that are no multiple of 4 leave
    the last quad, the last step and the last workgroup ragged), and against the same mesh
    under S = 1 on a plain persistent grid: every vertex is written exactly once whatever the partition."""
    from helpers import assert_parity
    B, M = 64, 12
    mesh = synth.make_mesh_range(verts, B, 0, verts)
    deltas, mw = synth.make_morphs_dense_range(verts, M, 0, verts)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(deltas)
    c.set_pose(mesh["world"], mw)
    assert (c.get_tuning("effective_split"), c.get_tuning("effective_grid")) == (split, grid)
    c.deform()
    pg, ng = c.read()
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    assert_parity(pg, ng, pr, nr, "%d vertices at S = %d / %d workgroups" % (verts, split, grid))
    c.set_tuning(morph_split=1, grid_cap=512)
    c.deform()
    p1, n1 = c.read()
    # (the S lanes of a quad add their partial sums in another order than one lane does: positions agree to rounding, normals follow)
    assert_parity(p1, n1, pr, nr, "%d vertices at S = 1" % verts)
    assert np.abs(p1 - pg).max() <= 2e-6 * max(1.0, np.abs(pr).max()) + 1e-5
    c.close()
