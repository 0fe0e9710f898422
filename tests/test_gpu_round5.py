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
