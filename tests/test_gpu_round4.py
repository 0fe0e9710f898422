"""GPU tests added in round 4: the LDS-staged sparse-row walk (pieces, long rows, > 256 morphs), the pointer-doubling hierarchy
solve (deep chains, > 256 bones, parents in any order, append bones), local poses that become resident and are prefetched like
world poses (forced hits through the gate hook of the tools-only build), shards cut at 256 vertices."""
import numpy as np
import pytest

from helpers import assert_parity, fk_reference
from reze_engine_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(rz):
    c = rz.DeformContext(0)
    yield c
    c.close()


def _frame(ctx, mesh, sparse, mw, world=None, **tuning):
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_sparse(*sparse)
    ctx.set_instances(1)
    ctx.set_tuning(**dict(dict(morph_split=0, grid_cap=0, fast=-1, out_cap=-1), **tuning))
    ctx.set_pose(mesh["world"] if world is None else world, mw)
    ctx.deform()
    return ctx.read()


@pytest.mark.parametrize("V,B,M,per_vertex,tuning", [
    (5000, 40, 300, 260, {}),                       # rows of ~260 entries: 64 x 260 entries per step >> the LDS buffer -> many pieces; > 256 morphs
    (5000, 40, 300, 260, {"morph_split": 1}),       # 256-vertex steps: four rounds per step, bounds loaded round by round
    (70001, 64, 48, 40, {"grid_cap": 7}),           # several steps per wave, ragged tail
    (300, 8, 12, 12, {"fast": 0}),                  # fewer vertices than one workgroup; prep-kernel frame
    (4097, 513, 20, 6, {}),                         # skeleton beyond 512 bones: early bones + late loop
])
def test_sparse_rows_staged_through_lds(ctx, oracle, V, B, M, per_vertex, tuning):
    """rz_deform_kernel MODE 2 (round 4): a step's CSR range is copied into LDS in pieces and every vertex walks its row out of
    LDS. Rows far longer than a piece, rows that straddle pieces, duplicate entries (rows longer than M), empty steps."""
    mesh = synth.make_mesh(V, B, seed=V + 3)
    rng = np.random.default_rng(V + 4)
    heavy = rng.choice(V, size=max(1, V // 20), replace=False)           # 5 % of the vertices carry almost every morph
    off, idx, d3 = [0], [], []
    for m in range(M):
        take = heavy[rng.random(len(heavy)) < per_vertex / M]
        extra = rng.choice(V, size=max(1, V // 200), replace=False)      # + a thin spread over the whole mesh
        dup = take[:3]                                                   # + duplicates inside the morph (both offsets add)
        ids = np.concatenate([take, extra, dup]).astype(np.uint32)
        idx.append(ids); d3.append(rng.normal(scale=0.02, size=(len(ids), 3)).astype(np.float32)); off.append(off[-1] + len(ids))
    off = np.array(off, np.uint32); idx = np.concatenate(idx); d3 = np.concatenate(d3)
    mw = (rng.random(M) * (rng.random(M) < 0.9)).astype(np.float32)
    assert np.bincount(idx, minlength=V).max() >= min(per_vertex, M) * 0.6
    pm = oracle.morph_sparse(V, off, idx, d3, mw, mesh["pos"])
    S = oracle.palette(mesh["world"], mesh["inv_bind"])
    pr, nr = oracle.skin(pm, mesh["nrm"], mesh["joints"], mesh["weights"], S)
    pg, ng = _frame(ctx, mesh, (off, idx, d3), mw, **tuning)
    assert_parity(pg, ng, pr, nr, "LDS-staged sparse rows V=%d M=%d %s" % (V, M, tuning))
    # the row sum does not depend on the launch shape: another grid, the same bits
    pg2, ng2 = _frame(ctx, mesh, (off, idx, d3), mw, **dict(tuning, grid_cap=3 if tuning.get("grid_cap") != 3 else 5))
    assert np.array_equal(pg, pg2) and np.array_equal(ng, ng2), "sparse frame depends on the launch shape"


def _random_tree(B, depth_chain, rng):
    """parents in ANY order (a child may come before its parent), one chain of `depth_chain` bones, the rest shallow"""
    order = rng.permutation(B)
    parents = np.full(B, -1, np.int32)
    for k in range(1, depth_chain):
        parents[order[k]] = order[k - 1]
    for k in range(depth_chain, B):
        parents[order[k]] = order[int(rng.integers(0, k))] if rng.random() < 0.9 else -1
    return parents


@pytest.mark.parametrize("B,depth", [(200, 12), (349, 40), (700, 3), (64, 64), (5, 1)])
def test_pointer_doubling_hierarchy_solve(rz, oracle, B, depth):
    """rz_fk_kernel / the fused prologue (model.ts:330-420): parent chains resolved by pointer doubling — deep chains (64 levels:
    six rounds), depths that are no power of two, skeletons beyond 256 / 512 bones (several bones per thread), parents listed
    after their children, a forest with a single level; append rotation + append move on a third of the bones; local
    translations. Against the float64 restatement of the reference's solve; fused == three-kernel frame bit for bit."""
    rng = np.random.default_rng(B * 7 + depth)
    V = 6000
    mesh = synth.make_mesh(V, B, seed=B)
    parents = _random_tree(B, min(depth, B), rng)
    bind = (rng.random((B, 3), dtype=np.float32) - 0.5).astype(np.float32)
    ap = np.where(rng.random(B) < 0.33, rng.integers(0, B, size=B), -1).astype(np.int32)
    ratio = (rng.random(B, dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    mv = (rng.random(B) < 0.5).astype(np.uint8)
    q = rng.normal(size=(B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    lt = ((rng.random((B, 3), dtype=np.float32) - 0.5) * 0.3).astype(np.float32)
    ref = fk_reference(parents, bind, q, lt, ap, ratio, mv)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    inv_bind = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (B, 1))
    c.upload_skeleton(inv_bind)
    c.upload_skeleton_topology(parents, bind, ap, ratio, mv)
    outs = {}
    for fuse in (1, 0):
        c.set_tuning(fuse_fk=fuse)
        c.set_pose_local(q, None, lt)
        c.deform()
        assert c.get_tuning("effective_fuse_fk") == fuse
        w = c.read_world(0)
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(w, ref.reshape(B, 16), rtol=0, atol=4e-5 * scale * max(1, depth // 8), err_msg="world matrices, fuse_fk=%d" % fuse)
        outs[fuse] = c.read()
        c.deform_n(3)                                   # replays read the pose workgroup 0 left in the device block
        assert c.get_tuning("pose_resident") == fuse       # fused frame: workgroup 0 kept the pose; rz_fk_kernel (ONE workgroup) reads the pinned slot each time
        p2, n2 = c.read()
        assert np.array_equal(p2, outs[fuse][0]) and np.array_equal(n2, outs[fuse][1]), "replay of a resident local pose"
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), "fused prologue vs rz_fk_kernel"
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], ref.reshape(B, 16).astype(np.float32), inv_bind)
    ep = np.linalg.norm(outs[1][0] - pr, axis=1) / np.maximum(np.linalg.norm(pr, axis=1), 1.0)
    assert ep.max() <= 1e-4 * max(1, depth // 8), ep.max()        # (f32 solve of a 64-deep chain vs float64: error grows with the depth)
    c.close()


@pytest.mark.parametrize("morphs", ["none", "sparse"])
def test_local_poses_are_prefetched_like_world_poses(rz, rzv, oracle, morphs):
    """Zero-copy LOCAL poses (rz_set_pose_local, one character, fused hierarchy solve): the frame of pose u carries a helper
    workgroup that stages pose u + 1, the staged copy is taken when its sequence number — which carries the pose KIND — matches,
    and on a miss workgroup 0 leaves the pose in the device block. Forced hits by construction (frames queued behind the gate
    hook of the tools-only build), forced misses (a sync after every frame), kinds alternating (world / rotations / rotations +
    translations: a helper never stages for a pose of another kind), every frame bit-identical to its pose run in isolation."""
    V, B = 60000, 150
    mesh = synth.make_mesh(V, B, seed=41)
    rng = np.random.default_rng(42)
    M = 0
    g = rzv.DeformContext(0)
    g.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    g.upload_skeleton(mesh["inv_bind"])
    if morphs == "sparse":
        M = 16
        off, vi, d3, _ = synth.make_morphs_sparse(V, M, seed=43)
        g.upload_morphs_sparse(off, vi, d3)
    g.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    P = 5
    qs = rng.normal(size=(P, B, 4)).astype(np.float32); qs /= np.linalg.norm(qs, axis=2, keepdims=True)
    lts = ((rng.random((P, B, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    mws = [(rng.random(M).astype(np.float32) * (rng.random(M) < 0.8)).astype(np.float32) if M else None for _ in range(P)]
    worlds = [synth.make_pose(mesh["parents"], mesh["bind"], B, seed=900 + k) for k in range(P)]

    def upload(kind, k):
        if kind == "world":
            g.set_pose(worlds[k], mws[k])
        elif kind == "rot":
            g.set_pose_local(qs[k], mws[k])
        else:
            g.set_pose_local(qs[k], mws[k], lts[k])

    g.set_tuning(pose_prefetch=0)
    iso = {}
    for kind in ("world", "rot", "rot+t"):
        for k in range(P):
            upload(kind, k); g.deform(); iso[(kind, k)] = g.read()
    assert g.get_tuning("effective_fuse_fk") == 1
    g.set_tuning(pose_prefetch=-1)

    def same(kind, k, what):
        p, n = g.read()
        assert np.array_equal(p, iso[(kind, k)][0]) and np.array_equal(n, iso[(kind, k)][1]), "%s: %s pose %d (%s)" % (what, kind, k, morphs)

    def gated(kind_a, a, kind_b, b):
        upload(kind_a, a); g.deform(); g.sync()
        try:
            g._chk(rzv.lib.rz_debug_gate(g._h, 1))
            upload(kind_a, a); g.deform()               # held by the gate; its helper will look at the slot the next upload lands in
            upload(kind_b, b)
        finally:
            g._chk(rzv.lib.rz_debug_gate(g._h, 0))
        staged = g.get_tuning("pose_staged")
        g.deform()
        same(kind_b, b, "gated %s -> %s" % (kind_a, kind_b))
        return staged

    for kind in ("rot", "rot+t", "world"):
        assert gated(kind, 0, kind, 1) == 1, "a %s pose behind a %s frame must be staged" % (kind, kind)
        assert gated(kind, 2, kind, 3) == 1
    for ka, kb in (("world", "rot"), ("rot", "world"), ("rot", "rot+t"), ("rot+t", "rot")):
        assert gated(ka, 1, kb, 4) == 0, "a helper of a %s frame staged a %s pose" % (ka, kb)
    for kind in ("rot", "rot+t"):                      # forced misses: the host is never ahead
        for k in (3, 0):
            g.sync(); upload(kind, k)
            assert g.get_tuning("pose_staged") == 0
            g.deform(); g.sync()
            same(kind, k, "pinned-slot pose")
            assert g.get_tuning("pose_resident") == 1
            g.deform_n(2); same(kind, k, "replay of the resident pose")
    checks = 0                                          # free-running mixture, copies in between (zero_copy = 0 starts a new epoch)
    for f in range(240):
        kind = ("world", "rot", "rot+t")[int(rng.integers(0, 3))]
        k = int(rng.integers(0, P))
        if rng.random() < 0.06:
            g.set_tuning(zero_copy=0); upload(kind, (k + 1) % P); g.deform(); g.set_tuning(zero_copy=-1)
        upload(kind, k); g.deform()
        if rng.random() < 0.3:
            g.deform_n(int(rng.integers(1, 4)))
        if f % 6 == 0:
            same(kind, k, "free-running frame %d" % f); checks += 1
    assert checks >= 40
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[2], mesh["inv_bind"],
                           None if not M else synth.sparse_to_dense(V, off, vi, d3), mws[2])
    assert_parity(iso[("world", 2)][0], iso[("world", 2)][1], pr, nr, "isolated world pose")
    g.close()


def test_shards_cut_at_256_vertices_equal_the_whole_mesh(rz, oracle):
    """rz_shard_range (round 4): shards of one mesh are multiples of 256 vertices (was 1024): 125 184 instead of 125 952 per GPU
    for C5. A shard's frame must hold the bits of the same vertices of the whole mesh's frame (dense and sparse morphs)."""
    V, B, M, G = 30000 + 257, 64, 10, 8
    b0, n0 = rz.shard_range(V, G, 0)
    assert n0 % 256 == 0 and n0 < (V + G - 1) // G + 256
    assert rz.shard_range(1000000, 8, 0)[1] == 125184 and rz.shard_range(1000000, 8, 7) == (876288, 123712)
    mesh = synth.make_mesh(V, B, seed=8)
    deltas, mw = synth.make_morphs_dense(V, M, seed=9)
    whole = rz.DeformContext(0)
    whole.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); whole.upload_skeleton(mesh["inv_bind"]); whole.upload_morphs_dense(deltas)
    whole.set_tuning(morph_split=4)
    whole.set_pose(mesh["world"], mw); whole.deform()
    pw, nw = whole.read()
    for r in (0, 3, G - 1):
        b, n = rz.shard_range(V, G, r)
        part, d = rz.shard.cut_mesh(mesh, deltas, b, n)
        c = rz.DeformContext(0)
        c.upload_mesh(part["pos"], part["nrm"], part["joints"], part["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(d)
        c.set_tuning(morph_split=4)
        c.set_pose(mesh["world"], mw); c.deform()
        pg, ng = c.read()
        assert np.array_equal(pg, pw[b:b + n]) and np.array_equal(ng, nw[b:b + n]), "shard %d of %d" % (r, G)
        c.close()
    whole.close()
