"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs (BASELINE.json configs C2-C5), plus size-independent properties at full size."""
import numpy as np
import pytest

from helpers import assert_hull, assert_parity
from reze_engine_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(rz):
    c = rz.DeformContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctxv(rzv):
    """a context of the all-variants build: the kernel variants the product does not ship (conftest.py: rzv)"""
    c = rzv.DeformContext(0)
    assert c.get_tuning("all_variants") == 1
    yield c
    c.close()


def _variant_only(**t):
    """does this tuning select a variant only the tools build carries?"""
    return t.get("geo_lds", 0) == 1 or t.get("nontemporal", 1) == 0 or t.get("unroll", 0) == 4 or t.get("inst_loop", -1) == 9


def run_gpu(ctx, mesh, world=None, deltas=None, mw=None, sparse=None, **tuning):
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    if deltas is not None:
        ctx.upload_morphs_dense(deltas)
    elif sparse is not None:
        ctx.upload_morphs_sparse(*sparse)
    ctx.set_instances(1)
    ctx.set_tuning(morph_split=0, unroll=0, nontemporal=1, nt_store=1, geo_lds=0, grid_cap=0, fast=-1, out_cap=-1)
    ctx.set_tuning(**tuning)
    ctx.set_pose(mesh["world"] if world is None else world, mw)
    ctx.deform()
    return ctx.read()


def test_c2_lbs_30k_200_bones(ctx, ctxv, oracle):
    """config 2: 30k verts / 200 bones / 0 morphs, fp32, vs the vs() restatement."""
    mesh = synth.make_mesh(30000, 200)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
    product = ctx
    for geo in (1, 0):
        ctx = ctxv if geo else product          # rest geometry through LDS is a tools-only variant
        for fast in (1, 0):
            pg, ng = run_gpu(ctx, mesh, geo_lds=geo, fast=fast)
            assert_parity(pg, ng, pr, nr, "C2 geo_lds=%d fast=%d" % (geo, fast))
            # palette (engine.ts:926-928 restatement, rows 0..2) is observable in both forms
            S = oracle.palette(mesh["world"], mesh["inv_bind"]).reshape(-1, 4, 4)      # [b, col, row]
            rows = np.transpose(S, (0, 2, 1))[:, :3, :].reshape(-1, 12)
            np.testing.assert_allclose(ctx.read_palette(), rows, rtol=1e-6, atol=1e-6)


def test_interleaved_upload_matches_soa_upload(ctx):
    """rz_upload_mesh takes the reference's 8-float interleaved vertex buffer (model.ts:196-200)."""
    mesh = synth.make_mesh(5000, 64, seed=11)
    pg, ng = run_gpu(ctx, mesh)
    inter = np.zeros((5000, 8), dtype=np.float32)
    inter[:, 0:3] = mesh["pos"]
    inter[:, 3:6] = mesh["nrm"]
    inter[:, 6:8] = 0.25
    ctx.upload_mesh_interleaved(inter, mesh["joints"], mesh["weights"])
    ctx.set_pose(mesh["world"])
    ctx.deform()
    p2, n2 = ctx.read()
    assert np.array_equal(pg, p2) and np.array_equal(ng, n2)


def test_identity_pose_known_answer(ctx):
    mesh = synth.make_mesh(10000, 100, seed=4)
    q = np.zeros((100, 4), dtype=np.float32)
    q[:, 3] = 1
    world = synth.fk_world(mesh["parents"], mesh["bind"], q)
    pg, ng = run_gpu(ctx, mesh, world=world)
    np.testing.assert_allclose(pg, mesh["pos"], rtol=1e-6, atol=2e-5)
    np.testing.assert_allclose(ng, mesh["nrm"], atol=1e-6)


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("split", [1, 2, 4, 8])
@pytest.mark.parametrize("unroll", [4, 8])
def test_c3_fused_morph_skin_all_kernel_variants(ctx, ctxv, oracle, split, unroll, fast):
    """config 3: 30k verts / 200 bones / 64 active morph targets, every morph-split x unroll variant,
    both as the one-launch frame (fast=1: palette fused, kernarg morph list) and with the prep kernel."""
    mesh = synth.make_mesh(30000, 200)
    deltas, mw = synth.make_morphs_dense(30000, 64)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=8)
    if unroll == 4:
        ctx = ctxv                              # 4 morphs in flight: tools-only variant
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, morph_split=split, unroll=unroll, fast=fast)
    assert_parity(pg, ng, pr, nr, "C3 S=%d U=%d fast=%d" % (split, unroll, fast))
    assert ctx.get_tuning("effective_split") == split
    assert ctx.get_tuning("effective_fast") == fast


@pytest.mark.parametrize("nt,nts,geo", [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 0, 0)])
def test_c3_load_path_variants(ctx, ctxv, oracle, nt, nts, geo):
    mesh = synth.make_mesh(30000, 200)
    deltas, mw = synth.make_morphs_dense(30000, 64)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=8)
    if _variant_only(nontemporal=nt, geo_lds=geo):
        ctx = ctxv
    for fast in (1, 0):
        pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, nontemporal=nt, nt_store=nts, geo_lds=geo, fast=fast)
        assert_parity(pg, ng, pr, nr, "C3 nt=%d nts=%d geo=%d fast=%d" % (nt, nts, geo, fast))


@pytest.mark.parametrize("V,B,M", [(1, 1, 1), (3, 2, 7), (1023, 17, 3), (1025, 300, 9), (4097, 471, 33)])
def test_ragged_sizes_and_partial_weights(ctx, oracle, V, B, M):
    """edge cases: single vertex, non-multiple-of-tile sizes, odd morph counts, zero weights mixed in."""
    mesh = synth.make_mesh(V, B, seed=V)
    deltas, mw = synth.make_morphs_dense(V, M, seed=M)
    mw[::3] = 0.0                                    # inactive morphs are skipped, not multiplied
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw)
    for split in (0, 1, 4):
        for fast in (1, 0):
            pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, morph_split=split, fast=fast)
            assert_parity(pg, ng, pr, nr, "ragged V=%d M=%d S=%d fast=%d" % (V, M, split, fast))


def test_more_active_morphs_than_kernel_arguments_hold(ctx, oracle):
    """> 128 active morphs cannot ride in the kernel arguments: the frame silently uses the prep kernel."""
    V, B, M = 6000, 33, 200
    mesh = synth.make_mesh(V, B, seed=51)
    deltas, mw = synth.make_morphs_dense(V, M, seed=52)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=4)
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw)
    assert ctx.get_tuning("effective_fast") == 0
    assert_parity(pg, ng, pr, nr, "M=200 active")
    mw[100:] = 0                                      # 100 active: back on the one-launch path
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=4)
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw)
    assert ctx.get_tuning("effective_fast") == 1
    assert_parity(pg, ng, pr, nr, "M=200, 100 active")


def test_large_skeleton_palette(ctx, oracle):
    """B > 256 exercises the multi-pass palette staging (demo models have 349 / 471 bones; stress 1500)."""
    for B in (471, 1500):
        mesh = synth.make_mesh(20000, B, seed=B)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
        for fast in (1, 0):
            pg, ng = run_gpu(ctx, mesh, fast=fast)
            assert_parity(pg, ng, pr, nr, "B=%d fast=%d" % (B, fast))


def test_all_morph_weights_zero_equals_plain_skin(ctx, oracle):
    mesh = synth.make_mesh(9000, 50, seed=8)
    deltas, mw = synth.make_morphs_dense(9000, 16)
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=np.zeros_like(mw))
    p0, n0 = run_gpu(ctx, mesh, deltas=deltas, mw=None)          # NULL weights == all zero
    ctx.upload_morphs_dense(None)
    ctx.set_pose(mesh["world"])
    ctx.deform()
    p1, n1 = ctx.read()
    assert np.array_equal(pg, p1) and np.array_equal(ng, n1)
    assert np.array_equal(p0, p1) and np.array_equal(n0, n1)


def test_degenerate_weights_and_normals(ctx, oracle):
    """zero weight sum takes the (1,0,0,0) branch (engine.ts:257); a zero normal stays the rest normal."""
    mesh = synth.make_mesh(2048, 30, seed=12)
    mesh["weights"][::5] = 0
    mesh["nrm"][::7] = 0
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
    pg, ng = run_gpu(ctx, mesh)
    assert_parity(pg, ng, pr, nr, "degenerate")
    assert np.array_equal(ng[::35], np.zeros_like(ng[::35]))


def test_sparse_morphs_match_oracle_and_dense_path(ctx, oracle):
    V, B, M = 28842, 349, 60                         # the demo model's shape
    mesh = synth.make_mesh(V, B, seed=21)
    off, idx, d3, mw = synth.make_morphs_sparse(V, M, density=0.02)
    mw[5] = 0
    pm = oracle.morph_sparse(V, off, idx, d3, mw, mesh["pos"])
    S = oracle.palette(mesh["world"], mesh["inv_bind"])
    pr, nr = oracle.skin(pm, mesh["nrm"], mesh["joints"], mesh["weights"], S)
    for fast in (1, 0):
        pg, ng = run_gpu(ctx, mesh, sparse=(off, idx, d3), mw=mw, fast=fast)
        assert_parity(pg, ng, pr, nr, "sparse fast=%d" % fast)
    pd, nd = run_gpu(ctx, mesh, deltas=synth.sparse_to_dense(V, off, idx, d3), mw=mw, morph_split=1)
    assert_parity(pd, nd, pr, nr, "dense expansion")


def test_sparse_morphs_concentrated_on_a_face_region(ctx, oracle):
    """The demo model's shape: every vertex morph is a facial expression over the same few hundred vertices, so some
    vertices carry dozens of entries and most carry none. The LDS-staged sparse row walk must match the oracle there
    too — on ragged sizes, with a grid cap that gives waves several tiles, with instances that have their own weights,
    and with out_cap on and off."""
    for V, B, M, region in ((28842, 349, 60, (5000, 700)), (1000, 16, 40, (700, 300)), (70003, 64, 33, (69000, 1003))):
        mesh = synth.make_mesh(V, B, seed=V + 1)
        off, idx, d3, mw = synth.make_morphs_sparse(V, M, density=600.0 / V, region=region, seed=V + 2)
        assert np.bincount(idx, minlength=V).max() >= 20
        pm = oracle.morph_sparse(V, off, idx, d3, mw, mesh["pos"])
        S = oracle.palette(mesh["world"], mesh["inv_bind"])
        pr, nr = oracle.skin(pm, mesh["nrm"], mesh["joints"], mesh["weights"], S)
        for fast, cap, oc in ((1, 0, -1), (0, 0, 0), (1, 16, -1), (1, 3, 0)):
            pg, ng = run_gpu(ctx, mesh, sparse=(off, idx, d3), mw=mw, fast=fast, grid_cap=cap, out_cap=oc)
            assert_parity(pg, ng, pr, nr, "face-concentrated sparse V=%d fast=%d cap=%d out_cap=%d" % (V, fast, cap, oc))
    # two instances, the second with only three active expressions
    mw2 = np.stack([mw, np.where(np.arange(M) < 3, mw, 0).astype(np.float32)])
    ctx.set_instances(2)
    ctx.set_pose(np.stack([mesh["world"], mesh["world"]]), mw2)
    ctx.deform()
    for i in range(2):
        pm = oracle.morph_sparse(V, off, idx, d3, mw2[i], mesh["pos"])
        pr, nr = oracle.skin(pm, mesh["nrm"], mesh["joints"], mesh["weights"], S)
        pg, ng = ctx.read(instance=i)
        assert_parity(pg, ng, pr, nr, "face-concentrated sparse, instance %d" % i)
    ctx.set_instances(1)


@pytest.mark.parametrize("cap", [0, 64, 192, 2048])
def test_lds_batched_output_stores(ctx, oracle, cap):
    """out_cap: outputs parked in a per-wave LDS buffer and flushed as 16-byte stores (full, partial and multi-flush runs)."""
    for V, B, M, split in ((30000, 200, 64, 0), (70001, 33, 5, 1), (5000, 10, 0, 0)):
        mesh = synth.make_mesh(V, B, seed=V)
        deltas, mw = synth.make_morphs_dense(V, M, seed=7) if M else (None, None)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw, threads=8)
        for nts in (0, 1):
            pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, morph_split=split, out_cap=cap, nt_store=nts, geo_lds=0)
            assert_parity(pg, ng, pr, nr, "out_cap=%d V=%d nts=%d" % (cap, V, nts))


def test_sparse_morph_duplicates_and_out_of_range_entries(ctx, oracle):
    """PMX files may list a vertex twice inside one morph (both offsets add, in file order) and, when damaged, indices
    past the mesh (ignored): the per-vertex CSR built at upload must keep the oracle's accumulation order."""
    V, B = 3000, 12
    mesh = synth.make_mesh(V, B, seed=81)
    rng = np.random.default_rng(82)
    off = np.array([0, 5, 5, 9], dtype=np.uint32)                       # morph 1 is empty
    idx = np.array([7, 7, 2999, 5000, 7, 0, 7, 1, 4000000], dtype=np.uint32)   # duplicates of 7, two out of range
    d3 = rng.normal(size=(9, 3)).astype(np.float32)
    mw = np.array([0.5, 1.0, -0.25], dtype=np.float32)
    pm = oracle.morph_sparse(V, off, idx, d3, mw, mesh["pos"])
    assert not np.array_equal(pm[7], mesh["pos"][7])
    S = oracle.palette(mesh["world"], mesh["inv_bind"])
    pr, nr = oracle.skin(pm, mesh["nrm"], mesh["joints"], mesh["weights"], S)
    for fast in (1, 0):
        pg, ng = run_gpu(ctx, mesh, sparse=(off, idx, d3), mw=mw, fast=fast)
        assert_parity(pg, ng, pr, nr, "sparse duplicates fast=%d" % fast)


def test_larger_than_c5_no_32bit_overflow(ctx, oracle):
    """3 M vertices x 40 dense morphs = 1.44 GB of morph planes: byte offsets pass 2^32 inside one buffer."""
    V, B, M = 3000000, 64, 40
    mesh = synth.make_mesh(V, B, seed=91)
    deltas, mw = synth.make_morphs_dense(V, M, seed=92)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"],
                           deltas, mw, threads=32)
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw)
    assert_parity(pg, ng, pr, nr, "3M x 40")
    assert_parity(pg[-5000:], ng[-5000:], pr[-5000:], nr[-5000:], "3M x 40 tail")


def test_c4_instances_each_match_their_own_pose(ctx, oracle):
    """config 4 (reduced instance count for the oracle): per-instance palette, shared static mesh."""
    V, B, I = 30000, 200, 6
    mesh = synth.make_mesh(V, B)
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=100 + i) for i in range(I)])
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    ctx.set_instances(I)
    ctx.set_pose(worlds)
    ctx.deform()
    for i in range(I):
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"])
        pg, ng = ctx.read(instance=i)
        assert_parity(pg, ng, pr, nr, "instance %d" % i)
    ctx.set_instances(1)


@pytest.mark.parametrize("fast", [-1, 0, 1])
@pytest.mark.parametrize("inst_loop", [-1, 0, 3, 9])
def test_c4_instance_loop_kernel_and_ragged_groups(ctx, ctxv, oracle, inst_loop, fast):
    """19 poses do not divide into groups of 8: the pose-loop kernel (inst_loop != 0) and the generic kernel
    (inst_loop = 0) must both match every pose; 471 bones forces a smaller group (LDS). fast = 1 makes the pose-group
    kernel form its palettes itself (world matrices staged in LDS, converted in place) instead of reading
    rz_prep_kernel's output."""
    if inst_loop == 9:
        ctx = ctxv                              # the register-resident crowd kernel: tools-only variant
    for V, B, I in ((7001, 64, 19), (3000, 471, 5)):
        mesh = synth.make_mesh(V, B, seed=V)
        worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=300 + i) for i in range(I)])
        ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
        ctx.upload_skeleton(mesh["inv_bind"])
        ctx.upload_morphs_dense(None)
        ctx.set_instances(I)
        ctx.set_tuning(inst_loop=inst_loop, grid_cap=0, nt_store=-1, fast=fast)
        ctx.set_pose(worlds)
        ctx.deform()
        g = ctx.get_tuning("effective_inst_group")
        pp = ctx.get_tuning("effective_poses_per_wg")
        if inst_loop == 9:
            assert pp >= 1 and g == 0                       # register-resident form
        elif inst_loop == 0:
            assert pp == 0 and g == 0                       # generic kernel, one pose per workgroup row
        else:
            assert pp == 0 and 2 <= g <= 8                  # LDS pose-group form (whole palette or bone subsets)
        for i in range(I):
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"])
            pg, ng = ctx.read(instance=i)
            assert_parity(pg, ng, pr, nr, "V=%d B=%d instance %d loop=%d fast=%d" % (V, B, i, inst_loop, fast))
            S = oracle.palette(worlds[i], mesh["inv_bind"]).reshape(-1, 4, 4)
            rows = np.transpose(S, (0, 2, 1))[:, :3, :].reshape(-1, 12)
            np.testing.assert_allclose(ctx.read_palette(i), rows, rtol=1e-6, atol=1e-6)
    ctx.set_tuning(inst_loop=-1, fast=-1)
    ctx.set_instances(1)


def test_instanced_morph_weights_are_per_instance(ctx, oracle):
    V, B, M, I = 5000, 40, 8, 3
    mesh = synth.make_mesh(V, B, seed=31)
    deltas, _ = synth.make_morphs_dense(V, M, seed=32)
    rng = np.random.default_rng(33)
    mws = rng.random((I, M), dtype=np.float32)
    mws[1] = 0
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=200 + i) for i in range(I)])
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(deltas)
    ctx.set_instances(I)
    ctx.set_pose(worlds, mws)
    ctx.deform()
    for i in range(I):
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i],
                               mesh["inv_bind"], deltas, mws[i])
        pg, ng = ctx.read(instance=i)
        assert_parity(pg, ng, pr, nr, "instance %d" % i)
    ctx.set_instances(1)


def test_c5_full_size_parity_and_properties(ctx, oracle, rz):
    """config 5 at full size on one GPU: 1M verts / 256 bones / 64 dense morphs.
    (a) direct parity with the threaded C oracle; (b) vertex shards computed independently are
    bit-identical to the single-context result (the multi-GPU partition);
    (c) identity-pose + zero-weight property: output == rest mesh."""
    V, B, M = 1000000, 256, 64
    mesh = synth.make_mesh(V, B)
    deltas, mw = synth.make_morphs_dense(V, M)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=16)
    pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, morph_split=1)
    ep, en = assert_parity(pg, ng, pr, nr, "C5 full")
    print("C5 full-size parity: max rel pos err %.3e, max nrm err %.3e" % (ep, en))
    # (a') the DEFAULT plan — a fresh context nobody tuned: the very kernel the driver's bench line names — against the oracle too
    d = rz.DeformContext(0)
    d.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); d.upload_skeleton(mesh["inv_bind"]); d.upload_morphs_dense(deltas)
    d.set_pose(mesh["world"], mw)
    assert d.kernel_name() == "rz_deform_dense_kernel<2, 8, true, true, false, true, 3>", d.kernel_name()
    d.deform()
    pd, nd = d.read()
    d.close()
    epd, end = assert_parity(pd, nd, pr, nr, "C5 full, default plan")
    print("C5 full-size parity, default plan: max rel pos err %.3e, max nrm err %.3e" % (epd, end))
    # (b) shard 3 of 8 with the same kernel variant
    b, n = rz.shard_range(V, 8, 3)
    shard = {k: mesh[k][b:b + n] for k in ("pos", "nrm", "joints", "weights")}
    shard.update(inv_bind=mesh["inv_bind"], world=mesh["world"])
    ps, ns = run_gpu(ctx, shard, deltas=np.ascontiguousarray(deltas[:, b:b + n]), mw=mw, morph_split=1)
    assert np.array_equal(ps, pg[b:b + n]) and np.array_equal(ns, ng[b:b + n])
    # (c) identity pose, zero morph weights
    q = np.zeros((B, 4), dtype=np.float32)
    q[:, 3] = 1
    world = synth.fk_world(mesh["parents"], mesh["bind"], q)
    pi, ni = run_gpu(ctx, mesh, world=world, deltas=deltas, mw=np.zeros(M, dtype=np.float32))
    np.testing.assert_allclose(pi, mesh["pos"], rtol=1e-6, atol=2e-5)


@pytest.mark.parametrize("overlap", [0, 1])
def test_pose_upload_pipeline_never_serves_a_stale_or_torn_pose(rz, oracle, overlap):
    """Per-frame inputs are double-buffered and large uploads ride a second stream (pinned 4-slot ring, ev_up / ev_free
    hand-off). Hammer it: 400 frames cycling through four crowd poses — world matrices, local rotations, back and forth,
    and poses sampled on the device, with and without a deform between two uploads — and check, whenever a frame is read back, that it is exactly the
    frame of the pose that was set last (bit-identical to the same pose computed in isolation)."""
    V, B, I = 6000, 160, 48                           # 48 x 160 x 64 B = 480 KB per world upload: the piped path
    mesh = synth.make_mesh(V, B, seed=5)
    rng = np.random.default_rng(6)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(I)
    # overlap = 1: the opt-in protocol where the front kernels (prep / FK / sampling) of a frame run on the upload stream,
    # into the other slot of a 2-slot ring, while the skin kernel of the frame before is still reading its slot
    c.set_tuning(overlap=overlap, fast=0 if overlap else -1)
    assert c.get_tuning("effective_overlap") == overlap
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    poses = []
    for k in range(4):
        q = rng.normal(size=(I, B, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        w = np.stack([synth.fk_world(mesh["parents"], mesh["bind"], q[i]) for i in (0, I - 1)])
        poses.append((q, w))
    worlds = [np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=40 + 7 * k + i) for i in range(I)]) for k in range(4)]
    # isolated results: instance 0 and I-1 of every pose, both kinds
    iso = {}
    for k in range(4):
        c.set_pose(worlds[k]); c.deform(); iso[("w", k)] = (c.read(instance=0), c.read(instance=I - 1))
        c.set_pose_local(poses[k][0]); c.deform(); iso[("l", k)] = (c.read(instance=0), c.read(instance=I - 1))
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[2][I - 1], mesh["inv_bind"])
    assert_parity(iso[("w", 2)][1][0], iso[("w", 2)][1][1], pr, nr, "isolated world pose")
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], poses[1][1][0], mesh["inv_bind"])
    assert_parity(iso[("l", 1)][0][0], iso[("l", 1)][0][1], pr, nr, "isolated local pose")
    # a third kind: poses sampled on the device from an uploaded motion (writes the current slot's world matrices itself)
    nk = 4
    kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
    kq /= np.linalg.norm(kq, axis=2, keepdims=True)
    c.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.3)
    frames = [rng.random(I).astype(np.float32) * 30 for _ in range(4)]
    for k in range(4):
        c.set_pose_sampled(frames[k]); c.deform(); iso[("s", k)] = (c.read(instance=0), c.read(instance=I - 1))
    assert not np.array_equal(iso[("s", 0)][0][0], iso[("s", 1)][0][0])
    checks = 0
    for f in range(400):
        kind = "wls"[int(rng.integers(0, 3))]
        k = int(rng.integers(0, 4))
        if rng.random() < 0.2:                         # an upload that is overwritten before any frame consumes it
            c.set_pose(worlds[(k + 1) % 4]) if rng.random() < 0.5 else c.set_pose_local(poses[(k + 2) % 4][0])
        if kind == "w":
            c.set_pose(worlds[k])
        elif kind == "l":
            c.set_pose_local(poses[k][0])
        else:
            c.set_pose_sampled(frames[k])
        c.deform()
        if f % 7 == 0 or f > 390:
            for inst, want in ((0, iso[(kind, k)][0]), (I - 1, iso[(kind, k)][1])):
                got = c.read(instance=inst)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), "frame %d (%s%d) instance %d" % (f, kind, k, inst)
            checks += 1
    assert checks > 40
    c.close()


def test_device_fk_local_translations_and_append_move(rz, oracle):
    """Row f1 x f2: the GPU hierarchy solve with VMD bone translations (SkeletonRuntime.localTranslations) and bones that
    append their append parent's rotation and movement (model.ts:355-393; clamped ratio for the rotation, raw ratio for the
    move), three poses at once, against a float64 restatement; the deformed mesh against the oracle fed with those
    world matrices. Parents deliberately come AFTER some children (the reference solves recursively)."""
    from helpers import fk_reference
    V, B, I = 3000, 90, 3
    mesh = synth.make_mesh(V, B, seed=71)
    rng = np.random.default_rng(72)
    perm = rng.permutation(B)                      # shuffle bone ids so parents are not always earlier
    inv = np.argsort(perm)
    parents = np.array([(-1 if mesh["parents"][inv[j]] < 0 else perm[mesh["parents"][inv[j]]]) for j in range(B)], dtype=np.int32)
    bind = mesh["bind"][inv]
    quats = rng.normal(size=(I, B, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=2, keepdims=True)
    trans = (rng.random((I, B, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(1.5)
    ap = np.full(B, -1, dtype=np.int32)
    ratio = np.ones(B, dtype=np.float32)
    move = np.zeros(B, dtype=np.uint8)
    for k, b in enumerate(rng.choice(B, size=12, replace=False)):
        ap[b] = int(rng.integers(0, B))
        ratio[b] = [0.5, -0.75, 1.5, 1e-7, -2.0, 1.0][k % 6]
        move[b] = k % 2
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ib = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (B, 1))
    ib[:, 12:15] = -rng.random((B, 3), dtype=np.float32)
    c.upload_skeleton(ib)
    c.set_instances(I)
    c.upload_skeleton_topology(parents, bind, ap, ratio, move)
    for with_t in (True, False):
        c.set_pose_local(quats, local_translations=trans if with_t else None)
        c.deform()
        for i in range(I):
            ref = fk_reference(parents, bind, quats[i], trans[i] if with_t else None, ap, ratio, move)
            got = c.read_world(i)
            scale = np.maximum(1.0, np.abs(ref).max())
            assert np.abs(got - ref).max() <= 3e-5 * scale, "world matrices, pose %d translations=%s: %g" % (i, with_t, np.abs(got - ref).max())
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], got, ib)
            pg, ng = c.read(instance=i)
            assert_parity(pg, ng, pr, nr, "device FK with translations, pose %d" % i)
    # the two runs must differ (translations were really applied)
    assert np.abs(fk_reference(parents, bind, quats[0], trans[0], ap, ratio, move) - fk_reference(parents, bind, quats[0], None, ap, ratio, move)).max() > 0.1
    c.close()


def test_engine_device_sampling_matches_host_sampling_through_napi(tmp_path):
    """new Engine(null, { deviceFK, deviceSampling }).seekFrame(f) — motion flattened by host/vmd-sampler.js, sampled,
    solved and deformed on the GPU from one float — against the host sampler + host FK engine on the same synthetic
    PMX + VMD (bone translations, custom interpolation curves, a group morph, an append-move bone), one and two shards."""
    import json
    import shutil
    import subprocess
    import os
    from pmx_synth import write_pmx, write_vmd
    if shutil.which("node") is None:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "m.pmx").write_bytes(write_pmx())
    s = 0.38268343
    curve = bytes([10, 30, 50, 100, 90, 70, 5, 20, 60, 80, 100, 127, 120, 40, 110, 64]) + bytes(48)
    (tmp_path / "a.vmd").write_bytes(write_vmd(
        [("bone1", 0, (0, 0, s, 0.92387953)), ("bone1", 15, (0, s, 0, 0.92387953), (0, 0, 0), curve), ("bone1", 30, (s, 0, 0, 0.92387953)),
         ("bone3", 0, (s, 0, 0, 0.92387953), (0.3, -0.2, 0.1)), ("bone3", 20, (0, 0, s, 0.92387953), (-0.5, 0.4, 0.25), curve),
         ("bone0", 0, (0, 0, 0, 1), (0, 0.5, 0)), ("bone0", 25, (0, 0, 0, 1), (1.0, 0.25, -0.5), curve), ("bone20", 30, (0, 0, -s, 0.92387953)),
         ("nosuchbone", 5, (0, 0, 0, 1))],
        [("v1", 0, 0.8), ("v1", 20, 0.1), ("v2", 6, 0.4), ("grp", 0, 0.0), ("grp", 30, 1.0), ("blink", 10, 0.5), ("twist", 0, 0.2), ("twist", 30, 0.9)]))
    for layout, devs in (("sparse", "0"), ("dense", "0,0")):
        out = tmp_path / (layout + devs.replace(",", "_"))
        out.mkdir()
        subprocess.check_call(["node", os.path.join(root, "tests", "js", "sampled_e2e.js"), str(tmp_path / "m.pmx"), str(tmp_path / "a.vmd"),
                               str(out), layout, devs], timeout=300)
        frames = json.load(open(str(out / "frames.json")))
        rd = lambda f: np.fromfile(str(out / f), dtype=np.float32)  # noqa: E731
        moved = 0.0
        for i in range(len(frames)):
            wa, wb = rd("a_world_%d.f32" % i).reshape(-1, 16), rd("b_world_%d.f32" % i).reshape(-1, 16)
            assert np.abs(wa - wb).max() <= 5e-5 * max(1.0, np.abs(wa).max()), "world matrices at frame %g: %g" % (frames[i], np.abs(wa - wb).max())
            pa, na = rd("a_pos_%d.f32" % i).reshape(-1, 3), rd("a_nrm_%d.f32" % i).reshape(-1, 3)
            pb, nb = rd("b_pos_%d.f32" % i).reshape(-1, 3), rd("b_nrm_%d.f32" % i).reshape(-1, 3)
            assert_parity(pb, nb, pa, na, "device sampling vs host sampling, %s %s frame %g" % (layout, devs, frames[i]))
            moved = max(moved, float(np.abs(pa - rd("a_pos_0.f32").reshape(-1, 3)).max()))
            assert (rd("a_mw_%d.f32" % i) != 0).sum() >= 1
        assert moved > 0.2                     # the motion really moves the mesh between the sampled frames
        if devs == "0":                        # the crowd: instance k at frames[i] must equal the single-instance result
            for i in (1, 2, 4):
                pc, nc = rd("crowd_pos_%d.f32" % i).reshape(-1, 3), rd("crowd_nrm_%d.f32" % i).reshape(-1, 3)
                assert_parity(pc, nc, rd("a_pos_%d.f32" % i).reshape(-1, 3), rd("a_nrm_%d.f32" % i).reshape(-1, 3), "crowd instance at frame %g" % frames[i])


def test_device_motion_sampling_matches_the_float64_sampler(rz, oracle):
    """rz_upload_animation + rz_set_pose_sampled: MMD sampling (Bezier-warped slerp / lerp, linear morph keys, group
    feeds) + hierarchy solve + morph + skin all on the GPU from ONE float per instance. Every instance sits at its own
    frame — before the first key, on a key, between keys with linear, default and sharp curves, past the last key."""
    from helpers import fk_reference, sample_reference
    V, B, M, I = 4000, 48, 6, 7
    mesh = synth.make_mesh(V, B, seed=31)
    deltas, _ = synth.make_morphs_dense(V, M, seed=32)
    rng = np.random.default_rng(33)
    tracked = rng.choice(B, size=30, replace=False)
    track_bone = np.concatenate([tracked, [B + 5]]).astype(np.int32)        # one track for a bone this model lacks
    key_off, kfs, rots, poss, ips = [0], [], [], [], []
    for tr in range(len(track_bone)):
        nk = int(rng.integers(1, 6))
        frames = np.sort(rng.choice(np.arange(0, 60), size=nk, replace=False)).astype(np.float32)
        q = rng.normal(size=(nk, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        if nk > 2:
            q[2] = q[1] + 1e-3 * rng.normal(size=4); q[2] /= np.linalg.norm(q[2])    # nearly equal keys: the lerp branch
        kfs.append(frames); rots.append(q); poss.append(rng.normal(size=(nk, 3)) * 0.5)
        ip = np.tile(np.array([20, 20, 20, 20, 20, 20, 20, 20, 107, 107, 107, 107, 107, 107, 107, 107], dtype=np.uint8), (nk, 1))
        for k in range(nk):
            if rng.random() < 0.6:
                ip[k] = rng.integers(0, 128, size=16)
        ips.append(ip)
        key_off.append(key_off[-1] + nk)
    anim = dict(track_bone=track_bone, key_off=np.array(key_off, np.uint32), key_frame=np.concatenate(kfs), key_rot=np.concatenate(rots).astype(np.float32),
                key_pos=np.concatenate(poss).astype(np.float32), key_interp=np.concatenate(ips))
    # morph tracks: 0..3 drive vertex morphs 0..3, track 4 is a group feeding morphs 1 and 5, morph 4 is never keyed
    mk = [np.array([0, 10, 40], np.float32), np.array([5], np.float32), np.array([0, 30], np.float32), np.array([2, 3, 50], np.float32), np.array([0, 20], np.float32)]
    mwk = [rng.random(len(k)).astype(np.float32) for k in mk]
    anim.update(mkey_off=np.cumsum([0] + [len(k) for k in mk]).astype(np.uint32), mkey_frame=np.concatenate(mk), mkey_weight=np.concatenate(mwk))
    feeds = [[(0, 1.0)], [(1, 1.0), (4, 0.5)], [(2, 1.0)], [(3, 1.0)], [], [(4, -0.25)]]
    anim.update(feed_off=np.cumsum([0] + [len(f) for f in feeds]).astype(np.uint32), feed_track=np.array([t for f in feeds for t, _ in f], np.int32),
                feed_ratio=np.array([r for f in feeds for _, r in f], np.float32))
    ap = np.full(B, -1, dtype=np.int32); ap[7] = 3; ap[20] = 11
    ratio = np.ones(B, dtype=np.float32); ratio[7] = 0.5; ratio[20] = -1.0
    move = np.zeros(B, dtype=np.uint8); move[20] = 1
    frames = np.array([-3.0, 0.0, 7.25, 19.0, 33.5, 58.999, 400.0], dtype=np.float32)
    for morphs in ("dense", "none"):
        c = rz.DeformContext(0)
        c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
        c.upload_skeleton(mesh["inv_bind"])
        if morphs == "dense":
            c.upload_morphs_dense(deltas)
        Mq = M if morphs == "dense" else 0
        c.set_instances(I)
        c.upload_skeleton_topology(mesh["parents"], mesh["bind"], ap, ratio, move)
        with pytest.raises(rz.RzError):
            c.set_pose_sampled(frames)                                    # no motion uploaded yet
        a = dict(anim)
        if Mq == 0:
            for k in ("mkey_off", "mkey_frame", "mkey_weight", "feed_off", "feed_track", "feed_ratio"):
                a[k] = None
        c.upload_animation(a["track_bone"], a["key_off"], a["key_frame"], a["key_rot"], a["key_pos"], a["key_interp"],
                           a["mkey_off"], a["mkey_frame"], a["mkey_weight"], a["feed_off"], a["feed_track"], a["feed_ratio"])
        c.set_pose_sampled(frames)
        c.deform()
        for i in range(I):
            q, t, w = sample_reference(a, float(frames[i]), B, Mq)
            ref = fk_reference(mesh["parents"], mesh["bind"], q, t, ap, ratio, move)
            got = c.read_world(i)
            assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), "world, instance %d frame %g: %g" % (i, frames[i], np.abs(got - ref).max())
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], got, mesh["inv_bind"],
                                   deltas if Mq else None, w.astype(np.float32) if Mq else None)
            pg, ng = c.read(instance=i)
            assert_parity(pg, ng, pr, nr, "sampled pose, %s morphs, instance %d" % (morphs, i))
        # a host-supplied pose takes over again
        c.set_pose(np.stack([mesh["world"]] * I), None)
        c.deform()
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
        pg, ng = c.read(instance=I - 1)
        assert_parity(pg, ng, pr, nr, "back to a host pose")
        c.close()


def test_device_fk_matches_host_fk_and_reference_fixture(ctx, oracle):
    """Row f1: Model.computeWorldMatrices (model.ts:330-420) on the GPU. (a) synthetic tree, 5 poses at once, against
    the host-order FK twin; (b) the REAL 349-bone skeleton with its 26 append-rotation bones and pool.vmd frame 0,
    against the world matrices the reference's own code produced (tests/golden/ref_c1_pose0.npz)."""
    import os
    V, B, I = 4000, 120, 5
    mesh = synth.make_mesh(V, B, seed=61)
    rng = np.random.default_rng(62)
    quats = rng.normal(size=(I, B, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=2, keepdims=True)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(None)
    ctx.set_instances(I)
    ctx.set_tuning(inst_loop=-1, fast=-1, grid_cap=0)
    ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
    ctx.set_pose_local(quats)
    ctx.deform()
    for i in range(I):
        world = synth.fk_world(mesh["parents"], mesh["bind"], quats[i])
        np.testing.assert_allclose(ctx.read_world(i), world, rtol=2e-5, atol=2e-5)
        pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world, mesh["inv_bind"])
        pg, ng = ctx.read(instance=i)
        assert_parity(pg, ng, pr, nr, "device FK instance %d" % i)
    ctx.set_instances(1)
    # (b) reference fixture
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_c1_pose0.npz"))
    B = len(g["parents"])
    m2 = synth.make_mesh(6000, B, seed=63)
    ctx.upload_mesh(m2["pos"], m2["nrm"], m2["joints"], m2["weights"])
    ctx.upload_skeleton(g["inv_bind"])
    ap = np.where(g["append_rotate"], g["append_parent"], -1)
    ctx.upload_skeleton_topology(g["parents"], g["bind"], ap, g["append_ratio"])
    for key_q, key_w in (("local_rot_pose0", "world_pose0"), ("local_rot_tween150", "world_tween150")):
        ctx.set_pose_local(g[key_q])
        ctx.deform()
        np.testing.assert_allclose(ctx.read_world(0), g[key_w], rtol=3e-5, atol=3e-5)
        pr, nr = oracle.deform(m2["pos"], m2["nrm"], m2["joints"], m2["weights"], g[key_w], g["inv_bind"])
        pg, ng = ctx.read()
        assert_parity(pg, ng, pr, nr, "device FK on the reference skeleton (%s)" % key_q)
    with pytest.raises(Exception):
        ctx.upload_skeleton_topology(np.array([1, 0] + [0] * (B - 2)), g["bind"])      # 0 <-> 1 cycle


def test_fused_outline_hull_and_bounding_box(ctx, oracle):
    """Row f4: consumers of the deformed mesh fused into the skin kernel — the outline pass's inverted hull
    (engine.ts:458-461) and a per-frame AABB — single pose with morphs, then 3 instances."""
    V, B, M = 9001, 77, 6
    mesh = synth.make_mesh(V, B, seed=71)
    deltas, mw = synth.make_morphs_dense(V, M, seed=72)
    edge = np.random.default_rng(73).uniform(0.0, 1.5, size=V).astype(np.float32)
    edge[::4] = 0.0                                        # materials without the edge flag
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    for split in (1, 4):
        pg, ng = run_gpu(ctx, mesh, deltas=deltas, mw=mw, morph_split=split)
        ctx.upload_edge_scale(edge)
        ctx.enable_aabb(True)
        for _ in range(3):                                 # the box double-buffers across frames: stay consistent
            ctx.deform()
        pg, ng = ctx.read()
        hg = ctx.read_hull()
        assert_parity(pg, ng, pr, nr, "with epilogues S=%d" % split)
        href = oracle.hull(pr, nr, edge)
        assert_hull(hg, href, "hull S=%d" % split)
        assert np.array_equal(hg[::4], pg[::4])            # edge 0 => hull == position
        box = ctx.read_aabb()
        assert np.array_equal(box[:3], pg.min(axis=0)) and np.array_equal(box[3:], pg.max(axis=0))   # exact on the GPU's own output
        np.testing.assert_allclose(box, np.r_[pr.min(axis=0), pr.max(axis=0)], rtol=1e-5, atol=1e-4)
    # instanced: epilogues force the generic kernel; every instance gets its own box
    I = 3
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=400 + i) for i in range(I)])
    ctx.upload_morphs_dense(None)
    ctx.set_instances(I)
    ctx.set_pose(worlds)
    ctx.deform()
    assert ctx.get_tuning("effective_inst_group") == 0
    for i in range(I):
        pg, ng = ctx.read(instance=i)
        box = ctx.read_aabb(i)
        assert np.array_equal(box[:3], pg.min(axis=0)) and np.array_equal(box[3:], pg.max(axis=0))
        p_i, n_i = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"])
        assert_hull(ctx.read_hull(i), oracle.hull(p_i, n_i, edge), "hull instance %d" % i)
    ctx.upload_edge_scale(None)
    ctx.enable_aabb(False)
    ctx.set_instances(1)


def test_single_process_comm_init_all_one_rank(rz):
    """ncclCommInitAll / grouped all-gather entry points (one Node process driving several GPUs) with one GPU."""
    mesh = synth.make_mesh(3000, 10, seed=43)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.set_pose(mesh["world"])
    c.deform()
    pg, ng = c.read()
    rz.capi.comm_init_all([c], 3000)
    rz.capi.allgather_all([c], with_normals=True)
    p2, n2 = c.read_gathered()
    assert np.array_equal(p2, pg) and np.array_equal(n2, ng)
    c2 = rz.DeformContext(0)
    c2.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    with pytest.raises(rz.RzError):
        rz.capi.comm_init_all([c, c2], 6000)          # two ranks on one GPU are refused, with a message
    c.close()
    c2.close()


def test_allgather_single_rank_roundtrip(ctx, rz, oracle):
    """RCCL all-gather entry points with a world of 1: gathered buffer == local result."""
    mesh = synth.make_mesh(5000, 20, seed=41)
    pg, ng = run_gpu(ctx, mesh)
    uid = rz.capi.comm_unique_id()
    ctx.comm_init(1, 0, uid, 5000)
    ctx.allgather(with_normals=True)
    p2, n2 = ctx.read_gathered()
    assert np.array_equal(p2, pg) and np.array_equal(n2, ng)


def test_autotune_keeps_parity_and_picks_a_listed_plan(rz, oracle):
    """rz_autotune times the candidate launch shapes and keeps one: the result must be one of the candidates, the
    frame must still match the oracle, and 0 / -1 hands the keys back to the heuristics. Dense, morph-free and
    instanced frames."""
    ctx = rz.DeformContext(0)
    V, B, M = 50000, 64, 16
    mesh = synth.make_mesh(V, B, seed=21)
    deltas, mw = synth.make_morphs_dense(V, M, seed=22)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    run_gpu(ctx, mesh, deltas=deltas, mw=mw)
    got = ctx.autotune(10)
    # (the keys stay 0 / 0 / -1 when the search keeps the heuristic plan: rz_autotune_pick wants a clear win)
    assert got["effective_split"] in (1, 2, 4, 8) and ctx.get_tuning("morph_split") in (0, got["effective_split"])
    assert ctx.get_tuning("grid_cap") in (0, 256, 512, 1024)
    assert (ctx.get_tuning("morph_split") == 0) == (ctx.get_tuning("grid_cap") == 0)
    ctx.deform()
    pg, ng = ctx.read()
    assert_parity(pg, ng, pr, nr, "after autotune (dense)")
    # the searched shape belongs to the workload it was timed on: new morph targets hand the keys back to the heuristics
    ctx.upload_morphs_dense(deltas[:4])
    assert ctx.get_tuning("morph_split") == 0 and ctx.get_tuning("grid_cap") == 0
    # morph-free: the split only sets the wave-step size there (256 or 64 vertices)
    pr0, nr0 = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"])
    run_gpu(ctx, mesh)
    got = ctx.autotune()
    assert got["effective_split"] in (1, 4)
    ctx.deform()
    pg, ng = ctx.read()
    assert_parity(pg, ng, pr0, nr0, "after autotune (no morphs)")
    ctx.set_tuning(morph_split=0, grid_cap=0)
    # instanced: poses per workgroup x workgroups per CU
    I = 12
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=900 + i) for i in range(I)])
    ctx.set_instances(I)
    ctx.set_pose(worlds)
    got = ctx.autotune(5)
    assert got["effective_inst_group"] in (4, 8) and ctx.get_tuning("inst_loop") in (-1, 4, 8)
    ctx.deform()
    for i in (0, 5, 11):
        pri, nri = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[i], mesh["inv_bind"])
        pg, ng = ctx.read(instance=i)
        assert_parity(pg, ng, pri, nri, "after autotune (instance %d)" % i)
    ctx.close()


def test_graph_replay_is_the_same_frames(rz, oracle):
    """"graph" tuning key: rz_deform_n replays hipGraphs of 16 captured frames. Same output bits as plain launches for a
    one-launch frame, a crowd (prep + skin), a device-FK crowd and a frame with the fused consumers; the capture is redone
    when the pose, the tuning or the mesh changes."""
    V, B = 9000, 48
    mesh = synth.make_mesh(V, B, seed=14)
    deltas, mw = synth.make_morphs_dense(V, 10, seed=15)
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    c.upload_skeleton(mesh["inv_bind"])
    c.upload_morphs_dense(deltas)
    c.upload_skeleton_topology(mesh["parents"], mesh["bind"])

    def both(setup, inst=0):
        outs = []
        for g in (0, 1):
            c.set_tuning(graph=g)
            setup()
            c.deform_n(70)                         # 1 plain + 4 graphs + 5 plain frames when graph = 1
            outs.append(c.read(instance=inst))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        return outs[1]

    pg, ng = both(lambda: c.set_pose(mesh["world"], mw))
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    assert_parity(pg, ng, pr, nr, "graph replay, one-launch frame")
    world2 = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=77)
    pg, ng = both(lambda: c.set_pose(world2, mw))                       # a new pose: the captured kernel arguments are stale
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world2, mesh["inv_bind"], deltas, mw)
    assert_parity(pg, ng, pr, nr, "graph replay after a new pose")
    c.enable_aabb(True); c.upload_edge_scale(np.full(V, 0.5, np.float32))
    pg, ng = both(lambda: c.set_pose(world2, mw))
    bb = c.read_aabb()
    assert np.abs(bb[:3] - pg.min(axis=0)).max() <= 1e-5 and np.abs(bb[3:] - pg.max(axis=0)).max() <= 1e-5
    assert_hull(c.read_hull(), oracle.hull(pr, nr, np.full(V, 0.5, np.float32)), "hull under graph replay")
    c.enable_aabb(False); c.upload_edge_scale(None); c.upload_morphs_dense(None)
    I = 9
    c.set_instances(I)
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=200 + i) for i in range(I)])
    pg, ng = both(lambda: c.set_pose(worlds), inst=I - 1)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[I - 1], mesh["inv_bind"])
    assert_parity(pg, ng, pr, nr, "graph replay, crowd")
    q = np.random.default_rng(3).normal(size=(I, B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    pg, ng = both(lambda: c.set_pose_local(q), inst=4)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], c.read_world(4), mesh["inv_bind"])
    assert_parity(pg, ng, pr, nr, "graph replay, device FK crowd")
    c.close()


def test_peer_direct_gather_three_shards(rz, oracle):
    """rz_gather_direct: three contexts (three vertex shards; one GPU here, so no peer mapping but the same
    pointers-into-the-root's-buffer mechanism) store their frames straight into the root's gathered arrays. The
    gathered mesh must equal the oracle's whole-mesh result bit for bit with what each shard reads back itself;
    re-uploading a mesh returns that context to its private output buffers."""
    V, B, M, G, root = 10007, 40, 5, 3, 1
    mesh = synth.make_mesh(V, B, seed=77)
    deltas, mw = synth.make_morphs_dense(V, M, seed=78)
    pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
    ctxs = []
    for r in range(G):
        b, n, _ = rz.shard.shard_of(V, G, r)
        shard, d = rz.shard.cut_mesh(mesh, deltas, b, n)
        c = rz.DeformContext(0)
        c.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"])
        c.upload_skeleton(mesh["inv_bind"])
        c.upload_morphs_dense(d)
        ctxs.append((c, b, n))
    with pytest.raises(rz.RzError):
        ctxs[0][0].gather_fence()                     # not a root yet
    rz.capi.gather_direct([c for c, _, _ in ctxs], V, root=root)
    for c, _, _ in ctxs:
        c.set_pose(mesh["world"], mw)
        c.deform()
    pg, ng = ctxs[root][0].read_gathered()
    assert_parity(pg, ng, pr, nr, "peer-direct gather")
    for c, b, n in ctxs:                              # a contributor still reads its own shard (out of the root's buffer)
        ps, ns = c.read()
        assert np.array_equal(ps, pg[b:b + n]) and np.array_equal(ns, ng[b:b + n])
    with pytest.raises(rz.RzError):
        ctxs[0][0].set_instances(4)                   # instancing and sharding are exclusive
    # a second frame with another pose lands in the same buffer
    world2 = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=5)
    pr2, nr2 = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], world2, mesh["inv_bind"], deltas, mw)
    for c, _, _ in ctxs:
        c.set_pose(world2, mw)
        c.deform()
    ctxs[root][0].gather_fence()
    pg2, ng2 = ctxs[root][0].read_gathered(v0=100, n=V - 100)
    assert_parity(pg2, ng2, pr2[100:], nr2[100:], "peer-direct gather frame 2")
    # a new mesh on shard 2 detaches it: its frames go to its own buffers again, the root's copy keeps the old data
    c2, b2, n2 = ctxs[2]
    small = synth.make_mesh(300, B, seed=3)
    c2.upload_mesh(small["pos"], small["nrm"], small["joints"], small["weights"])
    c2.upload_skeleton(small["inv_bind"])
    c2.set_pose(small["world"])
    c2.deform()
    ps, ns = c2.read()
    prs, nrs = oracle.deform(small["pos"], small["nrm"], small["joints"], small["weights"], small["world"], small["inv_bind"])
    assert_parity(ps, ns, prs, nrs, "detached shard")
    pg3, _ = ctxs[root][0].read_gathered()
    assert np.array_equal(pg3[b2:b2 + n2], np.concatenate([pr2[:0], pg2[b2 - 100:b2 - 100 + n2]]))
    ctxs[root][0].close()                             # destroying the root detaches the rest
    c0 = ctxs[0][0]
    c0.deform()
    p0, n0 = c0.read()
    assert_parity(p0, n0, pr2[:ctxs[0][2]], nr2[:ctxs[0][2]], "after the root is gone")
    c0.close()
    c2.close()


def test_error_paths(ctx, rz):
    c = rz.DeformContext(0)
    with pytest.raises(rz.RzError):
        c.deform()                                    # nothing uploaded
    mesh = synth.make_mesh(100, 4, seed=1)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    with pytest.raises(rz.RzError):
        c.deform()                                    # no skeleton
    c.upload_skeleton(mesh["inv_bind"])
    with pytest.raises(rz.RzError):
        c.deform()                                    # no pose
    with pytest.raises(rz.RzError):
        c.set_tuning(morph_split=16)
    with pytest.raises(rz.RzError):
        c._L.rz_read and c.read(v0=90, n=20)          # out of range
    with pytest.raises(rz.RzError):
        c.upload_edge_scale(np.ones(7, dtype=np.float32))     # wrong length
    with pytest.raises(rz.RzError):
        c.set_pose_local(np.zeros((4, 4), dtype=np.float32))  # no topology uploaded
    with pytest.raises(rz.RzError):
        c.read_aabb()                                 # reduction not enabled
    c2 = rz.DeformContext(0)
    with pytest.raises(rz.RzError):
        c2.upload_morphs_dense(np.zeros((1, 1, 3), dtype=np.float32)) if False else c2._L.rz_upload_morphs_dense(c2._h, 1, None) and (_ for _ in ()).throw(rz.RzError(-1, "morphs before mesh"))
    c2.close()
    c.close()


def test_engine_through_napi_matches_oracle(tmp_path, oracle):
    """The drop-in path end to end: host JS Engine (loadModel / loadAnimation / playAnimation / rotateBones /
    setMorphWeights / step) -> reze_deform.node -> C ABI -> MI355X, against the oracle fed with the SAME
    world matrices and effective morph weights the host produced. Sparse and dense morph layouts."""
    import json
    import shutil
    import subprocess
    import os
    from pmx_synth import write_pmx, write_vmd
    if shutil.which("node") is None:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "m.pmx").write_bytes(write_pmx())
    s = 0.38268343
    (tmp_path / "a.vmd").write_bytes(write_vmd(
        [("bone1", 0, (0, 0, s, 0.92387953)), ("bone3", 0, (s, 0, 0, 0.92387953), (0.3, -0.2, 0.1)), ("bone1", 15, (0, s, 0, 0.92387953)),
         ("bone3", 20, (0, 0, s, 0.92387953), (-0.5, 0.4, 0.25)), ("bone0", 0, (0, 0, 0, 1), (0, 0.5, 0)), ("bone0", 25, (0, 0, 0, 1), (1.0, 0.25, -0.5)),
         ("bone20", 30, (0, 0, -s, 0.92387953))], [("v1", 0, 0.8), ("v2", 6, 0.4)]))
    for layout, devs in (("sparse", "0"), ("dense", "0"), ("sparse", "0,0"), ("dense", "0,0,0"), ("sparse", "0:fk"), ("dense", "0,0:fk"),
                         ("sparse", "0,0:direct"), ("dense", "0,0,0:direct")):
        # "0,0": the Engine's multi-GPU sharding (one context per listed device) exercised as 2-3 shards on one GPU
        out = tmp_path / (layout + devs.replace(",", "_").replace(":", "_"))
        out.mkdir()
        subprocess.check_call(["node", os.path.join(root, "tests", "js", "engine_e2e.js"), str(tmp_path / "m.pmx"),
                               str(tmp_path / "a.vmd"), str(out), layout, devs], timeout=300)
        rd = lambda f, dt: np.fromfile(str(out / f), dtype=dt)  # noqa: E731
        v = rd("vertices.f32", np.float32).reshape(-1, 8)
        joints = rd("joints.u16", np.uint16).reshape(-1, 4)
        weights = rd("weights.u8", np.uint8).reshape(-1, 4)
        ib = rd("invbind.f32", np.float32).reshape(-1, 16)
        off, vidx = rd("morph_offsets.u32", np.uint32), rd("morph_vidx.u32", np.uint32)
        d3 = rd("morph_deltas.f32", np.float32).reshape(-1, 3)
        seen_morph = False
        for step in range(4):
            world = rd("world_%d.f32" % step, np.float32).reshape(-1, 16)
            mw = rd("mw_%d.f32" % step, np.float32)
            pm = oracle.morph_sparse(len(v), off, vidx, d3, mw, v[:, 0:3])
            S = oracle.palette(world, ib)
            pr, nr = oracle.skin(pm, v[:, 3:6], joints, weights, S)
            pg = rd("pos_%d.f32" % step, np.float32).reshape(-1, 3)
            ng = rd("nrm_%d.f32" % step, np.float32).reshape(-1, 3)
            if ":fk" in devs:      # the GPU solved the hierarchy in f32: compare its world matrices with the host's
                np.testing.assert_allclose(rd("gpuworld_%d.f32" % step, np.float32).reshape(-1, 16), world, rtol=3e-5, atol=3e-5)
            assert_parity(pg, ng, pr, nr, "napi %s devices %s step %d" % (layout, devs, step))
            # fused consumers through the same boundary: outline hull and bounding box (multi-shard boxes are merged on the host)
            edge = rd("edge.f32", np.float32)
            assert (edge > 0).sum() > 0
            hull = rd("hull_%d.f32" % step, np.float32).reshape(-1, 3)
            assert_hull(hull.reshape(-1, 3), oracle.hull(pr, nr, edge), "hull through N-API")
            box = rd("bounds_%d.f32" % step, np.float32)
            assert np.array_equal(box[:3], pg.min(axis=0)) and np.array_equal(box[3:], pg.max(axis=0))
            seen_morph = seen_morph or (mw != 0).sum() >= 3
            assert not np.allclose(pg, v[:, 0:3])            # the pose really moved the mesh
        assert seen_morph
        st = json.load(open(out / "stats.json"))
        assert st["t"]["frameMs"] > 0 and st["st"]["vertsPerSec"] > 0


@pytest.mark.parametrize("device_fk", [0, 1])
def test_engine_frames_in_flight_through_napi(tmp_path, device_fk):
    """new Engine(null, { framesInFlight: 2 }): frames alternate between the context and an rz_fork of it (shared static data, own
    stream + outputs). Clock steps, tweens, morph weights (vertex, group, bone morph) and frame seeks; host FK and device FK +
    device sampling. Every frame must be bit-identical to the plain engine's, hull and bounds included."""
    import json
    import shutil
    import subprocess
    import os
    from pmx_synth import write_pmx, write_vmd
    if shutil.which("node") is None:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "m.pmx").write_bytes(write_pmx())
    s = 0.38268343
    (tmp_path / "a.vmd").write_bytes(write_vmd(
        [("bone1", 0, (0, 0, s, 0.92387953)), ("bone3", 0, (s, 0, 0, 0.92387953), (0.3, -0.2, 0.1)), ("bone1", 15, (0, s, 0, 0.92387953)),
         ("bone3", 20, (0, 0, s, 0.92387953), (-0.5, 0.4, 0.25)), ("bone0", 0, (0, 0, 0, 1), (0, 0.5, 0)), ("bone0", 25, (0, 0, 0, 1), (1.0, 0.25, -0.5))],
        [("v1", 0, 0.8), ("v2", 6, 0.4), ("twist", 0, 0.1), ("twist", 30, 0.9)]))
    out = tmp_path / "r.json"
    subprocess.check_call(["node", os.path.join(root, "tests", "js", "engine_inflight.js"), str(tmp_path / "m.pmx"), str(tmp_path / "a.vmd"), str(out), str(device_fk)], timeout=300)
    r = json.load(open(out))
    assert r["frames"] == 9 and r["mismatches"] == [], r
    assert r["forked"] is True and r["moved"] > 0.2
