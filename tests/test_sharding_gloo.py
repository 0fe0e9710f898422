"""N > 1 host path: two processes over gloo shard the mesh exactly as bench.py does — every rank GENERATES only its own
range of the block-seeded synthetic mesh (synth.make_mesh_range) — deform their shard, all-gather padded chunks, and must
reproduce the unsharded result bit for bit.
  * CPU variant (runs everywhere): the CPU oracle stands in for the kernel — partition, per-rank generation and gather layout.
  * GPU variant (-m gpu): every rank drives the REAL HIP kernel through the C ABI on the box's GPU (two ranks share it),
    and the gathered mesh must equal a single context's whole-mesh frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, v_total, q, on_gpu=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import oracle
    import reze_engine_amd as rz
    from reze_engine_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, M = 24, 5
        b, n, chunk = rz.shard.shard_of(v_total, world, rank)
        part = synth.make_mesh_range(v_total, B, b, n, seed=77)          # this rank's range only
        d, mw = synth.make_morphs_dense_range(v_total, M, b, n, seed=78)
        if n == 0:
            pos = np.zeros((0, 3), dtype=np.float32)
        elif on_gpu:
            c = rz.DeformContext(0)                                      # both ranks share GPU 0 of the box
            c.upload_mesh(part["pos"], part["nrm"], part["joints"], part["weights"])
            c.upload_skeleton(part["inv_bind"])
            c.upload_morphs_dense(d)
            c.set_pose(part["world"], mw)
            c.deform()
            pos, _ = c.read()
            c.close()
        else:
            pos, _ = oracle.deform(part["pos"], part["nrm"], part["joints"], part["weights"], part["world"], part["inv_bind"], d, mw)
        send = torch.from_numpy(rz.shard.pad_to_chunk(pos, chunk))
        recv = torch.empty((world * chunk, 3), dtype=torch.float32)
        dist.all_gather_into_tensor(recv, send)
        full = rz.shard.gathered_to_mesh(recv.numpy(), v_total)
        # the unsharded result, from the whole mesh generated in one piece
        mesh = synth.make_mesh_range(v_total, B, 0, v_total, seed=77)
        deltas, mw_all = synth.make_morphs_dense_range(v_total, M, 0, v_total, seed=78)
        assert np.array_equal(mw, mw_all) and np.array_equal(mesh["pos"][b:b + n], part["pos"]) and np.array_equal(mesh["joints"][b:b + n], part["joints"])
        if on_gpu:
            c = rz.DeformContext(0)
            c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
            c.upload_skeleton(mesh["inv_bind"])
            c.upload_morphs_dense(deltas)
            c.set_pose(mesh["world"], mw)
            c.deform()
            ref, _ = c.read()
            c.close()
            pr, nr = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
            err = np.linalg.norm(ref.astype(np.float64) - pr, axis=1) / np.maximum(np.linalg.norm(pr, axis=1), 1.0)
            assert err.max() <= 1e-4, err.max()
        else:
            ref, _ = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], deltas, mw)
        ranges = [None] * world
        dist.all_gather_object(ranges, (b, n))
        res = (rank, bool(np.array_equal(full, ref)), ranges, chunk)
        if q is not None:
            q.put(res)
        return res
    finally:
        dist.destroy_process_group()


def _crowd_worker(rank, world, port, instances, q, on_gpu=False):
    """BASELINE config 4 over N ranks (last sentence): the crowd is cut along the INSTANCE axis — every rank holds the whole
    mesh, generates the poses of ITS instances only (seeded by the global instance number, like bench.py) and deforms them; nothing is
    exchanged by the path. The test gathers every rank's instances (test plumbing, not the path) and compares them with one rank posing
    the whole crowd: instance k must have the same bits wherever it ran."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # noqa: F401
    import torch.distributed as dist
    import oracle
    import reze_engine_amd as rz
    from reze_engine_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, B = 3000, 40
        mesh = synth.make_mesh(V, B, seed=91)
        pose = lambda i: synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i).astype(np.float32)     # noqa: E731
        b, n = rz.shard.instances_of(instances, world, rank)

        def run(first, count):
            worlds = np.stack([pose(first + k) for k in range(count)]) if count else np.zeros((0, B, 16), np.float32)
            if count == 0:
                return np.zeros((0, V, 3), np.float32)
            if on_gpu:
                c = rz.DeformContext(0)                                  # every rank shares GPU 0 of the box
                c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
                c.set_instances(count)
                c.set_pose(worlds); c.deform()
                out = np.stack([c.read(k)[0] for k in range(count)])
                c.close()
                return out
            return np.stack([oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], worlds[k], mesh["inv_bind"])[0] for k in range(count)]).astype(np.float32)
        mine = run(b, n)
        parts = [None] * world
        dist.all_gather_object(parts, (b, n, mine))
        parts.sort(key=lambda t: t[0])
        assert [t[0] for t in parts] == [sum(u[1] for u in parts[:k]) for k in range(world)] and sum(t[1] for t in parts) == instances
        whole = run(0, instances)
        ok = bool(np.array_equal(np.concatenate([t[2] for t in parts]), whole))
        if on_gpu and instances:
            pr, _ = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], pose(instances - 1), mesh["inv_bind"])
            err = np.linalg.norm(whole[-1].astype(np.float64) - pr, axis=1) / np.maximum(np.linalg.norm(pr, axis=1), 1.0)
            assert err.max() <= 1e-4, err.max()
        res = (rank, ok, [(t[0], t[1]) for t in parts])
        if q is not None:
            q.put(res)
        return res
    finally:
        dist.destroy_process_group()


def _run_crowd_ranks(instances, on_gpu, world=2):
    port = _free_port()
    per = -(-instances // world)
    want = [(min(instances, r * per), min(per, instances - min(instances, r * per))) for r in range(world)]
    if on_gpu:
        import json
        import subprocess
        code = ("import sys, json; sys.path.insert(0, %r); import test_sharding_gloo as t; "
                "r = t._crowd_worker(int(sys.argv[1]), %d, int(sys.argv[2]), int(sys.argv[3]), None, True); print('RESULT ' + json.dumps(r))" % (os.path.dirname(os.path.abspath(__file__)), world))
        procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(instances)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
        res = []
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, "rank process died with %d\n%s" % (p.returncode, err.decode()[-3000:])
            res.append(json.loads([ln for ln in out.decode().splitlines() if ln.startswith("RESULT ")][-1][7:]))
    else:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_crowd_worker, args=(r, world, port, instances, q, False)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for rank, ok, ranges in res:
        assert ok, "rank %d: an instance differs from the same instance of the whole crowd" % rank
        assert [tuple(r) for r in ranges] == want, (ranges, want)


def _check(res, v_total):
    for rank, ok, ranges, chunk in res:
        assert ok, "rank %d: gathered mesh differs from the unsharded result" % rank
        assert ranges[0][0] == 0 and ranges[0][1] + ranges[1][1] == v_total and (ranges[1][0] == ranges[0][1] or ranges[1][1] == 0)
        assert chunk % 256 == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_ranks(v_total, on_gpu):
    port = _free_port()
    if on_gpu:
        # Separate interpreters that import torch BEFORE the HIP library, like bench.py: PyTorch bundles its own ROCm
        # runtime libraries, and pulling them into a process that already runs on /opt/rocm's (the pytest process, after
        # the other GPU tests) ends in a double free at exit.
        import json
        import subprocess
        code = ("import sys, json; sys.path.insert(0, %r); import test_sharding_gloo as t; "
                "r = t._worker(int(sys.argv[1]), 2, int(sys.argv[2]), int(sys.argv[3]), None, True); print('RESULT ' + json.dumps(r))" % os.path.dirname(os.path.abspath(__file__)))
        procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(v_total)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
        res = []
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, "rank process died with %d\n%s" % (p.returncode, err.decode()[-3000:])
            line = [ln for ln in out.decode().splitlines() if ln.startswith("RESULT ")][-1]
            res.append(json.loads(line[7:]))
        _check(res, v_total)
        return
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, v_total, q, on_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _check(res, v_total)


@pytest.mark.parametrize("v_total", [5000, 1024, 2049, 40000])
def test_two_rank_shard_and_gather_equals_single_rank(v_total):
    _run_two_ranks(v_total, on_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("v_total", [40000, 2049])
def test_two_ranks_real_kernel_per_rank_sharing_the_gpu(v_total):
    _run_two_ranks(v_total, on_gpu=True)


@pytest.mark.parametrize("instances,world", [(7, 2), (8, 2), (5, 3)])
def test_crowd_sharded_along_the_instance_axis_over_gloo(instances, world):
    _run_crowd_ranks(instances, on_gpu=False, world=world)


@pytest.mark.gpu
@pytest.mark.parametrize("instances", [21, 2])
def test_two_ranks_pose_their_own_instances_of_a_crowd_sharing_the_gpu(instances):
    _run_crowd_ranks(instances, on_gpu=True)
