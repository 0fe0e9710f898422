"""N > 1 host path on CPU: two processes over gloo shard the mesh exactly as bench.py does, deform their
shard (CPU oracle standing in for the GPU kernel — this test is about partition + gather layout, not
arithmetic), all-gather padded chunks, and must reproduce the unsharded result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, v_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    import reze_engine_amd as rz
    from reze_engine_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, M = 24, 5
        mesh = synth.make_mesh(v_total, B, seed=77)
        deltas, mw = synth.make_morphs_dense(v_total, M, seed=78)
        b, n, chunk = rz.shard.shard_of(v_total, world, rank)
        part, d = rz.shard.cut_mesh(mesh, deltas, b, n)
        if n:
            pos, nrm = oracle.deform(part["pos"], part["nrm"], part["joints"], part["weights"], part["world"],
                                     part["inv_bind"], d, mw)
        else:
            pos = nrm = np.zeros((0, 3), dtype=np.float32)
        send = torch.from_numpy(rz.shard.pad_to_chunk(pos, chunk))
        recv = torch.empty((world * chunk, 3), dtype=torch.float32)
        dist.all_gather_into_tensor(recv, send) if hasattr(dist, "all_gather_into_tensor") else None
        full = rz.shard.gathered_to_mesh(recv.numpy(), v_total)
        ref, _ = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"],
                               deltas, mw)
        ranges = [None] * world
        dist.all_gather_object(ranges, (b, n))
        q.put((rank, bool(np.array_equal(full, ref)), ranges, chunk))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("v_total", [5000, 1024, 2049])
def test_two_rank_shard_and_gather_equals_single_rank(v_total):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, v_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ranges, chunk in res:
        assert ok, "rank %d: gathered mesh differs from the unsharded result" % rank
        assert ranges[0][0] == 0 and ranges[0][1] + ranges[1][1] == v_total and ranges[1][0] == ranges[0][1] or ranges[1][1] == 0
        assert chunk % 1024 == 0
