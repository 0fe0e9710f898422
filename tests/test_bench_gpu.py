"""bench.py's driver contract, exercised on the GPU box: one JSON line with the fields the driver reads, at N = 1 and —
rehearsed as two ranks sharing the one GPU of this box over gloo — through the torch.distributed launch line the driver
uses for N > 1 (the 8-GPU run itself is the driver's). Small workloads: this checks plumbing, not speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line, got %d:\n%s" % (len(lines), out[-2000:])
    return json.loads(lines[0])


def _check(d, n_gpus, steps, verts):
    for k in REQUIRED:
        assert k in d, "missing key " + k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"]           # the driver matches the line against BASELINE.json's metric
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["unit"] == "verts/s" and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = the vertices ALL ranks deformed / the slowest rank's time
    assert abs(d["value"] - verts * steps / (d["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert r["kernel"].startswith("rz_") and "traffic_source" in r and r["frac_of_measured_ceiling"] > 0
    ranks = d["config"]["ranks"]
    assert len(ranks) == n_gpus and sorted(x["rank"] for x in ranks) == list(range(n_gpus))
    assert sum(x["verts"] for x in ranks) == verts and all(x["kernel_ms"] > 0 and x["kernel"].startswith("rz_") for x in ranks)
    assert d["config"]["kernel_ms_max_over_ranks"] >= d["config"]["kernel_ms_min_over_ranks"] > 0
    assert d["config"]["frame_ms_with_pose_upload"] > 0 and d["config"]["frame_ms_device_sampled_pose"] > 0
    nb = d["config"]["numa_binding"]                # rank 0 ran on its GPU's node, or says that there was nothing to bind to
    assert nb is None or (nb["gpu_node"] >= 0 and nb["cpus"])


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--verts", "40000", "--bones", "64", "--morphs", "8",
                                   "--steps", "20", "--warmup", "3", "--cpu-sample-verts", "20000", "--clock-warm-seconds", "0.2"], cwd=ROOT, timeout=600).decode()
    d = _last_json(out)
    _check(d, 1, 20, 40000)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "verts/s" and cb["sample"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
           "--verts", "50000", "--bones", "64", "--morphs", "8", "--share-gpu", "--dist-backend", "gloo", "--no-cpu-baseline", "--clock-warm-seconds", "0.2"]
    out = subprocess.check_output(cmd, cwd=ROOT, env=env, timeout=900, stderr=subprocess.STDOUT).decode()
    d = _last_json(out)
    _check(d, 2, 20, 50000)
    assert d["scaling"] == "strong" and d["config"]["verts_per_gpu"] < 50000 and "x2" in d["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_is_a_two_rank_run():
    """Round-2 review: `python bench.py --gpus N` run PLAINLY (no torch.distributed.run around it) must be an N-rank run —
    it re-executes itself under the launcher — never one rank printing n_gpus 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--verts", "50000", "--bones", "64",
           "--morphs", "8", "--share-gpu", "--dist-backend", "gloo", "--no-cpu-baseline", "--clock-warm-seconds", "0.2"]
    out = subprocess.check_output(cmd, cwd=ROOT, env=env, timeout=900, stderr=subprocess.STDOUT).decode()
    d = _last_json(out)
    _check(d, 2, 20, 50000)
    assert "self" in d["config"]["launched_by"]
    # both modes at every N, and how to read a scaling ratio
    assert d["config"]["ms_per_step_one_stream"] > 0 and d["config"]["ms_per_step_two_frames_in_flight"] > 0 and d["config"]["speedup_basis"]
    # one plan on all ranks (the search reduces its table over the ranks and every rank adopts the same entry)
    ranks = d["config"]["ranks"]
    assert len({(r["kernel"], r["morph_split"]) for r in ranks}) == 1, ranks      # (the grid follows each rank's shard size)
    assert d["config"]["autotune_table"] and d["config"]["autotune_pick"] is not None
    assert d["roofline"]["kernel_ms_check"]["ok"], d["roofline"]


@pytest.mark.gpu
def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus 1 under a 2-rank environment (or the reverse) is an error, not a line with another n_gpus."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2"], cwd=ROOT, env=env, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 3 and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    # two ranks on a one-GPU box without --share-gpu: also an error (one process per GPU)
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu-baseline"], cwd=ROOT, env=env2,
                           timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode != 0 and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_single_rank_rccl_evidence_and_sparse_configs():
    """--allgather at N = 1 builds the RCCL communicator through rz_comm_init and reports what the communicator says
    (ncclCommCount / ncclCommUserRank); --config demo / sparse2 are the sparse real-shape lines of SURVEY 8d."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--verts", "40000", "--bones", "64", "--morphs", "8", "--steps", "20",
                                   "--warmup", "3", "--no-cpu-baseline", "--clock-warm-seconds", "0.2", "--allgather"], cwd=ROOT, timeout=600).decode()
    d = _last_json(out)
    rc = d["config"]["ranks"][0]["rccl"]
    assert rc["comm_count"] == 1 and rc["comm_user_rank"] == 0 and rc["version"] > 0 and d["config"]["allgather_ms"] > 0
    for cfg, verts in (("demo", 28842), ("sparse2", 28842)):
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "50", "--warmup", "5",
                                       "--clock-warm-seconds", "0.2", "--no-sampled-loop"], cwd=ROOT, timeout=600).decode()
        d = _last_json(out)
        assert d["n_gpus"] == 1 and d["config"]["verts_total"] == verts and d["config"]["morph_layout"] == "sparse CSR"
        assert "2, " in d["roofline"]["kernel"] or ", 2," in d["roofline"]["kernel"], d["roofline"]["kernel"]     # MODE 2 = sparse
        assert d["cpu_baseline"]["value"] > 0 and "sparse" in d["cpu_baseline"]["sample"]
        assert d["roofline"]["algorithmic_bytes_per_launch"] > verts * 60
