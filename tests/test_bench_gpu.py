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
