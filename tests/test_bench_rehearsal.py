"""First contact with a multi-GPU node, rehearsed WITHOUT one (round-4 review item 8). The one thing a 1-GPU box cannot exercise is what
an RCCL communicator that FAILS or HANGS does to `bench.py --gpus 8` — the driver needs its one JSON line either way. These tests run
the real bench.py as 8 ranks over gloo on the CPU, against a stand-in for the `reze_engine_amd` package whose DeformContext does no GPU
work (frames take time proportional to the shard's vertices, so the compute scaling of the line can be checked) and whose
rz_comm_init / rz_allgather fail, hang, or succeed as told. Everything else — argument handling, self-launch, sharding, the autotune
table reduction, the timed loops and their collectives, the per-rank records, the watchdog — is bench.py's own code."""
import json
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    """Stand-in for reze_engine_amd (tests/test_bench_rehearsal.py): the real synth / shard / capi modules, a DeformContext without a GPU."""
    import importlib.util, os, sys, time
    _real = os.path.join(%(root)r, "reze-engine_amd")
    _spec = importlib.util.spec_from_file_location("reze_engine_amd_real", os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
    _pkg = importlib.util.module_from_spec(_spec)
    sys.modules["reze_engine_amd_real"] = _pkg
    _spec.loader.exec_module(_pkg)
    sys.modules[__name__ + ".synth"], sys.modules[__name__ + ".shard"], sys.modules[__name__ + ".capi"] = _pkg.synth, _pkg.shard, _pkg.capi
    capi, shard, synth = _pkg.capi, _pkg.shard, _pkg.synth
    MODE = os.environ.get("REZE_STUB_RCCL", "ok")
    US_PER_KVERT = 40.0                      # a frame of 1000 vertices "takes" 40 us


    class DeformContext:
        def __init__(self, device=0, lib=None):
            self.V = 0; self.I = 1; self.busy_until = 0.0; self.tuning = {}
        def _frame_s(self): return self.V * self.I / 1000.0 * US_PER_KVERT * 1e-6
        def _enqueue(self, frames):
            self.busy_until = max(self.busy_until, time.perf_counter()) + frames * self._frame_s()
        def upload_mesh(self, pos, nrm, j, w): self.V = len(pos)
        def upload_skeleton(self, ib): pass
        def upload_morphs_dense(self, d): pass
        def upload_morphs_sparse(self, *a): pass
        def upload_skeleton_topology(self, *a, **k): pass
        def upload_animation(self, *a, **k): pass
        def set_instances(self, n): self.I = n
        def set_pose(self, *a): pass
        def set_pose_local(self, *a, **k): pass
        def set_pose_sampled(self, *a): pass
        def set_tuning(self, **kw): self.tuning.update(kw)
        def get_tuning(self, key): return {"effective_split": 4, "effective_grid": 489}.get(key, 0)
        def autotune_measure(self, frames=0):
            ms = self._frame_s() * 1e3
            return [dict(morph_split=0, grid_cap=0, inst_loop=-1, eff_split=4, eff_grid=489, eff_inst_group=0, same_as=-1, ms=ms, ms_min=ms * 0.99, ms_max=ms * 1.01),
                    dict(morph_split=2, grid_cap=512, inst_loop=0, eff_split=2, eff_grid=489, eff_inst_group=0, same_as=-1, ms=ms * 1.05, ms_min=ms * 1.04, ms_max=ms * 1.06)]
        def autotune_pick(self, table): return capi.load().rz_autotune_pick((capi.RzTuneEntry * len(table))(*[capi.RzTuneEntry(**e) for e in table]), len(table))
        def autotune_apply(self, entry): pass
        def deform(self): self._enqueue(1)
        def deform_n(self, frames): self._enqueue(frames)
        def deform_pair(self, other, frames):
            if os.environ.get("REZE_STUB_PAIR") == "fades":
                # two frames in flight look a tenth faster during the calibration (its three calls) and run a tenth slower afterwards
                self.pair_calls = getattr(self, "pair_calls", 0) + 1
                self._enqueue(frames * (0.9 if self.pair_calls <= 3 else 1.1))
                return
            self._enqueue((frames + 1) // 2); other._enqueue(frames // 2)
        def sync(self):
            d = self.busy_until - time.perf_counter()
            if d > 0: time.sleep(d)
        def time_span(self, frames, other=None, lead=0):
            # the stand-in's "events": virtual GPU time from the first TIMED frame's start to the last frame's end
            if lead: self.deform_n(lead)
            t0 = max(self.busy_until, time.perf_counter())
            if other is None: self.deform_n(frames)
            else: self.deform_pair(other, frames)
            self.sync()
            if other is not None: other.sync()
            return (max(self.busy_until, other.busy_until if other is not None else 0.0) - t0) * 1e3
        def fork(self):
            f = DeformContext(); f.V = self.V; f.I = self.I; return f
        def frame_call(self, kind, *a):
            return (lambda: self._enqueue(1)), (lambda: None)
        def time_frames(self, frames):
            ms = self._frame_s() * 1e3
            return dict(frame_ms=ms, deform_kernel_ms=ms * 0.98, prep_kernel_ms=0.0, verts_per_frame=self.V, algorithmic_bytes_per_frame=self.V * 828, frames=frames)
        def kernel_name(self): return "rz_deform_dense_kernel<4, 8, true, false, false, true>"
        def comm_init(self, nranks, rank, uid, v_total):
            if MODE == "fail": raise capi.RzError(-4, "ncclCommInitRank failed: unhandled system error (stand-in)")
            if MODE == "hang": time.sleep(3600)
            if MODE == "hang_rank3" and rank == 3: time.sleep(3600)
        def comm_info(self): return {"comm_count": int(os.environ["WORLD_SIZE"]), "comm_user_rank": int(os.environ["RANK"])}
        def allgather(self, with_normals=False):
            if MODE == "fail_allgather": raise capi.RzError(-4, "ncclAllGather failed: remote process exited (stand-in)")
        def close(self): pass

    capi.comm_unique_id = lambda: b"\\0" * 128
    capi.rccl_info = lambda: {"path": "stand-in", "version": 0, "reused": False}
    RzError, device_count, shard_range = capi.RzError, capi.device_count, capi.shard_range
''')


def _run(tmp_path, mode, extra=()):
    work = tmp_path / "bench_root"
    (work / "reze_engine_amd").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "bench.py"), work / "bench.py")
    (work / "reze_engine_amd" / "__init__.py").write_text(STUB % {"root": ROOT})
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(REZE_STUB_RCCL=mode, OMP_NUM_THREADS="1")
    extra = list(extra)
    if "--pair-fades" in extra:
        env["REZE_STUB_PAIR"] = "fades"
        extra.remove("--pair-fades")

    shape = ["--verts", "65536", "--bones", "16", "--morphs", "2"]
    if "--crowd" in extra:
        extra.remove("--crowd")
        shape = ["--verts", "1024", "--bones", "16", "--morphs", "0", "--instances", "100"]
    cmd = [sys.executable, str(work / "bench.py"), "--gpus", "8", "--share-gpu", "--rehearse-rccl", "--dist-backend", "gloo"] + shape + [
           "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-sampled-loop", "--clock-warm-seconds", "0", "--rccl-timeout", "6"] + list(extra)
    p = subprocess.run(cmd, cwd=str(work), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    out = p.stdout.decode()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected, got %d (rc %d)\nstdout tail: %s\nstderr tail: %s" % (len(lines), p.returncode, out[-1500:], p.stderr.decode()[-3000:])
    return json.loads(lines[0]), p


def _scaling_intact(d):
    # 8 ranks, 65 536 vertices: every rank holds 8 192 and a stand-in frame takes 40 us per 1 000 vertices = 0.33 ms; the whole mesh
    # on one rank would take 2.6 ms. The line must show the sharded time (slowest rank), i.e. the compute numbers survived the RCCL trouble.
    assert d["n_gpus"] == 8 and len(d["config"]["ranks"]) == 8 and sum(r["verts"] for r in d["config"]["ranks"]) == 65536
    assert 0.2 < d["ms_per_step"] < 0.8, d["ms_per_step"]
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["config"]["autotune_pick"] == 0 and d["config"]["ms_per_step_one_stream"] > 0
    # round 6: the K steps are timed by events on the stream; the host's clock around the same steps is on the line beside them and can
    # only be longer (it adds the fixed cost of a timed region)
    c = d["config"]
    assert "hipEvent" in d["timed_by"]
    assert c["ms_per_step_host_wall"] >= d["ms_per_step"] > 0 and c["host_fixed_cost_us_per_timed_region"] >= 0
    assert abs(c["value_host_wall"] - 65536 / (c["ms_per_step_host_wall"] * 1e-3)) <= 1e-6 * c["value_host_wall"] and c["lead_in_frames"] == 2


@pytest.mark.parametrize("mode", ["fail", "fail_allgather"])
def test_bench_8_ranks_with_a_failing_communicator_still_prints_its_line(tmp_path, mode):
    d, p = _run(tmp_path, mode)
    _scaling_intact(d)
    for r in d["config"]["ranks"]:
        assert r["rccl"] and "error" in r["rccl"] and "stand-in" in r["rccl"]["error"], r
    assert d["config"]["allgather_ms"] is None and p.returncode == 0


@pytest.mark.parametrize("mode", ["hang", "hang_rank3"])
def test_bench_8_ranks_with_a_hanging_communicator_is_cut_loose_by_the_watchdog(tmp_path, mode):
    d, p = _run(tmp_path, mode)
    _scaling_intact(d)
    for r in d["config"]["ranks"]:
        assert r["rccl"] and "watchdog" in r["rccl"]["error"], r
    assert d["config"]["allgather_ms"] is None


def test_bench_8_ranks_with_a_working_communicator(tmp_path):
    d, p = _run(tmp_path, "ok")
    _scaling_intact(d)
    for r in d["config"]["ranks"]:
        assert r["rccl"]["comm_count"] == 8 and r["rccl"]["expected_count"] == 8 and "error" not in r["rccl"]
    assert d["config"]["allgather_ms"] is not None and p.returncode == 0


def test_two_frames_in_flight_stay_the_headline_only_where_their_timed_steps_win(tmp_path):
    """Round 5 (seen on C4 --device-fk): the untimed calibration promised >= 3 % for two frames in flight, the K timed steps of the pair
    loop then ran SLOWER than the one-stream loop's. The headline must be the one-stream loop — both numbers stay on the line."""
    d, p = _run(tmp_path, "ok", extra=["--pair-fades"])
    c = d["config"]
    assert c["frames_in_flight_calibrated"] == 2 and c["frames_in_flight"] == 1, (c["frames_in_flight_calibrated"], c["frames_in_flight"], c["frames_in_flight_choice"])
    assert c["ms_per_step_two_frames_in_flight"] > c["ms_per_step_one_stream"]
    assert abs(d["ms_per_step"] - c["ms_per_step_one_stream"]) < 1e-12


def test_bench_c4_is_sharded_along_the_instance_axis(tmp_path):
    """SURVEY 8e, last sentence: a crowd shards along the instance axis — every rank holds the whole mesh and poses ceil(I / N) of the
    instances, nothing is exchanged and no communicator is made. 100 instances over 8 ranks: 13 x 7 + 9."""
    d, p = _run(tmp_path, "fail", extra=["--crowd"])            # (a communicator that would fail is never asked for)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["instances"] == 100 and c["parallelism"] == "instance-shard x8" and "instance-sharded" in c["workload"]
    got = [(r["instance_begin"], r["instances"], r["verts"]) for r in c["ranks"]]
    assert got == [(13 * k, 13 if k < 7 else 9, 1024) for k in range(8)], got
    for r in c["ranks"]:
        assert "skipped" in r["rccl"] and "instance axis" in r["rccl"]["skipped"], r
    # the slowest rank poses 13 of the 100 characters: 13 x 1024 vertices x 40 us per 1000 = 0.53 ms per step
    assert 0.4 < d["ms_per_step"] < 1.0 and abs(d["value"] - 1024 * 100 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert p.returncode == 0
