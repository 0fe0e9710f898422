"""What each GPU runs at N = 1, 2, 4, 8 (strong scaling of C5): one shard of rz_shard_range(1 M, N, 0) on this GPU.
Prints frame time, the projected N-GPU speed-up (t1 / tN) and the shard's algorithmic GB/s — for the heuristic plan and after rz_autotune.
Exit status 1 when the heuristic plan is more than 2 % slower than what the search adopts at any N (round 6)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B, M = 1000000, 256, 64
mesh = synth.make_mesh(V, B)
deltas, mw = synth.make_morphs_dense(V, M)
t1 = None
t1_wall = None
worst = 0.0
for N in (1, 2, 4, 8):
    b, n, _ = rz.shard.shard_of(V, N, 0)
    shard, d = rz.shard.cut_mesh(mesh, deltas, b, n)
    ctx = rz.DeformContext(0)
    ctx.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(d); ctx.set_pose(mesh["world"], mw)
    ctx.deform_n(50)
    t = min((ctx.time_frames(300) for _ in range(5)), key=lambda t: t["frame_ms"])
    us = t["frame_ms"] * 1e3
    t1 = t1 or us
    print(json.dumps({"N": N, "shard_verts": n, "frame_us": round(us, 2), "GBps": round(t["algorithmic_bytes_per_frame"] / us / 1e3, 1),
                      "projected_speedup": round(t1 / us, 2), "split": ctx.get_tuning("effective_split"), "grid": ctx.get_tuning("effective_grid"),
                      "out_cap": ctx.get_tuning("effective_out_cap")}))
    ctx.autotune()
    t = min((ctx.time_frames(300) for _ in range(5)), key=lambda t: t["frame_ms"])
    worst = max(worst, us / (t["frame_ms"] * 1e3) - 1.0)
    print(json.dumps({"N": N, "autotuned_frame_us": round(t["frame_ms"] * 1e3, 2), "split": ctx.get_tuning("effective_split"), "grid": ctx.get_tuning("effective_grid"), "projected_speedup_vs_untuned_t1": round(t1 / (t["frame_ms"] * 1e3), 2)}))
    # the same shard with two frames in flight (rz_fork: shared static data, own stream + outputs), wall clock per frame
    fk = ctx.fork()
    fk.set_pose(mesh["world"], mw)
    def wall(run, sync, n=1000):
        sync(); t0 = time.perf_counter(); run(n); sync(); return (time.perf_counter() - t0) / n * 1e6
    both = lambda: (ctx.sync(), fk.sync())          # noqa: E731
    one = min(wall(ctx.deform_n, ctx.sync) for _ in range(4))
    two = min(wall(lambda n: ctx.deform_pair(fk, n), both) for _ in range(4))
    t1_wall = one if N == 1 else t1_wall
    print(json.dumps({"N": N, "wall_us_one_stream": round(one, 2), "wall_us_two_frames_in_flight": round(two, 2),
                      "projected_speedup_best_mode_vs_N1_one_stream": round(t1_wall / min(one, two), 2)}))
    fk.close()
    ctx.close()
print(json.dumps({"heuristic_behind_search_worst_pct": round(100 * worst, 2), "tolerance_pct": 2.0}))
sys.exit(1 if worst > 0.02 else 0)
