#!/usr/bin/env python3
"""bench.py — deformed verts/s of the fused morph + skin path on N MI355X GPUs.

Workload (BASELINE.json `metric`, config C5): a 1,000,000-vertex / 256-bone / 64-dense-morph
synthetic PMX (generator: reze-engine_amd/synth.py, seed 0x5EED), fp32. One "step" = one frame =
palette/active-morph prep kernel + fused morph+skin kernel over the whole mesh, inputs resident
in HBM. With N > 1 (launched by `python -m torch.distributed.run`, one rank per GPU) the SAME
1 M-vertex mesh is vertex-sharded across the ranks (strong scaling, as the north star states);
there is no data-path collective inside the timed region — the optional RCCL all-gather of
deformed positions is timed separately and reported in `config`.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` (HIP-event
timing of the dominant kernel against the ~8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle —
all-core JavaScript skin when Node is available, else the threaded C port — on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--verts", type=int, default=1000000)
    ap.add_argument("--bones", type=int, default=256)
    ap.add_argument("--morphs", type=int, default=64)
    ap.add_argument("--instances", type=int, default=1, help="C4-style instancing (single GPU only)")
    ap.add_argument("--config", choices=["c5", "c4", "c3", "c2"], default=None,
                    help="BASELINE.json shortcut: c5 = 1M/256/64 (default), c4 = 256 x 30k/200/0, c3 = 30k/200/64, c2 = 30k/200/0")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --verts is the whole mesh, sharded over ranks; weak: --verts per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-verts", type=int, default=200000)
    ap.add_argument("--allgather", action="store_true", help="also time the RCCL all-gather of positions")
    ap.add_argument("--device-fk", action="store_true",
                    help="solve the bone hierarchy on the GPU: frames start from local rotations (rz_set_pose_local)")
    ap.add_argument("--device-sampling", action="store_true",
                    help="with --device-fk: a synthetic motion is uploaded once and every frame sends ONE float per instance (rz_set_pose_sampled)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for the barrier / max-reduce (gloo + --share-gpu lets a 1-GPU box rehearse N > 1)")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank onto GPU (local_rank %% visible devices)")
    ap.add_argument("--graph", action="store_true", help="replay captured hipGraphs of 16 frames in the timed loop (rz_set_tuning graph=1): for launch-bound small frames")
    ap.add_argument("--tune", default="", help="comma list key=value passed to rz_set_tuning (disables the autotune pass)")
    ap.add_argument("--no-autotune", action="store_true", help="skip rz_autotune (setup-time search over launch shapes) and use the built-in heuristics")
    return ap.parse_args()


def cpu_baseline(args, mesh, deltas, mw):
    """Bounded CPU sample of the same workload: first `cpu_sample_verts` vertices, all morphs.
    Preferred: the all-core JavaScript f32 skin (oracle/js/cpu_baseline.js, worker_threads).
    Fallback: the threaded C oracle, labelled as a stand-in."""
    import oracle
    n = min(args.cpu_sample_verts, len(mesh["pos"]))
    sub = {k: np.ascontiguousarray(mesh[k][:n]) for k in ("pos", "nrm", "joints", "weights")}
    d = None if deltas is None else np.ascontiguousarray(deltas[:, :n])
    cores = os.cpu_count() or 1
    js = os.path.join(ROOT, "oracle", "js", "cpu_baseline.js")
    node = None
    for cand in ("node", "/usr/bin/node"):
        try:
            subprocess.check_output([cand, "--version"], stderr=subprocess.STDOUT)
            node = cand
            break
        except Exception:
            continue
    if node and os.path.exists(js):
        try:
            with tempfile.TemporaryDirectory() as td:
                sub["pos"].tofile(os.path.join(td, "pos.f32"))
                sub["nrm"].tofile(os.path.join(td, "nrm.f32"))
                sub["joints"].tofile(os.path.join(td, "joints.u16"))
                sub["weights"].tofile(os.path.join(td, "weights.u8"))
                mesh["world"].astype(np.float32).tofile(os.path.join(td, "world.f32"))
                mesh["inv_bind"].astype(np.float32).tofile(os.path.join(td, "invbind.f32"))
                if d is not None:
                    d.tofile(os.path.join(td, "deltas.f32"))
                    mw.astype(np.float32).tofile(os.path.join(td, "mw.f32"))
                out = subprocess.check_output(
                    [node, js, td, str(n), str(len(mesh["world"])),
                     str(0 if d is None else d.shape[0]), str(cores), "15"],
                    stderr=subprocess.STDOUT, timeout=180).decode()
            r = json.loads(out.strip().splitlines()[-1])
            return {"value": r["verts_per_s"], "unit": "verts/s", "cores": r["threads"], "kind": "port",
                    "sample": "%d verts x %d morphs, %d frames, JavaScript f32 skin (oracle/js) on %d worker_threads; "
                              "single-thread %.3g verts/s" % (n, 0 if d is None else d.shape[0], r["frames"],
                                                               r["threads"], r["single_thread_verts_per_s"])}
        except Exception as e:   # fall through to the C port, say why
            sys.stderr.write("node cpu baseline failed (%s); using the C oracle\n" % e)
    frames = 0
    t0 = time.perf_counter()
    while True:
        oracle.deform(sub["pos"], sub["nrm"], sub["joints"], sub["weights"], mesh["world"], mesh["inv_bind"],
                      d, mw, threads=cores)
        frames += 1
        el = time.perf_counter() - t0
        if el > 10.0 or frames >= 50:
            break
    return {"value": n * frames / el, "unit": "verts/s", "cores": cores, "kind": "port",
            "sample": "%d verts x %d morphs, %d frames, threaded C oracle (C stand-in for the TypeScript baseline)"
                      % (n, 0 if d is None else d.shape[0], frames)}


def main():
    args = parse_args()
    if args.config == "c4":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 0, 256
    elif args.config == "c3":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 64, 1
    elif args.config == "c2":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 0, 1
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and world_size > 1:
        args.gpus = world_size
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    dist = None
    # REZE_BENCH_FORCE_DIST=1 exercises the torch.distributed (RCCL) path with a single rank
    if world_size > 1 or os.environ.get("REZE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)

    import reze_engine_amd as rz
    from reze_engine_amd import synth

    V_total = args.verts * (world_size if args.scaling == "weak" else 1)
    B, M, I = args.bones, args.morphs, args.instances
    b, n, _chunk = rz.shard.shard_of(V_total, world_size, rank)

    # every rank generates the same full mesh deterministically and keeps its shard
    mesh = synth.make_mesh(V_total, B)
    if M > 0:
        deltas_full, mw = synth.make_morphs_dense(V_total, M)
    else:
        deltas_full, mw = None, None
    shard, deltas = rz.shard.cut_mesh(mesh, deltas_full, b, n)

    ctx = rz.DeformContext(local_rank)
    ctx.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    if deltas is not None:
        ctx.upload_morphs_dense(deltas)
    worlds = mesh["world"]
    mws = mw
    if I > 1:
        ctx.set_instances(I)
        worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
        if mw is not None:
            mws = np.tile(mw, (I, 1))
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        ctx.set_tuning(**{k: int(v)})
    quats = None
    if args.device_fk:
        rng = np.random.default_rng(4242)
        quats = rng.normal(size=(I, B, 4)).astype(np.float32)
        quats /= np.linalg.norm(quats, axis=2, keepdims=True)
        ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])

    frames = None
    if args.device_sampling:
        if not args.device_fk:
            raise SystemExit("--device-sampling needs --device-fk")
        rng = np.random.default_rng(777)
        nk = 8                                              # keys per bone, every 10 frames, default (identity) curves
        kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
        kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        ctx.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq,
                             (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2, np.tile(np.array([20] * 8 + [107] * 8, np.uint8), B * nk))
        frames = rng.random(I).astype(np.float32) * 70.0
        tick = [0]

    def put_pose():
        if frames is not None:
            tick[0] += 1
            ctx.set_pose_sampled((frames + 0.5 * tick[0]) % 70.0)
        elif quats is not None:
            ctx.set_pose_local(quats, mws)
        else:
            ctx.set_pose(worlds, mws)
    put_pose()
    tuned = None
    if not args.no_autotune and not args.tune:
        try:
            tuned = ctx.autotune()      # setup-time search over launch shapes (untimed, like a GEMM library's find mode)
        except Exception as e:          # noqa: BLE001  (the heuristics are a complete fallback)
            sys.stderr.write("[bench] autotune failed, using the heuristics: %r\n" % (e,))
            ctx.set_tuning(morph_split=0, grid_cap=0, inst_loop=-1)

    if args.graph:
        ctx.set_tuning(graph=1)

    def barrier():
        ctx.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warmup, then EXACTLY K timed steps between barrier + synchronize on both sides ----
    # Each rank stamps t1 when ITS K steps have drained (stream sync + torch.cuda.synchronize()), the closing
    # barrier follows, and the reported time is the MAX over ranks: the wall time until the slowest GPU finished,
    # without charging the collective latency of the closing barrier itself to an 18-us-per-step workload.
    ctx.deform_n(args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.deform_n(args.steps)
    ctx.sync()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: HIP events on the context's own stream ----
    timing = ctx.time_frames(max(20, min(args.steps, 200)))
    kern_s = timing["deform_kernel_ms"] * 1e-3
    achieved = timing["algorithmic_bytes_per_frame"] / kern_s / 1e9
    traffic = None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            key = "V%d_B%d_M%d_I%d" % (n, B, M, I)
            if key in rec:
                traffic = rec[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    # per-frame pose upload included (PCIe-inclusive rate; never `value`)
    # (secondary numbers never take the line down with them: a failure here is reported as null)
    n_up = min(args.steps, 200)
    with_upload_ms = None
    try:
        for _ in range(400):            # the upload path's own warm-up: pinned ring, upload stream (the HIP runtime
                                        # stalls ~25 ms once, somewhere in the first few hundred two-stream frames)
            put_pose()
            ctx.deform()
        ctx.sync()
        tp0 = time.perf_counter()
        for _ in range(n_up):
            put_pose()
            ctx.deform()
        ctx.sync()
        with_upload_ms = (time.perf_counter() - tp0) * 1e3 / n_up
    except Exception as e:              # noqa: BLE001
        sys.stderr.write("[bench] per-frame upload timing failed: %r\n" % (e,))

    ag_ms = None
    if args.allgather and I == 1:
        uid = [rz.capi.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world_size, rank, uid[0], V_total)
        for _ in range(5):
            ctx.allgather()
        barrier()
        ta = time.perf_counter()
        for _ in range(50):
            ctx.allgather()
        barrier()
        ag_ms = (time.perf_counter() - ta) * 1e3 / 50

    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, mesh, deltas_full, mw)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] cpu baseline failed: %r\n" % (e,))

    if rank == 0:
        verts = V_total * I * args.steps
        out = {
            "metric": "deformed verts/sec at 1/2/4/8 GPU; achieved HBM GB/s vs ~8 TB/s roofline",
            "value": verts / elapsed,
            "unit": "verts/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d-vert / %d-bone / %d-dense-morph synthetic PMX%s, vertex-sharded over %d GPU(s)"
                            % ({(1000000, 256, 64, 1): "C5", (30000, 200, 0, 256): "C4", (30000, 200, 64, 1): "C3",
                                (30000, 200, 0, 1): "C2"}.get((V_total, B, M, I), "custom"), V_total, B, M,
                               (" x %d instances (per-instance palette in LDS)" % I) if I > 1 else "", world_size),
                "verts_total": V_total, "verts_per_gpu": n, "bones": B, "morphs": M, "instances": I,
                "parallelism": "vertex-shard x%d" % world_size,
                "bone_hierarchy_solve": ("device (motion sampling + hierarchy solve in rz_fk_kernel)" if args.device_sampling else "device (rz_fk_kernel)") if args.device_fk else "host",
                "autotune": tuned is not None,
                "graph_replay": bool(args.graph),
                "morph_split": ctx.get_tuning("effective_split"),
                "grid": ctx.get_tuning("effective_grid"),
                "frame_ms_events": timing["frame_ms"],
                "prep_kernel_ms": timing["prep_kernel_ms"],
                "frame_ms_with_pose_upload": with_upload_ms,
                "allgather_ms": ag_ms,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rz_deform_kernel (fused morph+skin)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": timing["algorithmic_bytes_per_frame"],
                "kernel_ms": timing["deform_kernel_ms"],
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
