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
    ap.add_argument("--clock-warm-seconds", type=float, default=2.5, help="untimed setup: run frames this long before the warmup steps so the GPU is at its sustained clocks")
    ap.add_argument("--frames-in-flight", choices=["auto", "1", "2"], default="auto",
                    help="2: the timed steps alternate between the context and a fork of it (rz_fork: shared static data, own stream and outputs), so the tail of frame f "
                         "overlaps the ramp of frame f + 1. auto (default): an untimed calibration during the warm-up picks 2 only when it is >= 3 %% faster on every rank's clock "
                         "(small frames: shards at N >= 4, single characters); the C5 frame at N = 1 stays on one stream")
    ap.add_argument("--no-pair-loop", action="store_true", help="skip the secondary loop with two frames in flight")
    ap.add_argument("--no-sampled-loop", action="store_true", help="skip the secondary per-frame loop with the motion sampled on the GPU")
    ap.add_argument("--no-autotune", action="store_true", help="skip rz_autotune (setup-time search over launch shapes) and use the built-in heuristics")
    return ap.parse_args()


def cpu_baseline(args, mesh, deltas, mw):
    """Bounded CPU sample of the same workload: first `cpu_sample_verts` vertices, all morphs.
    Preferred: the all-core JavaScript f32 skin (oracle/js/cpu_baseline.js, worker_threads).
    Fallback: the threaded C oracle, labelled as a stand-in."""
    import oracle
    n = min(args.cpu_sample_verts, len(mesh["pos"]))     # rank 0's shard = the head of the mesh
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

    # every rank generates ITS OWN shard of the same block-seeded mesh (synth.make_mesh_range): an 8-rank node never
    # builds eight copies of the 1 M-vertex mesh + 768 MB of morph targets on the host, and N = 1 ... 8 deform the same mesh
    shard = synth.make_mesh_range(V_total, B, b, n)
    deltas, mw = synth.make_morphs_dense_range(V_total, M, b, n) if M > 0 else (None, None)
    mesh = shard                         # skeleton / pose fields are the same on every rank

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

    def upload_motion():
        """A synthetic motion (8 keys per bone, every 10 frames, default curves) for the device sampler."""
        rng = np.random.default_rng(777)
        nk = 8
        kq = rng.normal(size=(B, nk, 4)).astype(np.float32)
        kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        extra = {}
        if M > 0:                           # every vertex morph keyed at its bench weight: the sampled frame streams the same 64 targets
            extra = dict(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 70.0], np.float32), M), mkey_weight=np.repeat(mw, 2),
                         feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
        ctx.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq,
                             (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2, np.tile(np.array([20] * 8 + [107] * 8, np.uint8), B * nk), **extra)
        return rng.random(I).astype(np.float32) * 70.0

    frames = None
    tick = [0]
    if args.device_sampling:
        if not args.device_fk:
            raise SystemExit("--device-sampling needs --device-fk")
        frames = upload_motion()

    def put_pose(c=None):
        c = ctx if c is None else c
        if frames is not None:
            tick[0] += 1
            c.set_pose_sampled((frames + 0.5 * tick[0]) % 70.0)
        elif quats is not None:
            c.set_pose_local(quats, mws)
        else:
            c.set_pose(worlds, mws)

    def frame_table(fr):
        """64 rows of per-instance frame numbers marching through the 70-frame motion (cycled by the per-frame loop)."""
        return np.stack([(fr + 0.5 * k) % 70.0 for k in range(64)]).astype(np.float32)
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

    # Setup, untimed: bring the GPU to its sustained clock / power state. Measured on MI355X: the first ~2 s of work after
    # idle run 7-8 % slower (C4 one-launch frame 38.5 us cold, 35.8 us after 3 s of frames), and the driver may ask for as
    # few as 20 timed steps. A long-running job lives in the warm state; this is the same kind of step as rz_autotune.
    if args.clock_warm_seconds > 0:
        t_end = time.perf_counter() + args.clock_warm_seconds
        while time.perf_counter() < t_end:
            ctx.deform_n(200)
            ctx.sync()

    def barrier():
        ctx.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def rank_max(x):
        """max over ranks of a host-side scalar (None stays None when every rank has None)."""
        if dist is None:
            return x
        vals = [None] * world_size
        dist.all_gather_object(vals, x)
        vals = [v for v in vals if v is not None]
        return max(vals) if vals else None

    # ---- warmup, then EXACTLY K timed steps between barrier + synchronize on both sides ----
    # Each rank stamps t1 when ITS K steps have drained (stream sync + torch.cuda.synchronize()), the closing
    # barrier follows, and the reported time is the MAX over ranks: the wall time until the slowest GPU finished,
    # without charging the collective latency of the closing barrier itself to an 18-us-per-step workload.
    # Two frames in flight (rz_fork): a second context that borrows this one's static buffers and owns its stream, pose
    # slots and outputs; frames alternate between the two, so the tail of frame f overlaps the launch ramp of frame f + 1 —
    # what a WebGPU queue does with consecutive command buffers. The roofline below is always that of ONE kernel on one
    # stream (rz_time_frames); with --frames-in-flight 2 the timed steps themselves alternate.
    fork = None
    if (args.frames_in_flight != "1" or not args.no_pair_loop) and not args.allgather:
        try:
            fork = ctx.fork()
            put_pose(fork)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] rz_fork failed: %r\n" % (e,))
            fork = None

    def timed(run, sync_all):
        barrier()
        t0 = time.perf_counter()
        run()
        sync_all()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def sync_pair():
        ctx.sync()
        fork.sync()

    # untimed calibration (part of the warm-up, like rz_autotune): the same number of frames in both modes, three rounds,
    # best of each, max over ranks; two frames in flight must win by 3 % to be chosen
    calib = None
    in_flight = 1
    if fork is not None:
        if args.frames_in_flight == "2":
            in_flight = 2
        elif args.frames_in_flight == "auto":
            nc = max(20, min(args.steps, 200))
            t1 = min(timed(lambda: ctx.deform_n(nc), ctx.sync) for _ in range(3)) / nc * 1e3
            t2 = min(timed(lambda: ctx.deform_pair(fork, nc), sync_pair) for _ in range(3)) / nc * 1e3
            in_flight = 2 if t2 < 0.97 * t1 else 1
            calib = {"frames": nc, "one_stream_ms": t1, "two_in_flight_ms": t2, "rule": "two in flight when >= 3 % faster (max over ranks, best of 3)"}
    one_stream = lambda: timed(lambda: ctx.deform_n(args.steps), ctx.sync)                     # noqa: E731
    paired = lambda: timed(lambda: ctx.deform_pair(fork, args.steps), sync_pair)                # noqa: E731
    other_ms = None
    if in_flight == 2:
        ctx.deform_pair(fork, args.warmup)
        sync_pair()
        elapsed = paired()
        if not args.no_pair_loop:      # secondary: the same K steps on one stream
            ctx.deform_n(args.warmup)
            other_ms = one_stream() / args.steps * 1e3
    else:
        ctx.deform_n(args.warmup)
        elapsed = one_stream()
        if fork is not None and not args.no_pair_loop:      # secondary: the same K steps with two frames in flight
            ctx.deform_pair(fork, args.warmup)
            sync_pair()
            other_ms = paired() / args.steps * 1e3
    if fork is not None:
        fork.close()
        fork = None

    # ---- roofline of the dominant kernel: HIP events on the context's own stream ----
    timing = ctx.time_frames(max(20, min(args.steps, 200)))
    kern_s = timing["deform_kernel_ms"] * 1e-3
    achieved = timing["algorithmic_bytes_per_frame"] / kern_s / 1e9
    kernel_name = ctx.kernel_name()
    # HBM bytes per launch from the PMC counters are collected OFFLINE (rocprofv3 cannot run inside this process):
    # tools/gpu_profile.sh + tools/parse_prof.py store them per workload shape; this line only looks the shape up.
    traffic, traffic_source = None, None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            key = "V%d_B%d_M%d_I%d" % (n, B, M, I)
            if key in rec:
                traffic = rec[key]["hbm_bytes_per_launch"]
                traffic_source = "stored: profiles/pmc_traffic.json[%s] (%s)" % (key, rec[key].get("command", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"))
        except Exception:
            traffic = None
    if traffic is None:
        traffic_source = "none stored for this workload shape (profiles/pmc_traffic.json)"
    # the streaming ceiling measured on this kind of box (tools/membench): reads for the morph-stream frames, writes for crowds
    ceiling, ceiling_what = None, None
    cj = os.path.join(ROOT, "profiles", "ceilings.json")
    if os.path.exists(cj):
        try:
            cz = json.load(open(cj))
            which = "write_GBps" if (M == 0 and I > 1) else "nt_read_GBps"
            ceiling, ceiling_what = cz[which], "%s (%s)" % (which, cz.get("source", "tools/membench"))
        except Exception:
            ceiling = None

    # per-frame pose upload included (PCIe-inclusive rate; never `value`)
    # (secondary numbers never take the line down with them: a failure here is reported as null)
    n_up = min(args.steps, 200)

    def per_frame_loop(frame, check):
        """frame() = one pose upload + rz_deform through the raw C ABI (DeformContext.frame_call: arrays and pointers are
        prepared up front, so the loop times the library and the GPU, not numpy conversions)."""
        try:
            for _ in range(400):        # the upload path's own warm-up: pinned ring, upload stream (the HIP runtime
                                        # stalls ~25 ms once, somewhere in the first few hundred two-stream frames)
                frame()
            barrier()
            tp0 = time.perf_counter()
            for _ in range(n_up):
                frame()
            ctx.sync()
            dt = (time.perf_counter() - tp0) * 1e3 / n_up
            check()
            return rank_max(dt)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] per-frame upload timing failed: %r\n" % (e,))
            return rank_max(None)
    if frames is not None:
        primary_call = ctx.frame_call("sampled", frame_table(frames))
    elif quats is not None:
        primary_call = ctx.frame_call("local", quats, mws)
    else:
        primary_call = ctx.frame_call("world", worlds, mws)
    with_upload_ms = per_frame_loop(*primary_call)
    # the same per-frame loop with two frames in flight: pose f + 1 is uploaded and its frame enqueued on the fork while frame f
    # still runs on the context (and the other way round)
    with_upload_pair_ms = None
    if not args.no_pair_loop and not args.allgather:
        fk2 = None
        try:
            fk2 = ctx.fork()
            kind = "sampled" if frames is not None else ("local" if quats is not None else "world")
            call_b = fk2.frame_call(kind, frame_table(frames)) if frames is not None else fk2.frame_call(kind, quats if quats is not None else worlds, mws)
            fa, fb, flip = primary_call[0], call_b[0], [0]

            def both():
                flip[0] ^= 1
                (fb if flip[0] else fa)()

            def check_both():
                fk2.sync()
                primary_call[1]()
                call_b[1]()
            with_upload_pair_ms = per_frame_loop(both, check_both)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] two-in-flight per-frame loop failed: %r\n" % (e,))
            with_upload_pair_ms = rank_max(None)
        finally:
            if fk2 is not None:
                fk2.close()
    # ... and the same loop when the motion lives on the GPU (rz_set_pose_sampled: ONE float per instance per frame, bones
    # sampled + hierarchy solved by rz_fk_kernel): the per-frame loop that does not pay for the pose upload at any N
    sampled_ms = None
    if not args.no_sampled_loop:
        try:
            if frames is None:
                if not args.device_fk:
                    ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
                fr2 = upload_motion()
            else:
                fr2 = frames

            sampled_ms = per_frame_loop(*ctx.frame_call("sampled", frame_table(fr2)))
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] device-sampling loop failed: %r\n" % (e,))
            sampled_ms = rank_max(None)
        put_pose()                      # back to the primary pose kind

    ag_ms, rccl = None, None
    if args.allgather and I == 1:
        uid = [rz.capi.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world_size, rank, uid[0], V_total)
        rccl = rz.capi.rccl_info()
        for _ in range(5):
            ctx.allgather()
        barrier()
        ta = time.perf_counter()
        for _ in range(50):
            ctx.allgather()
        barrier()
        ag_ms = (time.perf_counter() - ta) * 1e3 / 50

    # one record per rank, so a slow or oddly planned GPU is visible in the scaling file
    mine = {"rank": rank, "device": local_rank, "verts": n, "kernel_ms": timing["deform_kernel_ms"], "frame_ms": timing["frame_ms"],
            "kernel": kernel_name, "grid": ctx.get_tuning("effective_grid"), "morph_split": ctx.get_tuning("effective_split"),
            "autotuned": tuned is not None, "rccl": rccl}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world_size
        dist.all_gather_object(per_rank, mine)

    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, mesh, deltas, mw)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] cpu baseline failed: %r\n" % (e,))

    if rank == 0:
        verts = V_total * I * args.steps
        kms = [r["kernel_ms"] for r in per_rank]
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
                "frame_ms_device_sampled_pose": sampled_ms,
                "frame_ms_with_pose_upload_two_in_flight": with_upload_pair_ms,
                "frames_in_flight": in_flight,
                "frames_in_flight_choice": ("--frames-in-flight " + args.frames_in_flight) if calib is None else calib,
                "ms_per_step_one_stream": elapsed / args.steps * 1e3 if in_flight == 1 else other_ms,
                "ms_per_step_two_frames_in_flight": other_ms if in_flight == 1 else elapsed / args.steps * 1e3,
                "frames_in_flight_note": "2 = frames alternate between the context and an rz_fork of it (shared static data, own stream + outputs): the tail of frame f overlaps the ramp of frame f + 1; roofline.* is always one kernel on one stream",
                "per_frame_loops": "max over ranks, raw C ABI calls; with_pose_upload = rz_set_pose%s + rz_deform per frame, device_sampled_pose = rz_set_pose_sampled (1 float / instance) + rz_deform"
                                   % ("_local" if args.device_fk else ""),
                "allgather_ms": ag_ms,
                "kernel_ms_min_over_ranks": min(kms), "kernel_ms_max_over_ranks": max(kms),
                "ranks": per_rank,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "measured_ceiling": ceiling,
                "measured_ceiling_what": ceiling_what,
                "frac_of_measured_ceiling": (achieved / ceiling) if ceiling else None,
                "algorithmic_bytes_per_launch": timing["algorithmic_bytes_per_frame"],
                "kernel_ms": timing["deform_kernel_ms"],
                "frame_frac": timing["algorithmic_bytes_per_frame"] / (timing["frame_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
            "cpu_baseline": cpu,
        }
    else:
        out = None
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if out is not None:
        # The JSON line must be the LAST thing on stdout: RCCL writes its version banner through C stdio, which sits in a
        # buffer until exit when stdout is a pipe — flush that first, then print.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
