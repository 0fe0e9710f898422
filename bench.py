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

`--gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment) re-executes itself under
`python -m torch.distributed.run --nproc-per-node N`: a plain `python bench.py --gpus 8` IS an 8-rank run. A launcher whose
world size differs from --gpus, or fewer visible GPUs than ranks (without --share-gpu), is an error, never a silent N = 1.
At N > 1 every rank joins an RCCL communicator built through the library's own entry points (rz_comm_init) and reports what
the communicator says about itself (`config.ranks[].rccl`); the all-gather of deformed positions is timed OUTSIDE `value`.
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
    ap.add_argument("--instances", type=int, default=1, help="C4-style instancing; with --gpus N the crowd is sharded along the INSTANCE axis: every rank "
                                                              "holds the whole mesh and poses ceil(I / N) of the instances, no collective, no communicator (SURVEY 8e)")
    ap.add_argument("--config", choices=["c5", "c4", "c3", "c2", "demo", "sparse2"], default=None,
                    help="BASELINE.json shortcut: c5 = 1M/256/64 (default), c4 = 256 x 30k/200/0, c3 = 30k/200/64, c2 = 30k/200/0; "
                         "demo = the demo model's shape: 28 842 verts / 349 bones / 60 SPARSE vertex morphs with its statistics (36 397 offsets, "
                         "largest 1 718, all on one 1 800-vertex face region); sparse2 = the same with the offsets spread at 2 %% density (SURVEY 8d)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --verts is the whole mesh, sharded over ranks; weak: --verts per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-verts", type=int, default=0,
                    help="cpu_baseline: vertices of the mesh the CPU skin runs on; 0 (default) = the WHOLE mesh the GPU line deforms "
                         "(BASELINE north star: 'CPU skin of the same mesh'), timed for ~10 s of frames")
    ap.add_argument("--allgather", action="store_true", help="time the RCCL all-gather of positions at N = 1 too (at N > 1 it always is, outside `value`)")
    ap.add_argument("--no-allgather", action="store_true", help="N > 1: skip the RCCL communicator + all-gather timing")
    ap.add_argument("--rccl-timeout", type=float, default=120.0, help="watchdog of the RCCL phase (communicator + all-gather), seconds")
    ap.add_argument("--device-fk", action="store_true",
                    help="solve the bone hierarchy on the GPU: frames start from local rotations (rz_set_pose_local)")
    ap.add_argument("--device-sampling", action="store_true",
                    help="with --device-fk: a synthetic motion is uploaded once and every frame sends ONE float per instance (rz_set_pose_sampled)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for the barrier / max-reduce (gloo + --share-gpu lets a 1-GPU box rehearse N > 1)")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank onto GPU (local_rank %% visible devices)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="with --share-gpu: run the RCCL phase all the same. ncclCommInitRank refuses ranks that share a GPU, so this rehearses what a FAILING "
                         "communicator does to the line (it must still appear, with ranks[].rccl.error) on a box that cannot form a working one")
    ap.add_argument("--graph", action="store_true", help="replay captured hipGraphs of 16 frames in the timed loop (rz_set_tuning graph=1): for launch-bound small frames")
    ap.add_argument("--tune", default="", help="comma list key=value passed to rz_set_tuning (disables the autotune pass)")
    ap.add_argument("--clock-warm-seconds", type=float, default=2.5, help="untimed setup: run frames this long before the warmup steps so the GPU is at its sustained clocks")
    ap.add_argument("--frames-in-flight", choices=["auto", "1", "2"], default="auto",
                    help="2: the timed steps alternate between the context and a fork of it (rz_fork: shared static data, own stream and outputs), so the tail of frame f "
                         "overlaps the ramp of frame f + 1. auto (default): an untimed calibration during the warm-up picks 2 only when it is >= 3 %% faster on every rank's clock "
                         "(small frames: shards at N >= 4, single characters); the C5 frame at N = 1 stays on one stream")
    ap.add_argument("--no-pair-loop", action="store_true", help="skip the secondary loop with two frames in flight")
    ap.add_argument("--no-sampled-loop", action="store_true", help="skip the secondary per-frame loop with the motion sampled on the GPU")
    ap.add_argument("--no-numa-bind", action="store_true", help="leave the process where the launcher put it instead of binding it to the cores of its GPU's NUMA node")
    ap.add_argument("--no-autotune", action="store_true", help="skip rz_autotune (setup-time search over launch shapes) and use the built-in heuristics")
    return ap.parse_args()


def cpu_baseline(args, mesh, deltas, mw, sparse=None):
    """CPU baseline on the same workload: the whole mesh (or, with --cpu-sample-verts, its first vertices), all morphs, a bounded
    number of frames (~10 s).
    Preferred: the all-core JavaScript f32 skin (oracle/js/cpu_baseline.js, worker_threads).
    Fallback: the threaded C oracle, labelled as a stand-in. Sparse-morph workloads use the C oracle's sparse accumulate
    (the JavaScript baseline only knows dense targets) followed by its threaded skin."""
    import oracle
    if sparse is not None:
        cores = os.cpu_count() or 1
        V = len(mesh["pos"])
        frames = 0
        t0 = time.perf_counter()
        while True:
            pm = oracle.morph_sparse(V, sparse[0], sparse[1], sparse[2], mw, mesh["pos"])
            oracle.deform(pm, mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"], mesh["inv_bind"], None, None, threads=cores)
            frames += 1
            el = time.perf_counter() - t0
            if el > 10.0 or frames >= 2000:
                break
        return {"value": V * frames / el, "unit": "verts/s", "cores": cores, "kind": "port",
                "sample": "%d verts x %d sparse morphs (%d offsets), %d frames, C oracle: sparse morph accumulate on one thread + skin on %d threads "
                          "(C stand-in for the TypeScript baseline)" % (V, len(sparse[0]) - 1, int(sparse[0][-1]), frames, cores)}
    n = len(mesh["pos"]) if args.cpu_sample_verts <= 0 else min(args.cpu_sample_verts, len(mesh["pos"]))     # rank 0's shard = the head of the mesh
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
                     str(0 if d is None else d.shape[0]), str(cores), "10"],
                    stderr=subprocess.STDOUT, timeout=300).decode()
            r = json.loads(out.strip().splitlines()[-1])
            return {"value": r["verts_per_s"], "unit": "verts/s", "cores": r["threads"], "kind": "port",
                    "sample": "%d of the mesh's %d verts x %d morphs, %d timed frames, JavaScript f32 skin (oracle/js) on %d worker_threads; "
                              "single-thread %.3g verts/s" % (n, len(mesh["pos"]), 0 if d is None else d.shape[0], r["frames"],
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


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: become one. Re-executes this script under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU, and exits with its
    status — so a plain invocation can never quietly measure one rank and call it N."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: re-executing as %d ranks: %s\n" % (args.gpus, args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env["REZE_BENCH_SELF_LAUNCHED"] = "1"
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.config == "c4":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 0, 256
    elif args.config == "c3":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 64, 1
    elif args.config == "c2":
        args.verts, args.bones, args.morphs, args.instances = 30000, 200, 0, 1
    sparse_kind = None
    if args.config in ("demo", "sparse2"):
        args.verts, args.bones, args.morphs, args.instances = 28842, 349, 60, 1
        sparse_kind = args.config
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)               # never returns
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        # a launcher with another world size than --gpus (or --gpus left at its default under a launcher): refuse, do not guess
        if rank == 0:
            sys.stderr.write("[bench] --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line for a run "
                             "that is not the one asked for\n" % (args.gpus, world_size))
        sys.exit(3)

    import torch
    # a launcher may hand every rank ITS OWN GPU through *_VISIBLE_DEVICES (each rank then sees exactly one device, index 0)
    isolated = (world_size > 1 and torch.cuda.device_count() == 1 and
                any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")))
    if isolated:
        local_rank = 0
    elif world_size > 1 and not args.share_gpu and torch.cuda.device_count() < world_size:
        if rank == 0:
            sys.stderr.write("[bench] %d ranks but only %d GPU(s) visible: one process per GPU (--share-gpu rehearses N > 1 on fewer GPUs)\n"
                             % (world_size, torch.cuda.device_count()))
        sys.exit(4)
    dist = None
    # REZE_BENCH_FORCE_DIST=1 exercises the torch.distributed (RCCL) path with a single rank
    if world_size > 1 or os.environ.get("REZE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = local_rank % max(1, torch.cuda.device_count())
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)
    if dist is not None and world_size > 1 and not args.share_gpu:
        # one PHYSICAL GPU per rank: two ranks that resolve to the same device (a global *_VISIBLE_DEVICES = 0 on a one-GPU box
        # looks like per-rank isolation from inside one rank) would quietly halve each other's numbers
        pr = torch.cuda.get_device_properties(local_rank)
        me = (str(getattr(pr, "uuid", "")), getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None))
        ids = [None] * world_size
        dist.all_gather_object(ids, me)
        # an identity that says nothing (no uuid, PCI address all zero — some virtualised nodes) cannot prove a collision: ranks that
        # index distinct devices of one visible list are distinct GPUs by construction, so only an INFORMATIVE identity seen twice refuses
        def informative(t):
            u = "".join(ch for ch in t[0] if ch.isalnum()).strip("0")
            return bool(u) or any(x not in (None, 0) for x in t[1:])
        if len(set(ids)) != world_size and (isolated or all(informative(t) for t in ids)):
            if rank == 0:
                sys.stderr.write("[bench] %d ranks but they resolve to %d distinct GPU(s): one process per GPU (--share-gpu rehearses N > 1 on fewer GPUs)\n"
                                 % (world_size, len(set(ids))))
            sys.exit(4)

    # Every rank runs on the cores of ITS GPU's NUMA node (what numactl --cpunodebind does; rz_device_numa_node): per-frame inputs cross
    # the host link, and a thread on the other socket pays for every HIP call and every pulled byte (profiles/r5_crowd_upload_numa.txt:
    # a host-animated C4 frame 63 us from the GPU's node, 80-82 us from the other). AFTER torch has brought its HIP runtime up (the
    # library must bind to the copy the process already carries — loading it first put a second runtime under torch's RCCL and
    # ncclCommInitRank failed); sched_setaffinity moves the calling thread, which makes every HIP call from here on and first-touches
    # the pinned rings, and the threads it starts.
    full_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa = None
    if not args.no_numa_bind and torch.cuda.is_available():
        try:
            from reze_engine_amd import capi as _capi
            numa = _capi.bind_to_device_node(local_rank)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] rank %d: no NUMA binding (%r)\n" % (rank, e))

    import reze_engine_amd as rz
    from reze_engine_amd import synth

    B, M = args.bones, args.morphs
    crowd = args.instances > 1
    if crowd:
        # BASELINE config 4 across GPUs: the crowd is cut along the INSTANCE axis (last sentence) — every rank holds the whole
        # static mesh and poses its own contiguous range of instances; nothing is exchanged, no communicator is made
        V_total = args.verts
        I_total = args.instances * (world_size if args.scaling == "weak" else 1)
        inst_begin, I = rz.shard.instances_of(I_total, world_size, rank)
        if I == 0:
            if rank == 0:
                sys.stderr.write("[bench] %d instances over %d ranks leaves ranks without work\n" % (I_total, world_size))
            sys.exit(5)
        b, n = 0, V_total
    else:
        V_total = args.verts * (world_size if args.scaling == "weak" else 1)
        I_total, inst_begin, I = 1, 0, 1
        b, n, _chunk = rz.shard.shard_of(V_total, world_size, rank)

    # every rank generates ITS OWN shard of the same block-seeded mesh (synth.make_mesh_range): an 8-rank node never
    # builds eight copies of the 1 M-vertex mesh + 768 MB of morph targets on the host, and N = 1 ... 8 deform the same mesh
    shard = synth.make_mesh_range(V_total, B, b, n)
    deltas, mw, sparse = None, None, None
    if sparse_kind is not None:
        # sparse vertex morphs (PMX's on-disk form, engine/src/pmx-loader.ts:483-488) of the WHOLE mesh, cut to this rank's range
        if sparse_kind == "demo":
            off, idx, d3, mw = synth.make_morphs_demo_shape(V_total, M)
        else:
            off, idx, d3, mw = synth.make_morphs_sparse(V_total, M, density=0.02)
        if world_size > 1:
            keep = (idx >= b) & (idx < b + n)
            off = np.concatenate([[0], np.cumsum([int(keep[off[m]:off[m + 1]].sum()) for m in range(M)])]).astype(np.uint32)
            idx, d3 = (idx[keep] - b).astype(np.uint32), d3[keep]
        sparse = (off, idx, d3)
    elif M > 0:
        deltas, mw = synth.make_morphs_dense_range(V_total, M, b, n)
    mesh = shard                         # skeleton / pose fields are the same on every rank

    ctx = rz.DeformContext(local_rank)
    ctx.upload_mesh(shard["pos"], shard["nrm"], shard["joints"], shard["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    if deltas is not None:
        ctx.upload_morphs_dense(deltas)
    if sparse is not None:
        ctx.upload_morphs_sparse(*sparse)
    worlds = mesh["world"]
    mws = mw
    if crowd:
        ctx.set_instances(I)
        worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + inst_begin + i) for i in range(I)])   # instance k has the same pose at every N
        if mw is not None:
            mws = np.tile(mw, (I, 1))
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        ctx.set_tuning(**{k: int(v)})
    quats = None
    if args.device_fk:
        rng = np.random.default_rng(4242)
        quats = rng.normal(size=(I_total, B, 4)).astype(np.float32)[inst_begin:inst_begin + I]
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
        return (rng.random(I_total).astype(np.float32) * 70.0)[inst_begin:inst_begin + I]

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

    def gather(x):
        """one entry per rank (a list of length 1 without torch.distributed)"""
        if dist is None:
            return [x]
        vals = [None] * world_size
        dist.all_gather_object(vals, x)
        return vals

    # ---- launch-shape search (untimed setup, like a GEMM library's find mode) ----
    # Every rank measures the same candidate list on its own shard; the tables are reduced with MAX over the ranks (the frame
    # time of a sharded mesh is its slowest GPU's) and every rank adopts the SAME entry: the heuristic plan unless a
    # candidate beats it by >= 2 % with non-overlapping round ranges (rz_autotune_pick). config.autotune_table carries the reduced table.
    def clock_warm(seconds):
        # Setup, untimed: bring the GPU to its sustained clock / power state. Measured on MI355X: the first ~2 s of work after
        # idle run 7-8 % slower (C4 one-launch frame 38.5 us cold, 35.8 us after 3 s of frames), and the driver may ask for as
        # few as 20 timed steps. A long-running job lives in the warm state; this is the same kind of step as rz_autotune.
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            ctx.deform_n(200)
            ctx.sync()

    tuned, tune_table, tune_pick = None, None, None
    if args.clock_warm_seconds > 0:
        clock_warm(args.clock_warm_seconds)          # BEFORE the search: a cold first candidate would lose to a warm last one
    if not args.no_autotune and not args.tune:
        try:
            mine_tab = ctx.autotune_measure()
        except Exception as e:          # noqa: BLE001  (the heuristics are a complete fallback)
            sys.stderr.write("[bench] autotune failed on rank %d, using the heuristics: %r\n" % (rank, e))
            mine_tab = None
        tabs = gather(mine_tab)
        key = lambda t: [(e["morph_split"], e["grid_cap"], e["inst_loop"]) for e in t]      # noqa: E731
        if all(t is not None for t in tabs) and all(key(t) == key(tabs[0]) for t in tabs):
            tune_table = [dict(e) for e in tabs[0]]
            for k, e in enumerate(tune_table):
                e["ms"] = max(t[k]["ms"] for t in tabs)
                e["ms_min"] = min(t[k]["ms_min"] for t in tabs)
                e["ms_max"] = max(t[k]["ms_max"] for t in tabs)
                if any(t[k]["same_as"] != e["same_as"] for t in tabs):
                    e["same_as"] = -1           # not the same launch on every rank: judged on its own
            tune_pick = ctx.autotune_pick(tune_table)
            ctx.autotune_apply(tune_table[tune_pick])
            tuned = {k: ctx.get_tuning(k) for k in ("effective_split", "effective_grid", "effective_inst_group")}
        else:
            ctx.set_tuning(morph_split=0, grid_cap=0, inst_loop=-1)

    if args.graph:
        ctx.set_tuning(graph=1)

    if args.clock_warm_seconds > 0:
        clock_warm(min(0.5, args.clock_warm_seconds))      # the adopted plan, warm

    def barrier():
        ctx.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def rank_max(x):
        """max over ranks of a host-side scalar (None stays None when every rank has None)."""
        vals = [v for v in gather(x) if v is not None]
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
    if args.frames_in_flight != "1" or not args.no_pair_loop:
        try:
            fork = ctx.fork()
            put_pose(fork)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] rz_fork failed on rank %d: %r\n" % (rank, e))
            if fork is not None:
                fork.close()
            fork = None
        # every rank takes the same code path below (the timed loops contain collectives): one rank without a fork = nobody forks
        if not all(gather(fork is not None)):
            if fork is not None:
                fork.close()
            fork = None

    # What is timed, and by which clock (round 6). The K steps sit between barrier + synchronize on both sides, as the contract says. Inside
    # that bracket they are timed TWICE: by a hipEvent pair on the stream the frames run on (rz_time_span; SURVEY 8d prescribes event pairs
    # around the back-to-back frames) and by the host's clock around the same call + synchronize. The host clock carries a fixed cost per
    # timed region — the first launch, the wake-up from the final wait: ~30 us (profiles/r6_bench_shard8_steps20.json) — which is nothing
    # in 500 steps of C5 and 9 % of the driver's 20 steps of a 16.6 us shard frame. `ms_per_step` / `value` are the event span; the host
    # wall is reported beside it (config.ms_per_step_host_wall, config.host_fixed_cost_us_per_timed_region). Both are MAX over ranks.
    # LEAD untimed frames sit in front of the opening event in the same call (rz_time_span): the K timed steps are enqueued while the GPU
    # is busy with them and run back to back from the first one, as in a render loop — without them the first timed step waits ~5 us
    # behind the event for its own launch. The host wall covers LEAD + K frames and is divided by LEAD + K.
    LEAD = 2

    def timed(span_call, sync_all):
        barrier()
        t0 = time.perf_counter()
        span_ms = span_call()           # enqueues exactly the K steps between two events and blocks until the closing event
        sync_all()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        span = span_ms * 1e-3
        if dist is not None:
            dist.barrier()
            t = torch.tensor([span, wall], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            span, wall = float(t[0].item()), float(t[1].item())
        return span, wall

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
            nc = 200                    # (not a function of --steps: the driver's 20 steps must not decide the mode from 20 frames)
            t1 = min(timed(lambda: ctx.time_span(nc, lead=LEAD), ctx.sync)[0] for _ in range(3)) / nc * 1e3
            t2 = min(timed(lambda: ctx.time_span(nc, fork, lead=LEAD), sync_pair)[0] for _ in range(3)) / nc * 1e3
            in_flight = 2 if t2 < 0.97 * t1 else 1
            calib = {"frames": nc, "one_stream_ms": t1, "two_in_flight_ms": t2, "rule": "two in flight when >= 3 % faster (max over ranks, best of 3)"}

    def one_stream():
        ctx.deform_n(args.warmup)
        return timed(lambda: ctx.time_span(args.steps, lead=LEAD), ctx.sync)

    def paired():
        ctx.deform_pair(fork, args.warmup)
        sync_pair()
        return timed(lambda: ctx.time_span(args.steps, fork, lead=LEAD), sync_pair)

    kernel_reps = []

    def kernel_timing():
        """roofline of the dominant kernel: HIP events on the context's own stream, one kernel at a time — SURVEY 8d's protocol: event pairs
        around 200 back-to-back frames (whatever --steps is), the MEDIAN of 5 repetitions"""
        reps = sorted((ctx.time_frames(200) for _ in range(5)), key=lambda t: t["deform_kernel_ms"])
        kernel_reps[:] = [t["deform_kernel_ms"] for t in reps]
        return reps[2]

    # Order: the K steps on ONE stream (the timed steps themselves, or — when two frames in flight were chosen — the
    # secondary number), then the kernel's own timing straight after them, in the same state of the GPU, then anything that
    # runs two kernels at once. A kernel cannot take longer than the step that contains it: if the event timing disagrees
    # with the one-stream step by more than 3 % it is measured again, once, and the line says so.
    one_ms, pair_ms, pair_wall = None, None, None
    one_el, one_wall = one_stream()     # always: it is the headline whenever the overlapped step would be shorter than its own kernel
    one_ms = one_el / args.steps * 1e3
    timing = kernel_timing()
    kcheck = {"rule": "kernel_ms <= 1.03 x ms_per_step_one_stream", "remeasured": False}
    if one_ms is not None and timing["deform_kernel_ms"] > 1.03 * one_ms:
        timing = kernel_timing()
        kcheck["remeasured"] = True
    kcheck["ok"] = one_ms is None or timing["deform_kernel_ms"] <= 1.03 * one_ms
    if fork is not None and (in_flight == 2 or not args.no_pair_loop):
        pair_el, pair_wall = paired()
        pair_ms = pair_el / args.steps * 1e3
    # Headline rule (round 5): a step cannot be shorter than the kernel it contains. When two frames in flight bring the step UNDER the
    # event-timed kernel (small frames: C3 4.4 us per step against a 6.4 us kernel) that number is throughput of two overlapped frames,
    # not the time of a frame: `value` / `ms_per_step` then stay the ONE-STREAM loop and the overlap is reported beside it
    # (config.ms_per_step_two_frames_in_flight). Two in flight remains the headline only where it wins AND the step still holds its kernel.
    # "Where it wins" is decided by the K timed steps, not by the calibration alone: a calibration that promised >= 3 % and a timed
    # pair loop that then ran SLOWER than the timed one-stream loop (seen on C4 --device-fk: 34.5 us one stream, 37.0 two in flight)
    # leaves the one-stream loop as the headline.
    headline_in_flight = in_flight
    # (round 6: "faster" = by at least 1 % — a C4 line came out "2 in flight" at 33.58 us against 33.62 one stream, which says nothing)
    if in_flight == 2 and pair_ms is not None and (pair_ms < timing["deform_kernel_ms"] or pair_ms > 0.99 * one_ms):
        headline_in_flight = 1
    elapsed = pair_el if headline_in_flight == 2 else one_el
    elapsed_wall = pair_wall if headline_in_flight == 2 else one_wall
    if fork is not None:
        fork.close()
        fork = None

    kern_s = timing["deform_kernel_ms"] * 1e-3
    achieved = timing["algorithmic_bytes_per_frame"] / kern_s / 1e9
    kernel_name = ctx.kernel_name()
    # HBM bytes per launch from the PMC counters are collected OFFLINE (rocprofv3 cannot run inside this process):
    # tools/gpu_profile.sh + tools/parse_prof.py store them per (workload shape, kernel); this line looks up the kernel the
    # plan ACTUALLY launches and carries no traffic figure when that kernel was never measured.
    traffic, traffic_source = None, None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    shape_key = "V%d_B%d_M%d_I%d%s" % (n, B, M, I, "" if sparse_kind is None else "_" + sparse_kind)
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj))
            key = shape_key + "|" + kernel_name
            if key in rec:
                traffic = rec[key]["hbm_bytes_per_launch"]
                traffic_source = "stored: profiles/pmc_traffic.json[%s] (%s)" % (key, rec[key].get("command", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"))
            else:
                others = sorted(k.split("|", 1)[1] for k in rec if k.startswith(shape_key + "|"))
                traffic_source = "none stored for this kernel on this workload shape (profiles/pmc_traffic.json has %s)" % (others or "nothing for the shape")
        except Exception:
            traffic = None
    if traffic_source is None:
        traffic_source = "profiles/pmc_traffic.json missing"
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

    def per_frame_loop(frame, check, sync_all=None):
        """frame() = one pose upload + rz_deform through the raw C ABI (DeformContext.frame_call: arrays and pointers are
        prepared up front, so the loop times the library and the GPU, not numpy conversions). The clock stops when EVERY
        context the loop fed has drained (sync_all)."""
        sync_all = sync_all or ctx.sync
        dt = None
        try:
            for _ in range(400):        # the upload path's own warm-up: pinned ring, upload stream (the HIP runtime
                                        # stalls ~25 ms once, somewhere in the first few hundred two-stream frames)
                frame()
            sync_all()
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] per-frame loop warm-up failed on rank %d: %r\n" % (rank, e))
            frame = None
        if not all(gather(frame is not None)):      # the barrier below is a collective: all ranks or none
            return None
        try:
            barrier()
            tp0 = time.perf_counter()
            for _ in range(n_up):
                frame()
            sync_all()
            dt = (time.perf_counter() - tp0) * 1e3 / n_up
            check()
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] per-frame upload timing failed on rank %d: %r\n" % (rank, e))
            dt = None
        vals = gather(dt)
        return None if any(v is None for v in vals) else max(vals)
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
    if not args.no_pair_loop:
        fk2, call_b = None, None
        try:
            fk2 = ctx.fork()
            kind = "sampled" if frames is not None else ("local" if quats is not None else "world")
            call_b = fk2.frame_call(kind, frame_table(frames)) if frames is not None else fk2.frame_call(kind, quats if quats is not None else worlds, mws)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] two-in-flight per-frame loop: rz_fork failed on rank %d: %r\n" % (rank, e))
            call_b = None
        if all(gather(call_b is not None)):
            fa, fb, flip = primary_call[0], call_b[0], [0]

            def both():
                flip[0] ^= 1
                (fb if flip[0] else fa)()

            def check_both():
                primary_call[1]()
                call_b[1]()

            def sync_both():
                ctx.sync()
                fk2.sync()
            with_upload_pair_ms = per_frame_loop(both, check_both, sync_both)
        if fk2 is not None:
            fk2.close()
    # Caller-written poses (rz_map_pose / rz_commit_pose, ABI 7), crowds only: the per-frame loop of a host that writes its world matrices
    # straight into the pinned ring slot the pull kernel reads — 48 B per bone, no pack, no copy in the library. `fill` stands in for the
    # caller's pose solve: one memmove of the pre-packed rows into the slot (a solve writes them there in the first place);
    # _protocol_only writes nothing and is what the library itself costs per frame.
    mapped_ms, mapped_bare_ms, mapped_pair_ms = None, None, None
    if crowd and quats is None and frames is None and hasattr(ctx, "mapped_frame_call"):
        import ctypes
        rows = np.ascontiguousarray(worlds.reshape(I, B, 4, 4)[:, :, :, :3], dtype=np.float32)
        src = rows.ctypes.data

        def fill(dst, nbytes):
            ctypes.memmove(dst, src, nbytes)
        try:
            ROWS = rz.capi.POSE_ROWS12
            mapped_ms = per_frame_loop(*ctx.mapped_frame_call(ROWS, fill))
            mapped_bare_ms = per_frame_loop(*ctx.mapped_frame_call(ROWS, None))
            put_pose()                  # (the bare loop left whatever the ring slots held as the resident pose)
            if not args.no_pair_loop:
                fk3 = ctx.fork()
                ca, cb, flip3 = ctx.mapped_frame_call(ROWS, fill), fk3.mapped_frame_call(ROWS, fill), [0]

                def both3():
                    flip3[0] ^= 1
                    (cb if flip3[0] else ca)[0]()

                def check3():
                    ca[1]()
                    cb[1]()
                mapped_pair_ms = per_frame_loop(both3, check3, lambda: (ctx.sync(), fk3.sync()))
                fk3.close()
        except Exception as e:          # noqa: BLE001  (secondary numbers never take the line down)
            sys.stderr.write("[bench] mapped-pose loop failed on rank %d: %r\n" % (rank, e))
        # every rank ran the same collectives inside per_frame_loop or raised before the first: a rank that failed reports None
        if not all(gather(mapped_ms is not None)):
            mapped_ms = mapped_bare_ms = mapped_pair_ms = None
    # ... and the same loop when the motion lives on the GPU (rz_set_pose_sampled: ONE float per instance per frame, bones
    # sampled + hierarchy solved by rz_fk_kernel): the per-frame loop that does not pay for the pose upload at any N
    sampled_ms = None
    if not args.no_sampled_loop:
        sampled_call = None
        try:
            if frames is None:
                if not args.device_fk:
                    ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
                fr2 = upload_motion()
            else:
                fr2 = frames
            sampled_call = ctx.frame_call("sampled", frame_table(fr2))
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] device-sampling loop setup failed on rank %d: %r\n" % (rank, e))
        if all(gather(sampled_call is not None)):
            sampled_ms = per_frame_loop(*sampled_call)
        put_pose()                      # back to the primary pose kind

    # one record per rank, so a slow or oddly planned GPU is visible in the scaling file — gathered BEFORE the RCCL phase below,
    # so that the line has them whatever happens there
    eff = {k: ctx.get_tuning(k) for k in ("effective_split", "effective_grid", "effective_inst_group", "effective_subsets", "effective_subset_bones", "effective_inst_lds")}
    mine = {"rank": rank, "device": local_rank, "verts": n, "instances": I, "instance_begin": inst_begin, "kernel_ms": timing["deform_kernel_ms"], "frame_ms": timing["frame_ms"],
            "kernel": kernel_name, "grid": eff["effective_grid"], "morph_split": eff["effective_split"], "autotuned": tuned is not None, "rccl": None}
    per_rank = gather(mine)

    # ---- RCCL: at N > 1 every rank joins a communicator made by the library's own entry points, says what the
    # communicator reports about itself, and the all-gather of deformed positions is timed — OUTSIDE `value`.
    # It runs LAST and under a watchdog: a multi-GPU collective is the one thing a 1-GPU box cannot rehearse, and a hang in it
    # must cost the line its RCCL fields, not the line itself. ----
    ag_ms, hard_exit = None, False
    want_comm = not crowd and ((world_size > 1 and not args.no_allgather and (not args.share_gpu or args.rehearse_rccl)) or args.allgather)
    if want_comm:
        import threading
        box = {"done": False, "ranks": None, "ag_ms": None}

        def rccl_phase():
            rec, ms = None, None
            # The current device is PER THREAD (a new thread starts on device 0) and torch's object collectives put their tensors on
            # "cuda" = the calling thread's current device: without this every rank but 0 would hand NCCL tensors of a GPU that is not
            # its communicator's — a failure only a real multi-GPU node can show (gloo and the one-rank RCCL line never see it).
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
            try:
                uid = [rz.capi.comm_unique_id() if rank == 0 else None]
                if dist is not None:
                    dist.broadcast_object_list(uid, src=0)
                ctx.comm_init(world_size, rank, uid[0], V_total)
                rec = rz.capi.rccl_info()
                rec.update(ctx.comm_info())
                rec["expected_count"] = world_size
                for _ in range(5):
                    ctx.allgather()
                barrier()
                ta = time.perf_counter()
                for _ in range(50):
                    ctx.allgather()
                barrier()
                ms = (time.perf_counter() - ta) * 1e3 / 50
            except Exception as e:          # noqa: BLE001
                sys.stderr.write("[bench] RCCL communicator / all-gather failed on rank %d: %r\n" % (rank, e))
                rec = {"error": repr(e)}
            try:
                got = gather((rec, ms))
                box["ranks"] = [g[0] for g in got]
                times = [g[1] for g in got]
                box["ag_ms"] = None if any(t is None for t in times) else max(times)
            except Exception as e:          # noqa: BLE001  (the exchange of the records itself failed: this rank reports what it has)
                sys.stderr.write("[bench] rank %d could not exchange the RCCL records: %r\n" % (rank, e))
                box["ranks"] = [rec if r == rank else {"error": "records not exchanged: %r" % (e,)} for r in range(world_size)]
                box["ag_ms"] = None
            box["done"] = True
        th = threading.Thread(target=rccl_phase, daemon=True)
        th.start()
        th.join(timeout=args.rccl_timeout)
        if box["done"]:
            ag_ms = box["ag_ms"]
            for r, rec in zip(per_rank, box["ranks"]):
                r["rccl"] = rec
        else:
            sys.stderr.write("[bench] rank %d: the RCCL phase did not finish within %g s — reporting the line without it\n" % (rank, args.rccl_timeout))
            for r in per_rank:
                r["rccl"] = {"error": "RCCL communicator / all-gather did not finish within %g s (watchdog)" % args.rccl_timeout}
            hard_exit = True            # a collective is stuck: no orderly shutdown is possible
    elif world_size > 1:
        why = ("a crowd is sharded along the instance axis: no exchange, no communicator (SURVEY 8e)" if crowd else
               "--share-gpu: ranks share a GPU, RCCL needs one GPU per rank" if args.share_gpu else "--no-allgather")
        for r in per_rank:
            r["rccl"] = {"skipped": why}

    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        try:
            if full_affinity is not None:
                os.sched_setaffinity(0, full_affinity)      # the CPU baseline runs on ALL the host's cores, not on one node's
            cpu = cpu_baseline(args, mesh, deltas, mw, sparse)
        except Exception as e:          # noqa: BLE001
            sys.stderr.write("[bench] cpu baseline failed: %r\n" % (e,))

    if rank == 0:
        verts = V_total * I_total * args.steps
        kms = [r["kernel_ms"] for r in per_rank]
        names = {(1000000, 256, 64, 1): "C5", (30000, 200, 0, 256): "C4", (30000, 200, 64, 1): "C3", (30000, 200, 0, 1): "C2"}
        shard_how = "instance-sharded" if crowd else "vertex-sharded"
        if sparse_kind == "demo":
            wl = ("demo-shaped: %d-vert / %d-bone / %d SPARSE vertex morphs with the demo model's statistics (%d offsets, largest %d, "
                  "all on one 1 800-vertex face region), vertex-sharded over %d GPU(s)" % (V_total, B, M, int(sparse[0][-1]) if world_size == 1 else -1,
                                                                                      int(np.diff(sparse[0]).max()) if world_size == 1 else -1, world_size))
        elif sparse_kind == "sparse2":
            wl = "sparse-2%%: %d-vert / %d-bone / %d SPARSE vertex morphs at 2 %% density spread over the mesh, vertex-sharded over %d GPU(s)" % (V_total, B, M, world_size)
        else:
            wl = "%s: %d-vert / %d-bone / %d-dense-morph synthetic PMX%s, %s over %d GPU(s)" % (
                names.get((V_total, B, M, I_total), "custom"), V_total, B, M, (" x %d instances (per-instance palette in LDS)" % I_total) if crowd else "", shard_how, world_size)
        out = {
            "metric": "deformed verts/sec at 1/2/4/8 GPU; achieved HBM GB/s vs ~8 TB/s roofline",
            "value": verts / elapsed,
            "unit": "verts/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "timed_by": "hipEvent pair on the frames' stream around exactly K steps (rz_time_span), %d untimed lead-in frames in front of the opening event, all inside "
                        "barrier + synchronize on both sides, MAX over ranks; the host clock around the same region (lead-in + K frames, divided by their count): "
                        "config.ms_per_step_host_wall" % LEAD,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "verts_total": V_total, "verts_per_gpu": n, "bones": B, "morphs": M, "instances": I_total, "instances_per_gpu": I,
                "morph_layout": "sparse CSR" if sparse_kind else ("dense planes" if M else "none"),
                "parallelism": ("instance-shard x%d" if crowd else "vertex-shard x%d") % world_size,
                "launched_by": "self (python -m torch.distributed.run, re-executed by bench.py)" if os.environ.get("REZE_BENCH_SELF_LAUNCHED") == "1"
                               else ("external launcher (WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else "single process"),
                "numa_binding": numa,       # rank 0's: {"gpu_node", "cpus"}, null = not bound (one node, unknown, or --no-numa-bind)
                "bone_hierarchy_solve": ("device (motion sampling + hierarchy solve in rz_fk_kernel)" if args.device_sampling else "device (rz_fk_kernel)") if args.device_fk else "host",
                "autotune": tuned is not None,
                "autotune_rule": "heuristic plan (entry 0) unless a candidate's median-of-5-rounds time, MAX over ranks, is >= 2 % faster and its slowest round beats the heuristic's fastest; every rank adopts the same entry",
                "autotune_pick": tune_pick,
                "autotune_table": None if tune_table is None else [{k: (round(e[k], 6) if isinstance(e[k], float) else e[k]) for k in e} for e in tune_table],
                "graph_replay": bool(args.graph),
                "morph_split": eff["effective_split"],
                "grid": eff["effective_grid"],
                "inst_group": eff["effective_inst_group"],
                "inst_subsets": eff["effective_subsets"],
                "inst_subset_bones": eff["effective_subset_bones"],
                "inst_lds_bytes": eff["effective_inst_lds"],
                "frame_ms_events": timing["frame_ms"],
                "prep_kernel_ms": timing["prep_kernel_ms"],
                "frame_ms_with_pose_upload": with_upload_ms,
                "frame_ms_device_sampled_pose": sampled_ms,
                "frame_ms_with_pose_upload_two_in_flight": with_upload_pair_ms,
                "frames_in_flight": headline_in_flight,
                "frames_in_flight_calibrated": in_flight,
                "frames_in_flight_rule": "value / ms_per_step are the one-stream loop unless two frames in flight were calibrated faster (>= 3 %), their K timed steps were faster than the one-stream loop's too (by >= 1 %), AND their step is not shorter than the event-timed kernel (a step never undercuts its own kernel: such an overlap is reported in ms_per_step_two_frames_in_flight only)",
                "frames_in_flight_choice": ("--frames-in-flight " + args.frames_in_flight) if calib is None else calib,
                "lead_in_frames": LEAD,
                "ms_per_step_host_wall": elapsed_wall * 1e3 / (args.steps + LEAD),
                "value_host_wall": verts / elapsed_wall * (args.steps + LEAD) / args.steps,
                "host_fixed_cost_us_per_timed_region": (elapsed_wall - elapsed * (args.steps + LEAD) / args.steps) * 1e6,
                "ms_per_step_one_stream_host_wall": one_wall * 1e3 / (args.steps + LEAD),
                "frame_ms_with_pose_mapped": mapped_ms,
                "frame_ms_with_pose_mapped_protocol_only": mapped_bare_ms,
                "frame_ms_with_pose_mapped_two_in_flight": mapped_pair_ms,
                "pose_mapped_note": "crowds: rz_map_pose(ROWS12) + the caller's write (one memmove of pre-packed 48 B rows into the pinned ring slot, standing in for a pose solve that writes there) + rz_commit_pose + rz_deform per frame; _protocol_only = without the write",
                "ms_per_step_one_stream": one_ms,
                "ms_per_step_two_frames_in_flight": pair_ms,
                "speedup_basis": "compare N-GPU lines mode for mode: ms_per_step_one_stream(1) / ms_per_step_one_stream(N), or the _two_frames_in_flight pair; "
                                 "ms_per_step / value are the mode `frames_in_flight` names (one stream unless the rule in frames_in_flight_rule admits two)",
                "frames_in_flight_note": "2 = frames alternate between the context and an rz_fork of it (shared static data, own stream + outputs): the tail of frame f overlaps the ramp of frame f + 1; roofline.* is always one kernel on one stream",
                "per_frame_loops": "max over ranks, raw C ABI calls, clock stopped after every context drained; with_pose_upload = rz_set_pose%s + rz_deform per frame, device_sampled_pose = rz_set_pose_sampled (1 float / instance) + rz_deform"
                                   % ("_local" if args.device_fk else ""),
                "allgather_ms": ag_ms,
                "allgather_note": "ncclAllGather of the deformed positions over the library's own communicator, timed after and outside the K steps",
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
                "kernel_ms_repetitions": list(kernel_reps),       # 5 x 200 event-timed frames, ascending; kernel_ms is their median
                "kernel_ms_check": kcheck,
                "frame_frac": timing["algorithmic_bytes_per_frame"] / (timing["frame_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
            "cpu_baseline": cpu,
        }
    else:
        out = None
    if not hard_exit:
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
    if hard_exit:
        os._exit(0)


if __name__ == "__main__":
    main()
