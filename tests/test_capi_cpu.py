"""CPU-side checks of the C ABI: the library builds/loads, exports every symbol the header
declares, the pure host helpers behave, and the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "reze_deform.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rz_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_header_symbol(rz):
    L = rz.capi.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libreze_deform.so does not export %s" % n
    assert sorted(rz.capi.SYMBOLS) == names
    header = open(os.path.join(ROOT, "include", "reze_deform.h")).read()
    assert L.rz_abi_version() == int(re.search(r"#define RZ_ABI_VERSION (\d+)", header).group(1))


def test_shard_ranges_tile_the_mesh(rz):
    for v_total in (1, 1023, 1024, 30000, 1000000, 1000001):
        for n in (1, 2, 4, 8):
            spans = [rz.shard_range(v_total, n, r) for r in range(n)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == v_total
            for (b0, c0), (b1, _c1) in zip(spans, spans[1:]):
                assert b1 == b0 + c0 or (c0 == 0 and b1 == v_total) or b1 == v_total
            full = [c for _, c in spans if c and c != spans[0][1]]
            assert len(full) <= 1                      # only the last non-empty shard may be short
            assert spans[0][1] % 256 == 0 or n == 1 or spans[0][1] == v_total
    with pytest.raises(rz.RzError):
        rz.shard_range(10, 0, 0)


def test_no_gpu_means_loud_failure_not_fallback(rz):
    if rz.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(rz.RzError) as e:
        rz.DeformContext(0)
    assert e.value.code == -3
    assert "device" in str(e.value).lower()


def test_product_never_imports_the_oracle():
    """③: nothing under reze-engine_amd/ may reference oracle/."""
    pkg = os.path.join(ROOT, "reze-engine_amd")
    bad = []
    for d, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".js", ".c", ".cpp", ".h", ".hip", ".ts")):
                src = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"\boracle[/.]|from oracle|import oracle|rzo_|librz_oracle", src):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_bench_refuses_a_mismatched_world_size_and_never_prints_a_line_without_gpus():
    """bench.py's launcher logic runs before anything touches a GPU, so it is testable here: a world size that is not --gpus
    is refused (exit 3); `--gpus 2` run plainly launches its own two ranks, which — on a box without GPUs — fail loudly:
    non-zero status and NO JSON line (never a quiet N = 1)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 3 and b"refusing" in p.stderr and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    import torch
    if not torch.cuda.is_available():
        env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--no-cpu-baseline"], env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode != 0 and b"re-executing as 2 ranks" in p.stderr
        assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_autotune_pick_keeps_the_heuristic_unless_a_candidate_is_clearly_faster(rz):
    """rz_autotune_pick is pure host logic (include/reze_deform.h): entry 0 unless a candidate's median is >= 2 % lower AND its
    slowest round is under entry 0's fastest round; tables without spreads (ms_min = 0) are judged on the median alone."""
    capi, L = rz.capi, rz.capi.load()

    def pick(rows):
        tab = (capi.RzTuneEntry * len(rows))()
        for i, (ms, lo, hi, same) in enumerate(rows):
            tab[i].ms, tab[i].ms_min, tab[i].ms_max, tab[i].same_as = ms, lo, hi, same
        return L.rz_autotune_pick(tab, len(rows))

    assert pick([(33.3, 33.0, 33.7, -1), (32.5, 32.2, 32.7, -1)]) == 1          # 2.4 % and ranges apart
    assert pick([(33.3, 33.0, 33.7, -1), (32.5, 32.2, 33.1, -1)]) == 0          # 2.4 % but the ranges overlap: noise
    assert pick([(33.3, 33.0, 33.7, -1), (32.9, 32.8, 32.95, -1)]) == 0         # apart, but only 1.2 %
    assert pick([(33.3, 0.0, 0.0, -1), (32.5, 0.0, 0.0, -1)]) == 1              # no spreads recorded: the median rule alone
    assert pick([(33.3, 33.0, 33.7, -1), (30.0, 29.9, 30.1, 0)]) == 0           # an alias of entry 0 is never "another plan"
    assert pick([(33.3, 33.0, 33.7, -1), (32.5, 32.2, 32.7, -1), (31.9, 31.8, 32.0, -1)]) == 2     # the fastest qualifying entry


def test_sanitized_host_library_cpu_driver():
    """make asan + tools/asan_run.py cpu: the host library (csrc/*.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer,
    driven through every C-ABI path that needs no GPU — every export refuses a NULL context with a message, shard arithmetic
    over edge sizes, the launch-shape pick rule on synthetic tables, rz_create without a device. Any report aborts the child.
    (The GPU half — misuse script, fuzz walks, ring / fork / graph soak — runs on the GPU box: profiles/r4_asan.txt.)"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_run.py"), "cpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    if p.returncode == 77:          # the toolchain for the sanitized build is not on this machine (tools/asan_run.py says which part)
        pytest.skip(out.strip().splitlines()[-1] if out.strip() else "sanitized build unavailable")
    assert p.returncode == 0 and "ASAN-CPU-OK" in out, out[-3000:]


def test_crowd_pose_packing_loop_on_the_host():
    """The host half of a crowd's pose upload (csrc/pose.cpp: pack_rows_avx512, reached through the tools-only build's rz_debug_pack_rows —
    no GPU involved): 4 x 4 column-major world matrices (engine.ts:2383-2389 uploads all sixteen floats) become 48 B per bone — the four
    columns' x y z — for every bone count (the loop takes four bones per step, the tail one by one), and ANY bottom row that is not
    exactly 0 0 0 1 — a projective entry, a NaN, a negative zero: bit patterns — is reported, because the device would write 0 0 0 1 back."""
    import ctypes
    import numpy as np
    import reze_engine_amd as rz
    if not os.path.exists(rz.capi.VARIANTS_LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "reze-engine_amd", "csrc"), "variants"])
    L = ctypes.CDLL(rz.capi.VARIANTS_LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    L.rz_debug_pack_rows.argtypes = [fp, ctypes.c_uint32, fp]
    rng = np.random.default_rng(3)
    probe = np.zeros(16, np.float32); probe[15] = 1.0
    if L.rz_debug_pack_rows(probe.ctypes.data_as(fp), 1, np.zeros(16, np.float32).ctypes.data_as(fp)) < 0:
        pytest.skip("this host has no AVX-512: poses travel unpacked")
    for bones in (1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 1001):
        w = rng.normal(size=(bones, 4, 4)).astype(np.float32)         # [bone][column][row]
        w[:, :, 3] = (0.0, 0.0, 0.0, 1.0)
        out = np.full(bones * 12 + 16, 7.0, np.float32)               # guard words behind the rows
        ok = L.rz_debug_pack_rows(w.ctypes.data_as(fp), bones, out.ctypes.data_as(fp))
        assert ok == 1, bones
        assert np.array_equal(out[:bones * 12].reshape(bones, 4, 3), w[:, :, :3]), bones
        assert (out[bones * 12:] == 7.0).all(), "the packing loop wrote past its %d bones" % bones
        for bad_bone in {0, bones // 2, bones - 1}:
            for col, val in ((0, 0.25), (3, 2.0), (1, np.float32("nan")), (2, np.float32(-0.0))):
                v = w.copy()
                v[bad_bone, col, 3] = val
                assert L.rz_debug_pack_rows(v.ctypes.data_as(fp), bones, out.ctypes.data_as(fp)) == 0, (bones, bad_bone, col, val)


def test_launch_shape_heuristics_of_dense_frames():
    """make_plan for the one-launch dense frame is a pure function of the sizes; the tools-only build exports it (rz_debug_plan_dense, no GPU
    involved). The table is tests/test_gpu_round6.py::test_heuristic_plan_at_the_shard_sizes_of_c5's — what the round-6 sweeps found best
    on MI355X (profiles/r6_plan_sweep.txt, r6_fresh_plans*.txt, NOTEBOOK R6.2 / R6.9) — and over a sweep of 400 sizes every plan covers the
    mesh, fits the kernel's limits and follows the rules the table came from: a wave's run is a whole number of S = 2 steps or the frame
    runs at S = 4 / 8; a run between one and two steps does not exist at S = 2 / 4; the shard sizes of N = 1, 2, 4, 8 keep their plans."""
    import ctypes
    import reze_engine_amd as rz
    if not os.path.exists(rz.capi.VARIANTS_LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "reze-engine_amd", "csrc"), "variants"])
    L = ctypes.CDLL(rz.capi.VARIANTS_LIB_PATH)
    ci = ctypes.c_int

    def plan(V, B=256, M=64, ncu=256):
        s, g, q, o = ci(), ci(), ci(), ci()
        assert L.rz_debug_plan_dense(V, B, M, ncu, ctypes.byref(s), ctypes.byref(g), ctypes.byref(q), ctypes.byref(o)) == 0
        return s.value, g.value, q.value, o.value

    table = [(1000000, 2, 489), (875008, 2, 428), (797440, 4, 480), (625152, 4, 489), (530432, 4, 461), (500224, 2, 489), (400128, 2, 391), (375040, 2, 733),
             (333568, 4, 435), (313856, 4, 491), (281600, 4, 440), (250112, 4, 489), (156416, 4, 611), (125184, 4, 489), (93952, 8, 734)]
    for V, S, grid in table:
        assert plan(V)[:2] == (S, grid), (V, plan(V))
    assert plan(30000, B=200)[:2] == (8, 235)                                  # C3
    assert plan(1000000, M=1)[0] == 1 and plan(1000000, M=2)[0] == 2         # the split never exceeds the morph count
    assert L.rz_debug_plan_dense(0, 256, 64, 256, ctypes.byref(ci()), ctypes.byref(ci()), ctypes.byref(ci()), None) < 0
    for V in range(20000, 1220000, 3000):
        S, grid, qpw, cap = plan(V)
        nq, step = (V + 3) // 4, 64 // S
        assert S in (2, 4, 8) and qpw % 8 == 0 and qpw >= 8 and grid >= 1
        assert grid * 4 * qpw >= nq > (grid - 1) * 4 * qpw, (V, S, grid, qpw)           # the grid covers the mesh and no workgroup is idle
        if S == 2:
            assert qpw % step == 0, (V, qpw)                                           # whole 128-vertex steps, or the frame is not at S = 2
        if S in (2, 4):
            assert not (step < qpw < 2 * step), (V, S, qpw)                              # never between one and two steps
        if S == 8:
            assert qpw == 8                                                            # one 32-vertex step per wave
        assert cap == 0 or (cap % 64 == 0 and cap <= 640 and cap >= qpw * 4)           # a parked run fits its buffer
