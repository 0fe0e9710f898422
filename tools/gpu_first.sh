#!/bin/bash
# First GPU contact: environment probe, GPU parity tests, default bench, kernel sweep, rocprof stats.
mkdir -p gpurun_out
{
  echo "== env"; nproc; node --version 2>&1; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
  python - <<'P'
import os; print("cpus", os.cpu_count())
P
} > gpurun_out/env.txt 2>&1
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "== bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== sweep"
timeout 900 python tools/sweep.py > gpurun_out/sweep.txt 2>&1; tail -3 gpurun_out/sweep.txt
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.txt 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r1 | head -20
