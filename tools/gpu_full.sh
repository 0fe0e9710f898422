#!/bin/bash
# Full confirmation run: GPU tests, smoke, default bench + C4 bench, profiles.
mkdir -p gpurun_out
echo "== pytest gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt
echo "== bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench.json
for c in c4 c3 c2; do
  timeout 600 python bench.py --config $c --no-cpu-baseline 2>>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench_$c.json
done
bash tools/gpu_profile.sh 2>&1 | tail -5
