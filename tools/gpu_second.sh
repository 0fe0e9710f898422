#!/bin/bash
mkdir -p gpurun_out
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.txt
echo "== membench"
timeout 300 ./tools/membench > gpurun_out/membench.txt 2>&1; tail -3 gpurun_out/membench.txt
echo "== bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
echo "== sweep"
timeout 1200 python tools/sweep.py > gpurun_out/sweep.txt 2>&1; tail -2 gpurun_out/sweep.txt
