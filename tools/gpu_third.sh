#!/bin/bash
mkdir -p gpurun_out
echo "== pytest gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.txt
echo "== bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
echo "== bench C4"
timeout 600 python bench.py --verts 30000 --bones 200 --morphs 0 --instances 256 --no-cpu-baseline 2>>gpurun_out/bench.err | tee gpurun_out/bench_c4.json
echo "== bench allgather (1 rank)"
timeout 600 python bench.py --allgather --no-cpu-baseline --steps 50 2>>gpurun_out/bench.err | tee gpurun_out/bench_ag.json
echo "== membench"
timeout 300 ./tools/membench > gpurun_out/membench.txt 2>&1; tail -2 gpurun_out/membench.txt
bash tools/gpu_profile.sh 2>&1 | tail -50
