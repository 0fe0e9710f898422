#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.txt
timeout 900 python tools/sweep.py c4 > gpurun_out/sweep.txt 2>&1; tail -2 gpurun_out/sweep.txt
