#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
echo "== pytest round4 + prefetch"
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "round4 or prefetch or staged or doubling or local_poses or shards_cut" 2>&1 | tail -15 | tee $O/pytest_r4.txt
echo "== asan"
timeout 900 python tools/asan_run.py gpu 2 2>&1 | tail -25 | tee $O/asan_gpu.txt
echo "== A/B small"
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
