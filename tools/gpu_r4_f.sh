#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
echo "== dmabench"; timeout 120 tools/dmabench 2>&1 | tee $O/dmabench.txt
echo "== pytest (largest skeleton + sparse)"
timeout 900 python -m pytest tests -m gpu -q -x -k "largest or sparse" 2>&1 | tail -3 | tee $O/pytest_subset.txt
echo "== A/B"
for rep in 1 2; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py dense 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py dense 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
