#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
echo "== pytest (sparse)"
timeout 900 python -m pytest tests -m gpu -q -x -k "sparse or fuzz" 2>&1 | tail -3 | tee $O/pytest_subset.txt
for c in demo; do timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline.txt; done
echo "== A/B"
for rep in 1 2 3; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
