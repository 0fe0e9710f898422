#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
echo "== pytest sparse"
timeout 900 python -m pytest tests -m gpu -q -x -k "sparse or fuzz or prefetch" 2>&1 | tail -5 | tee $O/pytest_subset.txt
for c in sparse2 demo; do timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline.txt; done
echo "== A/B small"
for rep in 1 2; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
echo "== HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_devkernarg.txt
HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_devkernarg.txt
HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/timeline.py c2 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline_devkernarg.txt
HIP_FORCE_DEV_KERNARG=0 timeout 200 python tools/timeline.py c2 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline_devkernarg.txt
