#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4l; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "sampl or fk or FK or anim or motion or bone_morph or host_engine or e2e or node or napi" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee $O/pytest_subset.txt
for c in sampled-demo sampled-c2; do timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids\|per XCD\|late wave" | tee -a $O/timeline.txt; done
for rep in 1 2; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout 300 python tools/node_frame_bench.py 2>&1 | tail -2 | tee $O/node_frame_bench.txt
