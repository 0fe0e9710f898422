#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5f; rm -rf $O; mkdir -p $O
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -rf 2>&1 | tail -25 | tee $O/pytest_round5.txt
echo "== device-animated tests of the earlier rounds (the solve's arithmetic changed: explicit FMA chains)"
timeout 900 python -m pytest tests -m gpu -q -x -k "hierarchy or fk or sampl or local or fused or bone_morph or override or anim or physics" 2>&1 | tail -8 | tee $O/pytest_fk.txt
for c in local-c4 sampled-c4 c4 sampled-c2 sampled-demo; do timeout 300 python tools/timeline.py $c 2>&1 | grep -v Warning | tee $O/timeline_$c.txt; done
