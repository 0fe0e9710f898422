#!/bin/bash
# Round 5, session B: the split build (host units + kernels/ files) on the GPU — full parity suite, then fresh-context A/B of the
# split product against the pre-split build of the same commit, and the SGPR-pinned morph list (S = 2 only / S = 2 + 4).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
F=tools/_tmp/flavors
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q -x -rf 2>&1 | tail -15 | tee $O/pytest_gpu.txt
echo "== A/B fresh contexts: c5"
timeout 900 python tools/ab_inproc.py c5 f8 presplit=$F/libreze_deform_presplit.so split=reze-engine_amd/libreze_deform.so pin2=$F/libreze_deform_pin2.so pin=$F/libreze_deform_pin.so old=tools/_tmp/old/libreze_deform_old.so 2>&1 | grep -v Warning | tee $O/ab_c5.txt
echo "== A/B fresh contexts: shard, c3, c2, demo"
timeout 900 python tools/ab_inproc.py shard,c3 f10 presplit=$F/libreze_deform_presplit.so split=reze-engine_amd/libreze_deform.so pin=$F/libreze_deform_pin.so old=tools/_tmp/old/libreze_deform_old.so 2>&1 | grep -v Warning | tee $O/ab_dense_small.txt
timeout 900 python tools/ab_inproc.py c2,demo,c4 f10 presplit=$F/libreze_deform_presplit.so split=reze-engine_amd/libreze_deform.so 2>&1 | grep -v Warning | tee $O/ab_small.txt
