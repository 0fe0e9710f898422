#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5c; mkdir -p $O
for q in 1 2 4 8; do GPU_MAX_HW_QUEUES=$q timeout 300 python tools/placement.py c5 nooffset 2>&1 | grep -v Warning | tee $O/placement_c5_q$q.txt; done
timeout 300 python tools/placement.py c5 nooffset extrastream 2>&1 | grep -v Warning | tee $O/placement_c5_extrastream.txt
