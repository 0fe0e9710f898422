#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r3h; mkdir -p $O
for v in base u2 u4 base u2 u4; do
echo "== $v"
REZE_LIB=$R/tools/_tmp/libreze_deform_$v.so timeout 300 python tools/c4_subsets.py mini 2>&1 | grep -v library | tee -a $O/c4_clamp_$v.txt
done
