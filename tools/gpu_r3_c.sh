#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
rocm-smi --showcomputepartition --showmemorypartition --showclocks --showpower --showperflevel 2>&1 | grep -v "^$" | head -40 | tee $O/smi.txt
for i in 1 2; do
echo "== c4 subsets sweep (product) pass $i"
timeout 600 python tools/c4_subsets.py quick 2>&1 | tee $O/c4_subsets_$i.txt | tail -16
done
rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -30
