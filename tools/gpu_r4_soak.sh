#!/bin/bash
# Round 4 soak: the ABI state-machine fuzz over 120 seeds (odd seeds the product library, even seeds the all-variants build), the sharded walks, the host-engine walks
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/soak; rm -rf $O; mkdir -p $O
REZE_FUZZ_SEEDS=120 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -6 | tee $O/soak.txt
