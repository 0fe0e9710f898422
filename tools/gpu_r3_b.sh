#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
echo "== pytest round3 + c4 + bench"
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py::test_c4_full_size_256x30k_200b tests/test_bench_gpu.py -x -q 2>&1 | tail -25 | tee $O/pytest.txt
echo "== c4 subsets sweep (product)"
timeout 600 python tools/c4_subsets.py quick 2>&1 | tee $O/c4_subsets.txt | tail -30
echo "== c4 subsets sweep (occupancy 6 build)"
REZE_LIB=$R/tools/_tmp/libreze_deform_occ6.so timeout 600 python tools/c4_subsets.py quick 2>&1 | tee $O/c4_subsets_occ6.txt | tail -30
echo "== bench c4 x3"
for i in 1 2 3; do timeout 300 python bench.py --config c4 --no-cpu-baseline 2>$O/bench_c4.err | tail -1 > $O/bench_c4_$i.json
python -c "
import json; d=json.load(open('$O/bench_c4_$i.json')); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frame_frac'], d['config']['autotune_pick'], d['config']['grid'], d['config']['inst_group'])
"; done
tail -5 $O/bench_c4.err
