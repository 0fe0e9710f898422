#!/bin/bash
# round 3, first contact: new GPU tests (bone subsets, bench launcher), C4 subset sweep, quick bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
echo "== pytest round3 + c4 + bench"
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py::test_c4_full_size_256x30k_200b tests/test_bench_gpu.py -x -q 2>&1 | tail -25 | tee $O/pytest.txt
echo "== c4 subsets sweep"
timeout 600 python tools/c4_subsets.py 2>&1 | tee $O/c4_subsets.txt | tail -50
echo "== bench c4"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json
python -c "
import json; d=json.load(open('$O/bench_c4.json')); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frame_frac'], d['config']['autotune_pick'])
for e in d['config']['autotune_table']: print(e)
"
tail -5 $O/bench_c4.err
