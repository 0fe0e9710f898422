#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "prefetch or zero_copy or envelope or fork" 2>&1 | tail -25 | tee $O/pytest.txt
timeout 600 python tools/live_loop.py 2>&1 | tee $O/live_loop.txt
timeout 300 python bench.py --verts 125952 --no-cpu-baseline --frames-in-flight 1 --no-sampled-loop 2>$O/bench.err | tail -1 > $O/bench_shard8.json
python -c "
import json; d=json.load(open('$O/bench_shard8.json')); c=d['config']; print(d['ms_per_step'], c['frame_ms_with_pose_upload'], c['frame_ms_with_pose_upload_two_in_flight'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
tail -3 $O/bench.err
