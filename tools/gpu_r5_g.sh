#!/bin/bash
# Round 5, session G: a crowd's pose pulled out of the pinned ring by rz_pull_pose_kernel (world matrices as three rows per bone) —
# the new parity tests, the per-frame loops pulled vs copied, the C4 bench lines with their upload loops, and where an XCD's lag in
# the C4 frame comes from (per-XCD medians of every stamp, both workgroup orders).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -rf 2>&1 | tail -25 | tee $O/pytest_round5.txt
echo "== per-frame loops of a host-animated crowd: pulled vs copied"
timeout 600 python tools/crowd_upload.py 256,64 2>&1 | grep -v Warning | tee $O/crowd_upload.txt
echo "== C4 lines"
for extra in "" "--device-fk"; do
  timeout 600 python bench.py --config c4 $extra --no-cpu-baseline 2>>$O/bench.err | tail -1 > "$O/bench_c4_$(echo $extra | tr -d ' -').json"
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5g/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']
        print(f.split('/')[-1], 'kernel', d['roofline']['kernel'], 'one', c.get('ms_per_step_one_stream'), 'two', c.get('ms_per_step_two_frames_in_flight'), 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'upload', c.get('frame_ms_with_pose_upload'), 'upload2', c.get('frame_ms_with_pose_upload_two_in_flight'), 'sampled', c.get('frame_ms_device_sampled_pose'))
    except Exception as e:
        print(f, 'unreadable', e)
P
tail -5 $O/bench.err
echo "== C4 timelines"
timeout 300 python tools/timeline.py c4 2>&1 | grep -v Warning | tee $O/timeline_c4.txt
timeout 300 python tools/timeline.py c4 inst_order=0 2>&1 | grep -v Warning | tee $O/timeline_c4_order0.txt
timeout 300 python tools/timeline.py c5 2>&1 | grep -v Warning | tee $O/timeline_c5.txt
