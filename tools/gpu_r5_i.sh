#!/bin/bash
# Round 5, session I: big-pose ring (no per-frame "slot free" hand-off), pull for world matrices / copy engine for local rotations.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
echo "== host"
lscpu | grep -E "Model name|Socket|NUMA" | tee $O/host.txt
for d in /sys/class/drm/card*/device/numa_node; do echo "$d: $(cat $d)"; done | tee -a $O/host.txt
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], '...')" | tee -a $O/host.txt
echo "== packbench"
./tools/packbench | tee $O/packbench.txt
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -rf 2>&1 | tail -25 | tee $O/pytest_round5.txt
echo "== per-frame loops of a host-animated crowd: pulled vs copied"
timeout 600 python tools/crowd_upload.py 256 2>&1 | grep -v Warning | tee $O/crowd_upload.txt
echo "== C4 lines"
for extra in "" "--device-fk"; do
  timeout 600 python bench.py --config c4 $extra --no-cpu-baseline 2>>$O/bench.err | tail -1 > "$O/bench_c4_$(echo $extra | tr -d ' -').json"
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5i/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']
        print(f.split('/')[-1], 'kernel', d['roofline']['kernel'], 'one', c.get('ms_per_step_one_stream'), 'two', c.get('ms_per_step_two_frames_in_flight'), 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'upload', c.get('frame_ms_with_pose_upload'), 'upload2', c.get('frame_ms_with_pose_upload_two_in_flight'), 'sampled', c.get('frame_ms_device_sampled_pose'))
    except Exception as e:
        print(f, 'unreadable', e)
P
tail -3 $O/bench.err
echo "== the whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q -x -rf 2>&1 | tail -8 | tee $O/pytest_gpu.txt
