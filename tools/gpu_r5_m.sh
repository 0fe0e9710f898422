#!/bin/bash
# Round 5, session M: where do the kernel arguments live? HIP_FORCE_DEV_KERNARG=1 puts the kernarg segment in device memory (the scalar
# loads of `p` then do not cross the host link) — small frames with the variable 0 / 1 / unset.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5m; rm -rf $O; mkdir -p $O
for kv in unset 0 1; do
  for c in c2 demo c3; do
    if [ $kv = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$kv; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_${c}_kernarg_$kv.json
  done
done
python - <<'P' | tee gpurun_out/r5m/summary.txt
import json, glob
for f in sorted(glob.glob('gpurun_out/r5m/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']
        print(f.split('/')[-1], 'one', round(c.get('ms_per_step_one_stream')*1e3,2), 'two', round(c.get('ms_per_step_two_frames_in_flight')*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), 'upload', round(c.get('frame_ms_with_pose_upload')*1e3,2), 'sampled', round(c.get('frame_ms_device_sampled_pose')*1e3,2), 'numa', c.get('numa_binding'))
    except Exception as e:
        print(f, 'unreadable', e)
P
tail -3 $O/bench.err
