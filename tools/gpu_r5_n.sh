#!/bin/bash
# Round 5, session N: the staging of the hierarchy solve — key loads out of divergent branches, deep-round ancestor tables asked for at the
# top, own-bone local matrices ahead of the barrier. Parity of everything device-animated, timelines, the sampled per-frame loops.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "hierarchy or fk or sampl or local or fused or bone_morph or override or anim or physics or crowd or node" 2>&1 | tail -8 | tee $O/pytest_fk.txt
for c in sampled-c2 sampled-demo local-c2; do timeout 300 python tools/timeline.py $c 2>&1 | grep -v Warning | tee $O/timeline_$c.txt; done
for c in c2 demo; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_$c.json; done
python - <<'P' | tee gpurun_out/r5n/summary.txt
import json, glob
for f in sorted(glob.glob('gpurun_out/r5n/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']
        print(f.split('/')[-1], 'one', round(c.get('ms_per_step_one_stream')*1e3,2), 'two', round(c.get('ms_per_step_two_frames_in_flight')*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), 'upload', round(c.get('frame_ms_with_pose_upload')*1e3,2), 'sampled', round(c.get('frame_ms_device_sampled_pose')*1e3,2))
    except Exception as e:
        print(f, 'unreadable', e)
P
REZE_LIB=reze-engine_amd/libreze_deform.so timeout 600 python tools/ab_r4.py anim 2>&1 | grep -v Warning | tee $O/ab_anim.txt
