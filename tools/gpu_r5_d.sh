#!/bin/bash
# Round 5, session D: the reworked hierarchy solve (records requested ahead of the kernel arguments, one record per morph, radix-4 doubling) —
# parity of everything device-animated, per-frame loops against the pre-split build; and the C5 kernel-argument / morph-list question
# on ONE box with two contexts per build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
F=tools/_tmp/flavors
echo "== pytest gpu (device-animated subset first, then everything)"
timeout 1500 python -m pytest tests -m gpu -q -x -rf 2>&1 | tail -15 | tee $O/pytest_gpu.txt
echo "== per-frame loops: pre-split build vs HEAD"
for L in $F/libreze_deform_presplit.so reze-engine_amd/libreze_deform.so $F/libreze_deform_presplit.so reze-engine_amd/libreze_deform.so; do REZE_LIB=$L timeout 600 python tools/ab_r4.py anim 2>&1 | grep -v Warning | tee -a $O/ab_anim.txt; done
echo "== C5: which build (10 rounds, one context per build, then the same builds again as second contexts)"
timeout 900 python tools/ab_inproc.py c5 8 old=tools/_tmp/old/libreze_deform_old.so head=reze-engine_amd/libreze_deform.so pin2=$F/libreze_deform_pin2.so np=$F/libreze_deform_np.so np_pin2=$F/libreze_deform_np_pin2.so old_b=tools/_tmp/old/libreze_deform_old.so head_b=reze-engine_amd/libreze_deform.so pin2_b=$F/libreze_deform_pin2.so 2>&1 | grep -v Warning | tee $O/ab_c5.txt
for c in c2 demo; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_$c.json; done
timeout 600 python bench.py --config c4 --device-fk --device-sampling --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4_sampled.json
timeout 600 python bench.py --config c4 --device-fk --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4_devicefk.json
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5d/bench_*.json')):
    d = json.load(open(f)); c = d['config']
    print(f.split('/')[-1], 'one', c.get('ms_per_step_one_stream'), 'two', c.get('ms_per_step_two_frames_in_flight'), 'kernel', d['roofline']['kernel_ms'], 'prep', c.get('prep_kernel_ms'), 'upload', c.get('frame_ms_with_pose_upload'), 'sampled', c.get('frame_ms_device_sampled_pose'), 'headline', d['ms_per_step'], c.get('frames_in_flight'))
P
tail -3 $O/bench.err
