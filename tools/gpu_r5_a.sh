#!/bin/bash
# Round 5, session A: (1) pullbench — pose upload by copy vs pull kernel vs helper workgroups; (2) the C5 question settled in ONE process:
# round-3 library vs HEAD vs morph-list-in-SGPRs vs no kernel-argument preload, alternating; (3) why two frames in flight do not hide
# rz_fk_kernel for --device-fk (autotuned shape vs heuristic shape).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
F=tools/_tmp/flavors
echo "== pullbench"; timeout 300 tools/pullbench 2>&1 | tee $O/pullbench.txt
echo "== A/B c5 (10 rounds)"
timeout 900 python tools/ab_inproc.py c5 10 old=tools/_tmp/old/libreze_deform_old.so head=reze-engine_amd/libreze_deform.so pin=$F/libreze_deform_pin.so nopreload=$F/libreze_deform_nopreload.so 2>&1 | grep -v Warning | tee $O/ab_c5.txt
echo "== A/B shard, c3 (12 rounds)"
timeout 900 python tools/ab_inproc.py shard,c3 12 old=tools/_tmp/old/libreze_deform_old.so head=reze-engine_amd/libreze_deform.so pin=$F/libreze_deform_pin.so pinall=$F/libreze_deform_pinall.so pin2=$F/libreze_deform_pin2.so 2>&1 | grep -v Warning | tee $O/ab_small_dense.txt
echo "== device-fk crowd: autotuned vs heuristic shape, one stream vs two in flight"
for extra in "" "--no-autotune"; do
  timeout 600 python bench.py --config c4 --device-fk --no-cpu-baseline --no-sampled-loop $extra 2>>$O/bench.err | tail -1 > $O/bench_c4_devicefk${extra// /_}.json
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5a/bench_*.json')):
    d = json.load(open(f)); c = d['config']
    print(f.split('/')[-1], 'pick', c.get('autotune_pick'), 'grid', c.get('grid'), 'group', c.get('inst_group'), 'one', c.get('ms_per_step_one_stream'), 'two', c.get('ms_per_step_two_frames_in_flight'), 'kernel', d['roofline']['kernel_ms'], 'prep', c.get('prep_kernel_ms'), 'upload', c.get('frame_ms_with_pose_upload'), 'upload2', c.get('frame_ms_with_pose_upload_two_in_flight'))
P
tail -3 $O/bench.err
