#!/bin/bash
# Round 5, session E: crowd frames with the hierarchy solved in the skin kernel's front (new tests, device-fk / sampled C4 lines) and the
# dense kernel's entry: leading preloaded arguments vs everything out of `p` (round-3 style), on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
F=tools/_tmp/flavors
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -rf 2>&1 | tail -25 | tee $O/pytest_round5.txt
echo "== C4 device-animated lines"
for extra in "--device-fk" "--device-fk --device-sampling" "--device-fk --tune fuse_fk=0"; do
  timeout 600 python bench.py --config c4 $extra --no-cpu-baseline 2>>$O/bench.err | tail -1 > "$O/bench_c4_$(echo $extra | tr -d ' -' | tr '=' '_').json"
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5e/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']
        print(f.split('/')[-1], 'kernel', d['roofline']['kernel'], 'one', c.get('ms_per_step_one_stream'), 'two', c.get('ms_per_step_two_frames_in_flight'), 'kernel_ms', d['roofline']['kernel_ms'], 'prep', c.get('prep_kernel_ms'), 'upload', c.get('frame_ms_with_pose_upload'), 'upload2', c.get('frame_ms_with_pose_upload_two_in_flight'), 'sampled', c.get('frame_ms_device_sampled_pose'), 'pick', c.get('autotune_pick'))
    except Exception as e:
        print(f, 'unreadable', e)
P
tail -5 $O/bench.err
echo "== C5: entry of the dense kernel"
timeout 900 python tools/ab_inproc.py c5 8 old=tools/_tmp/old/libreze_deform_old.so head=reze-engine_amd/libreze_deform.so fromp=$F/libreze_deform_fromp.so fromp_np=$F/libreze_deform_fromp_np.so np=$F/libreze_deform_np.so 2>&1 | grep -v Warning | tee $O/ab_c5.txt
timeout 900 python tools/ab_inproc.py shard,c3 8 old=tools/_tmp/old/libreze_deform_old.so head=reze-engine_amd/libreze_deform.so fromp=$F/libreze_deform_fromp.so fromp_np=$F/libreze_deform_fromp_np.so 2>&1 | grep -v Warning | tee $O/ab_shard_c3.txt
