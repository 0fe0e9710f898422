#!/bin/bash
# Round 4, first contact: parity of the row-cooperative sparse walk / pointer-doubling hierarchy solve / local-pose prefetch,
# then A/B against the round-3 library (tools/_tmp/old/libreze_deform_old.so, built from the previous commit).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
echo "== pytest (sparse, device FK, prefetch, fuzz)"
timeout 900 python -m pytest tests -m gpu -q -x -k "sparse or fk or FK or prefetch or fuzz or bone_morph or sampled or local or smoke or physics or override" 2>&1 | tail -15 | tee $O/pytest_subset.txt
echo "== A/B"
for rep in 1 2; do
  REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py 2>&1 | tee -a $O/ab.txt
  timeout 300 python tools/ab_r4.py 2>&1 | tee -a $O/ab.txt
done
echo "== bench lines"
for c in demo sparse2 c2; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_$c.json; done
timeout 300 python bench.py --config c4 --device-fk --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4_devicefk.json
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4c/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']; r = d['roofline']
        print('%-26s ms/step %.5f (one %s two %s) kernel %s %.5f ms frac %.3f traffic %s | upload loop %s sampled loop %s' % (
            f.split('/')[-1], d['ms_per_step'], c.get('ms_per_step_one_stream'), c.get('ms_per_step_two_frames_in_flight'), r['kernel'], r['kernel_ms'], r['frac'], r.get('traffic'), c['frame_ms_with_pose_upload'], c['frame_ms_device_sampled_pose']))
    except Exception as e:
        print(f, 'unreadable', e)
P
echo "== node frame loop"
timeout 300 python tools/node_frame_bench.py 2>&1 | tail -3 | tee $O/node_frame_bench.txt
tail -3 $O/bench.err
