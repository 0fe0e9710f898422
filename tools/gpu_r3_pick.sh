#!/bin/bash
# after a change to rz_autotune_pick: the lines whose plan the rule decides (C4, the 8-rank rehearsal) and the stability runs again
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/full; mkdir -p $O
timeout 600 python bench.py --config c4 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4.json
timeout 900 python bench.py --gpus 8 --share-gpu --dist-backend gloo --steps 50 --warmup 5 --no-cpu-baseline --no-sampled-loop --clock-warm-seconds 0.5 2>>$O/bench.err | grep '^{' | tail -1 > $O/bench_rehearse8.json
bash tools/gpu_r3_stab.sh
python - <<'P'
import json
for n in ('bench_c4', 'bench_rehearse8'):
    d = json.load(open('gpurun_out/full/%s.json' % n)); c = d['config']; r = d['roofline']
    print(n, 'ms/step %.5f pick %s kernel_ms %.5f frac %.3f frame_frac %.3f' % (d['ms_per_step'], c.get('autotune_pick'), r['kernel_ms'], r['frac'], r['frame_frac']))
P
