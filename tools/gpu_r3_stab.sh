#!/bin/bash
# search stability: consecutive bench runs of C5 and C4 on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/full; mkdir -p $O; rm -f $O/stab_*.json
for i in 1 2 3 4 5; do timeout 600 python bench.py --no-cpu-baseline --no-sampled-loop --no-pair-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/stab_c5_$i.json; done
for i in 1 2 3 4 5; do timeout 600 python bench.py --config c4 --no-cpu-baseline --no-sampled-loop --no-pair-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/stab_c4_$i.json; done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/full/stab_*.json')):
    d = json.load(open(f)); c = d['config']; r = d['roofline']; t = c['autotune_table']
    print('%-16s ms/step %.5f pick %2d kernel %s %.5f frac %.3f | heuristic %.5f [%.5f %.5f] best other %.5f' % (f.split('/')[-1], d['ms_per_step'], c['autotune_pick'], r['kernel'], r['kernel_ms'], r['frac'], t[0]['ms'], t[0]['ms_min'], t[0]['ms_max'], min(e['ms'] for e in t[1:] if e['same_as'] != 0)))
P
