#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
echo "== pytest gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_gpu.txt
echo "== asan"
timeout 900 python tools/asan_run.py gpu 2 2>&1 | tail -8 | tee $O/asan_gpu.txt
echo "== c4"
timeout 200 python tools/timeline.py c4 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline.txt
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/bench_c4.json
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 --tune nt_store=1 2>>$O/bench.err | tail -1 > $O/bench_c4_nts.json
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/bench_c4_b.json
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4i/bench_*.json')):
    try:
        d = json.load(open(f)); c = d['config']; r = d['roofline']
        print('%-26s ms/step %.5f kernel %s %.5f ms frac %.3f frame_frac %.3f pick %s' % (f.split('/')[-1], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['frame_frac'], c.get('autotune_pick')))
    except Exception as e:
        print(f, 'unreadable', e)
P
tail -3 $O/bench.err
