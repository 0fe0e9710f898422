#!/usr/bin/env python3
"""Pretty-print gpurun_out/sweep.json (top rows per config by frame time)."""
import collections, json, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sweep.json'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = json.load(open(path))
by = collections.OrderedDict()
for r in rows: by.setdefault(r['config'], []).append(r)
keep = ('inst_loop','morph_split','unroll','nontemporal','nt_store','geo_lds','grid_cap','fast','kernel_ms','frame_ms','gbps','S','U','F','grid')
fmt = lambda r: ' '.join('%s=%s' % (k[:6], ('%.4f' % r[k] if isinstance(r[k], float) else r[k])) for k in keep if k in r)
for k, v in by.items():
    print('==', k)
    errs = [r for r in v if 'error' in r]
    v = sorted([r for r in v if 'error' not in r], key=lambda r: r['frame_ms'])
    for r in v[:top]: print('  ', fmt(r))
    if v: print('   worst', fmt(v[-1]))
    if errs: print('   errors', len(errs), errs[0].get('error'))
