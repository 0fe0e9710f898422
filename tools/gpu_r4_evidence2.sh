#!/bin/bash
# Round 4, second evidence session: rocprofv3 stats + PMC traffic (tools/gpu_profile.sh), SQ counters of the sparse kernel,
# per-wave timelines of every tracked workload, same-session A/B against the round-3 library, the sanitized host library, dmabench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/ev2; rm -rf $O; mkdir -p $O/profiles
bash tools/gpu_profile.sh 2>&1 | tail -20
# condense on the box (the CSVs are hundreds of MB; gpurun_out/ only travels back under 64 MiB) and keep what profiles/ tracks
cd $R; mkdir -p /tmp/prof_keep; cp profiles/pmc_traffic.json /tmp/prof_keep/ 2>/dev/null
python tools/parse_prof.py r4 > $O/parse_prof.log 2>&1 || tail -5 $O/parse_prof.log
cp profiles/r4_kernel_stats_*.txt profiles/r4_bench_under_rocprof_*.json profiles/pmc_traffic.json profiles/r4_pmc_hbm_traffic.txt $O/profiles/ 2>/dev/null
rm -rf gpurun_out/prof
cd $R; bash tools/gpu_sparse_counters.sh 2>&1 | tail -12
cd $R; cp gpurun_out/spcnt/summary.txt $O/profiles/sq_counters_sparse_raw.txt 2>/dev/null; rm -rf gpurun_out/spcnt
for c in c2 c3 sparse2 demo sampled-c2 local-c2 sampled-demo shard c4 c5; do
  timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids" > $O/timeline_$c.txt; head -3 $O/timeline_$c.txt | cut -c1-200
done
for rep in 1 2 3; do
  REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 400 python tools/ab_r4.py small dense c4 anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  timeout 400 python tools/ab_r4.py small dense c4 anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout 900 python tools/asan_run.py gpu 3 2>&1 | tail -8 | tee $O/asan_gpu.txt
timeout 120 tools/dmabench 2>&1 | tee $O/dmabench.txt
