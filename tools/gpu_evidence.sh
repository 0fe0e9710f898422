#!/bin/bash
# Second evidence session of a round (the first is tools/gpu_full.sh): rocprofv3 kernel stats + PMC traffic (tools/gpu_profile.sh) condensed
# ON the box (the CSVs are hundreds of MB; gpurun_out/ travels back under 64 MiB), per-wave timelines of every tracked workload, the
# counters of the dominant kernels (tools/archive/gpu_counters.sh) and the round's micro-benchmarks.
#   usage: tools/gpu_evidence.sh <tag> [profile] [timelines] [counters] [micro]      default: all four parts      -> gpurun_out/ev/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-r6}; shift
PARTS=${@:-profile timelines counters micro}
O=gpurun_out/ev; rm -rf $O; mkdir -p $O/profiles
for part in $PARTS; do
case $part in
profile)
  bash tools/gpu_profile.sh 2>&1 | tail -20
  cd $R
  python tools/parse_prof.py $TAG > $O/parse_prof.log 2>&1 || tail -5 $O/parse_prof.log
  cp profiles/${TAG}_kernel_stats_*.txt profiles/${TAG}_bench_under_rocprof_*.json profiles/pmc_traffic.json profiles/${TAG}_pmc_hbm_traffic.txt $O/profiles/ 2>/dev/null
  rm -rf gpurun_out/prof ;;
timelines)
  for c in c2 c3 sparse2 demo sampled-c2 local-c2 sampled-demo shard c4 local-c4 sampled-c4 c5; do
    timeout 200 python tools/archive/timeline.py $c 2>&1 | grep -v "amdgpu.ids" > $O/timeline_$c.txt; head -1 $O/timeline_$c.txt | cut -c1-200
  done ;;
counters)
  rm -rf gpurun_out/counters
  PMC_GROUPS="1 2 3" bash tools/archive/gpu_counters.sh c5 shard demo 2>&1 | grep -v "^$" | tail -20      # (the write-path groups matter for the store-bound crowd kernels)
  bash tools/archive/gpu_counters.sh c4 c4fk 2>&1 | grep -v "^$" | tail -20
  cd $R; mkdir -p $O/counters; cp gpurun_out/counters/summary_*.txt $O/counters/ 2>/dev/null ;;
micro)
  timeout 120 tools/archive/overlapbench 2>&1 | tee $O/overlapbench.txt | tail -4
  timeout 120 tools/pullbench 2>&1 | tee $O/pullbench.txt | tail -3
  timeout 60 tools/archive/packbench 2>&1 | tee $O/packbench.txt | tail -6
  timeout 300 python tools/archive/crowd_upload.py 256 2>&1 | grep -v Warning | tee $O/crowd_upload.txt ;;
esac
done
