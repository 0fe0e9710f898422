#!/bin/bash
# Profiles for profiles/: kernel-trace stats of the default bench and HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only — MI355X_MICROARCH.md §HBM),
# with a calibration pass over membench's known-byte kernels in the same counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline"
echo "== kernel trace + stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $BENCH > $O/trace.log 2>&1
echo "== pmc FETCH_SIZE"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- $BENCH > $O/fetch.log 2>&1
echo "== pmc WRITE_SIZE"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- $BENCH > $O/write.log 2>&1
echo "== calibration (membench quick) FETCH_SIZE / WRITE_SIZE"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o mb -- $R/tools/membench quick > $O/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o mb -- $R/tools/membench quick > $O/cal_write.log 2>&1
echo "== shard (1/8) + C4 traces"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_shard -o bench -- python $R/bench.py --verts 125952 --steps 200 --warmup 20 --no-cpu-baseline > $O/trace_shard.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -o bench -- python $R/bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline > $O/trace_c4.log 2>&1
cd $R; find gpurun_out/prof -name "*.csv" | head -40; du -sh gpurun_out/prof
