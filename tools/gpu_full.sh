#!/bin/bash
# Full confirmation run of a round: GPU tests, smoke, the tracked bench lines (C5 default + C4 / C3 / C2 / one 1/8 shard / sparse
# real-shape lines / device-animated C4 / RCCL at N = 1 / 8-rank rehearsal), search stability (consecutive runs), shard scaling,
# the per-frame upload loop, the Node frame loop, the C4 sweep. Output: gpurun_out/full/ (tools/collect_profiles.py copies it to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/full; rm -rf $O; mkdir -p $O
if [ "$1" != "nobench" ]; then
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" > $O/pytest_rc.txt
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu_full.txt | tail -12 | tee $O/pytest_gpu.txt; cat $O/pytest_rc.txt | tee -a $O/pytest_gpu.txt; rm -f $O/pytest_gpu_full.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
fi
echo "== bench"
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_c5.json
for c in c4 c3 c2; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_$c.json; done
for c in demo sparse2; do timeout 600 python bench.py --config $c 2>>$O/bench.err | tail -1 > $O/bench_$c.json; done
timeout 600 python bench.py --verts 125184 --no-cpu-baseline --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/bench_shard8.json
timeout 600 python bench.py --verts 125184 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_shard8_auto.json
timeout 600 python bench.py --verts 250112 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_shard4.json
timeout 600 python bench.py --verts 500224 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_shard2.json
# the driver's own flags, at every shard size of N = 1, 2, 4, 8 (round 6: the timed region is 20 steps; the step is an event span)
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c5_steps20.json
for n in 2:500224 4:250112 8:125184; do timeout 600 python bench.py --verts ${n#*:} --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_shard${n%:*}_steps20.json; done
timeout 600 python bench.py --config c4 --device-fk --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4_devicefk.json
timeout 600 python bench.py --config c4 --device-fk --device-sampling --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c4_sampled.json
REZE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --allgather --steps 100 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_c5_allgather1.json
echo "== 8 ranks on the one GPU (plumbing rehearsal; plain invocation: bench.py launches itself)"
timeout 900 python bench.py --gpus 8 --share-gpu --dist-backend gloo --steps 50 --warmup 5 --no-cpu-baseline --no-sampled-loop --clock-warm-seconds 0.5 2>>$O/bench.err | grep '^{' | tail -1 > $O/bench_rehearse8.json
timeout 900 python bench.py --config c4 --gpus 8 --share-gpu --dist-backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-sampled-loop --clock-warm-seconds 0.5 2>>$O/bench.err | grep '^{' | tail -1 > $O/bench_c4_rehearse8.json
echo "== search stability: consecutive runs"
for i in 1 2 3 4 5; do timeout 600 python bench.py --no-cpu-baseline --no-sampled-loop --no-pair-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/stab_c5_$i.json; done
for i in 1 2 3 4 5; do timeout 600 python bench.py --config c4 --no-cpu-baseline --no-sampled-loop --no-pair-loop --frames-in-flight 1 2>>$O/bench.err | tail -1 > $O/stab_c4_$i.json; done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/full/bench_*.json')) + sorted(glob.glob('gpurun_out/full/stab_*.json')):
    try:
        d = json.load(open(f)); c = d['config']; r = d['roofline']
        print('%-30s value %.4g verts/s  ms/step %.5f (in flight %d; one %s two %s) pick %s kernel %s %.5f ms frac %.3f frame_frac %.3f | upload loop %s (pair %s) sampled loop %s' % (
            f.split('/')[-1], d['value'], d['ms_per_step'], c.get('frames_in_flight', 1), c.get('ms_per_step_one_stream'), c.get('ms_per_step_two_frames_in_flight'), c.get('autotune_pick'), r['kernel'], r['kernel_ms'], r['frac'], r['frame_frac'], c['frame_ms_with_pose_upload'], c.get('frame_ms_with_pose_upload_two_in_flight'), c['frame_ms_device_sampled_pose']))
    except Exception as e:
        print(f, 'unreadable', e)
P
if [ "$1" != "nobench" ]; then
echo "== shard scaling (exit status 1 = the heuristic plan is more than 2 % behind the search somewhere)"
timeout 600 python tools/shard_scaling.py 2>&1 | tee $O/shard_scaling.txt; echo "shard_scaling rc=${PIPESTATUS[0]}" | tee -a $O/shard_scaling.txt
echo "== the heuristic plan among the launch shapes, at the shard sizes and between them"
timeout 900 python tools/plan_sweep.py --sizes 1000000,875008,750080,625152,500224,437760,375040,312576,250112,218880,187648,156416,125184,93952,62720 --steps-per-wave 1,2,3,4,6,8 --rounds 3 --frames 150 --out $O/plan_sweep.json > $O/plan_sweep.txt 2>&1; echo "plan_sweep rc=$?" >> $O/plan_sweep.txt; grep heuristic_behind $O/plan_sweep.txt
echo "== launch shapes in fresh processes (placement sensitivity)"
timeout 600 python tools/fresh_plans.py 250112,281600,218880 3 > $O/fresh_plans.txt 2>&1; tail -4 $O/fresh_plans.txt
timeout 600 python tools/fresh_plans.py 313856,333568,375040,530432,625152,797440 2 2:512,4:512,4:x1,2:x1 > $O/fresh_plans_final.txt 2>&1; tail -2 $O/fresh_plans_final.txt
echo "== the persistent grid against one step per wave, 46 sizes (final heuristics)"
timeout 600 python tools/onestep_sweep.py > $O/onestep_sweep.txt 2>>$O/bench.err; wc -l $O/onestep_sweep.txt
echo "== parity of every BASELINE config (max / p99.9)"
timeout 900 python tools/parity_report.py > $O/parity.txt 2>$O/parity.err; tail -14 $O/parity.txt
echo "== pullbench"
timeout 300 tools/pullbench > $O/pullbench.txt 2>&1; grep "(e)" $O/pullbench.txt | tail -40
echo "== live loop"
timeout 300 python tools/live_loop.py 2>&1 | tee $O/live_loop.txt
echo "== node frame loop"
timeout 300 python tools/node_frame_bench.py 2>&1 | tail -8 | tee $O/node_frame_bench.txt
fi
tail -3 $O/bench.err
