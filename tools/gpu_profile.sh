# kernel-level profile of the default bench (run through gpurun from the repo root): tools/gpu_profile.sh TAG [pmc]
#   - un-contended single-batch timeline (HIP events), both solver modes
#   - rocprofv3 --kernel-trace --stats of the default bench, timed pipeline only (4 batches in flight) -> <TAG>_kernel_stats_bench_20_5.csv
#   - the same with --in-flight 1 (one batch alone on the chip: clean per-kernel durations) -> <TAG>_kernel_stats_bench_inflight1.csv
#   - with `pmc`: PMC passes at the batch sizes the bench runs (tools/gpu_pmc.sh: config 2 / 512 frames, config 5 / 64 frames)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python tools/dev_batch_timeline.py 1 30 2>/dev/null | tail -1 | tee $R/gpurun_out/${TAG}_timeline_grid.json
python tools/dev_batch_timeline.py 0 30 2>/dev/null | tail -1 | tee $R/gpurun_out/${TAG}_timeline_reference.json
cd /tmp && export TMPDIR=/tmp
for MODE in "20_5:--no-alone-leg" "inflight1:--in-flight 1"; do
  N=${MODE%%:*}; A=${MODE#*:}
  ILCC_BENCH_GEN_WORKERS=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats_$N -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs $A > $R/gpurun_out/prof_${TAG}_bench_$N.json 2> /dev/null
  F=$(find $R/gpurun_out/prof_${TAG}_stats_$N -name "*kernel_stats.csv" | head -1)
  cp $F $R/gpurun_out/${TAG}_kernel_stats_bench_$N.csv; cut -c1-150 $F | head -18
  python -c "
import json; d=json.load(open('$R/gpurun_out/prof_${TAG}_bench_$N.json')); print('BENCH under rocprof ($N)', round(d['value']), d['ms_per_step'], d['roofline']['k6_ms_pipelined'])"
done
cd $R
# BASELINE's roofline run (config 5): the same two kernel traces
for MODE in "config5:--no-alone-leg" "config5_inflight1:--in-flight 1"; do
  N=${MODE%%:*}; A=${MODE#*:}
  ILCC_BENCH_GEN_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats_$N -- python $R/bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs $A > $R/gpurun_out/prof_${TAG}_bench_$N.json 2> /dev/null
  F=$(find $R/gpurun_out/prof_${TAG}_stats_$N -name "*kernel_stats.csv" | head -1)
  cp $F $R/gpurun_out/${TAG}_kernel_stats_$N.csv; cut -c1-150 $F | head -18
  python -c "
import json; d=json.load(open('$R/gpurun_out/prof_${TAG}_bench_$N.json')); print('BENCH under rocprof ($N)', round(d['value']), d['ms_per_step'], d['roofline']['k6_ms_pipelined'])"
done
if [ "$2" = "pmc" ]; then
  tools/gpu_pmc.sh $TAG 1024 2 | tail -14 | cut -c1-220
  tools/gpu_pmc.sh $TAG 128 5 | tail -16 | cut -c1-220
fi
