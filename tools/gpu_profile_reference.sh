# profile of ILCC_SOLVER_REFERENCE_LOCAL (the mode with reference-path parity; K7a = k7a_local_solve dominates it), through gpurun
# from the repo root:  tools/gpu_profile_reference.sh TAG [pmc] [timing]
#   - un-contended 128-frame and 1024-frame batch timelines (HIP events)
#   - rocprofv3 --kernel-trace --stats of `bench.py --solver reference`, pipelined and --in-flight 1
#   - with `pmc`: PMC passes of tools/pmc_target.py 1024 2 0 (tools/gpu_pmc.sh)
#   - with `timing`: K7a's per-iteration cycle split (dogleg / evaluate<false> / evaluate<true>) from a -DILCC_K7_TIMING build
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python tools/dev_batch_timeline.py 0 30 128 2>/dev/null | tail -1 | tee $R/gpurun_out/${TAG}_timeline_reference.json
python tools/dev_batch_timeline.py 0 12 1024 2>/dev/null | tail -1 | tee $R/gpurun_out/${TAG}_timeline_reference_1024f.json
cd /tmp && export TMPDIR=/tmp
for MODE in "reference:--no-alone-leg" "reference_inflight1:--in-flight 1"; do
  N=${MODE%%:*}; A=${MODE#*:}
  ILCC_BENCH_GEN_WORKERS=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats_$N -- python $R/bench.py --solver reference --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs $A > $R/gpurun_out/prof_${TAG}_bench_$N.json 2> /dev/null
  F=$(find $R/gpurun_out/prof_${TAG}_stats_$N -name "*kernel_stats.csv" | head -1)
  cp $F $R/gpurun_out/${TAG}_kernel_stats_$N.csv; cut -c1-150 $F | head -14
  python -c "
import json; d=json.load(open('$R/gpurun_out/prof_${TAG}_bench_$N.json')); print('BENCH --solver reference under rocprof ($N)', round(d['value']), d['ms_per_step'])"
done
cd $R
for A in "$@"; do
  if [ "$A" = "pmc" ]; then tools/gpu_pmc.sh $TAG 1024 2 0 | tail -14 | cut -c1-260; fi
  if [ "$A" = "timing" ] && [ -f build/ab/libilcc_hip_k7t.so ]; then
    ILCC_HIP_LIB=$R/build/ab/libilcc_hip_k7t.so python tools/dev_batch_timeline.py 0 3 128 2>&1 | grep "K7a f0" | tail -4 | tee $R/gpurun_out/${TAG}_k7a_cycles.txt
  fi
done
