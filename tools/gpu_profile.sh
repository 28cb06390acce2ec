# kernel-level profile of the default bench (run through gpurun from the repo root): rocprofv3 kernel stats of the
# timed pipeline only (no extra legs), the un-contended single-batch timeline, optional PMC passes
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python tools/dev_batch_timeline.py 1 30 2>/dev/null | tail -1
python tools/dev_batch_timeline.py 0 30 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
ILCC_BENCH_GEN_WORKERS=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof_${TAG}_bench.json 2> /dev/null
cd $R
find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cp {} gpurun_out/${TAG}_kernel_stats.csv; cut -c1-160 {} | head -16"
python -c "
import json; d=json.load(open('gpurun_out/prof_${TAG}_bench.json')); print('BENCH under rocprof', round(d['value']), d['ms_per_step'], d['roofline']['launch_ms'])"
if [ "$2" = "pmc" ]; then
  cd /tmp
  for C in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/prof_${TAG}_pmc_$N -- python $R/tools/pmc_target.py > /dev/null 2>&1
  done
  cd $R
  python tools/pmc_summary.py gpurun_out/prof_${TAG}_pmc_ > gpurun_out/${TAG}_pmc_summary.csv; cat gpurun_out/${TAG}_pmc_summary.csv
fi
