# pipeline-depth study (run through gpurun from the repo root; needs build/ab/libilcc_hip_s8.so = tools/build_variant.sh s8 -DILCC_SLOTS=8):
#   frames/s at 2..8 batches in flight, and a rocprofv3 kernel trace at depths 4 / 6 / 8 summarised by tools/trace_timeline.py
#   (hardware queue of every stream, kernels running concurrently, K6 full passes in the window)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for D in $([ "$1" = trace ] || echo 2 3 4 5 6 8); do
  for RUN in 1 2; do
  GPU_MAX_HW_QUEUES=12 ILCC_BENCH_MAX_DEPTH=$D ILCC_HIP_LIB=$R/build/ab/libilcc_hip_s8.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --in-flight $D > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('DEPTH $D', round(d['value']), round(d['ms_per_step'],3), 'k6 pipelined', round(r['k6_ms_pipelined'],4), 'full', round(r['k6_full_pass_ms_pipelined'],4), d['stage_ms_last_batch_overlapped'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for D in 4 6 8; do
  rm -rf $R/gpurun_out/prof_depth$D
  GPU_MAX_HW_QUEUES=12 ILCC_BENCH_GEN_WORKERS=1 ILCC_BENCH_MAX_DEPTH=$D ILCC_HIP_LIB=$R/build/ab/libilcc_hip_s8.so timeout 420 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_depth$D -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-alone-leg --in-flight $D > /tmp/o.json 2>/dev/null
  F=$(find $R/gpurun_out/prof_depth$D -name "*kernel_trace.csv" | head -1)
  echo "=== depth $D: $(python -c "import json; d=json.load(open('/tmp/o.json')); print(round(d['value']), d['ms_per_step'])")"
  python $R/tools/trace_timeline.py $F 8 4 summary > $R/gpurun_out/r05_depth${D}_timeline.txt
  python $R/tools/trace_timeline.py $F 3 6 >> $R/gpurun_out/r05_depth${D}_timeline.txt
  head -32 $R/gpurun_out/r05_depth${D}_timeline.txt
  rm -rf $R/gpurun_out/prof_depth$D
done
