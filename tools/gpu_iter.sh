# one iteration on the GPU while working on a kernel (run through gpurun from the repo root):  tools/gpu_iter.sh TAG ["pytest -k expr" | all | none]
#   parity tests (all of -m gpu by default), then the default bench and the config-5 bench, resident legs only, twice each
TAG=${1:-it}
KEXPR=${2:-all}
mkdir -p gpurun_out
if [ "$KEXPR" = "all" ]; then timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
elif [ "$KEXPR" != "none" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -8; fi
for R in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench.err || tail -5 gpurun_out/${TAG}_bench.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); r=d['roofline']; print('BENCH ${TAG}', round(d['value']), round(d['ms_per_step'],3), 'k6', round(r['k6_ms_pipelined'],4), 'full', round(r['k6_full_pass_ms_pipelined'],4), 'frac', round(r['frac'],4), d['frames_ok'], d['median_corner_error_mm_vs_ground_truth'], d['stage_ms_last_batch_overlapped'])"
done
for R in 1 2; do
timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench5.json 2>gpurun_out/${TAG}_bench5.err || tail -5 gpurun_out/${TAG}_bench5.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench5.json')); r=d['roofline']; print('CONFIG5 ${TAG}', round(d['value']), round(d['ms_per_step'],3), 'k6', round(r['k6_ms_pipelined'],4), 'full', round(r['k6_full_pass_ms_pipelined'],4), d['frames_ok'], d['median_corner_error_mm_vs_ground_truth'], d['stage_ms_last_batch_overlapped'])"
done
python tools/dev_batch_timeline.py 1 30 2>/dev/null | tail -1
