# quick GPU check while iterating on a kernel (run through gpurun from the repo root):
#   tools/gpu_quick.sh TAG ["pytest -k expression"]  -> parity subset, un-contended timeline, default bench (no extra legs)
TAG=${1:-q}
KEXPR=${2:-"grid_cost or every_stage or branch_and_bound or near_tie"}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$KEXPR" 2>&1 | tail -4
python tools/dev_batch_timeline.py 1 30 2>/dev/null | tail -1
for R in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); r=d['roofline']; print('BENCH ${TAG}', round(d['value']), round(d['ms_per_step'],3), 'k6', round(r['k6_ms_pipelined'],4), 'frac', round(r['frac'],4), 'exec', round(r['executed_fraction'],4), d['frames_ok'], d['median_corner_error_mm_vs_ground_truth'])"
done
