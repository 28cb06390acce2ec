# one iteration on ILCC_SOLVER_REFERENCE_LOCAL (K7a) through gpurun from the repo root: tools/gpu_ref_iter.sh TAG [sweep frames]
TAG=${1:-ref}; NF=${2:-1024}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "local or reference or smoke or solver" 2>&1 | tail -5
timeout 600 python tools/grid_parity_sweep.py $NF 777 local gpurun_out/${TAG}_parity_sweep_local.json 2>&1 | tail -3
python tools/dev_batch_timeline.py 0 30 128 2>/dev/null | tail -1
python tools/dev_batch_timeline.py 0 12 1024 2>/dev/null | tail -1
timeout 300 python bench.py --solver reference --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench_reference.json 2>gpurun_out/${TAG}_bench_reference.err || tail -5 gpurun_out/${TAG}_bench_reference.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_reference.json')); print('BENCH --solver reference', round(d['value']), round(d['ms_per_step'],3), d['frames_ok'], d['stage_ms_last_batch_overlapped'])"
