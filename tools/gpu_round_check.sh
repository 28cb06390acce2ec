# full GPU verification + profiles for the round (run through gpurun from the repo root)
TAG=${1:-r01h}
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 300 gpurun_out/bench_$TAG.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cd $R
find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-150 {} | head -14"
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$TAG.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['valu']['executed_fraction'], d['pcie_inclusive']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'], d['cpu_baseline']['gpu_vs_cpu_corner_deviation_mm']['max'], d['frames_ok'])
PY
