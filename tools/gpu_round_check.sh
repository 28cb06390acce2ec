# full GPU verification + profiles for the round (run through gpurun from the repo root)
set -x
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
python bench.py > gpurun_out/bench_r01g.json 2> gpurun_out/bench_r01g.err; tail -c 300 gpurun_out/bench_r01g.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g_stats -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_g_fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_g_write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/prof_g_sq -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
cd $R
find gpurun_out/prof_g_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cut -c1-150 {} | head -12"
python - <<'PY'
import csv, glob, collections, json
d=json.load(open('gpurun_out/bench_r01g.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['valu'], d.get('pcie_inclusive',{}).get('value'), d['cpu_baseline']['value'], d['cpu_baseline']['gpu_vs_cpu_corner_deviation_mm'])
for tag in ('fetch','write','sq'):
    fs=glob.glob('gpurun_out/prof_g_%s/**/*counter_collection.csv'%tag, recursive=True)
    if not fs: print(tag,'no file'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        key=(r['Kernel_Name'][:36], r.get('Grid_Size'))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k6_grid' in k[0] or 'k1_roi_count' in k[0]:
            print(tag,k,{c:(sum(x)/len(x)) for c,x in v.items()}, 'n', len(next(iter(v.values()))))
PY
