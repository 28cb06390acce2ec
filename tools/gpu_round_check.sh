# full GPU verification for the round (run through gpurun from the repo root): tests, smoke, the driver's bench command
# (--steps 20 --warmup 5) and a long run beside it
TAG=${1:-r06}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_20_5.json 2> gpurun_out/bench_${TAG}_20_5.err; tail -c 600 gpurun_out/bench_${TAG}_20_5.err
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_${TAG}_200_20.json 2> gpurun_out/bench_${TAG}_200_20.err; tail -c 300 gpurun_out/bench_${TAG}_200_20.err
python - <<PY
import json
for n in ("20_5","200_20"):
    try:
        d=json.load(open('gpurun_out/bench_${TAG}_%s.json'%n))
        print('BENCH',n, round(d['value']), round(d['ms_per_step'],3), 'h2d', d.get('value_h2d_inclusive'), 'k6 ms alone', round(d['roofline']['k6_ms_alone'],4), 'pipelined', round(d['roofline']['k6_ms_pipelined'],4), 'valu frac', round(d['roofline']['frac'],4), 'exec', d['roofline']['executed_fraction'], d['frames_ok'], 'amb', d['frames_flagged_ambiguous'], 'max', d['max_corner_error_mm_vs_ground_truth'], 'med', d['median_corner_error_mm_vs_ground_truth'], 'stages', d['stage_ms_last_batch_overlapped'])
        if 'grid_vs_reference_path_mm' in d: print('   GRIDvsREF', {k:v for k,v in d['grid_vs_reference_path_mm'].items() if k!='what'})
        if 'cpu_baseline' in d: print('   CPU', d['cpu_baseline']['value'], d['cpu_baseline']['runs_frames_per_s'], d['cpu_baseline']['all_cores_value'], d['cpu_baseline']['gpu_vs_cpu_corner_deviation_mm'])
        if 'pcie_inclusive' in d: print('   PCIE', {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='how'}) for k,v in d['pcie_inclusive'].items()})
    except Exception as e: print('BENCH',n,'failed',e)
PY
timeout 900 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_config5.json 2> gpurun_out/bench_${TAG}_config5.err; tail -c 300 gpurun_out/bench_${TAG}_config5.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_${TAG}_config5.json')); r=d['roofline']
print('CONFIG5', round(d['value']), round(d['ms_per_step'],2), 'h2d', d.get('value_h2d_inclusive'), 'k6 ms alone', round(r['k6_ms_alone'],3), 'frac', round(r['frac'],4), 'traffic', r['traffic'], d['frames_ok'], 'med', d['median_corner_error_mm_vs_ground_truth'], 'max', d['max_corner_error_mm_vs_ground_truth'], d['stage_ms_last_batch_overlapped'])
d=json.load(open('gpurun_out/bench_${TAG}_20_5.json')); r=d['roofline']
print('ROOFLINE', {k: r[k] for k in ('achieved','frac','frac_pipelined','frac_rocprof_alone','traffic','k6_ms_alone','k6_ms_pipelined','executed_fraction','valu_instr_per_eval','uncredited_share','hbm_frac','k1_hbm_frac','h2d_inclusive_frames_per_s','link_frac')})
print('NOISE', d.get('noise_floor_mm', {}).get('variants')); print('ACCEPT', d.get('accept_rule')); print('WARM', d.get('warmup_extra_steps_until_steady'), d.get('warmup_s'))
PY
