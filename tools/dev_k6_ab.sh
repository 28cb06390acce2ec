for cfg in "ILCC_K6_REFINE_RADIUS=6" "ILCC_K6_REFINE_RADIUS=3" "ILCC_K6_REFINE_RADIUS=4" "ILCC_K6_REFINE_RADIUS=8" "ILCC_SEED_STRIDE_TH=8 ILCC_K6_REFINE_RADIUS=4" "ILCC_SEED_STRIDE_TH=6 ILCC_K6_REFINE_RADIUS=3"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3), 'exec', round(d['roofline']['valu']['executed_fraction'],4), d['frames_ok'])"
done
