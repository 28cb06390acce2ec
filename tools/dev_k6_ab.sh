for cfg in "ILCC_K6_CHAIN=1" "ILCC_K6_CHAIN=0"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3))"
done
