for cfg in "ILCC_K6_PRIO=0" "ILCC_K6_PRIO=1"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3), d['frames_ok'])"
done
