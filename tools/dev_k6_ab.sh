for cfg in "3 " "4 build/ab/libilcc_slots_4.so" "5 build/ab/libilcc_slots_5.so"; do
  set -- $cfg
  for q in 8 16; do
  echo "== depth $1 lib $2 queues $q"
  GPU_MAX_HW_QUEUES=$q ILCC_BENCH_MAX_DEPTH=8 ILCC_HIP_LIB=$2 timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --in-flight $1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3))"
  done
done
