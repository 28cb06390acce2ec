for spec in "build/ab/libilcc_hip_s6.so:5" "build/ab/libilcc_hip_s6.so:6" "build/ab/libilcc_hip_s8.so:8" ":4"; do
  L=${spec%%:*}; D=${spec##*:}
  for R in 1 2; do
  GPU_MAX_HW_QUEUES=12 ILCC_BENCH_MAX_DEPTH=$D ILCC_HIP_LIB=${L:+$PWD/$L} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --in-flight $D > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('AB lib=${L:-base} depth $D', round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['k6_ms_pipelined'],4))"
  done
done
