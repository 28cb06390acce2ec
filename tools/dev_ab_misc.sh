run() { ILCC_HIP_LIB=${1:+$PWD/$1} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs $2 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('AB lib=${1:-base} $2', round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['k6_ms_pipelined'],4))"; }
for R in 1 2; do
run "" ""
run build/ab/libilcc_hip_nochain.so ""
run "" "--frames-per-batch 128 --batches-per-step 8"
run "" "--frames-per-batch 192 --batches-per-step 5"
run "" "--frames-per-batch 384 --batches-per-step 3"
run "" "--frames-per-batch 512 --batches-per-step 2"
done
