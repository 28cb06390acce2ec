# scratch A/B driver: variant libraries from build/ab/ through ILCC_HIP_LIB (tools/build_variant.sh builds them)
for L in "" $@; do
  for R in 1 2; do
  ILCC_HIP_LIB=${L:+$PWD/$L} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('AB lib=${L:-base}', round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['k6_ms_pipelined'],4), 'frac', round(d['roofline']['frac'],4), 'exec', round(d['roofline']['executed_fraction'],4))"
  done
done
