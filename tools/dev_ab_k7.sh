for L in "" build/ab/libilcc_hip_k7w4.so; do
  for R in 1 2; do
  ILCC_HIP_LIB=${L:+$GRAFT_REPO_ROOT/$L} python bench.py --solver reference --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > /tmp/o.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('K7A lib=${L:-base}', round(d['value']), round(d['ms_per_step'],3), d['frames_ok'])"
  done
  ILCC_HIP_LIB=${L:+$GRAFT_REPO_ROOT/$L} python tools/dev_batch_timeline.py 0 12 1024 2>/dev/null | tail -1
done
