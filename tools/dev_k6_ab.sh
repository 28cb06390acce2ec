for fpg in 128 256 512; do
  echo "== frames-per-gpu $fpg"
  ILCC_SEED_STRIDE_T=5 timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --frames-per-gpu $fpg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3), d['frames_ok'], d['stage_ms_last_step_overlapped'])"
done
