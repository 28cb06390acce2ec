for lib in "" build/ab/libilcc_k7_512.so build/ab/libilcc_k7_1024.so; do
  echo "== lib $lib"
  ILCC_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['frames_ok'], d['stage_ms_last_step_overlapped'])"
done
ILCC_HIP_LIB=build/ab/libilcc_k7_1024.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "local or stage or fixture" 2>&1 | tail -3
