for lib in "" build/ab/libilcc_u2.so build/ab/libilcc_u3.so build/ab/libilcc_u6.so; do
  echo "== lib $lib"
  ILCC_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'k6', round(d['roofline']['launch_ms'],3), 'exec', round(d['roofline']['valu']['executed_fraction'],4), d['frames_ok'])"
done
