# A/B on the default bench (config 2, run through gpurun from the repo root): tools/dev_ab2.sh NAME...  -> build/ab/libilcc_hip_NAME.so
# (NAME = base: the in-tree library), resident leg only, twice each
for V in "$@"; do
  for R in 1 2; do
    ILCC_HIP_LIB=$([ "$V" = base ] && echo lidar_camera_calibration_amd/libilcc_hip.so || echo build/ab/libilcc_hip_$V.so) timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('CONFIG2 $V', round(d['value']), 'k6', round(r['launch_ms'],4), 'full', round(r['full_pass_ms'],4), d['frames_ok'])"
  done
done
