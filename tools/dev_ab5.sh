# A/B on BASELINE config 5 (run through gpurun from the repo root): tools/dev_ab5.sh NAME...  -> build/ab/libilcc_hip_NAME.so (base = the in-tree library), resident leg only, twice each
for V in "$@"; do
  for R in 1 2; do
    L=$([ "$V" = base ] && echo lidar_camera_calibration_amd/libilcc_hip.so || echo build/ab/libilcc_hip_$V.so)
    ILCC_HIP_LIB=$PWD/$L timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('CONFIG5 $V', round(d['value']), 'k6 alone', round(r['k6_ms_alone'],3), 'locate', round(r['k6_locate_ms_alone'],3), 'prepass', round(r['k6_prepass_ms_alone'],3), 'full', round(r['k6_full_pass_ms_alone'],3), 'full pipelined', round(r['k6_full_pass_ms_pipelined'],3), 'evals', round(r['evals_executed_per_batch']/1e6,1), 'box', round(r['box_evals_per_batch']/1e6,1), d['frames_ok'])"
  done
done
