# steady-state period of the pipeline at several depths, no profiler (through gpurun from the repo root; needs build/ab/libilcc_hip_s8.so =
# tools/build_variant.sh s8 -DILCC_SLOTS=8):  tools/dev_depth_period.sh [depths...]
for D in ${@:-4 5 6}; do
  ILCC_HIP_LIB=$PWD/build/ab/libilcc_hip_s8.so python tools/dev_depth_timeline.py $D 80 gpurun_out/depth${D}_s8.json > /dev/null 2>&1
  python - $D <<'PY'
import json, sys
D = sys.argv[1]
t = json.load(open("gpurun_out/depth%s_s8.json" % D))
print("DEPTH", D, round(t["frames_per_s"]), "period", round(t["period_ms_start_to_start"], 3), "full", round(t["full_pass_ms"], 3), "idle",
      round(t["chain_idle_ms_between_full_passes"], 3), "own-front-end share", round(t["full_pass_waited_for_its_own_front_end_share"], 2), "life",
      round(t["batch_life_ms_first_to_last_kernel"], 2), {k: round(v, 2) for k, v in t["span_ms"].items()})
PY
done
