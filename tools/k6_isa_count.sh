#!/bin/sh
# VALU instructions of ONE (point, candidate) evaluation of k6_grid_cost, counted in the gfx950 assembly of the probe
# kernels in csrc/k6_grid_cost.hip (-DILCC_K6_ISA_PROBE), compiled with the library's own flags (csrc/Makefile):
#   per evaluation = (VALU instructions of the 3-call kernel - those of the 1-call kernel) / 2
# Prints a JSON object; `tools/k6_isa_count.sh > profiles/rNN_k6_isa_count.json` is what bench.py's K6_VALU_OPS_* constants
# are checked against (tests/test_host_logic.py::test_k6_credit_matches_the_isa).  Needs hipcc only (no GPU).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd "$R/lidar_camera_calibration_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -fno-honor-nans -DILCC_K6_ISA_PROBE \
    -S --cuda-device-only k6_grid_cost.hip -o "$T/k6.s" 2>/dev/null
count() {   # VALU instructions (v_*) of the probe kernel <N, BORDER>; v_readfirstlane / v_readlane are data movement to the scalar side
  awk -v pat="k6_isa_probeILi$1ELb$2EE" '
    $0 ~ "^_ZN4ilcc.*" pat "[^:]*:" {on = 1; next}
    on && /^\.Lfunc_end/ {on = 0}
    on && $1 ~ /^v_/ && $1 !~ /^v_readfirstlane|^v_readlane/ {n++}
    END {print n + 0}' "$T/k6.s"
}
countbox() {   # the box pre-pass's term: k6_isa_probe_box<N>
  awk -v pat="k6_isa_probe_boxILi$1EE" '
    $0 ~ "^_ZN4ilcc.*" pat "[^:]*:" {on = 1; next}
    on && /^\.Lfunc_end/ {on = 0}
    on && $1 ~ /^v_/ && $1 !~ /^v_readfirstlane|^v_readlane/ {n++}
    END {print n + 0}' "$T/k6.s"
}
B1=$(count 1 1); B3=$(count 3 1); I1=$(count 1 0); I3=$(count 3 0); X1=$(countbox 1); X3=$(countbox 3)
python3 - "$B1" "$B3" "$I1" "$I3" "$X1" "$X3" <<'PY'
import json, sys
b1, b3, i1, i3, x1, x3 = map(int, sys.argv[1:7])
print(json.dumps({"border_valu_per_eval": (b3 - b1) / 2.0, "interior_valu_per_eval": (i3 - i1) / 2.0,
                  "box_valu_per_tile_eval": (x3 - x1) / 2.0,
                  "probe_counts": {"border_1": b1, "border_3": b3, "interior_1": i1, "interior_3": i3, "box_1": x1, "box_3": x3},
                  "how": "hipcc -O3 --offload-arch=gfx950 -fno-honor-nans -DILCC_K6_ISA_PROBE -S k6_grid_cost.hip; v_* instructions of "
                         "k6_isa_probe<3,*> minus k6_isa_probe<1,*>, halved"}))
PY
rm -rf "$T"
