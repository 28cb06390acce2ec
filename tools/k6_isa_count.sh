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
hist() {   # opcode histogram "op:count ..." of the kernel whose mangled name contains $1
  awk -v pat="$1" '
    $0 ~ "^_ZN4ilcc.*" pat "[^:]*:" {on = 1; next}
    on && /^\.Lfunc_end/ {on = 0}
    on && $1 ~ /^v_/ && $1 !~ /^v_readfirstlane|^v_readlane/ {n[$1]++}
    END {for (k in n) printf "%s:%d ", k, n[k]}' "$T/k6.s"
}
B1=$(count 1 1); B3=$(count 3 1); I1=$(count 1 0); I3=$(count 3 0); X1=$(countbox 1); X3=$(countbox 3)
python3 - "$B1" "$B3" "$I1" "$I3" "$X1" "$X3" "$(hist k6_isa_probeILi1ELb1EE)" "$(hist k6_isa_probeILi3ELb1EE)" "$(hist k6_isa_probeILi1ELb0EE)" \
    "$(hist k6_isa_probeILi3ELb0EE)" "$(hist k6_isa_probe_boxILi1EE)" "$(hist k6_isa_probe_boxILi3EE)" <<'PY'
import json, sys
b1, b3, i1, i3, x1, x3 = map(int, sys.argv[1:7])
def parse(t):
    return {kv.split(":")[0]: int(kv.split(":")[1]) for kv in t.split()}
def per_eval(h1, h3):   # opcode histogram of ONE evaluation: (3 calls - 1 call) / 2
    h1, h3 = parse(h1), parse(h3)
    return {k: (h3.get(k, 0) - h1.get(k, 0)) / 2.0 for k in sorted(set(h1) | set(h3)) if h3.get(k, 0) != h1.get(k, 0)}
# issue classes on gfx950 (tools/ubench/valu_rate2.hip, profiles/r03a_ubench_valu_rates.txt): fp32 add / sub / mul / fma / fmac and the
# integer logic ops issue a wave64 instruction in 2 cycles ("full": 1 unit); floor / fract / rndne, min / max / med3, v_cmp, v_cndmask,
# v_bfi, v_lshlrev, cvt and DPP forms measure at half of that ("half": 2 units)
HALF = ("v_floor", "v_fract", "v_rndne", "v_min", "v_max", "v_med3", "v_cmp", "v_cndmask", "v_bfi", "v_lshl", "v_cvt", "v_pk_")
def units(h):
    full = sum(n for k, n in h.items() if not k.startswith(HALF))
    half = sum(n for k, n in h.items() if k.startswith(HALF))
    return {"full_rate_instr": full, "half_rate_instr": half, "issue_units": full + 2.0 * half,
            "mix_ceiling_over_peak": (full + half) / (full + 2.0 * half) if full + half else None}
hb, hi, hx = per_eval(sys.argv[7], sys.argv[8]), per_eval(sys.argv[9], sys.argv[10]), per_eval(sys.argv[11], sys.argv[12])
print(json.dumps({"border_valu_per_eval": (b3 - b1) / 2.0, "interior_valu_per_eval": (i3 - i1) / 2.0,
                  "box_valu_per_tile_eval": (x3 - x1) / 2.0,
                  "probe_counts": {"border_1": b1, "border_3": b3, "interior_1": i1, "interior_3": i3, "box_1": x1, "box_3": x3},
                  "opcodes": {"border": hb, "interior": hi, "box": hx},
                  "issue": {"border": units(hb), "interior": units(hi), "box": units(hx)},
                  "how": "hipcc -O3 --offload-arch=gfx950 -fno-honor-nans -DILCC_K6_ISA_PROBE -S k6_grid_cost.hip; v_* instructions of "
                         "k6_isa_probe<3,*> minus k6_isa_probe<1,*>, halved; issue units: 1 per full-rate, 2 per half-rate instruction"}))
PY
rm -rf "$T"
