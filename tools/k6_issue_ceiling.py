"""The K6 term's mix-weighted issue ceiling: the opcode histogram of one (point, candidate) evaluation
(tools/k6_isa_count.sh: gfx950 assembly of the probe kernels) priced opcode by opcode at the issue rates that
tools/ubench/valu_rate2.hip measured on the MI355X (profiles/r03a_ubench_valu_rates.txt), beside the nominal 2-cycle / 4-cycle
model.  Needs hipcc only.  usage: python tools/k6_issue_ceiling.py > profiles/r06_k6_issue_ceiling.json"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_T = 78.6   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz: one wave64 fp32 instruction per SIMD every 2 cycles (MI355X_MICROARCH.md)
isa = json.loads(subprocess.check_output([os.path.join(ROOT, "tools", "k6_isa_count.sh")], text=True))
rates = {}
for line in open(os.path.join(ROOT, "profiles", "r03a_ubench_valu_rates.txt")):
    m = re.match(r"(.+?)\s+([0-9.]+) ms\s+([0-9.]+) T lane-instr/s", line)
    if m:
        rates[m.group(1).strip()] = float(m.group(3))
# opcode (assembly mnemonic) -> row of the rate table
TABLE = {"v_add_f32_e32": "v_add_f32", "v_add_f32_e64": "v_add_f32_e64 |a|,|b|", "v_sub_f32_e32": "v_sub_f32", "v_subrev_f32_e32": "v_sub_f32",
         "v_sub_f32_e64": "v_sub_f32_e64 |a|, b", "v_mul_f32_e32": "v_mul_f32", "v_fma_f32": "v_fma_f32", "v_fmac_f32_e32": "v_fmac_f32",
         "v_floor_f32_e32": "v_floor_f32", "v_fract_f32_e32": "v_fract_f32", "v_min_f32_e32": "v_min_f32", "v_max_f32_e32": "v_max_f32",
         "v_med3_f32": "v_med3_f32", "v_cmp_le_f32_e32": "v_cmp_f32 (vcc)", "v_cndmask_b32_e32": "v_cndmask_b32_e64 (sgpr pair)",
         "v_cndmask_b32_e64": "v_cndmask_b32_e64 (sgpr pair)", "v_pk_add_f32": "v_pk_add_f32", "v_pk_fma_f32": "v_pk_fma_f32",
         "v_pk_mul_f32": "v_pk_mul_f32"}
out = {"peak_T_lane_instr_per_s": PEAK_T, "isa": isa, "rate_table": "profiles/r03a_ubench_valu_rates.txt", "classes": {}}
for cls, hist in isa["opcodes"].items():
    n = sum(hist.values())
    t_meas = sum(cnt / rates[TABLE[op]] for op, cnt in hist.items())          # ps per lane-evaluation at the measured rates (1 / T = ps)
    iu = isa["issue"][cls]
    out["classes"][cls] = {
        "valu_instr": n, "issue_units_nominal": iu["issue_units"],
        "mix_ceiling_nominal_T": PEAK_T * n / iu["issue_units"],              # every full-rate instruction at 78.6 T, every half-rate one at 39.3 T
        "mix_ceiling_measured_T": n / t_meas,                                  # every opcode at the rate the micro-benchmark measured for it
        "share_of_issue_cycles_half_rate": 2.0 * iu["half_rate_instr"] / iu["issue_units"],
        "opcode_rates_T": {op: rates[TABLE[op]] for op in hist}}
out["note"] = ("v_cndmask_b32 with vcc measured 6.9 T in a chain of dependent selects on one vcc (r03a): the kernel's selects read SGPR pairs "
               "written by distinct v_cmp -- priced at the SGPR-pair rate.  The measured full-rate figure (56-62 T) is 72-79 % of the "
               "2.4 GHz nominal: the chip does not hold 2.4 GHz under a VALU-saturating load")
print(json.dumps(out, indent=1))
