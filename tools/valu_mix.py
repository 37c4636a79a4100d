"""Prices a kernel's VALU instruction mix (SQ_INSTS_VALU_* classes from tools/pmc_sweep_anatomy.sh) with the issue costs measured by
tools/valu_issue.hip: the time the SIMDs need just to issue that mix.   tools/valu_mix.py profiles/r02c/anat_report_sweeps.txt profiles/valu_issue.json > profiles/valu_mix.json

Cost classes (cycles per wave64 instruction on one SIMD = 4 SIMDs x 256 CUs x 2.4 GHz / measured G inst/s at 4 waves per SIMD):
  plain    v_add / v_mul / v_fma f32, integer ALU with VGPR or inline-constant operands          -> v_fma_f32_3_distinct_vgpr
  trans    v_sqrt / v_rcp / v_rsq                                                               -> v_rcp_f32
  other    everything else the class counters do not name: v_cmp, v_cndmask, v_max, v_mov, shifts with an SGPR operand ...
           priced as an SGPR-operand instruction                                                -> v_mul_f32_sgpr
"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import build as hip_build
report, issue = sys.argv[1], json.load(open(sys.argv[2]))
rate = lambda name: issue["results"][name]["4_waves_per_simd"]
simd_ghz = issue["compute_units"] * 4 * 2.4
cyc = {"plain": simd_ghz / rate("v_fma_f32_3_distinct_vgpr"), "trans": simd_ghz / rate("v_rcp_f32"), "other": simd_ghz / rate("v_mul_f32_sgpr")}
per = {}
for line in open(report):
    m = re.match(r"\s+(k_\w+<[^>]*>)\s+n=\d+\s+(.*)", line)
    if m:
        d = per.setdefault(m.group(1), {})
        for kv in m.group(2).split():
            k, v = kv.split("=")
            d[k] = float(v) * 32            # the report holds means per (launch, XCD x SE instance): 32 instances per launch
out = {"csrc_sha256": hip_build.sources_sha256(), "issue_cycles_per_instruction": cyc, "source": "SQ_INSTS_VALU_* (rocprofv3 --pmc, %s) priced with tools/valu_issue.hip (profiles/valu_issue.json)" % report, "kernels": {}}
for k, d in per.items():
    if "SQ_INSTS_VALU" not in d or "SQ_INSTS_VALU_FMA_F32" not in d:
        continue
    plain = d["SQ_INSTS_VALU_ADD_F32"] + d["SQ_INSTS_VALU_MUL_F32"] + d["SQ_INSTS_VALU_FMA_F32"] + d["SQ_INSTS_VALU_INT32"]
    trans = d["SQ_INSTS_VALU_TRANS_F32"]
    other = d["SQ_INSTS_VALU"] - plain - trans
    cycles = plain * cyc["plain"] + trans * cyc["trans"] + other * cyc["other"]
    out["kernels"][k] = {"wave_insts_per_launch": d["SQ_INSTS_VALU"], "plain": plain, "trans": trans, "other": other,
                         "issue_floor_us": cycles / (issue["compute_units"] * 4) / 2.4e3,
                         "lane_utilisation": d.get("SQ_THREAD_CYCLES_VALU", 0) / d["SQ_INSTS_VALU"] / 64 if d.get("SQ_THREAD_CYCLES_VALU") else None}
print(json.dumps(out, indent=1))
