"""Collapse the rocprofv3 --pmc passes of scripts/gpu_attn_pmc.sh into profiles/<out>.json (per launch of
vit_attention_persistent_kernel at B = 64: 512 workgroups x 4 waves x 2 items, 12.08 GFLOP).   python scripts/pmc_attn_to_json.py [out_json]"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "pmc_attn")
acc, dur = defaultdict(list), []
for p in ("sq", "sq2"):
    f = os.path.join(base, p, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    seen = set()
    for r in csv.DictReader(open(f)):
        if "vit_attention_persistent_kernel" not in r["Kernel_Name"]:      # B = 64 runs on the persistent form since run O of round 2
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p == "sq" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
res = {k: sum(v) / len(v) for k, v in acc.items()}
waves = 512 * 4                          # 512 persistent workgroups x 4 waves, two (crop, head) items each
mfma_per_wave = 2 * 1440                 # 16x16x4 fp32, 8 passes = 32 cycles each on one SIMD
out = {"kernel": "vit_attention_persistent_kernel<0>", "B": 64, "workgroups": 512, "items": 1024, "flop_per_launch": 4.0 * 64 * 16 * 192 * 192 * 80,
       "profiled_dur_us": sum(dur) / max(len(dur), 1), **res}
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is per-launch wall cycles
    out["mfma_util_profiled"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] * 1024 / 8) if res["GRBM_GUI_ACTIVE"] else None
    out["mfma_busy_cycles_expected"] = waves * mfma_per_wave * 32
if "SQ_WAIT_ANY" in res and "SQ_WAVE_CYCLES" in res:
    out["wait_any_frac"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
if "SQ_INSTS_VALU" in res:
    out["valu_insts_per_wave_incl_mfma"] = res["SQ_INSTS_VALU"] / waves
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r2_pmc_attention.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
