"""Collapse the four rocprofv3 --pmc passes of scripts/gpu_profile.sh (gpurun_out/pmc3/{sq,fetch,write,l2}) into
profiles/r1_pmc_gemm.json: per GEMM shape the mean counter values per launch, the gfx950-corrected HBM-side traffic
(FETCH_SIZE is in KiB and counts 64 B per 128-B request for 16 B/lane streams: x1024 x2; WRITE_SIZE KiB x1024) and derived ratios.
    python scripts/pmc_to_json.py [pmc_dir] [out_json]"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc3")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r1_pmc_gemm.json")

M = 64 * 192
# (EPI id in the kernel name, N, K) of the shapes scripts/gemm_bench.py launches; proj and fc2 share the bias+residual kernel
SHAPES = {"qkv": ("5", 3840, 1280), "fc1": ("2", 5120, 1280)}
res = {}
for shape, (epi, N, K) in SHAPES.items():
    tag = f"gemm_f32_kernel<4, 1, 1, 5, true, {epi},"
    acc, durs = defaultdict(list), {}
    for p in ("sq", "fetch", "write", "l2"):
        f = os.path.join(pmc, p, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        d = []
        seen = set()
        for r in csv.DictReader(open(f)):
            if tag not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if d:
            durs[p] = round(sum(d) / len(d), 3)
    if not acc:
        continue
    e = {"kernel": f"gemm_f32_kernel<4, 1, 1, 5, true, {epi}>", "M": M, "N": N, "K": K}
    e.update({k: sum(v) / len(v) for k, v in sorted(acc.items())})
    e["profiled_dur_us"] = durs
    e["algorithmic_bytes"] = 4 * (M * K + N * K + M * N)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        e["traffic_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
        e["traffic_over_algorithmic"] = e["traffic_bytes"] / e["algorithmic_bytes"]
    if "TCC_HIT_sum" in e:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        # busy cycles are summed over the 1024 SIMDs... per MI355X_MICROARCH.md: util = MFMA_BUSY / (GUI_ACTIVE * 4 SIMD * 256 CU / 8 XCD-normalisation)
        e["mfma_util_profiled"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128)
        e["wait_any_frac"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAVE_CYCLES") else None
    res[shape] = e
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in v.items() if x in
              ("traffic_bytes", "traffic_over_algorithmic", "l2_hit_rate", "mfma_util_profiled", "wait_any_frac", "SQ_LDS_BANK_CONFLICT", "profiled_dur_us")})
