"""Collapse rocprofv3 --pmc passes (one directory per pass, each holding p_counter_collection.csv) into one JSON: per kernel whose
name contains one of the given substrings, the mean counter values PER LAUNCH over the last N launches (warm-up launches dropped),
the profiled duration per pass, the gfx950-corrected memory-side traffic (FETCH_SIZE is KiB and counts 64 B per 128-B request for
16-B-per-lane streams: x 1024 x 2; WRITE_SIZE KiB x 1024 — MI355X_MICROARCH.md, HBM section) and a few derived ratios.
    python scripts/pmc_kernels_to_json.py <pmc_dir> <out.json> <last_n> <name substring> [<name substring> ...]"""
import csv
import json
import os
import sys
from collections import defaultdict

pmc, out, last_n, subs = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
res = {}
for sub in subs:
    acc, durs, meta = {}, {}, {}
    for p in sorted(os.listdir(pmc)):
        f = os.path.join(pmc, p, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        per_disp = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            d = per_disp[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])     # summed over XCDs / instances
            d["__dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            meta = {"kernel": r["Kernel_Name"][:160], "grid_threads": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                    "lds_bytes": int(r["LDS_Block_Size"]), "vgprs": int(r["VGPR_Count"]), "agprs": int(r["Accum_VGPR_Count"]), "sgprs": int(r["SGPR_Count"])}
        ids = sorted(per_disp)[-last_n:]
        if not ids:
            continue
        durs[p] = round(sum(per_disp[i]["__dur"] for i in ids) / len(ids), 2)
        for c in per_disp[ids[0]]:
            if c != "__dur":
                acc[c] = sum(per_disp[i].get(c, 0.0) for i in ids) / len(ids)
    if not acc:
        continue
    e = dict(meta)
    e["launches_averaged"] = last_n
    e["profiled_dur_us"] = durs
    e.update({k: acc[k] for k in sorted(acc)})
    if "FETCH_SIZE" in e:
        e["read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024
    if "read_bytes_corrected" in e and "write_bytes" in e:
        e["traffic_bytes"] = e["read_bytes_corrected"] + e["write_bytes"]
        d = durs.get("fetch") or next(iter(durs.values()))
        e["traffic_GBps_profiled"] = round(e["traffic_bytes"] / (d * 1e-6) / 1e9, 1)
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in e:
                e[k.lower() + "_frac_of_wave_cycles"] = round(e[k] / e["SQ_WAVE_CYCLES"], 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["mfma_util_profiled"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 1024 / 8), 4)
    if "SQ_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        e["sq_busy_over_gui_active"] = round(e["SQ_BUSY_CYCLES"] / e["GRBM_GUI_ACTIVE"], 4)
    res[sub] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
