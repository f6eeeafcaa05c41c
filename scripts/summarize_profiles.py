#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_round.sh (gpurun_out/<tag>_*) into the files committed under
profiles/: per-workload kernel stats CSV, a PMC summary JSON (HBM traffic per launch of the WaveNet kernels with the
gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled;
units KiB) and MFMA utilisation."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
summary = {}


def short(name):
    if "wn_gate_winograd" in name or ("conv1d_mfma_dma_kernel" in name and ", 1>" in name):
        return "gate"
    if "wn_resskip_f16_kernel" in name:
        return "res_skip_f16"
    if "wn_resskip_kernel" in name or "wn_resskip_wide_kernel" in name or "wn_resskip_wave_kernel" in name or ("conv1d_mfma_kernel" in name and ", 2, true" in name):
        return "res_skip"
    return None


for wl in ("config2_sp_b1_10s", "config3_si_b16_10s", "config5_sp_stream64"):
    entry = {}
    stats = sorted(glob.glob(os.path.join(src, f"{tag}_trace_{wl}", "*", "*kernel_stats.csv")), key=os.path.getmtime,
                   reverse=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"), "w") as fo:
            fo.write(open(stats[0]).read())
        # several kernels can map to one category (since round 4 mbx_create's calibration forwards run the F(2,3) and the
        # direct form a few times each): the category is represented by the kernel with the largest total time
        for rr in rows:
            kk = short(rr["Name"])
            if kk and float(rr["TotalDurationNs"]) > entry.get(kk, {}).get("total_ns", -1.0):
                entry[kk] = {"avg_us_trace": float(rr["AverageNs"]) / 1e3, "calls": int(rr["Calls"]),
                             "total_ns": float(rr["TotalDurationNs"]), "kernel": rr["Name"]}
    for counter_dir, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", None)):
        files = sorted(glob.glob(os.path.join(src, f"{tag}_{counter_dir}_{wl}", "*", "*counter_collection.csv")),
                       key=os.path.getmtime, reverse=True)
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for rr in csv.DictReader(open(files[0])):
            kk = short(rr["Kernel_Name"])
            if kk and rr["Kernel_Name"] == entry.get(kk, {}).get("kernel", rr["Kernel_Name"]):
                agg[kk][rr["Counter_Name"]].append(float(rr["Counter_Value"]))
        for kk, counters in agg.items():
            for cname, vals in counters.items():
                entry.setdefault(kk, {})[cname] = sum(vals) / len(vals)
    for kk, ee in entry.items():
        if "FETCH_SIZE" in ee and "WRITE_SIZE" in ee:
            ee["hbm_read_bytes_per_launch"] = 2.0 * ee["FETCH_SIZE"] * 1024.0      # gfx950: FETCH_SIZE = 1/2 of the bytes
            ee["hbm_write_bytes_per_launch"] = ee["WRITE_SIZE"] * 1024.0
            ee["hbm_bytes_per_launch"] = ee["hbm_read_bytes_per_launch"] + ee["hbm_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ee and "GRBM_GUI_ACTIVE" in ee:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            ee["mfma_util"] = ee["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (ee["GRBM_GUI_ACTIVE"] / 8.0)
    summary[wl] = entry
# kernel stats of the builder-run secondaries (trace only)
for wl in ("config3_split_f16", "variant_blocks2", "config5_sp_stream64_80ms"):
    stats = sorted(glob.glob(os.path.join(src, f"{tag}_trace_{wl}", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if stats:
        with open(os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"), "w") as fo:
            fo.write(open(stats[0]).read())
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fo:
    json.dump(summary, fo, indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
