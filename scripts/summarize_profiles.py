#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_round.sh (gpurun_out/<tag>_*) into the files committed under
profiles/:

  <tag>_kernel_stats_<workload>.csv   rocprofv3's own --stats table of the traced run (every launch of the process)
  <tag>_steady_<workload>.csv         per kernel: mean / median / p10 / p90 of the STEADY-STATE launches of the traced run
  <tag>_pmc_summary.json              per workload: the WaveNet kernels' steady-state times, executed-FLOP fraction of the
                                      fp32 MFMA peak from the trace, HBM traffic per launch and matrix-core utilisation

"Steady state" (VERDICT round 4, weak #3): the trace of a bench run also holds mbx_create's calibration forwards (2 x 40
frames: the same kernels on tiny grids) and the warm-up steps during which the clocks ramp.  Per kernel name the launches
are grouped by grid size, the group with the largest total time is the workload's own; of that group the launches of the
warm-up steps (the first warmup / (warmup + steps + stage-timing pass) of them) are dropped.  The PMC passes are filtered to
the same grid size.  HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, units KiB, FETCH_SIZE doubled on gfx950
(MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
summary = {}
FP32_MATRIX_PEAK_TFLOPS = 157.3
PMC_WORKLOADS = ("config1_sp_b1_3s", "config2_sp_b1_10s", "config3_si_b16_10s", "config5_sp_stream64", "config5_sp_stream64_80ms")
TRACE_ONLY = ("config3_split_f16", "variant_blocks2", "geometry_sweep")


def short(name):
    if "wn_gate_winograd" in name or ("conv1d_mfma_dma_kernel" in name and ", 1>" in name):
        return "gate"
    if "wn_gate_f16_kernel" in name or "wn_gate_f16w_kernel" in name:
        return "gate_f16"
    if "wn_resskip_f16_kernel" in name:
        return "res_skip_f16"
    if "wn_resskip_kernel" in name or "wn_resskip_wide_kernel" in name or "wn_resskip_wave_kernel" in name or ("conv1d_mfma_kernel" in name and ", 2, true" in name):
        return "res_skip"
    if "wn_gate0_kernel" in name:
        return "gate0"
    return None


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)
    return files[0] if files else None


def bench_line(wl, kind="trace"):
    """The JSON line the bench printed in the traced run (steps, warm-up, FLOPs per launch)."""
    path = os.path.join(src, f"{tag}_{kind}_{wl}.log")
    if not os.path.exists(path):
        return {}
    for ln in reversed(open(path).read().strip().splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                pass
    return {}


def steady_launches(trace_csv, line):
    """{kernel name: {"grid": g, "durations_us": [...], "dropped_warmup": n, "other_grids": n}} of a per-call kernel trace."""
    groups = collections.defaultdict(lambda: collections.defaultdict(list))
    for rr in csv.DictReader(open(trace_csv)):
        if rr.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
            continue
        grid = (int(rr["Grid_Size_X"]), int(rr["Grid_Size_Y"]), int(rr["Grid_Size_Z"]))
        groups[rr["Kernel_Name"]][grid].append((int(rr["Start_Timestamp"]), (int(rr["End_Timestamp"]) - int(rr["Start_Timestamp"])) / 1e3))
    warm, steps = int(line.get("warmup", 0)), int(line.get("steps", 0))
    # forwards of the run: warm-up + timed steps + the stage-timing pass of bench.py (max(3, min(steps, 10)) forwards)
    extra = max(3, min(steps, 10)) if "roofline" in line else 0
    total_fwd = warm + steps + extra
    out = {}
    for name, by_grid in groups.items():
        grid, calls = max(by_grid.items(), key=lambda kv: sum(dd for _, dd in kv[1]))
        calls.sort()
        drop = int(round(len(calls) * warm / total_fwd)) if total_fwd else 0
        kept = [dd for _, dd in calls[drop:]] or [dd for _, dd in calls]
        out[name] = {"grid": grid[0] * grid[1] * grid[2], "durations_us": kept, "dropped_warmup": drop,
                     "other_grids": sum(len(vv) for gg, vv in by_grid.items() if gg != grid)}
    return out


def pct(vals, q):
    vals = sorted(vals)
    return vals[min(len(vals) - 1, int(q * len(vals)))]


def write_steady(wl, steady):
    with open(os.path.join(dst, f"{tag}_steady_{wl}.csv"), "w") as fo:
        fo.write('"Name","Grid","Calls","MeanUs","MedianUs","P10Us","P90Us","TotalUs","DroppedWarmup","CallsOnOtherGrids"\n')
        for name, ee in sorted(steady.items(), key=lambda kv: -sum(kv[1]["durations_us"])):
            dd = ee["durations_us"]
            fo.write(f'"{name}",{ee["grid"]},{len(dd)},{statistics.fmean(dd):.3f},{statistics.median(dd):.3f},{pct(dd, 0.1):.3f},'
                     f'{pct(dd, 0.9):.3f},{sum(dd):.1f},{ee["dropped_warmup"]},{ee["other_grids"]}\n')


for wl in PMC_WORKLOADS + TRACE_ONLY:
    entry = {}
    line = bench_line(wl)
    stats = newest(os.path.join(src, f"{tag}_trace_{wl}", "*", "*kernel_stats.csv"))
    if stats:
        with open(os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"), "w") as fo:
            fo.write(open(stats).read())
    trace = newest(os.path.join(src, f"{tag}_trace_{wl}", "*", "*kernel_trace.csv"))
    if not trace:
        continue
    steady = steady_launches(trace, line)
    write_steady(wl, steady)
    main_grid = {}
    for name, ee in steady.items():
        kk = short(name)
        dd = ee["durations_us"]
        if kk and sum(dd) > entry.get(kk, {}).get("total_us", -1.0):
            entry[kk] = {"kernel": name, "grid": ee["grid"], "calls_steady": len(dd), "mean_us_trace": statistics.fmean(dd),
                         "median_us_trace": statistics.median(dd), "total_us": sum(dd), "dropped_warmup": ee["dropped_warmup"],
                         "calls_on_other_grids": ee["other_grids"]}
            main_grid[kk] = (name, ee["grid"])
    roof = line.get("roofline") or {}
    if "gate" in entry and roof.get("mfma_flop_executed_per_launch"):
        ee = entry["gate"]
        ee["flop_executed_per_launch"] = roof["mfma_flop_executed_per_launch"]
        ee["frac_from_trace_mean"] = roof["mfma_flop_executed_per_launch"] / (ee["mean_us_trace"] * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
        ee["frac_from_trace_median"] = roof["mfma_flop_executed_per_launch"] / (ee["median_us_trace"] * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
        ee["avg_launch_us_events_same_run"] = roof.get("avg_launch_ms", 0.0) * 1e3
        ee["frac_events_same_run"] = roof.get("frac")
    if "res_skip" in entry and (roof.get("res_skip") or {}).get("flop_per_launch"):
        ee = entry["res_skip"]
        ee["flop_per_launch"] = roof["res_skip"]["flop_per_launch"]
        ee["frac_from_trace_mean"] = ee["flop_per_launch"] / (ee["mean_us_trace"] * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS
        ee["avg_launch_us_events_same_run"] = roof["res_skip"].get("avg_launch_ms", 0.0) * 1e3
    if line:
        entry["_run"] = {"ms_per_step": line.get("ms_per_step"), "steps": line.get("steps"), "warmup": line.get("warmup")}
    for counter_dir in ("pmc_fetch", "pmc_write", "pmc_sq"):
        cfile = newest(os.path.join(src, f"{tag}_{counter_dir}_{wl}", "*", "*counter_collection.csv"))
        if not cfile:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for rr in csv.DictReader(open(cfile)):
            kk = short(rr["Kernel_Name"])
            if kk and main_grid.get(kk) == (rr["Kernel_Name"], int(rr["Grid_Size"])):
                agg[kk][rr["Counter_Name"]].append(float(rr["Counter_Value"]))
        for kk, counters in agg.items():
            for cname, vals in counters.items():
                entry[kk][cname] = sum(vals) / len(vals)
    for kk, ee in entry.items():
        if kk.startswith("_"):
            continue
        if "FETCH_SIZE" in ee and "WRITE_SIZE" in ee:
            ee["hbm_read_bytes_per_launch"] = 2.0 * ee["FETCH_SIZE"] * 1024.0      # gfx950: FETCH_SIZE = 1/2 of the bytes
            ee["hbm_write_bytes_per_launch"] = ee["WRITE_SIZE"] * 1024.0
            ee["hbm_bytes_per_launch"] = ee["hbm_read_bytes_per_launch"] + ee["hbm_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ee and "GRBM_GUI_ACTIVE" in ee:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            ee["mfma_util"] = ee["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (ee["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in ee:
            ee["flop_executed_from_mops"] = ee["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
        if "SQ_WAIT_INST_ANY" in ee and ee.get("SQ_WAVE_CYCLES"):
            ee["wait_inst_any_over_wave_cycles"] = ee["SQ_WAIT_INST_ANY"] / ee["SQ_WAVE_CYCLES"]
    summary[wl] = entry
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fo:
    json.dump(summary, fo, indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
