#!/bin/bash
# PMC pass of the geometry sweep (round 6): matrix-core utilisation of the F(4,3) kernels at dilations above 16
#   gpurun -- 'bash scripts/profile_sweep_pmc.sh r06'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_geometry_sweep -- \
    python3 $R/bench.py --workload geometry_sweep --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_pmc_sq_geometry_sweep.log 2>&1
python3 - <<PY
import csv, glob, collections, json
files = glob.glob("$R/gpurun_out/${TAG}_pmc_sq_geometry_sweep/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for rr in csv.DictReader(open(files[0])):
    nm = rr["Kernel_Name"]
    if "wn_gate_winograd4" in nm or "conv1d_mfma_dma_kernel" in nm or "wn_resskip_wide" in nm:
        agg[(nm.split("(")[0], int(rr["Grid_Size"]))][rr["Counter_Name"]].append(float(rr["Counter_Value"]))
out = {}
for (nm, grid), cc in sorted(agg.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cc and "GRBM_GUI_ACTIVE" in cc and len(cc["GRBM_GUI_ACTIVE"]) >= 4:
        busy = sum(cc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cc["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sum(cc["GRBM_GUI_ACTIVE"]) / len(cc["GRBM_GUI_ACTIVE"])
        out[f"{nm} grid {grid}"] = {"launches": len(cc["GRBM_GUI_ACTIVE"]), "mfma_util": round(busy / 1024.0 / (act / 8.0), 4),
                                    "flop_executed_from_mops": sum(cc["SQ_INSTS_VALU_MFMA_MOPS_F32"]) / len(cc["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * 512.0}
json.dump(out, open("$R/gpurun_out/${TAG}_sweep_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
