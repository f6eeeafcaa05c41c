#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/bench.py --workload config3_si_b16_10s --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True)
print(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"][:50]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[n] += 1
for n, d in acc.items():
    c = max(cnt[n], 1)
    print(n, {k: round(v / c) for k, v in d.items()})
PY
