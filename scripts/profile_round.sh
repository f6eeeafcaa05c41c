#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash scripts/profile_round.sh r01'
# kernel-trace/stats and the PMC passes are separate runs (gpurun refuses --pmc together with trace domains other
# than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries land in gpurun_out/<tag>_*;
# scripts/summarize_profiles.py turns them into the small files committed under profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for WL in config2_sp_b1_10s config3_si_b16_10s config5_sp_stream64; do
  STEPS=20; [ $WL = config3_si_b16_10s ] && STEPS=5; [ $WL = config5_sp_stream64 ] && STEPS=30
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$WL -- \
      python3 $R/bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_trace_$WL.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_fetch_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_write_$WL.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_sq_$WL.log 2>&1
  tail -1 $R/gpurun_out/${TAG}_trace_$WL.log | cut -c1-200
done
# builder-run secondaries (kernel trace only): the opt-in split half precision of the res/skip layers and the two-block variant
for WL in config3_split_f16 variant_blocks2 config5_sp_stream64_80ms; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$WL -- \
      python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_trace_$WL.log 2>&1
  tail -1 $R/gpurun_out/${TAG}_trace_$WL.log | cut -c1-200
done
