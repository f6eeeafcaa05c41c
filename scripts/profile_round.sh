#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash scripts/profile_round.sh r05'
# kernel-trace/stats and the PMC passes are separate runs (gpurun refuses --pmc together with trace domains other
# than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).  The program sits directly behind `--` (no env /
# bash -c hop).  Raw output lands in gpurun_out/<tag>_*; scripts/summarize_profiles.py turns it into the small files
# committed under profiles/ (steady-state launches only: calibration launches and warm-up steps dropped).
# Traced runs: >= 10 warm-up + >= 20 timed steps for config 3, >= 50 timed steps for the single utterances, so that the
# clock ramp of the first launches does not sit in the averages.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for WL in config3_si_b16_10s config2_sp_b1_10s config1_sp_b1_3s config5_sp_stream64 config5_sp_stream64_80ms; do
  STEPS=50; WARM=10
  [ $WL = config3_si_b16_10s ] && STEPS=20
  [ $WL = config5_sp_stream64 ] && STEPS=40
  [ $WL = config5_sp_stream64_80ms ] && STEPS=40
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$WL -- \
      python3 $R/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_trace_$WL.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_fetch_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_write_$WL.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_$WL -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_pmc_sq_$WL.log 2>&1
  echo "$WL: $(tail -1 $R/gpurun_out/${TAG}_trace_$WL.log | cut -c1-200)"
done
# builder-run secondaries (kernel trace only): the opt-in split half precision, the two-block variant, the geometry sweep
for WL in config3_split_f16 variant_blocks2 geometry_sweep; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace_$WL -- \
      python3 $R/bench.py --workload $WL --steps 10 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_trace_$WL.log 2>&1
  echo "$WL: $(tail -1 $R/gpurun_out/${TAG}_trace_$WL.log | cut -c1-200)"
done
# the raw per-call traces are large: keep what summarize_profiles.py reads (it runs here too, so that the summaries exist
# even if the merge back of gpurun_out/ is cut short)
python3 $R/scripts/summarize_profiles.py $TAG > $R/gpurun_out/${TAG}_summary.json 2> $R/gpurun_out/${TAG}_summary.err
echo "summary written: $(wc -c < $R/gpurun_out/${TAG}_summary.json) bytes"
