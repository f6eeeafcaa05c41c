R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_a -- python3 $R/bench.py --workload config3_si_b16_10s --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_b -- python3 $R/bench.py --workload config3_si_b16_10s --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_b.log 2>&1
tail -2 $R/gpurun_out/pmc_a.log; find $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b -name "*.csv" | head
