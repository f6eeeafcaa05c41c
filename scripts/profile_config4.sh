#!/bin/bash
# kernel stats of the config-4 workload (C = 340, 256 ragged utterances on one GPU): gpurun -- bash scripts/profile_config4.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_trace_config4_vo_256utt -- python3 $R/bench.py --workload config4_vo_256utt --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02_trace_config4.log 2>&1
tail -c 200 $R/gpurun_out/r02_trace_config4.log
