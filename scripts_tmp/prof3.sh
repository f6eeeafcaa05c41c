R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stream -- python3 $R/bench.py --workload config5_sp_stream64 --steps 10 --warmup 2 > $R/gpurun_out/prof_stream.log 2>&1
tail -1 $R/gpurun_out/prof_stream.log | cut -c1-200
