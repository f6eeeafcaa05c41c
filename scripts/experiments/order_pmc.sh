#!/bin/bash
# on the GPU box: launch time (runvariants.sh) and HBM-side reads (rocprofv3 --pmc FETCH_SIZE, own pass) of the F(4,3)
# gate kernel under different block orders (mkexp.py order_* variants); profiles/README.md quotes the pairs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/experiments/runvariants.sh "$@"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export MBX_LIB_PATH=$R/scripts/experiments/libs/lib_$v.so
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_order_fetch_$v -- \
      python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/r03_order_fetch_$v.log 2>&1 || echo "pmc $v failed"
done
