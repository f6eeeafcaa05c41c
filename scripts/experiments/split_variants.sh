#!/bin/bash
# on the GPU box: gate / res-skip launch times of the split half precision kernels for each ablation library built by
# `EXP_FILE=wn_resskip_f16.hip|wn_gate_f16.hip python scripts/experiments/mkexp.py <name>:<patches> ...` (timing only)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo -n "$lib  "
  MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_$lib.so python scripts/experiments/split_probe.py 2>/dev/null | grep "16 800 split" | cut -c1-140
done
