#!/bin/bash
# on the GPU box: res/skip launch time of the split half precision kernel for each ablation library built by
# `EXP_FILE=wn_resskip_f16.hip python scripts/experiments/mkexp.py rh_base: rh_nomfma:rh_nomfma ...` (timing only)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo -n "$lib  "
  MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_$lib.so python scripts/experiments/split_probe.py 2>/dev/null | grep "16 800 split" | cut -c1-120
done
