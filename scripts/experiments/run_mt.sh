# usage: bash scripts/experiments/run_mt.sh variant ...   (libraries of scripts/experiments/mkexp.py; timing only)
for v in "$@"; do echo "== $v"; MBX_EXPERIMENT=1 MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_$v.so python scripts/experiments/mel_conv_probe.py 2>&1 | grep "PS_1\|wn.cond\|PS_final"; done
