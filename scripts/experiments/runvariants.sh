#!/bin/bash
# on the GPU box (through gpurun, from the repo root): for each variant built by mkexp.py run the bench (config 3 unless
# BENCH_ARGS says otherwise) with the engine pointed at it through MBX_LIB_PATH (engine.load_library) and print the
# per-kernel times.  The product library is never overwritten: a killed run cannot leave a wrong-output build in the tree.
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_$lib.so timeout -k 5 120 python bench.py --no-cpu-baseline --no-secondary $BENCH_ARGS > /tmp/b.json 2>/tmp/b.err || { echo "$lib FAILED"; tail -3 /tmp/b.err; }
  python - "$lib" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f"{sys.argv[1]:24s} step {d['ms_per_step']:.3f} ms gate {r['avg_launch_ms']*1e3:.1f} us frac {r['frac']:.3f} resskip {r['res_skip']['avg_launch_ms']*1e3:.1f} us " + ' '.join(f"{st['stage']} {st['avg_launch_ms']*1e3:.0f}" for st in r['stages']) + f" fe {r['frontend_ms_per_step']*1e3:.0f}")
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
