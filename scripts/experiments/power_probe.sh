#!/bin/bash
# GPU box: sample power / clocks / temperature (rocm-smi, read-only queries) while config 3 runs in a loop -- is the 2.24-2.33 GHz
# measured under the gate kernel a power cap, a thermal limit or something else?
cd $GRAFT_REPO_ROOT
(python scripts/experiments/stage_probe.py 16 800 600 > gpurun_out/power_probe_run.txt 2>&1) &
PID=$!
sleep 25
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -i "power\|sclk\|mclk\|temp\|perf\|fclk" | tr '\n' ';'
  echo
  sleep 1
done
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3
wait $PID
tail -1 gpurun_out/power_probe_run.txt
