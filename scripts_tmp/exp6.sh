cd $GRAFT_REPO_ROOT
MBX_GATE_CFG=10 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
MBX_GATE_CFG=11 python -m pytest tests/test_gpu_parity.py -q -x -k "forward" 2>&1 | tail -2
MBX_GATE_CFG=12 python -m pytest tests/test_gpu_parity.py -q -x -k "forward" 2>&1 | tail -2
for cfg in 1 10 11 12; do
echo "== MBX_GATE_CFG=$cfg"
MBX_GATE_CFG=$cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'], d['roofline']['avg_launch_ms'], d['roofline']['res_skip_avg_launch_ms'], d['roofline']['frac'])"
MBX_GATE_CFG=$cfg python bench.py --workload config3_si_b16_10s --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'], d['roofline']['avg_launch_ms'], d['roofline']['res_skip_avg_launch_ms'], d['roofline']['frac'])"
done
