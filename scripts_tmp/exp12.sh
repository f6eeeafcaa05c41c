cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'], d['roofline']['avg_launch_ms'], d['roofline']['res_skip_avg_launch_ms'], d['roofline']['frac'])"
python bench.py --workload config3_si_b16_10s --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'], d['roofline']['avg_launch_ms'], d['roofline']['res_skip_avg_launch_ms'], d['roofline']['frac'])"
