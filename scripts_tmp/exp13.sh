cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for no in 0 1; do
if [ $no = 1 ]; then export MBX_NO_OVERLAP=1; fi
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'])"
python bench.py --workload config1_sp_b1_3s --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'])"
python bench.py --workload config3_si_b16_10s --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'])"
done
