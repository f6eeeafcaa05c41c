cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
for sm in 1 0; do
echo "== MBX_NO_SMALL=$sm"
MBX_NO_SMALL=$sm python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'])"
MBX_NO_SMALL=$sm python bench.py --workload config1_sp_b1_3s --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'])"
done
