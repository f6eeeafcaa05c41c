cd $GRAFT_REPO_ROOT
python bench.py --workload config4_vo_256utt --steps 2 --warmup 1 2>&1 | tail -1
python bench.py --steps 20 --warmup 3 2>&1 | tail -1
