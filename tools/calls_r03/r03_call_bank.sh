cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_pwconv.py tests/test_gpu_optim.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
for c in cfg3 cfg4; do timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150; done
