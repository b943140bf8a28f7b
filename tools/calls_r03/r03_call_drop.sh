set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bnact.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_gpu_optim.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
for c in cfg3 cfg4; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
