cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03zg
timeout 300 python tools/step_profile.py --config cfg3 --rows 90 --torch-only > gpurun_out/r03zg/step_profile_cfg3.txt 2>gpurun_out/r03zg/err.txt
tail -3 gpurun_out/r03zg/err.txt
