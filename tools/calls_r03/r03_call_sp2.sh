cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03zh
timeout 300 python tools/step_profile.py --config cfg2 --rows 60 --torch-only > gpurun_out/r03zh/step_profile_cfg2_torch.txt 2>/dev/null
