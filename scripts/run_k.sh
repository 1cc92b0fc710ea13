cd $GRAFT_REPO_ROOT
timeout -k 10 500 python scripts/mw_configs.py --no-cpu ce_8_15 ns_8_15_2 sdpa_x64 threepoint_3_8_8 polyopt40 2>&1 | grep -v amdgpu.ids
bash scripts/run_h.sh ce_8_15 | grep "k_mw_factor\|k_mw_potrf_q\|k_mw_potrf_x\|iteration time"
