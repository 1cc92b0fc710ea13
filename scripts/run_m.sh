cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py -m gpu -x -q -k "exact" 2>&1 | tail -4 &&
timeout -k 10 500 python scripts/mw_configs.py --no-cpu threepoint_3_8_8 threepoint_4 polyopt40 ce_8_15 2>&1 | grep -v amdgpu.ids
