cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py -m gpu -x -q -k "dense_blocks or large_block" 2>&1 | tail -8 &&
timeout -k 10 500 python scripts/mw_configs.py --no-cpu sdpa_x64 sdpa_example 2>&1 | grep -v amdgpu.ids
