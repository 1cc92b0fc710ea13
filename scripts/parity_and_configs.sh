# Multi-word parity tests and the configuration table (scripts/mw_configs.py --no-cpu) in one GPU call
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py tests/test_reference_vectors.py -m gpu -x -q 2>&1 | tail -4 &&
timeout -k 10 500 python scripts/mw_configs.py --no-cpu 2>&1 | grep -v amdgpu.ids
