cd $GRAFT_REPO_ROOT
python scripts/mw_iter_profile.py ce_8_15 3 2>&1 | tail -2
timeout 900 python -m pytest tests/test_mw_parity.py tests/test_reference_vectors.py -m gpu -x -q --durations=6 2>&1 | tail -14
