cd $GRAFT_REPO_ROOT
for m in off on; do for c in 128 1024; do python scripts/mw_roofline.py 5 $c 5 $m 2>&1 | grep "exact products"; done; done
timeout 900 python -m pytest tests/test_mw_parity.py tests/test_reference_vectors.py -m gpu -x -q 2>&1 | tail -3
python scripts/mw_iter_profile.py ce_8_15 3 2>&1 | tail -2
