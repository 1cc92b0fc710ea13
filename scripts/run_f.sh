cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py -m gpu -x -q -k "exact or assembl or sharded" 2>&1 | tail -4 &&
timeout -k 10 300 python scripts/mw_roofline.py 5 1024 3 on 2>&1 | tail -2 &&
timeout -k 10 300 python scripts/mw_roofline.py 5 128 5 on 2>&1 | tail -2 &&
timeout -k 10 300 python scripts/mw_roofline.py 5 1024 3 off 2>&1 | tail -1 &&
timeout -k 10 300 python scripts/mw_roofline.py 6 1024 3 on 2>&1 | tail -2
