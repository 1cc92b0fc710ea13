cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mw_parity.py tests/test_reference_vectors.py -m gpu -x -q 2>&1 | tail -4 &&
timeout -k 10 300 python scripts/mw_roofline.py 5 1024 5 on 2>&1 | tail -2 | head -1 &&
timeout -k 10 300 python scripts/mw_roofline.py 5 128 5 on 2>&1 | tail -2 | head -1 &&
timeout -k 10 300 python scripts/mw_configs.py --no-cpu ce_8_15 polyopt40 delsarte_3_10 2>&1 | grep -v amdgpu.ids
