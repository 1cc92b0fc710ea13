cd $GRAFT_REPO_ROOT
timeout -k 10 800 python scripts/mw_configs.py --no-cpu ns_8_15_2 threepoint_4 threepoint_3_8_8 sdpa_x64 polyopt40 ce_8_15 2>&1 | grep -v amdgpu.ids
bash scripts/run_h.sh threepoint_3_8_8 > /dev/null; bash scripts/run_h.sh ns_8_15_2 > /dev/null; bash scripts/run_h.sh sdpa_x64 > /dev/null; echo done
