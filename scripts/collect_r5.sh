mkdir -p gpurun_out/r5f
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
timeout -k 5 100 scripts/micro/micro_hop > $O/m_micro_hop.txt 2>&1
timeout -k 5 100 scripts/micro/micro_clock > $O/m_micro_clock.txt 2>&1
timeout -k 5 100 scripts/micro/micro_step > $O/m_micro_step.txt 2>&1
bash scripts/iter_timeline.sh ce_8_15 > /dev/null 2>&1; cp gpurun_out/timeline_h_ce_8_15.txt $O/g_iteration_timeline_final.txt
bash scripts/iter_timeline.sh ns_8_15_3 > /dev/null 2>&1; cp gpurun_out/timeline_h_ns_8_15_3.txt $O/g_iteration_timeline_nsphere_N3.txt
timeout 600 python scripts/mw_configs.py > $O/g_configs_at_256_bits.txt 2>&1
timeout 300 python scripts/fp64_assembly_shapes.py ce_8_15 8192 polyopt40 2048 polyopt40 8192 ns_8_15_2 512 > $O/h_fp64_assembly_shapes.txt 2>&1
echo done
