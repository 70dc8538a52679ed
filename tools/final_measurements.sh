cd /root/repo
R=${ROUND:-r6}
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${R}f_tests.txt
timeout -k 10 1200 bash tools/bench_and_kernel_stats.sh > gpurun_out/${R}f_bench_stats.log 2>&1
timeout -k 10 700 bash tools/pmc_traffic.sh > gpurun_out/${R}f_traffic.log 2>&1
timeout -k 10 700 bash tools/pmc_layers.sh > gpurun_out/${R}_pmc_per_layer.txt 2>&1
timeout -k 10 500 bash tools/pmc_sq.sh ${R}sq1 "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" > /dev/null 2>&1
timeout -k 10 500 bash tools/pmc_sq.sh ${R}sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" > /dev/null 2>&1
timeout -k 10 500 bash tools/pmc_sq.sh ${R}sq3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > /dev/null 2>&1
cat gpurun_out/${R}f_tests.txt; tail -c 600 gpurun_out/bench_default.json
