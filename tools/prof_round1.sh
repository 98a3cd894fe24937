cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|SQ_|TCC_|GRBM|FETCH|WRITE_SIZE|TCP_" | head -150 > gpurun_out/counters.txt
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1_trace -- $B > gpurun_out/r1_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/r1_pmc1 -- $B > gpurun_out/r1_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/r1_pmc2 -- $B > gpurun_out/r1_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r1_pmc3 -- $B > gpurun_out/r1_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r1_pmc4 -- $B > gpurun_out/r1_pmc4.log 2>&1
ls -R gpurun_out | head -50
