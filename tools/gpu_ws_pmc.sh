# PMC passes of the default bench line (counters only): bash tools/gpu_ws_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; mkdir -p gpurun_out/$T
B="python3 bench.py --no-train --no-extras --no-cpu-baseline --no-breakdown --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- $B > gpurun_out/$T/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/$T/pmc1 -- $B > gpurun_out/$T/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d gpurun_out/$T/pmc2 -- $B > gpurun_out/$T/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/pmc3 -- $B > gpurun_out/$T/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$T/pmc4 -- $B > gpurun_out/$T/pmc4.log 2>&1
