# rocprofv3 passes for a bench configuration (run on the GPU box through gpurun; outputs under gpurun_out/<tag>_*)
#   bash tools/prof_round2.sh <tag> [bench.py args, e.g. --dtype bf16]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_trace -- $B > gpurun_out/${T}_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/${T}_pmc1 -- $B > gpurun_out/${T}_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/${T}_pmc2 -- $B > gpurun_out/${T}_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${T}_pmc3 -- $B > gpurun_out/${T}_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${T}_pmc4 -- $B > gpurun_out/${T}_pmc4.log 2>&1
git rev-parse HEAD > gpurun_out/${T}_commit.txt 2>/dev/null || true
