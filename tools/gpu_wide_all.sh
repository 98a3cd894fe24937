# wide layers: tests, bench lines, traces, PMC passes of both width sets:  bash tools/gpu_wide_all.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r5w}
bash tools/gpu_wide.sh $T
bash tools/gpu_wide_pmc.sh ${T}_pmc_64_128_256_512 64,128,256,512 > /dev/null
bash tools/gpu_wide_pmc.sh ${T}_pmc_128_256_512_1024 128,256,512,1024 > /dev/null
