cd $GRAFT_REPO_ROOT
python tools/variants.py --rounds 4 dgnn_amd/variants/base.so dgnn_amd/variants/young.so dgnn_amd/variants/maxilp.so dgnn_amd/variants/maxmem.so dgnn_amd/variants/nw8.so > gpurun_out/r2g_variants.log 2>&1
tail -14 gpurun_out/r2g_variants.log
