#!/bin/bash
# One GPU lease: the -m gpu suite on the in-tree library, then an interleaved A/B of library variants (tools/build_variant.sh) and phase traces.
#   LIBS="base late ..." TRACE="base late" [TESTS=0] [ROUNDS=5] tools/gpu_ab.sh <tag>
tag=${1:-ab}
mkdir -p gpurun_out
if [ "${TESTS:-1}" != "0" ]; then python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} 2>&1 | tail -5 > gpurun_out/${tag}_tests.txt; cat gpurun_out/${tag}_tests.txt; fi
libs=""; for v in $LIBS; do libs="$libs dgnn_amd/variants/$v.so"; done
[ -n "$libs" ] && python tools/variants.py --rounds ${ROUNDS:-5} ${VARIANT_ARGS} $libs 2>&1 | tee gpurun_out/${tag}_variants.txt
for v in $TRACE; do echo "== trace $v"; python tools/trace_fused.py 150000 dgnn_amd/variants/$v.so 2>&1 | tail -6; done | tee gpurun_out/${tag}_trace.txt
