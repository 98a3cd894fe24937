"""Build-level guard (no GPU needed): the fused layer kernels must not fall into scratch memory.

A harmless-looking edit (a data-dependent branch in the tile map) once made the compiler stop unrolling a row-block loop of the first
layer's kernel: its row registers went to 640 bytes of scratch per lane and the layer ran 8x slower while every parity test stayed green
(DESIGN.md 5a).  The device assembly tells: `.private_segment_fixed_size` and `.vgpr_spill_count` of every instantiation that the default
path launches must be zero."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_default_fused_kernels_use_no_scratch(tmp_path):
    src = os.path.join(ROOT, "dgnn_amd", "csrc", "fused_mfma.hip")
    out = str(tmp_path / "fused_mfma.s")
    # the flags of dgnn_amd/csrc/Makefile for this file
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                    src, "-o", out], check=True, capture_output=True)
    text = open(out).read()
    kernels = re.findall(r"\.name:\s+(\S*k_sage_fused_mfma\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                         r"(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) >= 15, len(kernels)
    seen = 0
    for name, scratch, vgprs, spills in kernels:
        # template arguments <CIN_PAD, COUT, NW, KS, DSP, FSP>: DSP == FSP == 2 is gemm mode f16x2 (default), 3/3 the bf16 x 3 form
        m = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        assert m, name
        cin, cout, nw, ks, dsp, fsp = map(int, m.groups())
        if (dsp, fsp) in ((2, 2), (3, 3)):
            seen += 1
            assert int(scratch) == 0 and int(spills) == 0, (name, scratch, spills)
            assert int(vgprs) <= 256, (name, vgprs)
    assert seen >= 11, seen
    assert any("ELb1E" in k[0] for k in kernels), "the decoder-carrying instantiation (DEC) of the 128 -> 128 layer is missing"
