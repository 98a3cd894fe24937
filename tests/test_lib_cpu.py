"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/dgnn_hip.h declares, the ctypes table mirrors the header, and the product refuses to run
without a GPU (no fallback).  No compute calls here."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dgnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dgnn_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "dgnn_plan_build" in syms and "dgnn_sage_layer_fused_fwd" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    import ctypes
    from dgnn_amd._lib import LIB_PATH, SIGNATURES, lib
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = lib()
    raw = ctypes.CDLL(LIB_PATH)
    for s in declared_symbols():
        assert hasattr(raw, s), "libdgnn_hip.so does not export %s" % s
        assert s in SIGNATURES, "ctypes table lacks %s" % s
    assert set(SIGNATURES) == set(declared_symbols())
    assert l.dgnn_version() == 100
    # pure host query entry points may be called without a GPU
    assert l.dgnn_plan_scratch_elems(16, 4) >= 20
    assert l.dgnn_linear_wgrad_scratch_elems(1000, 64, 28) >= 64 * 28


def test_no_cpu_fallback():
    from dgnn_amd import ops
    from dgnn_amd._lib import DgnnError
    with pytest.raises(DgnnError):
        ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4))
    from dgnn_amd.config import Config, reconbench_pretrained
    from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
    net = SurfaceNet(reconbench_pretrained(device="cpu"))
    with pytest.raises(RuntimeError):
        net.inference_layer(Config(x=torch.zeros(4, 29), edge_attr=torch.zeros(16, 20), edge_index=torch.zeros(2, 16, dtype=torch.long)))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dgnn_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M), f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/dgnn_hip.h is a C header (no C++/torch types) and the library links from a plain C program."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "dgnn_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %s\\n", dgnn_version(), dgnn_last_error_string());\n'
                   '  return dgnn_plan_scratch_elems(16, 4) >= 20 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    libdir = os.path.join(root, "dgnn_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-ldgnn_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) >= 1 and out[1] == "ok"
