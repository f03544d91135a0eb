"""The C-ABI library loads without a GPU and exports exactly what include/avs.h declares."""
import ctypes
import os
import re

import pytest

from adaptiveviscositysolver_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "avs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+\w+\s*\(\*\w+\)\s*\([^;]*\);", "", src)   # callback types (avs_allreduce_i32_fn) are not entry points
    return sorted(set(re.findall(r"\b(avs_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(capi.EXPORTED_SYMBOLS)


def probe_header_functions():
    src = open(os.path.join(ROOT, "include", "avs_probe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+\w+\s*\(\*\w+\)\s*\([^;]*\);", "", src)   # callback types (avs_allreduce_i32_fn) are not entry points
    return sorted(set(re.findall(r"\b(avs_[a-z0-9_]+)\s*\(", src)))


def test_probe_entries_live_in_the_probe_library_only(built_lib):
    """include/avs_probe.h (single SpMV launches, kernel sweeps, SELL / stream probes, the brick-form probe) is exported by libavs_probe.so --
    the -DAVS_PROBES build of the same sources -- and by nothing a plugin links (round-3 review: measurement code shipped in the product)."""
    assert probe_header_functions() == sorted(capi.PROBE_SYMBOLS)
    product, probe = ctypes.CDLL(capi.LIB_PATH), ctypes.CDLL(capi.PROBE_LIB_PATH)
    for name in capi.PROBE_SYMBOLS:
        assert not hasattr(product, name), f"{name} is exported by the product library"
        assert hasattr(probe, name), name
    for name in header_functions():     # the probe build is a superset
        assert hasattr(probe, name), name
    import subprocess
    syms = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    assert "probe" not in syms and "sell" not in syms.lower() and "stream_probe" not in syms


def test_library_exports_every_symbol(built_lib):
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), name
    assert b"gfx950" in built_lib.avs_version()


def test_struct_layouts_match_header(tmp_path):
    """sizeof of the ctypes mirrors == sizeof in C (gcc compiles include/avs.h as plain C)."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "avs.h"\nint main(void){printf("%zu %zu %zu\\n", '
                   'sizeof(avs_desc), sizeof(avs_solve_info), sizeof(avs_assembly_info));'
                   'printf("%zu %zu\\n", sizeof(avs_matrix_format), sizeof(avs_dist_info));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(capi.Desc), ctypes.sizeof(capi.SolveInfo), ctypes.sizeof(capi.AssemblyInfo),
                     ctypes.sizeof(capi.MatrixFormat), ctypes.sizeof(capi.DistInfo)]


def test_no_cpu_fallback_without_gpu(built_lib):
    """On a box without a GPU the product must fail loudly (AVS_EHIP), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from adaptiveviscositysolver_amd import ViscositySolve
    with pytest.raises(capi.AvsError) as e:
        ViscositySolve((32, 32, 32), 1 / 32, 0.01, 2)
    assert e.value.status == capi.EHIP
    import numpy as np
    from adaptiveviscositysolver_amd import pcg_csr
    with pytest.raises(capi.AvsError):
        pcg_csr(np.array([0, 1], np.int32), np.array([0], np.int32), np.array([1.0]), np.array([1.0]), np.array([0.0]))


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "adaptiveviscositysolver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("the oracle", "").replace("cpu oracle", "") or \
                    "import" not in "".join(l for l in text.splitlines() if "oracle" in l.lower()), f
    for line in open(os.path.join(ROOT, "include", "avs.h")):
        assert "oracle" not in line.lower()


def test_slab_margins_in_the_header_match_the_source():
    """include/avs.h promises the slab-local window (slab +- 12 cells of every level, stencils within 4): the constants of the source."""
    hdr = open(os.path.join(ROOT, "include", "avs.h")).read()
    src = open(os.path.join(ROOT, "adaptiveviscositysolver_amd", "csrc", "avs_internal.hpp")).read()
    index_margin = int(re.search(r"kSlabIndexMargin\s*=\s*(\d+)", src).group(1))
    stencil_margin = int(re.search(r"kSlabStencilMargin\s*=\s*(\d+)", src).group(1))
    assert f"slab +- {index_margin} level-l cells" in hdr
    assert f"within {stencil_margin} cells (of their level)" in hdr
    assert index_margin >= 2 * (stencil_margin + 1) + 1     # a stencil of the next coarser level reads 2 (margin + 1) + 1 cells of this one
