"""The brick-structured form of the solve matrix (csrc/avs_brick.hip, built on the device by csrc/avs_brick_build.hip): rows of one 8^3 brick
stored as geometric row patterns, x of the brick + halo in LDS.  Lossless: y must equal the plain CSR kernel's bit for bit (avs_bench_spmv
checks that on the device and fails otherwise), a solve through it must agree with the solve through the 4-B stream form, and the torch
reference builder (tools/brick_build.py) must give the same product through the same kernel."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from util import ROOT, build_pyramid, feed, rel_l2

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def _solver(sc, monkeypatch, brick, probe=True):
    monkeypatch.setenv("AVS_BRICK", "1" if brick else "0")
    monkeypatch.setenv("AVS_CG_RESIDENT", "0")   # the brick form serves the launch-per-phase loop
    pyr = build_pyramid(sc)
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=probe)   # probe build: avs_bench_spmv / avs_brick_spmv_probe
    feed(s, pyr)
    s.set_scene_fields(dsc)
    return s


SCENES = {
    "beam128_L4": lambda: scenes.fat_beam(128, 4),
    "beam64_L3_wall": lambda: scenes.fat_beam(64, 3, wall=True),
    "beam96x64x64_L3": lambda: scenes.fat_beam(64, 3, res=(128, 64, 64)),
    "sheet128_L4": lambda: scenes.thin_sheet(128, 4, thickness_cells=12),
    "sphere64_L3": lambda: scenes.sphere(64, 3),
    "beam64_L2": lambda: scenes.fat_beam(64, 2),
    "beam256_L5": lambda: scenes.fat_beam(256, 5),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_brick_form_is_lossless(name, monkeypatch, built_lib):
    s = _solver(SCENES[name](), monkeypatch, True)
    ai = s.assemble()
    fmt = s.matrix_format()
    if name.startswith("beam") or name.startswith("sheet"):
        assert fmt.brick_tiles > 0, "the form was not built for a flat-faced scene"
        assert fmt.brick_pattern_rows >= 0.6 * ai.n_velocity
        assert fmt.brick_bytes < 4 * ai.nnz
    # plain and fused-dot launches: avs_bench_spmv compares y with the plain CSR kernel bit for bit and fails on any difference
    s.bench_spmv(0, 3)
    s.bench_spmv(100, 3)
    s.close()


@pytest.mark.parametrize("env", [{"AVS_VALUE_PACK": "0"},                                   # -> windowed columns, the 1024^3 sheet's form
                                 {"AVS_VALUE_PACK": "0", "AVS_COLUMN_WINDOWS": "0"}])      # -> 6-B form (2-B code + int32 column)
@pytest.mark.parametrize("name", ["beam128_L4", "sheet128_L4"])
def test_brick_form_with_wide_streamed_words(name, env, monkeypatch, built_lib):
    """matrices whose code and column do not share 32 bits (the 1024^3 thin sheet: 25 + 8 bits) keep their streamed rows as 64-bit words
    (column | code << 32); forced here on small systems by switching the packed form off"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = _solver(SCENES[name](), monkeypatch, True)
    ai = s.assemble()
    fmt = s.matrix_format()
    assert fmt.column_bits == 0, "the packed form was not switched off"
    assert fmt.brick_tiles > 0 and fmt.brick_pattern_rows >= 0.6 * ai.n_velocity
    s.bench_spmv(0, 3)
    s.bench_spmv(100, 3)
    info = s.solve(tol=1e-9, max_iters=4000)
    assert info.converged
    s.close()


@pytest.mark.parametrize("name", ["beam128_L4", "beam64_L3_wall", "sheet128_L4"])
def test_solve_through_the_brick_form(name, monkeypatch, built_lib):
    out = {}
    for brick in (False, True):
        s = _solver(SCENES[name](), monkeypatch, brick, probe=False)   # the PRODUCT library
        s.assemble()
        assert (s.matrix_format().brick_tiles > 0) == brick
        info = s.solve(tol=1e-9, max_iters=4000)
        assert info.converged and not info.resident
        out[brick] = (info.iterations, np.asarray(s.solution()))
        s.close()
    # same products and row sums; the partial sums of p.Ap are added in another order
    assert abs(out[True][0] - out[False][0]) <= 3, (out[True][0], out[False][0])
    assert rel_l2(out[True][1], out[False][1]) < 1e-8


def test_reference_builder_and_device_builder_agree(monkeypatch, built_lib):
    """tools/brick_build.py (torch) builds the same form from the CSR + dof table; its arrays through the same kernel give plain CSR's y"""
    import brick_build as bb
    sc = scenes.fat_beam(128, 4)
    s = _solver(sc, monkeypatch, True)
    ai = s.assemble()
    n, nnz = int(ai.n_velocity), int(ai.nnz)
    dev = torch.device("cuda:0")
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE))
    tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
    capi.check(s.lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE))
    fmt = s.matrix_format()
    L = s.lib
    perm, geo = bb.brick_major(tab, sc.res)
    rp2, col2, val2 = bb.permute_csr(rp.long(), col.long(), val, perm)
    table, code = torch.unique(val2, return_inverse=True)
    col_bits = max(1, (n - 1).bit_length())
    form = bb.build(rp2, col2, code, geo, len(table), col_bits)
    st = form["stats"]
    # the two builders agree on what is regular (pattern ids and orders may differ)
    assert abs(st["regular_rows"] - fmt.brick_pattern_rows) <= 0.01 * n, (st["regular_rows"], fmt.brick_pattern_rows)

    class Arr(C.Structure):
        _fields_ = [("ntiles", C.c_int32)] + [(k, C.c_void_p) for k in ("tile_blk", "blocks", "rdesc", "ownslot", "pwords", "sdesc", "swords", "table")] + \
                   [("table_size", C.c_int32), ("col_bits", C.c_int32)]
    arr = Arr(form["ntiles"], *[form[k].data_ptr() for k in ("tile_blk", "blocks", "rdesc", "ownslot", "pwords", "sdesc", "swords")],
              table.data_ptr(), len(table), col_bits)
    L.avs_brick_spmv_probe.argtypes = [C.POINTER(Arr), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    ms = C.c_double()
    capi.check(L.avs_brick_spmv_probe(C.byref(arr), x.data_ptr(), y.data_ptr(), None, 2, None, C.byref(ms)))
    yref = torch.empty_like(y)
    capi.check(L.avs_spmv_csr(n, rp2.to(torch.int32).data_ptr(), col2.to(torch.int32).data_ptr(), val2.data_ptr(), x.data_ptr(), yref.data_ptr(), 14, 1, None))
    torch.cuda.synchronize()
    assert int((y.view(torch.int64) != yref.view(torch.int64)).sum()) == 0
    s.close()


def test_auto_mode_keeps_the_faster_form(monkeypatch, built_lib):
    """AVS_BRICK_AUTO (the default) measures both forms at the first assembly of a matrix size and keeps the faster -- the brick form on
    the fat 512^3 beam (~490 rows per shell brick: 0.7x the word stream's time; on a thin sheet it is 0.87x since the tiles are dealt to
    the XCDs in interleaved chunks, so whichever the measurement picks there is accepted) -- and the verdict is kept for later assemblies"""
    from adaptiveviscositysolver_amd import DevicePrepass
    monkeypatch.delenv("AVS_BRICK", raising=False)
    dev = torch.device("cuda:0")
    for make, brick in ((lambda: scenes.fat_beam(512, 4, device=dev), True), (lambda: scenes.thin_sheet(512, 4, thickness_cells=32, device=dev), None)):
        sc = make()
        pp = DevicePrepass(sc.res, sc.dx, sc.levels)
        pi = pp.run(sc.liquid, sc.solid)
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0, probe=True)
        pp.apply(s); s.set_scene_fields(sc); pp.close()
        del sc
        torch.cuda.empty_cache()
        first = None
        for _ in range(2):
            ai = s.assemble()
            fmt = s.matrix_format()
            assert ai.n_velocity >= 2_000_000
            if brick is not None:
                assert (fmt.brick_tiles > 0) == brick, (fmt.brick_tiles, brick)
            if first is None:
                first = fmt.brick_tiles > 0
            assert (fmt.brick_tiles > 0) == first   # the verdict of the first assembly holds
        s.bench_spmv(100, 2)   # bit-identical to plain CSR either way
        for mode, want in ((capi.BRICK_ALWAYS, True), (capi.BRICK_NEVER, False)):
            s.set_solver_option(capi.OPTION_BRICK_FORM, mode)
            s.assemble()
            assert (s.matrix_format().brick_tiles > 0) == want
            s.bench_spmv(100, 2)
        s.close()
        torch.cuda.empty_cache()


def test_brick_form_is_reproducible(monkeypatch, built_lib):
    """the builder assigns global pattern numbers with atomics; everything the arithmetic order depends on (a tile's local pattern numbers,
    the execution order of its rows, hence which lane adds a row's share of p.Ap) is canonical: two builds of the same matrix give the same
    iteration count and the same solution bit for bit"""
    got = []
    for _ in range(3):
        s = _solver(SCENES["beam128_L4"](), monkeypatch, True, probe=False)
        s.assemble()
        assert s.matrix_format().brick_tiles > 0
        info = s.solve(tol=1e-9, max_iters=4000)
        got.append((info.iterations, np.asarray(s.solution()).tobytes()))
        s.close()
    assert got[0][0] == got[1][0] == got[2][0]
    assert got[0][1] == got[1][1] == got[2][1]
