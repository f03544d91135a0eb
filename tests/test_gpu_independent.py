"""The independent checks of tests/independent.py against the HIP path (stencil lists, CSR, rhs, x0 all read back
through the C ABI): scatter-form assembly and the linear-shear known answer.  Inputs come from the HIP pre-pass."""
import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from independent import check_linear_shear, check_scatter_form, sampled_field_runs
from util import build_pyramid, feed

pytestmark = pytest.mark.gpu

CASES = {
    "beam32_L3": lambda: scenes.fat_beam(32, 3),
    "beam64_L3_wall_varvisc": lambda: scenes.fat_beam(64, 3, wall=True, variable_viscosity=True),
    "sphere64_L4": lambda: scenes.sphere(64, 4),
    "sheet64_L3": lambda: scenes.thin_sheet(64, 3, thickness_cells=12),
    "noncubic_L3": lambda: scenes.fat_beam(64, 3, res=(64, 32, 32)),
    "beam128_L3": lambda: scenes.fat_beam(128, 3),                      # BASELINE configs[1]
    "beam64_L3_wall_rho_usolid": lambda: scenes.with_sampled_fields(scenes.fat_beam(64, 3, wall=True)),
    "sphere64_obstacle_rho_usolid": lambda: scenes.with_sampled_fields(scenes.sphere_with_obstacle(64, 4)),
    "tank64_L3_usolid": lambda: scenes.with_sampled_fields(scenes.tank(64, 3)),   # liquid on the domain border, walls as collision SDF
}


def assembled(sc, enhanced):
    sc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, use_enhanced_gradients=enhanced, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    s.assemble()
    return s, pyr


@pytest.mark.parametrize("enhanced", [True, False])
@pytest.mark.parametrize("name", list(CASES))
def test_scatter_form_equals_gathered_assembly(name, enhanced, built_lib):
    s, pyr = assembled(CASES[name](), enhanced)
    rp, col, val, rhs = s.csr()
    r = check_scatter_form(rp, col, val, rhs, s.initial_guess(), s.edge_stencils(), s.center_stencils(), pyr.n_center)
    assert r["mass_min"] >= 0.0 and r["mass_positive_fraction"] > 0.5 and r["dups"] == 0


@pytest.mark.parametrize("enhanced", [True, False])
@pytest.mark.parametrize("name", ["sphere64_L4", "beam32_L3", "sheet64_L3", "beam128_L3"])
def test_linear_shear_known_answer(name, enhanced, built_lib):
    sc = CASES[name]()
    s, pyr = assembled(sc, enhanced)
    a = 3.0
    r = check_linear_shear(s.dof_table(capi.INDEX_VELOCITY), s.dof_table(capi.INDEX_EDGE), sc.dx, s.edge_stencils(),
                           s.center_stencils(), pyr.n_center, a=a)
    print(name, "enhanced" if enhanced else "plain", r)
    assert r["edge_uniform_n"] > 0 and r["edge_uniform_max"] <= 1e-9 * a
    assert r["center_n"] > 0 and r["center_max"] <= 1e-9 * a
    assert r["edge_transition_n"] > 0
    if enhanced:
        assert r["edge_transition_max"] <= 1e-9 * a, r
    else:
        assert r["edge_transition_bad"] > 0


@pytest.mark.parametrize("name", ["sphere_obstacle", "beam_wall"])
def test_density_and_solid_velocity_are_sampled_where_the_reference_samples_them(name, built_lib):
    """tests/independent.py (iii) against the HIP path: linear density / solid-velocity fields reveal the sample positions."""
    def run(sc):
        s, pyr = assembled(sc, True)
        rp, col, val, _ = s.csr()
        return dict(vel_table=s.dof_table(capi.INDEX_VELOCITY), edge_table=s.dof_table(capi.INDEX_EDGE),
                    center_table=s.dof_table(capi.INDEX_CENTER), n_center=pyr.n_center, csr=(rp, col, val),
                    edge=s.edge_stencils(), center=s.center_stencils())
    r = sampled_field_runs(name, 64, run)
    print(name, r)
    assert r["density_n"] > 1000 and len(r["density_levels"]) >= 2 and r["density_max_rel"] < 1e-6
    assert r["edge_boundary_n"] > 500 and r["edge_boundary_bad"] == 0
    assert r["center_boundary_n"] > 500 and r["center_boundary_bad"] == 0
