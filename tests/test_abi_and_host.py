"""CPU-side checks: the C-ABI library loads and exports every symbol include/cfd_hip.h
declares (no compute calls without a GPU); host-side logic (mesh contract, BC table packing,
schedules, log formats)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import REPO, load_golden
from oracle import cases
from oracle import cfd_oracle as orc


def _declared():
    text = (REPO / "include" / "cfd_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cfd_amd import _lib
    names = _declared()
    assert len(names) >= 45
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cfd_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    loaded = _lib.load()
    assert loaded.cfd_version() >= 100
    assert loaded.cfd_workspace_bytes(64, 64) > 54 * 66 * 96 * 8
    assert loaded.cfd_workspace_bytes(0, 5) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cfd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.HipError):
        _lib.load()


def test_product_mesh_contract_bitwise():
    from cfd_amd import mesh
    ref = load_golden("ref_runs/Coarse_mesh.npz")
    mine = mesh.generate_rect_mesh(0.02, 0.05, 5, 20, 1.0, 1.03)
    assert set(mine) == set(ref)
    for k in ref:
        assert np.array_equal(np.asarray(mine[k]), ref[k]), k
    g = load_golden("func_annulus_12x16.npz")
    mine = mesh.generate_annulus_mesh(0.5, 6.0, 12, 16, 0.0, float(np.pi), 1.15, 1.0)
    for k in mine:
        assert np.array_equal(np.asarray(mine[k]), g["mesh_" + k]), k
    seg = mesh._graded(2.0, 10, [(0.5, 1.2), (0.5, 1.0 / 1.2)])
    assert np.array_equal(seg, orc.stretch_1d(2.0, 10, [(0.5, 1.2), (0.5, 1.0 / 1.2)]))


def _emulate_flow_tables(tables, fields, ni, nj):
    """numpy replay of the per-index tables (what k_flow_bc does)."""
    from cfd_amd import _lib as L
    f = {k: v.copy() for k, v in fields.items()}
    sel = {0: (lambda a: (a[1, 1:-1], (0, slice(1, -1)), a[ni, 1:-1])),
           1: (lambda a: (a[ni, 1:-1], (ni + 1, slice(1, -1)), a[1, 1:-1])),
           2: (lambda a: (a[1:-1, 1], (slice(1, -1), 0), a[1:-1, nj])),
           3: (lambda a: (a[1:-1, nj], (slice(1, -1), nj + 1), a[1:-1, 1]))}
    for side in range(4):
        t = tables.flow[side]
        wrap_p = tables.wrap_p_i if side < 2 else tables.wrap_p_j
        uin, ug, uop = sel[side](f["u"])
        vin, vg, vop = sel[side](f["v"])
        pin_, pg, pop = sel[side](f["p"])
        uin, vin, pin_, uop, vop, pop = (x.copy() for x in (uin, vin, pin_, uop, vop, pop))
        kuv, kp = t["kind_uv"], t["kind_p"]
        un = uin * t["nx"] + vin * t["ny"]
        newu = np.where(kuv == L.BC_PERIODIC, uop, np.where(kuv == L.BC_AFFINE, t["ua"] * uin + t["ub"],
                        np.where(kuv == L.BC_SYMMETRY, uin - 2.0 * un * t["nx"], f["u"][ug])))
        newv = np.where(kuv == L.BC_PERIODIC, vop, np.where(kuv == L.BC_AFFINE, t["va"] * vin + t["vb"],
                        np.where(kuv == L.BC_SYMMETRY, vin - 2.0 * un * t["ny"], f["v"][vg])))
        newp = np.where((kp == L.BC_PERIODIC) & wrap_p, pop,
                        np.where(kp == L.BC_AFFINE, t["pa"] * pin_ + t["pb"], f["p"][pg]))
        f["u"][ug], f["v"][vg], f["p"][pg] = newu, newv, newp
        for name in ("k", "omega", "nu_t"):
            a_in, a_g, a_op = sel[side](f[name])
            f[name][a_g] = np.where(kuv == L.BC_PERIODIC, a_op.copy(), f[name][a_g])
    return f


@pytest.mark.parametrize("tag,bc_cfg", [
    ("annulus_12x16", cases._half_cylinder_bc(12, 16, True)),
    ("rect_8x10", {"i_min": {"type": "periodic", "range": (0, 9)}, "i_max": {"type": "periodic", "range": (0, 9)},
                   "j_min": {"type": "wall", "range": (0, 7)}, "j_max": {"type": "symmetry", "range": (0, 7)},
                   "pressure_reference": {"cell": (1, 2), "value": 0.25}})])
def test_bc_tables_reproduce_reference_ghosts(tag, bc_cfg):
    """The host-packed tables, replayed in numpy, give the reference's ghost cells bit for bit."""
    from cfd_amd.bc import BoundaryGeometry, BoundaryTables
    g = load_golden(f"func_{tag}.npz")
    mesh = {k[5:]: v for k, v in g.items() if k.startswith("mesh_")}
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    nu = float(g["params"][0])
    geo = BoundaryGeometry.from_nodes(mesh["node"])
    for side in ("i_min", "i_max", "j_min", "j_max"):
        unit, s = orc.edge_geometry(mesh, side, None)
        assert np.array_equal(geo.n_hat[side], unit) and np.array_equal(geo.s[side], s), side
    tables = BoundaryTables(bc_cfg, ni, nj, geo, nu, True)
    fields = {k: g[f"in_{k}"] for k in ("u", "v", "p", "k", "omega", "nu_t")}
    out = _emulate_flow_tables(tables, fields, ni, nj)
    for k in fields:
        assert np.array_equal(out[k], g[f"bc_{k}"]), k
    masks, has = orc.pressure_dirichlet_masks(bc_cfg, ni, nj)
    assert has == tables.has_dirichlet
    for q, side in enumerate(("i_min", "i_max", "j_min", "j_max")):
        assert np.array_equal(tables.p_masks[q].astype(bool), masks[side])
    ppi, ppj, _, _ = orc.pressure_periodicity(bc_cfg, ni, nj)
    assert (ppi, ppj) == (tables.pp_per_i, tables.pp_per_j)
    assert len(tables.turb_ops) > 0 and tables.turb_values.size == sum(o[4] - o[3] + 1 for o in tables.turb_ops if o[0] >= 2)


def test_bc_validation_errors_match_reference():
    from cfd_amd.bc import get_periodic_pairs, _parse_index_range
    with pytest.raises(ValueError, match="both i_min and i_max"):
        get_periodic_pairs({"i_min": {"type": "periodic"}, "i_max": {"type": "wall"}}, 4, 4)
    with pytest.raises(ValueError, match="span the full"):
        get_periodic_pairs({"i_min": {"type": "periodic", "range": (0, 2)}, "i_max": {"type": "periodic", "range": (0, 2)}}, 4, 4)
    with pytest.raises(ValueError, match="cannot be mixed"):
        get_periodic_pairs({"i_min": [{"type": "periodic"}, {"type": "wall"}], "i_max": {"type": "periodic"}}, 4, 4)
    with pytest.raises(ValueError, match="out of bounds"):
        _parse_index_range((0, 9), 5)


def test_schedules_and_logs(tmp_path):
    from cfd_amd import solver
    sched = solver._schedule_of([(0.4, 4), (0.2, 0)], "cfl_schedule")
    assert sched == [(0.2, 0), (0.4, 4)]
    assert solver._value_at(0, 9.0, sched) == 0.2
    assert solver._value_at(2, 9.0, sched) == pytest.approx(0.3)
    assert solver._value_at(99, 9.0, sched) == 0.4
    assert solver._value_at(3, 9.0, None) == 9.0
    with pytest.raises(ValueError):
        solver._schedule_of([(1.0, 3), (2.0, 3)], "x")
    p = solver.log_residual(tmp_path, "c", 10, 1.5e-3, 2, 9.251522e-4, 1e-12, 2e-13)
    solver.log_residual(tmp_path, "c", 20, 3.0e-3, 2, 1.0e-4, 1e-12, 2e-13)
    lines = p.read_text().splitlines()
    assert lines[0] == "step time inner_iter continuity_l2 mom_u_l2 mom_v_l2"
    assert lines[1] == "10 1.500000e-03 2 9.251522e-04 1.000000e-12 2.000000e-13"
    solver._cut_log_at(tmp_path, "c", 20)
    assert len(p.read_text().splitlines()) == 2
    f = solver.log_friction(tmp_path, "c", 0, 0.0, 0.01, 0.5, bulk_velocity=0.8, re=1e4, cf_dpdx=0.02)
    assert f.read_text().splitlines()[1] == "0 0.000000e+00 4.000000e-02 8.000000e-02 5.000000e-01 8.000000e-01 1.000000e+04"
    with pytest.raises(ValueError, match="time.dt must be provided"):
        solver.StepControl({"time": {"max_steps": 1}})
    with pytest.raises(ValueError, match="dt_max must be provided"):
        solver.StepControl({"time": {"max_steps": 1, "cfl": 0.5}})


def test_state_roundtrip(tmp_path):
    from cfd_amd import mesh, solver
    m = mesh.generate_rect_mesh(1.0, 1.0, 4, 3)
    fields = solver.initialize_fields(m, {"u": 1.0, "v": np.arange(12.0).reshape(4, 3), "p": 0.0, "scalars": {"k": 2.0}})
    fields["p_corr"] = np.ones((6, 5))
    path = solver.save_fields(m, fields, tmp_path, 7, 0.25)
    assert path.name == "fields_000007.npz"
    back, step, time = solver.load_state(path, m, {"u": 0.0, "v": 0.0, "p": 0.0})
    assert step == 7 and time == 0.25 and "p_corr" not in back
    for k in ("u", "v", "p", "k"):
        assert np.array_equal(back[k], fields[k])
    ref = load_golden("ref_runs/k_omega_fields_000300.npz")
    m2 = mesh.generate_rect_mesh(0.02, 0.05, 5, 20, 1.0, 1.03)
    f2, s2, t2 = solver.load_state(REPO / "tests/golden/ref_runs/k_omega_fields_000300.npz", m2, {"u": 0, "v": 0, "p": 0})
    assert s2 == 300 and set(("u", "v", "p", "k", "omega", "nu_t")) <= set(f2)
    assert np.array_equal(f2["omega"], ref["omega"])


def test_friction_matches_reference_vectors():
    from cfd_amd import post_processing as pp
    for tag in ("annulus_12x16", "rect_8x10"):
        g = load_golden(f"func_{tag}.npz")
        mesh = {k[5:]: v for k, v in g.items() if k.startswith("mesh_")}
        f = {"u": g["bc_u"], "v": g["bc_v"], "nu_t": g["nu_t"]}
        nu = float(g["params"][0])
        for side in ("j_min", "i_min"):
            fr = pp.friction_coefficient_array(f, mesh, side, None, nu=nu, hydraulic_diameter=0.7,
                                               pressure_gradient=(-0.3, 0.0))
            got = np.array([fr["cf"], fr["tau_avg"], fr["bulk_velocity"], fr["re"], fr["cf_dpdx"]])
            np.testing.assert_allclose(got, g[f"friction_{side}"], rtol=1e-14)
