"""Thin object wrapper over the C ABI (include/cfd_hip.h): one Device = one GPU context
holding every mesh / field plane of one j-slab in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dptr, iptr


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Device:
    def __init__(self, ni, nj, device=0, workspace_ptr=None, workspace_bytes=0):
        self.lib = _lib.load()
        self.ni, self.nj = int(ni), int(nj)
        self._h = C.c_void_p()
        check(self.lib.cfd_create(C.byref(self._h), self.ni, self.nj, int(device),
                                  C.c_void_p(workspace_ptr) if workspace_ptr else None,
                                  int(workspace_bytes)))
        self.turbulent = False

    # j-slab attributes (cfd_amd.slab.slab_device overrides them on a multi-rank run): this
    # context holds rows [j_offset, j_offset + nj) of a mesh with nj_global rows.
    rank, nranks, j_offset = 0, 1, 0
    comm = None
    transport = None

    @property
    def pitch(self):
        """Row pitch of the device planes in doubles (cell (i, j) sits at (j + CFD_GHOST_ROWS) * pitch + i + CFD_COL_OFFSET)."""
        return int(self.lib.cfd_pitch(self._h))

    def comm_counts(self):
        """(halo exchanges, all-reduces, all-gathers) this context has issued so far (0, 0, 0 on one rank)."""
        cnt = (C.c_longlong * 3)()
        native = C.c_int(0)
        check(self.lib.cfd_comm_counts(self._h, cnt, C.byref(native)))
        return tuple(int(v) for v in cnt)

    def comm_counts_by_phase(self):
        """{phase: (halo exchanges, all-reduces, all-gathers)}; phases as in phase_times plus "other"."""
        cnt = (C.c_longlong * 27)()
        check(self.lib.cfd_comm_counts_by_phase(self._h, cnt))
        names = ("bc", "adv", "mom", "press", "corr", "cont", "turb", "dt", "other")
        return {n: tuple(int(cnt[3 * p + k]) for k in range(3)) for p, n in enumerate(names)}

    def comm_native(self):
        """True when the library's own RCCL communicator carries the slab collectives."""
        native = C.c_int(0)
        check(self.lib.cfd_comm_counts(self._h, None, C.byref(native)))
        return bool(native.value)

    def extra_device_bytes(self):
        """Device bytes the context holds outside its workspace (SST planes, line-multigrid data of stretched meshes)."""
        n = C.c_size_t(0)
        check(self.lib.cfd_extra_device_bytes(self._h, C.byref(n)))
        return int(n.value)

    def rccl_ranks(self):
        """Ranks of the library's own RCCL communicator as RCCL reports them (0: none)."""
        n = C.c_int(0)
        check(self.lib.cfd_rccl_ranks(self._h, C.byref(n)))
        return int(n.value)

    @property
    def nj_global(self):
        return self.__dict__.get("_nj_global", self.nj)

    @nj_global.setter
    def nj_global(self, v):
        self._nj_global = int(v)

    def close(self):
        if self._h:
            self.lib.cfd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- shapes
    def shape(self, kind, whole=False):
        ni, nj = self.ni, (self.nj_global if whole else self.nj)
        return {(_lib.KIND_GHOSTED): (ni + 2, nj + 2), (_lib.KIND_CELL): (ni, nj),
                (_lib.KIND_FACE_I): (ni + 1, nj), (_lib.KIND_FACE_J): (ni, nj + 1)}[kind]

    # ---------------------------------------------------------------- mesh
    def upload_mesh(self, mesh):
        wl, wr = np.asarray(mesh["w_left_i"]), np.asarray(mesh["w_right_i"])
        ws, wn = np.asarray(mesh["w_south_j"]), np.asarray(mesh["w_north_j"])
        if not (np.array_equal(wl, 1.0 - wr) and np.array_equal(ws, 1.0 - wn)):
            raise ValueError("mesh interpolation weights are not complementary (w_left != 1 - w_right); "
                             "regenerate the mesh with precompute_from_nodes")
        if self.nranks > 1 and np.asarray(mesh["cell_volume"]).shape == (self.ni, self.nj_global):
            from .slab import slab_mesh
            mesh = slab_mesh(mesh, self.j_offset, self.nj)
        arrs = [_f64(mesh[k]) for k in ("cell_volume", "w_right_i", "w_north_j", "face_normal_i",
                                        "face_normal_j", "diff_d_i", "diff_d_j", "cell_spacing_i",
                                        "cell_spacing_j")]
        if arrs[0].shape != (self.ni, self.nj):
            raise ValueError(f"mesh is {arrs[0].shape}, context is {(self.ni, self.nj)}")
        check(self.lib.cfd_upload_mesh(self._h, *[dptr(a) for a in arrs]))

    def mesh_from_nodes(self, node):
        node = _f64(node)
        if node.shape != (self.ni + 1, self.nj + 1, 2):
            raise ValueError("node must have shape (Ni+1, Nj+1, 2)")
        check(self.lib.cfd_mesh_from_nodes(self._h, dptr(node)))

    def mesh_from_rect(self, x_nodes, y_nodes):
        """y_nodes: the node axis of the WHOLE mesh (Nj_global + 1 entries); a slab context takes
        its own rows plus one node on either side."""
        x, y = _f64(x_nodes), _f64(y_nodes)
        if x.shape != (self.ni + 1,) or y.shape != (self.nj_global + 1,):
            raise ValueError("node axes must have Ni+1 and Nj+1 entries")
        if self.nranks > 1:
            idx = np.clip(np.arange(self.j_offset - 1, self.j_offset + self.nj + 2), 0, self.nj_global)
            y = np.ascontiguousarray(y[idx])
        check(self.lib.cfd_mesh_from_rect(self._h, dptr(x), dptr(y)))

    MESH_PLANES = ("vol", "wr_i", "wn_j", "nix", "niy", "njx", "njy", "s_i", "s_j", "d_i", "d_j", "lmin")
    _MESH_KINDS = (1, 2, 3, 2, 2, 3, 3, 2, 3, 2, 3, 1)

    def download_mesh_plane(self, name):
        pid = self.MESH_PLANES.index(name)
        out = np.empty(self.shape(self._MESH_KINDS[pid]))
        check(self.lib.cfd_download_mesh_plane(self._h, pid, dptr(out)))
        return out

    # ---------------------------------------------------------------- fields
    def upload(self, field_id, arr, kind=_lib.KIND_GHOSTED):
        a = np.asarray(arr)
        if self.nranks > 1 and a.shape == self.shape(kind, whole=True):
            from .slab import slab_of
            a = slab_of(a, self.j_offset, self.nj, kind)     # ghost rows at a cut = the neighbour's rows
        a = _f64(a)
        if a.shape != self.shape(kind):
            raise ValueError(f"Field shape {a.shape} does not match target {self.shape(kind)}.")
        check(self.lib.cfd_upload_field(self._h, field_id, dptr(a), kind))

    def download(self, field_id, kind=_lib.KIND_GHOSTED, out=None, whole=False):
        """whole=True on a j-slab run: every rank receives the assembled global array (collective)."""
        if whole and self.nranks > 1:
            from .slab import gather_slabs
            mine = np.empty(self.shape(kind))
            check(self.lib.cfd_download_field(self._h, field_id, dptr(mine), kind))
            full = gather_slabs(mine, kind, self.nj, self.nranks, self.comm.group)
            if out is None:
                return full
            out[...] = full
            return out
        if out is None:
            out = np.empty(self.shape(kind))
        check(self.lib.cfd_download_field(self._h, field_id, dptr(out), kind))
        return out

    def exchange(self, field_id):
        """j-slab runs: refresh the ghost rows at the cuts from the neighbour ranks (collective)."""
        check(self.lib.cfd_exchange_field(self._h, int(field_id)))

    def upload_fields(self, fields):
        for name, fid in _lib.FIELD_IDS.items():
            if name in fields and name != "p_corr":
                self.upload(fid, fields[name])
        if getattr(self, "sst", False) and all(name in fields for name in _lib.SST_FIELD_IDS):
            # a restart state of an SST run carries the caches (solver.py:1296-1310 saves every entry of fields)
            for name, fid in _lib.SST_FIELD_IDS.items():
                self.upload(fid, fields[name])
            check(self.lib.cfd_sst_set_cached(self._h, 1))

    def download_fields(self, fields):
        """Refresh the host dict in place (arrays are caller visible, solver.py:333)."""
        for name, fid in _lib.FIELD_IDS.items():
            if name in fields:
                self.download(fid, out=fields[name], whole=True)
        if getattr(self, "sst", False) and self.sst_cached():
            # the reference's SST caches live inside `fields` (turbulence.py:473, 505, 521, 346)
            for name, fid in _lib.SST_FIELD_IDS.items():
                if name not in fields:
                    fields[name] = np.empty(self.shape(_lib.KIND_GHOSTED, whole=True))
                self.download(fid, out=fields[name], whole=True)
        return fields

    # ---------------------------------------------------------------- Menter SST
    def set_turbulence_model(self, model, mesh=None, bc_cfg=None):
        """model: "k_omega" | "sst".  SST needs mesh["cell_center"], mesh["node"] and the boundary config
        (wall face centres, turbulence.py:69-93)."""
        self.sst = model == "sst"
        check(self.lib.cfd_set_turbulence_model(self._h, _lib.TURB_SST if self.sst else _lib.TURB_KOMEGA))
        if not self.sst:
            return
        if mesh is None or bc_cfg is None:
            raise ValueError("the SST closure needs the mesh (cell_center, node) and the boundary config")
        cc = np.asarray(mesh["cell_center"])
        if self.nranks > 1 and cc.shape == (self.ni, self.nj_global, 2):
            cc = cc[:, self.j_offset:self.j_offset + self.nj]          # this slab's rows of the whole mesh
        cc = _f64(cc)
        if cc.shape != (self.ni, self.nj, 2):
            raise ValueError(f"cell_center is {cc.shape}, context is {(self.ni, self.nj, 2)}")
        check(self.lib.cfd_upload_cell_centers(self._h, dptr(cc)))
        from .bc import wall_face_centers
        xy = _f64(wall_face_centers(mesh["node"], bc_cfg))
        check(self.lib.cfd_set_wall_faces(self._h, int(xy.shape[0]), dptr(xy) if xy.shape[0] else None))

    def sst_cached(self):
        out = C.c_int(0)
        check(self.lib.cfd_sst_get_cached(self._h, C.byref(out)))
        return bool(out.value)

    # ---------------------------------------------------------------- parameters
    def set_physics(self, nu, pressure_gradient=(0.0, 0.0), rhie_chow=True, turbulent=False):
        self.turbulent = bool(turbulent)
        check(self.lib.cfd_set_physics(self._h, float(nu), float(pressure_gradient[0]),
                                       float(pressure_gradient[1]), int(bool(rhie_chow)), int(self.turbulent)))

    def set_solver(self, mom_tol=1e-8, mom_maxiter=200, p_tol=1e-8, p_maxiter=200, turb_tol=1e-8,
                   turb_maxiter=200, freeze_pressure=True):
        check(self.lib.cfd_set_solver(self._h, float(mom_tol), int(mom_maxiter), float(p_tol), int(p_maxiter),
                                      float(turb_tol), int(turb_maxiter), int(bool(freeze_pressure))))

    def set_multigrid(self, nu_pre=3, nu_post=3, omega=None, coarse_sweeps=40, fused=True, omega2=None, omegas=None):
        """omegas: list of 1-4 Jacobi damping factors used cyclically by the sweeps (default (1.25, 0.5, 1.0));
        omega / omega2: the older two-value form."""
        if omegas is None:
            if omega is None:
                omegas = (1.25, 0.5, 1.0) if nu_pre == 3 else (1.0, 0.6)
            else:
                omegas = (omega,) if not omega2 else (omega, omega2)
        om = np.ascontiguousarray(omegas, dtype=np.float64)
        check(self.lib.cfd_set_multigrid(self._h, int(nu_pre), int(nu_post), float(min(om[0], 1.0)), int(coarse_sweeps),
                                         int(bool(fused))))
        check(self.lib.cfd_set_smoother_omegas(self._h, int(om.size), dptr(om)))

    def set_transport_solver(self, warm_start=True, chebyshev=True, rho_max=0.95, max_iters=80):
        check(self.lib.cfd_set_transport_solver(self._h, int(bool(warm_start)), int(bool(chebyshev)), float(rho_max),
                                                int(max_iters)))

    def set_pressure_nullspace(self, on=True):
        check(self.lib.cfd_set_pressure_nullspace(self._h, int(bool(on))))

    def set_flux_kernels(self, marching=True, lookahead=True):
        check(self.lib.cfd_set_flux_kernels(self._h, int(bool(marching)), int(bool(lookahead))))

    def set_turbulence_clip(self, k_min, omega_min):
        check(self.lib.cfd_set_turbulence_clip(self._h, float(k_min), float(omega_min)))

    # ---------------------------------------------------------------- BC tables
    def set_boundary(self, tables):
        """tables: cfd_amd.bc.BoundaryTables"""
        for side in range(4):
            t = tables.flow[side]
            check(self.lib.cfd_set_flow_bc(self._h, side, len(t["kind_uv"]), iptr(t["kind_uv"]), iptr(t["kind_p"]),
                                           dptr(t["ua"]), dptr(t["ub"]), dptr(t["va"]), dptr(t["vb"]),
                                           dptr(t["pa"]), dptr(t["pb"]), dptr(t["nx"]), dptr(t["ny"])))
        check(self.lib.cfd_set_periodicity(self._h, int(tables.per_i), int(tables.per_j),
                                           int(tables.wrap_p_i), int(tables.wrap_p_j)))
        m = tables.p_masks
        check(self.lib.cfd_set_pressure_bc(self._h, iptr(m[0]), iptr(m[1]), iptr(m[2]), iptr(m[3]),
                                           int(tables.pp_per_i), int(tables.pp_per_j), int(tables.has_dirichlet),
                                           int(tables.pin[0]), int(tables.pin[1]), float(tables.pin_value)))
        ops = (_lib.TurbOp * max(1, len(tables.turb_ops)))()
        for q, o in enumerate(tables.turb_ops):
            ops[q] = _lib.TurbOp(*o)
        vals = _f64(tables.turb_values) if len(tables.turb_values) else np.zeros(1)
        check(self.lib.cfd_set_turb_bc(self._h, len(tables.turb_ops), ops, len(tables.turb_values), dptr(vals)))

    # ---------------------------------------------------------------- the step
    def apply_flow_bc(self):
        check(self.lib.cfd_apply_flow_bc(self._h))

    def compute_dt(self, cfl, cfl_diff, dt_min, dt_max):
        out = np.zeros(3)
        check(self.lib.cfd_compute_dt(self._h, int(cfl is not None), float(cfl or 0.0),
                                      int(cfl_diff is not None), float(cfl_diff or 0.0),
                                      float(dt_min if dt_min is not None else -np.inf),
                                      int(dt_max is not None), float(dt_max or 0.0), dptr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def simple_iter(self, dt, relax_u, relax_v, relax_p):
        res = np.zeros(3)
        check(self.lib.cfd_simple_iter(self._h, float(dt), float(relax_u), float(relax_v), float(relax_p), dptr(res)))
        return float(res[0]), float(res[1]), float(res[2])

    def piso_step(self, dt, n_correctors, relax_u, relax_v, relax_p):
        res = np.zeros(3)
        check(self.lib.cfd_piso_step(self._h, float(dt), int(n_correctors), float(relax_u), float(relax_v),
                                     float(relax_p), dptr(res)))
        return float(res[0]), float(res[1]), float(res[2])

    def turbulence_step(self, dt):
        check(self.lib.cfd_turbulence_step(self._h, float(dt)))

    def stats(self):
        s = _lib.SolveStats()
        check(self.lib.cfd_get_stats(self._h, C.byref(s)))
        return {"mom_iters": s.mom_iters, "p_iters": s.p_iters, "k_iters": s.k_iters,
                "p_solver": "mg-pcg" if s.p_solver == 0 else "mg-bicgstab",
                "mom_solver": "chebyshev" if s.mom_solver == 1 else "bicgstab",
                "k_solver": "chebyshev" if s.k_solver == 1 else "bicgstab",
                "mom_relres": tuple(s.mom_relres), "p_relres": s.p_relres, "k_relres": tuple(s.k_relres),
                "p_unconverged": s.p_unconverged, "transport_stalled": s.transport_stalled}

    def bulk_velocity(self, field_id=_lib.F_U):
        out = C.c_double()
        check(self.lib.cfd_bulk_velocity(self._h, field_id, C.byref(out)))
        return out.value

    def synchronize(self):
        check(self.lib.cfd_synchronize(self._h))

    # ---------------------------------------------------------------- fine-grained
    def eddy_viscosity(self):
        check(self.lib.cfd_eddy_viscosity(self._h))

    def komega_sources(self, grad_u, grad_v):
        """(source_k, source_omega) of the Wilcox closure from the k, omega, nu_t planes and the caller's gradients."""
        gu, gv = _f64(grad_u), _f64(grad_v)
        if gu.shape != (self.ni, self.nj, 2) or gv.shape != (self.ni, self.nj, 2):
            raise ValueError(f"velocity gradients must be {(self.ni, self.nj, 2)} arrays")
        sk, sw = np.empty((self.ni, self.nj)), np.empty((self.ni, self.nj))
        check(self.lib.cfd_komega_sources(self._h, dptr(gu), dptr(gv), dptr(sk), dptr(sw)))
        return sk, sw

    def effective_diffusivity(self, nu, sigma):
        out = np.empty((self.ni + 2, self.nj + 2))
        check(self.lib.cfd_effective_diffusivity(self._h, float(nu), float(sigma), dptr(out)))
        return out

    def momentum_diagonal(self, dt, nut_coef=1.0):
        check(self.lib.cfd_momentum_diagonal(self._h, float(dt), float(nut_coef)))

    def face_flux(self, rhie_chow=True, use_peff=False):
        fi, fj = np.empty(self.shape(2)), np.empty(self.shape(3))
        check(self.lib.cfd_face_flux(self._h, int(rhie_chow), int(use_peff), dptr(fi), dptr(fj)))
        return fi, fj

    def grad(self, field_id):
        g = np.empty((self.ni, self.nj, 2))
        check(self.lib.cfd_grad(self._h, field_id, dptr(g)))
        return g

    def advect(self, field_id, flux_i, flux_j):
        out = np.empty((self.ni, self.nj))
        check(self.lib.cfd_advect(self._h, field_id, dptr(_f64(flux_i)), dptr(_f64(flux_j)), dptr(out)))
        return out

    def momentum_rhs(self, dt):
        check(self.lib.cfd_momentum_rhs(self._h, float(dt)))

    def momentum_rhs_from(self, dt, adv_u, adv_v, extra=None, pressure_gradient=(0.0, 0.0)):
        """b_u, b_v planes from the caller's advective terms (solver.solve_momentum, solver.py:868-883)."""
        au, av = _f64(adv_u), _f64(adv_v)
        if au.shape != (self.ni, self.nj) or av.shape != (self.ni, self.nj):
            raise ValueError(f"advective terms must be {(self.ni, self.nj)} arrays")
        ex = None
        if extra is not None:
            ex = _f64(extra)
            if ex.shape != (self.ni + 2, self.nj + 2):
                raise ValueError(f"extra_pressure_field must be a {(self.ni + 2, self.nj + 2)} array")
        check(self.lib.cfd_momentum_rhs_from(self._h, float(dt), dptr(au), dptr(av), dptr(ex),
                                             float(pressure_gradient[0]), float(pressure_gradient[1])))

    def wall_friction(self, side, lo, hi, t_hat, dn, seg_len, nu, rho=1.0, use_nu_t=False, want_tau=False):
        """(sum tau * len, sum len[, tau]) over the faces lo..hi of one side, from the planes in HBM
        (include/cfd_hip.h: cfd_wall_friction)."""
        n = int(hi) - int(lo) + 1
        t, d, ln = _f64(t_hat), _f64(dn), _f64(seg_len)
        if t.shape != (n, 2) or d.shape != (n,) or ln.shape != (n,):
            raise ValueError("t_hat (n, 2), dn (n), seg_len (n) must cover the faces lo..hi")
        tau = np.empty(n) if want_tau else None
        out = np.zeros(2)
        check(self.lib.cfd_wall_friction(self._h, int(side), int(lo), int(hi), dptr(t), dptr(d), dptr(ln), float(nu),
                                         float(rho), int(bool(use_nu_t)), dptr(tau), dptr(out)))
        return (float(out[0]), float(out[1]), tau) if want_tau else (float(out[0]), float(out[1]))

    def momentum_solve(self):
        res = np.zeros(2)
        check(self.lib.cfd_momentum_solve(self._h, dptr(res)))
        return float(res[0]), float(res[1])

    def pressure_solve(self):
        check(self.lib.cfd_pressure_solve(self._h))

    def pressure_system(self):
        ci, cj = np.empty(self.shape(2)), np.empty(self.shape(3))
        d, b = np.empty(self.shape(1)), np.empty(self.shape(1))
        check(self.lib.cfd_download_pressure_system(self._h, dptr(ci), dptr(cj), dptr(d), dptr(b)))
        return ci, cj, d, b

    def snapshot(self):
        check(self.lib.cfd_snapshot(self._h))

    def correct(self, relax_u, relax_v, relax_p):
        check(self.lib.cfd_correct(self._h, float(relax_u), float(relax_v), float(relax_p)))

    def continuity(self):
        out = C.c_double()
        check(self.lib.cfd_continuity(self._h, C.byref(out)))
        return out.value

    def apply_turb_bc(self):
        check(self.lib.cfd_apply_turb_bc(self._h))

    def spmv(self, which, x):
        y = np.empty((self.ni, self.nj))
        check(self.lib.cfd_spmv(self._h, int(which), dptr(_f64(x)), dptr(y)))
        return y

    def vcycle(self, r):
        z = np.empty((self.ni, self.nj))
        check(self.lib.cfd_vcycle(self._h, dptr(_f64(r)), dptr(z)))
        return z

    def pressure_krylov(self, b, tol=1e-10, maxiter=200):
        x = np.empty((self.ni, self.nj))
        it, rr = C.c_int(), C.c_double()
        check(self.lib.cfd_pressure_krylov(self._h, dptr(_f64(b)), dptr(x), float(tol), int(maxiter),
                                           C.byref(it), C.byref(rr)))
        return x, it.value, rr.value

    def time_kernel(self, kernel, reps=20):
        out = C.c_double()
        check(self.lib.cfd_time_kernel(self._h, int(kernel), int(reps), C.byref(out)))
        return out.value

    def probe(self, kernel):
        check(self.lib.cfd_probe(self._h, int(kernel)))

    def probe_read(self, with_bytes=False, kernel=-1):
        """(launches timed, mean duration in us[, mean algorithmic bytes per cell of those launches]); kernel >= 0: of that
        kernel id alone (after probe(_lib.PROBE_ALL))."""
        n, us, bpc = C.c_int(), C.c_double(), C.c_double()
        check(self.lib.cfd_probe_read_id(self._h, int(kernel), C.byref(n), C.byref(us), C.byref(bpc)))
        return (n.value, us.value, bpc.value) if with_bytes else (n.value, us.value)

    def level_info(self):
        """(multigrid levels of the built pressure hierarchy, index of the first level that is replicated on every rank
        of a slab run -- levels below it are slab-distributed; 0 levels before the first pressure solve)."""
        n, first = C.c_int(), C.c_int()
        check(self.lib.cfd_level_info(self._h, C.byref(n), C.byref(first)))
        return n.value, first.value

    def level_smoothers(self):
        """[(ni, nj, strong / weak coupling ratio, line-smoothed?)] per multigrid level of the built pressure hierarchy."""
        n, first = C.c_int(), C.c_int()
        check(self.lib.cfd_level_info(self._h, C.byref(n), C.byref(first)))
        out = []
        for lvl in range(n.value):
            ni, nj, an, ln = C.c_int(), C.c_int(), C.c_double(), C.c_int()
            check(self.lib.cfd_level_smoother(self._h, lvl, C.byref(ni), C.byref(nj), C.byref(an), C.byref(ln)))
            out.append((ni.value, nj.value, an.value, bool(ln.value)))
        return out

    def set_option(self, name, value):
        """Tuning / A-B switch of a kernel form (include/cfd_hip.h: cfd_set_option)."""
        check(self.lib.cfd_set_option(self._h, str(name).encode(), float(value)))

    KERNEL_COUNT_NAMES = ("cheb4_2rhs", "cheb4_1rhs", "cheb2", "cheb1", "bicgstab_iters", "flux_march",
                          "flux_march_multistrip", "flux_tile", "mg_fused_levels_max", "mg_deferred_resid", "cg_pq",
                          "vcycles", "mg_line_levels_max", "mg_tail_levels", "correct_flux", "mg_semi_levels",
                          "speculative_iterations", "speculative_redone", "pcg_tail", "cheb_tail",
                          "cheb6_1rhs", "mg_float_levels", "mesh_form", "pcg_xsum")

    def kernel_counts(self, reset=False):
        """Launches per kernel family since the last reset (include/cfd_hip.h: cfd_kernel_counts)."""
        out = (C.c_longlong * 24)()
        check(self.lib.cfd_kernel_counts(self._h, out, 24, int(reset)))
        return dict(zip(self.KERNEL_COUNT_NAMES, [int(v) for v in out]))

    def host_counters(self, reset=False):
        """(kernel launches, blocking stream synchronisations) of the library in this process so far."""
        out = (C.c_longlong * 2)()
        check(self.lib.cfd_host_counters(out, int(reset)))
        return int(out[0]), int(out[1])

    SYNC_PHASES = ("bc", "adv", "mom", "press", "corr", "cont", "turb", "dt", "other")

    def host_syncs_by_phase(self, reset=False):
        """Blocking stream synchronisations of the library in this process by the phase that issued them."""
        out = (C.c_longlong * 9)()
        check(self.lib.cfd_host_syncs_by_phase(out, int(reset)))
        return dict(zip(self.SYNC_PHASES, [int(v) for v in out]))

    def enable_phase_timing(self, on=True):
        check(self.lib.cfd_enable_phase_timing(self._h, int(on)))

    def phase_times(self, reset=True):
        out = np.zeros(8)
        check(self.lib.cfd_phase_times(self._h, dptr(out), int(reset)))
        return dict(zip(("bc", "adv", "mom", "press", "corr", "cont", "turb", "dt"), out / 1e6))
