"""One rank of a j-slab run (started by tests/test_gpu_slab.py, once per rank).

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment
    python tests/slab_worker.py <case> <out_dir> <mode>

Every rank shares GPU 0 (the test box has one GPU).  mode selects process-group backend and slab transport:
    gloo        gloo group, torch.distributed callbacks staged through the host
    rccl        gloo group for the host side, the library's native RCCL transport for the slabs
    nccl        nccl group, native RCCL transport (what a multi-GPU node runs by default)
    nccl-fallback   as nccl, but the native transport is reported unavailable on the last rank
    nccl-torch  nccl group, torch.distributed callbacks on device views
The RCCL modes rehearse on one device with one fake host id per rank (cfd_amd.slab.rehearsal_env).
Rank 0 writes <out_dir>/result.npz.
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def build_case(name, work):
    from oracle import cases
    from cfd_amd import mesh as hip_mesh
    table = {
        "channel_komega_48x64": (lambda: cases.channel_forced(n=64, steps=3, turbulent=True, ratio_j=1.02), cases.perturbed_parabola, {"count_i": 48}),
        "channel_sst_48x64": (lambda: cases.channel_forced(n=64, steps=3, turbulent=True, ratio_j=1.03, model="sst"), cases.perturbed_parabola, {"count_i": 48}),
        "half_cylinder_sst_24x32": (lambda: cases.half_cylinder(nr=24, nt=32, steps=3, turbulent=True, model="sst"), None, {}),
        "channel_lam_64x128": (lambda: cases.channel_forced(n=64, steps=2, ratio_j=1.0), cases.perturbed_parabola, {"count_j": 128}),
        # large enough for two fused multigrid levels above the tail: the first coarse level keeps single-precision vectors
        "channel_lam_256x256": (lambda: cases.channel_forced(n=256, steps=2, ratio_j=1.0), cases.perturbed_parabola, {}),
        "channel_lam_512x512": (lambda: cases.channel_forced(n=512, steps=2, ratio_j=1.0), cases.perturbed_parabola, {}),
        "channel_lam_64x256": (lambda: cases.channel_forced(n=64, steps=2, ratio_j=1.0), cases.perturbed_parabola, {"count_j": 256}),
        "inlet_outlet_40x32": (lambda: cases.inlet_outlet_channel(ni=40, nj=32, steps=4), None, {}),
        "couette_piso_24x32": (lambda: cases.couette_piso(ni=24, nj=32, steps=4), cases.perturbed_rest, {}),
        "half_cylinder_komega_24x32": (lambda: cases.half_cylinder(nr=24, nt=32, steps=3, turbulent=True), None, {}),
    }
    builder, fixup, mp_over = table[name]
    cfg, mp, kind = builder()
    if mp_over:
        # non-square variants of the square channel builder: patch the counts and the BC ranges
        mp.update(mp_over)
        ni, nj = mp["count_i"], mp["count_j"]
        for side, n in (("i_min", nj), ("i_max", nj), ("j_min", ni), ("j_max", ni)):
            cfg["boundary"][side]["range"] = (0, n - 1)
    gen = hip_mesh.generate_rect_mesh if kind == "rect" else hip_mesh.generate_annulus_mesh
    cfg["mesh"] = {"generator": gen, "params": mp, "path": work / "mesh" / "mesh.npz"}
    cfg["paths"] = {"case_dir": work, "mesh_dir": work / "mesh", "states_dir": work / "states",
                    "post_dir": work / "postproc"}
    mesh = hip_mesh.make_mesh(cfg["mesh"])
    if fixup is not None:
        fixup(cfg, mesh)
    cfg["solver"]["turbulence_solver"] = {"tol": 1e-13, "maxiter": 2000}
    # small test meshes: keep the multigrid levels slab-distributed as long as possible
    cfg["solver"]["slab"] = {"min_rows": int(os.environ.get("SLAB_MIN_ROWS", "16"))}
    if name in ("channel_lam_256x256", "channel_lam_512x512"):
        # (default: coarse levels of more than 2^18 cells; here the 128^2 level, not the 64^2 one a single rank's tail takes)
        cfg["solver"]["tuning"] = {"mg_c32_min_cells": 4096}
    if os.environ.get("SLAB_UNFUSED"):
        cfg["solver"]["multigrid"] = {"fused": False}
    return cfg


def mesh_from_rect_check(out_dir, rank, world):
    """Slab metrics computed on the device from the node axes (one halo node per side, cut face rows
    from the neighbour) against the slices of the host mesh contract."""
    import torch.distributed as dist
    from cfd_amd import mesh as hip_mesh
    from cfd_amd.slab import slab_device, slab_mesh
    mp = dict(length_i=2.0, length_j=1.0, count_i=40, count_j=48, ratio_i=1.03, ratio_j=1.05)
    x, y = hip_mesh.rect_node_axes(**mp)
    host = hip_mesh.precompute_from_nodes(hip_mesh.generate_rect_nodes(**mp))
    dev = slab_device(mp["count_i"], mp["count_j"], 0)
    dev.mesh_from_rect(x, y)
    sl = slab_mesh(host, dev.j_offset, dev.nj)
    want = {"vol": sl["cell_volume"], "wr_i": sl["w_right_i"], "wn_j": sl["w_north_j"],
            "nix": sl["face_normal_i"][..., 0], "niy": sl["face_normal_i"][..., 1],
            "njx": sl["face_normal_j"][..., 0], "njy": sl["face_normal_j"][..., 1],
            "s_i": np.hypot(sl["face_normal_i"][..., 0], sl["face_normal_i"][..., 1]),
            "s_j": np.hypot(sl["face_normal_j"][..., 0], sl["face_normal_j"][..., 1]),
            "d_i": sl["diff_d_i"], "d_j": sl["diff_d_j"],
            "lmin": np.minimum(sl["cell_spacing_i"], sl["cell_spacing_j"])}
    worst = 0.0
    for name, ref in want.items():
        got = dev.download_mesh_plane(name)
        scale = max(1.0, float(np.abs(ref).max()))
        worst = max(worst, float(np.abs(got - ref).max() / scale))
    dev.close()
    allw = [None] * world
    if world > 1:
        dist.all_gather_object(allw, worst)
    else:
        allw = [worst]
    if rank == 0:
        np.savez(out_dir / "result.npz", worst=np.array(allw))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_check(name, out_dir, rank, world):
    """One fused V-cycle on a random right-hand side and one SIMPLE iteration with the solver statistics:
    what the cut handling of the fused kernels (three sweeps / four Chebyshev iterations per pass, halo rows from
    the neighbour) must reproduce exactly, not just up to the solver tolerance."""
    import torch.distributed as dist
    from cfd_amd import _lib as L
    from cfd_amd import solver as S
    from cfd_amd.bc import BoundaryGeometry
    from cfd_amd.slab import gather_slabs, slab_device, slab_of
    work = out_dir / f"rank{rank}"
    cfg = build_case(name, work)
    mesh = S.load_mesh(cfg["mesh"])
    ctl = S.StepControl(cfg)
    fields = S.initialize_fields(mesh, cfg["initial"])
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    fields.setdefault("nu_t", np.zeros((ni + 2, nj + 2)))
    r = np.random.default_rng(3).standard_normal((ni, nj))
    dev = slab_device(ni, nj, 0, min_slab_rows=int(os.environ.get("SLAB_MIN_ROWS", "16")))
    dev.upload_mesh(mesh)
    S.configure_device(dev, cfg, BoundaryGeometry.from_nodes(mesh["node"]), cfg["physics"]["nu"], ctl.turb_on)
    dev.upload_fields(fields)
    dev.apply_flow_bc()
    dt = dev.compute_dt(0.5, 0.5, 0.0, 1.0e-3)[0]
    dev.momentum_diagonal(dt)
    z = dev.vcycle(slab_of(r, dev.j_offset, dev.nj, L.KIND_CELL))
    if world > 1:
        z = gather_slabs(z, L.KIND_CELL, dev.nj, world)
    res = dev.simple_iter(dt, 0.7, 0.7, 0.3)
    st = dev.stats()
    native, counts = dev.comm_native(), dev.comm_counts()
    dev.close()
    if rank == 0:
        np.savez(out_dir / "result.npz", z=z, res=np.array(res), mom_iters=st["mom_iters"], p_iters=st["p_iters"],
                 cheb=int(st["mom_solver"] == "chebyshev"), native=int(native), counts=np.array(counts))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def friction_check(name, out_dir, rank, world):
    """The in-loop friction diagnostic (cfd_wall_friction) on slabs: an i-side segment is split over the ranks, a j-side
    segment belongs to the first / last rank; sums and the per-face shear stresses must be those of the single rank."""
    import torch.distributed as dist
    from cfd_amd import solver as S
    from cfd_amd.bc import BoundaryGeometry
    from cfd_amd.post_processing import WallFriction, friction_coefficient_device
    from cfd_amd.slab import slab_device
    work = out_dir / f"rank{rank}"
    cfg = build_case(name, work)
    mesh = S.load_mesh(cfg["mesh"])
    ctl = S.StepControl(cfg)
    fields = S.initialize_fields(mesh, cfg["initial"])
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    fields.setdefault("nu_t", np.zeros((ni + 2, nj + 2)))
    dev = slab_device(ni, nj, 0, min_slab_rows=int(os.environ.get("SLAB_MIN_ROWS", "16")))
    dev.upload_mesh(mesh)
    S.configure_device(dev, cfg, BoundaryGeometry.from_nodes(mesh["node"]), cfg["physics"]["nu"], ctl.turb_on)
    dev.upload_fields(fields)
    dev.apply_flow_bc()
    dt = dev.compute_dt(0.5, 0.5, 0.0, 1.0e-3)[0]
    res = [dev.simple_iter(dt, 0.7, 0.7, 0.3) for _ in range(2)]
    out = {}
    for side, rng in (("i_min", None), ("i_max", (5, nj - 4)), ("j_min", None), ("j_max", (3, ni - 2))):
        wall = WallFriction(mesh, side, rng)
        got = friction_coefficient_device(dev, wall, nu=cfg["physics"]["nu"], want_tau=True)
        out[f"{side}_cf"], out[f"{side}_tau"], out[f"{side}_tau_avg"] = got["cf"], got["tau"], got["tau_avg"]
        # the scalars alone (no tau array: no extra collective)
        again = friction_coefficient_device(dev, wall, nu=cfg["physics"]["nu"])
        assert again["cf"] == got["cf"]
    # the diagnostic leaves the step's look-ahead alone: the next iteration is the one a run without it takes
    res.append(dev.simple_iter(dt, 0.7, 0.7, 0.3))
    dev.close()
    if rank == 0:
        np.savez(out_dir / "result.npz", res=np.array(res), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    name, out_dir, mode = sys.argv[1], Path(sys.argv[2]), sys.argv[3]
    backend, transport = {"gloo": ("gloo", "torch"), "rccl": ("gloo", "rccl"), "nccl": ("nccl", None),
                          "nccl-torch": ("nccl", "torch"), "nccl-fallback": ("nccl", None)}[mode]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    work = out_dir / f"rank{rank}"
    work.mkdir(parents=True, exist_ok=True)
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        if mode != "gloo":
            from cfd_amd.slab import rehearsal_env
            rehearsal_env(rank)
        if transport:
            os.environ["CFD_SLAB_TRANSPORT"] = transport
        else:
            os.environ.pop("CFD_SLAB_TRANSPORT", None)
        kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        if mode == "nccl-fallback" and rank == world - 1:
            # the library cannot reach RCCL on the last rank: every rank must agree on the torch.distributed callbacks
            import cfd_amd.slab
            cfd_amd.slab._rccl_probe = lambda dev: "simulated: RCCL not loadable"
    if name == "mesh_from_rect":
        return mesh_from_rect_check(out_dir, rank, world)
    if name.startswith("friction:"):
        return friction_check(name.split(":", 1)[1], out_dir, rank, world)
    if name.startswith("kernels:"):
        return kernel_check(name.split(":", 1)[1], out_dir, rank, world)
    from cfd_amd import solver as hip_solver
    cfg = build_case(name, work)
    cfg["solver"]["device"] = 0
    fields, step, time = hip_solver.run_case(cfg)
    if rank == 0:
        log = np.loadtxt(work / "postproc" / f"{cfg['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
        np.savez(out_dir / "result.npz", step=step, time=time, log=log, correct_flux=hip_solver.LAST_RUN["kernel_counts"]["correct_flux"],
                 spec_iters=hip_solver.LAST_RUN["kernel_counts"]["speculative_iterations"],
                 float_levels=hip_solver.LAST_RUN["kernel_counts"]["mg_float_levels"],
                 mesh_form=hip_solver.LAST_RUN["kernel_counts"]["mesh_form"],
                 **{f"field_{k}": v for k, v in fields.items()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
