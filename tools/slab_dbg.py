"""debug: step a slab case manually and print solver stats per inner iteration"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import slab_worker as W

def main():
    name, out_dir = sys.argv[1], Path(sys.argv[2])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    work = out_dir / f"rank{rank}"; work.mkdir(parents=True, exist_ok=True)
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfd_amd import solver as S, _lib as L
    from cfd_amd.bc import BoundaryGeometry
    from cfd_amd.slab import slab_device
    cfg = W.build_case(name, work)
    mesh = S.load_mesh(cfg["mesh"])
    ctl = S.StepControl(cfg)
    fields = S.initialize_fields(mesh, cfg["initial"])
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    fields.setdefault("nu_t", np.zeros((ni + 2, nj + 2)))
    dev = slab_device(ni, nj, 0, min_slab_rows=16)
    dev.upload_mesh(mesh)
    S.configure_device(dev, cfg, BoundaryGeometry.from_nodes(mesh["node"]), cfg["physics"]["nu"], ctl.turb_on)
    dev.upload_fields(fields)
    out = {}
    dev.apply_flow_bc()
    dt, a, b = dev.compute_dt(0.5, 0.5, 0.0, 1e-3)
    if rank == 0: print("dt", repr(dt), a, b)
    for it in range(2):
        res = dev.simple_iter(dt, 0.7, 0.7, 0.3)
        st = dev.stats()
        if rank == 0: print("iter", it, [repr(r) for r in res], st)
        for nm, fid in (("u", L.F_U), ("v", L.F_V), ("p", L.F_P), ("pc", L.F_PCORR), ("ap", L.F_AP), ("bu", L.F_BU), ("nut", L.F_NUT)):
            out[f"{nm}{it}"] = dev.download(fid, whole=True)
    if ctl.turb_on:
        dev.turbulence_step(dt)
        if rank == 0: print("turb", dev.stats())
        out["k"] = dev.download(L.F_K, whole=True); out["om"] = dev.download(L.F_OMEGA, whole=True)
    if rank == 0:
        np.savez(out_dir / "dbg.npz", **out)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
main()
