"""debug: one V-cycle on a slab run vs fused/unfused"""
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
    from cfd_amd.slab import slab_device, slab_of, gather_slabs
    cfg = W.build_case(name, work)
    mesh = S.load_mesh(cfg["mesh"])
    ctl = S.StepControl(cfg)
    fields = S.initialize_fields(mesh, cfg["initial"])
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    fields.setdefault("nu_t", np.zeros((ni + 2, nj + 2)))
    rng = np.random.default_rng(3)
    r = rng.standard_normal((ni, nj))
    out = {}
    for fused in (0, 1):
        dev = slab_device(ni, nj, 0, min_slab_rows=16)
        dev.upload_mesh(mesh)
        cfg["solver"]["multigrid"] = {"fused": bool(fused)}
        S.configure_device(dev, cfg, BoundaryGeometry.from_nodes(mesh["node"]), cfg["physics"]["nu"], ctl.turb_on)
        dev.upload_fields(fields)
        dev.apply_flow_bc()
        dev.momentum_diagonal(1e-3)
        z = dev.vcycle(slab_of(r, dev.j_offset, dev.nj, L.KIND_CELL))
        if world > 1:
            z = gather_slabs(z, L.KIND_CELL, dev.nj, world)
        out[f"z{fused}"] = z
        dev.close()
    if rank == 0:
        np.savez(out_dir / "vc.npz", **out)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
main()
