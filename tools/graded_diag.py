#!/usr/bin/env python3
"""Pressure operator of the wall-graded bench workload at a small size: seam asymmetry, solver taken, level table.
python tools/graded_diag.py [n]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, graded=True)
for step in range(2):
    advance_one_step(dev, ctl, step)
dev.synchronize()
st = dev.stats()
print({k: st[k] for k in ("p_solver", "p_iters", "p_relres", "mom_iters")})
ci, cj, diag, _ = dev.pressure_system()
a = np.abs(ci[0, :] - ci[-1, :])
print("seam: sum |ci[0]-ci[Ni]| =", a.sum(), " sum |ci[0]| =", np.abs(ci[0, :]).sum(), " ratio", a.sum() / np.abs(ci[0, :]).sum())
jm = int(np.argmax(a))
print("worst row", jm, ci[0, jm], ci[-1, jm], " zero rows:", int((ci[0, :] == 0).sum()))
from cfd_amd import _lib as L
ap = dev.download(L.F_AP)           # ghosted [i][j]
ai = dev.download(L.F_AI, L.KIND_FACE_I)
aj = dev.download(L.F_AJ, L.KIND_FACE_J)
vol = dev.download_mesh_plane("vol")
d_i = dev.download_mesh_plane("d_i"); s_i = dev.download_mesh_plane("s_i"); wr = dev.download_mesh_plane("wr_i")
print("a_P row jm: ghost,0,1,..,Ni-1,ghost", ap[0, jm + 1], ap[1, jm + 1], ap[2, jm + 1], ap[-2, jm + 1], ap[-1, jm + 1])
print("a_P i-variation rel in row:", np.ptp(ap[1:-1, jm + 1]) / ap[1, jm + 1], " argmax", int(np.argmax(ap[1:-1, jm + 1])), int(np.argmin(ap[1:-1, jm + 1])))
print("a_i row:", ai[0, jm], ai[1, jm], ai[-2, jm], ai[-1, jm], " ptp", np.ptp(ai[:, jm]))
print("a_j :", aj[0, jm], aj[1, jm], aj[-1, jm], " ptp rel", np.ptp(aj[:, jm]) / aj[0, jm])
print("vol ptp rel", np.ptp(vol[:, jm]) / vol[0, jm], " d_i", d_i[0, jm], d_i[1, jm], d_i[-1, jm], " s_i ptp", np.ptp(s_i[:, jm]), " wr", wr[0, jm], wr[1, jm], wr[-1, jm])
s_j = dev.download_mesh_plane("s_j"); d_j = dev.download_mesh_plane("d_j"); nut = dev.download(L.F_NUT)
print("s_j ptp rel", np.ptp(s_j[:, jm]) / s_j[0, jm], " d_j ptp rel", np.ptp(d_j[:, jm]) / d_j[0, jm], " nut max", np.abs(nut).max())
print("d_j row:", d_j[0, jm], d_j[1, jm], d_j[2, jm], d_j[-1, jm], "argmax/min", int(np.argmax(d_j[:, jm])), int(np.argmin(d_j[:, jm])))
print("d_j[:, jm+1]:", d_j[0, jm + 1], d_j[-1, jm + 1], " vol row:", vol[0, jm], vol[1, jm], vol[-1, jm])
dev.close()
