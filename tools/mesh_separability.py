"""Which metric planes of the bench meshes are bitwise functions of one index?  (What the axis-aligned / separable forms
of the flux kernels rely on is checked on the device per mesh; this lists all twelve planes for the record.)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for graded in (False, True):
    dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, graded=graded)
    print("graded" if graded else "uniform", n)
    for name in dev.MESH_PLANES:
        a = dev.download_mesh_plane(name)        # [i][j]
        along_j = bool(np.all(a == a[:, :1]))    # function of i only
        along_i = bool(np.all(a == a[:1, :]))    # function of j only
        const = bool(np.all(a == a.flat[0]))
        ni_bad = int(np.sum(np.any(a != a[:, :1], axis=1)))
        print(f"  {name:5s} shape {a.shape}  f(i) only: {along_j}  f(j) only: {along_i}  constant: {const}  columns that vary along j: {ni_bad}")
    dev.close()
