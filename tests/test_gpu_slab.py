"""j-slab decomposition on the GPU: N ranks, each with its own libcfd_hip
context holding one slab, must reproduce the single-rank run.  The box has one GPU, so the
ranks share it: most cases exchange through gloo with host staging (the halo / all-reduce /
all-gather logic and every kernel's cut handling are the ones the RCCL path uses), and the RCCL
transports themselves -- the library's native one and the torch.distributed nccl callbacks -- are
rehearsed on the shared device with one fake host id per rank (RCCL's loopback socket transport)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import mask_corners

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(case, world, out_dir, min_rows=16, mode="gloo"):
    out_dir.mkdir(parents=True, exist_ok=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", SLAB_MIN_ROWS=str(min_rows))
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank), LOCAL_RANK="0")
        log = open(out_dir / f"rank{rank}.log", "w")
        procs.append((subprocess.Popen([sys.executable, "-u", str(ROOT / "tests" / "slab_worker.py"), case, str(out_dir), mode],
                                       env=e, stdout=log, stderr=subprocess.STDOUT, cwd=str(ROOT)), log))
    failed = []
    for rank, (p, log) in enumerate(procs):
        try:
            rc = p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        log.close()
        if rc != 0:
            failed.append((rank, rc))
    if failed:
        for _, (p, _) in enumerate(procs):
            if p.poll() is None:
                p.kill()
        tail = "\n".join(f"--- rank {r} (exit {rc})\n" + (out_dir / f"rank{r}.log").read_text()[-3000:] for r, rc in failed)
        raise AssertionError(f"slab run failed: {tail}")
    return np.load(out_dir / "result.npz")


@pytest.mark.parametrize("case,world,min_rows", [
    ("channel_komega_48x64", 2, 16), ("channel_komega_48x64", 4, 16), ("channel_lam_64x128", 2, 16),
    ("channel_lam_64x128", 2, 512),      # production threshold: only the finest level is slab-distributed
    ("channel_lam_256x256", 2, 16), ("channel_lam_256x256", 4, 16),     # float vectors of the first coarse level across the cuts
    ("inlet_outlet_40x32", 2, 16), ("couette_piso_24x32", 2, 16), ("half_cylinder_komega_24x32", 2, 16),
    # Menter SST on slabs: caches, wall distance and cell-centre differences across the cuts
    ("channel_sst_48x64", 2, 16), ("channel_sst_48x64", 4, 16), ("half_cylinder_sst_24x32", 2, 16),
])
def test_slab_run_matches_single_rank(case, world, min_rows, tmp_path):
    """min_rows = 16 keeps the multigrid levels slab-distributed as deep as the mesh allows."""
    one = _run(case, 1, tmp_path / "w1")
    many = _run(case, world, tmp_path / f"w{world}", min_rows)
    assert int(one["step"]) == int(many["step"])
    assert abs(float(one["time"]) - float(many["time"])) <= 1e-12 * max(1.0, abs(float(one["time"])))
    unorm = max(1.0, np.linalg.norm(one["field_u"].ravel()))
    for key in [k for k in one.files if k.startswith("field_") and k != "field_p_corr"]:
        a, b = mask_corners(one[key]), mask_corners(many[key])
        err, den = np.linalg.norm((a - b).ravel()), np.linalg.norm(a.ravel())
        assert err <= 1e-10 * den + 1e-12 * unorm, (key, err, den)
    la, lb = one["log"], many["log"]
    assert la.shape == lb.shape
    assert np.array_equal(la[:, 0], lb[:, 0]) and np.array_equal(la[:, 2], lb[:, 2])     # steps, inner iterations
    assert np.all(np.abs(la[:, 3] - lb[:, 3]) <= 1e-6 * np.abs(la[:, 3]) + 1e-13)        # continuity history
    # the slabs close their iterations with the same fused correction / flux pass as the single rank (a cut is a segment
    # boundary to that kernel); the half cylinder's j-sides are periodic / not local, so it keeps the separate kernels
    assert (int(many["correct_flux"]) > 0) == (int(one["correct_flux"]) > 0), (int(one["correct_flux"]), int(many["correct_flux"]))
    if "channel" in case or "couette" in case or "inlet" in case:
        assert int(many["correct_flux"]) > 0
    # slab contexts replay their solver schedules like a single rank does (one closing read per iteration; the sums the
    # stopping tests look at are all-reduced before that read)
    assert (int(many["spec_iters"]) > 0) == (int(one["spec_iters"]) > 0), (int(one["spec_iters"]), int(many["spec_iters"]))
    # ... with the same storage of the coarse multigrid vectors and the same form of the flux kernels (the mesh check is
    # agreed over the ranks)
    assert int(many["float_levels"]) == int(one["float_levels"]) and int(many["mesh_form"]) == int(one["mesh_form"])
    if case == "channel_lam_256x256":
        assert int(many["float_levels"]) == 1 and int(many["mesh_form"]) == 2


@pytest.mark.parametrize("mode", ["rccl", "nccl-torch"])
def test_sst_slabs_over_rccl(mode, tmp_path):
    """The SST planes live outside the workspace: both RCCL transports must reach them."""
    case = "channel_sst_48x64"
    one = _run(case, 1, tmp_path / "w1")
    many = _run(case, 2, tmp_path / "w2", 16, mode)
    unorm = max(1.0, np.linalg.norm(one["field_u"].ravel()))
    keys = [k for k in one.files if k.startswith("field_") and k != "field_p_corr"]
    assert "field__F1" in keys and "field__d_wall" in keys
    for key in keys:
        a, b = mask_corners(one[key]), mask_corners(many[key])
        err, den = np.linalg.norm((a - b).ravel()), np.linalg.norm(a.ravel())
        assert err <= 1e-10 * den + 1e-12 * unorm, (key, err, den)


@pytest.mark.parametrize("mode,world", [("rccl", 2), ("rccl", 4), ("nccl", 2), ("nccl-torch", 2)])
def test_rccl_transports(mode, world, tmp_path):
    """The same k-omega slab run over the RCCL transports (native: ncclSend / ncclRecv groups, ncclAllReduce,
    ncclAllGather enqueued by the library; nccl-torch: the callbacks on device views): fields and logs of the
    single-rank run, and the fused kernels' iteration counts."""
    case = "channel_komega_48x64"
    one = _run(case, 1, tmp_path / "w1")
    many = _run(case, world, tmp_path / f"w{world}", 16, mode)
    assert int(one["step"]) == int(many["step"])
    unorm = max(1.0, np.linalg.norm(one["field_u"].ravel()))
    for key in [k for k in one.files if k.startswith("field_") and k != "field_p_corr"]:
        a, b = mask_corners(one[key]), mask_corners(many[key])
        err, den = np.linalg.norm((a - b).ravel()), np.linalg.norm(a.ravel())
        assert err <= 1e-10 * den + 1e-12 * unorm, (key, err, den)
    assert np.array_equal(one["log"][:, 2], many["log"][:, 2])
    k1 = _run("kernels:" + case, 1, tmp_path / "k1")
    kn = _run("kernels:" + case, world, tmp_path / f"k{world}", 16, mode)
    assert np.abs(k1["z"] - kn["z"]).max() <= 1e-12 * np.abs(k1["z"]).max()
    assert int(k1["mom_iters"]) == int(kn["mom_iters"]) and int(k1["p_iters"]) == int(kn["p_iters"])
    assert int(kn["native"]) == (mode != "nccl-torch") and int(k1["native"]) == 0
    assert np.all(kn["counts"] > 0) and not np.any(k1["counts"])


def test_default_transport_falls_back_to_the_callbacks_together(tmp_path):
    """Default transport under an nccl group is the library's own communicator; when one rank cannot build it, all
    ranks take the torch.distributed callbacks (agreed by an all-reduce before anybody enters an RCCL collective of
    the library) and the run is the single-rank run."""
    case = "kernels:channel_komega_48x64"
    k1 = _run(case, 1, tmp_path / "k1")
    kn = _run(case, 2, tmp_path / "k2", 16, "nccl-fallback")
    assert np.abs(k1["z"] - kn["z"]).max() <= 1e-12 * np.abs(k1["z"]).max()
    assert int(k1["mom_iters"]) == int(kn["mom_iters"]) and int(k1["p_iters"]) == int(kn["p_iters"])
    assert int(kn["native"]) == 0 and np.all(kn["counts"] > 0)


@pytest.mark.parametrize("mode,min_rows", [("rccl", 16), ("rccl", 512), ("gloo", 16)])
def test_world8_rehearsal_on_one_gpu(mode, min_rows, tmp_path):
    """Eight ranks (the world size of BASELINE's config) sharing this box's one GPU: a 64 x 256 channel as eight
    32-row slabs against the single-rank run.  A rehearsal of the code path (collectives of a world of 8, replicated
    coarse levels gathered from 8 slabs, V-cycle tail on every rank), not a measurement."""
    case = "channel_lam_64x256"
    one = _run(case, 1, tmp_path / "w1")
    many = _run(case, 8, tmp_path / "w8", min_rows, mode)
    assert int(one["step"]) == int(many["step"])
    unorm = max(1.0, np.linalg.norm(one["field_u"].ravel()))
    for key in [k for k in one.files if k.startswith("field_") and k != "field_p_corr"]:
        a, b = mask_corners(one[key]), mask_corners(many[key])
        err, den = np.linalg.norm((a - b).ravel()), np.linalg.norm(a.ravel())
        assert err <= 1e-10 * den + 1e-12 * unorm, (key, err, den)
    la, lb = one["log"], many["log"]
    assert la.shape == lb.shape and np.array_equal(la[:, 2], lb[:, 2])
    assert np.all(np.abs(la[:, 3] - lb[:, 3]) <= 1e-6 * np.abs(la[:, 3]) + 1e-13)


@pytest.mark.parametrize("world", [1, 3])
def test_slab_mesh_from_rect(world, tmp_path):
    """cfd_mesh_from_rect on slabs (halo node below, cut face row from the rank above) reproduces
    the slices of the host mesh contract on every rank."""
    res = _run("mesh_from_rect", world, tmp_path / f"m{world}")
    assert res["worst"].shape == (world,)
    assert np.all(res["worst"] <= 1e-12), res["worst"]


@pytest.mark.parametrize("case,world", [("channel_lam_64x128", 2), ("channel_komega_48x64", 2), ("channel_komega_48x64", 4),
                                        # two fused levels above the tail, the first coarse one with float vectors (mg_c32)
                                        ("channel_lam_256x256", 2), ("channel_lam_256x256", 4),
                                        ("channel_lam_512x512", 2)])       # ... the first two coarse levels
def test_fused_kernels_across_cuts(case, world, tmp_path):
    """The fused kernels read up to four halo rows of the neighbour slab: one V(3,3) cycle on a random
    right-hand side agrees with the single-rank cycle to rounding, and one SIMPLE iteration takes the same
    Chebyshev / MG-PCG iteration counts (a wrong halo would only show up there: the solves still converge)."""
    one = _run("kernels:" + case, 1, tmp_path / "w1")
    many = _run("kernels:" + case, world, tmp_path / f"w{world}")
    scale = np.abs(one["z"]).max()
    assert np.abs(one["z"] - many["z"]).max() <= 1e-12 * scale
    assert int(one["mom_iters"]) == int(many["mom_iters"]) and int(one["p_iters"]) == int(many["p_iters"])
    assert int(one["cheb"]) == int(many["cheb"]) == 1
    assert np.allclose(one["res"], many["res"], rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("case,world", [("inlet_outlet_40x32", 2), ("channel_komega_48x64", 4)])
def test_wall_friction_on_slabs(case, world, tmp_path):
    """cfd_wall_friction on slabs (the friction log of run_case on every rank): i-side segments cross the cuts, j-side
    segments sit on one rank; sub-ranges start and end inside a slab.  Sums to rounding (other order), the per-face
    shear stresses exactly (each face is computed by one rank, the others add zeros)."""
    one = _run("friction:" + case, 1, tmp_path / "w1")
    many = _run("friction:" + case, world, tmp_path / f"w{world}")
    for side in ("i_min", "i_max", "j_min", "j_max"):
        a, b = one[f"{side}_tau"], many[f"{side}_tau"]
        assert a.shape == b.shape and a.size > 0
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(a).max(), 1e-300), side      # fields equal to the solver tolerance
        # (the mean over an inlet plane cancels to rounding: measured against the largest face value)
        assert abs(float(one[f"{side}_tau_avg"]) - float(many[f"{side}_tau_avg"])) <= 1e-9 * np.abs(a).max() + 1e-300, side
    # continuity sums equal to rounding; the momentum residuals are what each run's solver stopped at (one rank: the one-launch
    # Chebyshev solve runs the count its bound predicts, the slabs' fused form leaves at the first iterate that meets 1e-13)
    assert np.allclose(one["res"][:, 0], many["res"][:, 0], rtol=1e-6, atol=1e-14)
    assert np.all(one["res"][:, 1:] <= 1e-12) and np.all(many["res"][:, 1:] <= 1e-12)
