#!/usr/bin/env python3
"""bench.py -- outer SIMPLE step throughput on MI355X (contract: see task / DESIGN.md).

One "step" = one outer SIMPLE step (exactly `max_iter` inner iterations, continuity_tol = 0)
of the synthetic periodic channel of SURVEY.md 8(d) on an N x N cell mesh, all planes
resident in HBM before the timed region.  metric = cell-updates/s = Ni*Nj*n_gpus / step time.

    python bench.py --gpus 1 --steps 20 --warmup 5            (default size 8192)
    python bench.py --gpus N ...                              (starts N ranks itself, one per GPU, relays rank 0's line)
    torchrun ... bench.py --gpus N ...                        (ranks from the launcher's environment)

The JSON line also carries
  roofline     : the probed hot kernel (instantiations of one kernel grouped) with the largest share of device time,
                 found live from HIP-event pairs around EVERY launch of every candidate inside the timed steps;
                 achieved = sum of the algorithmic bytes of those launches / sum of their durations (the bytes differ
                 per launch for some kernels: the first two PCG direction steps of a solve skip planes, the first
                 Chebyshev launch has no x_{k-1}; the library reports them with the times), peak 8.0 TB/s.  --probe N
                 forces a kernel.  All candidates are reported in roofline_other_kernels (same timed steps).
                 traffic = HBM bytes per launch from the committed PMC passes of this command, quoted only for the
                 single-GPU default size they were measured on.
  N > 1        : the headline is BASELINE.json's configuration, the --size x --size channel split into N j-slabs
                 ("scaling": "strong"); the weak-scaling variant (one --size x --size slab per GPU of a channel N
                 times as high) is timed afterwards and reported as weak_scaling_leg (--weak makes it the headline).
  variants     : N = 1, after the timed region: ms per step of the wall-graded channel, the half annulus and the k-omega
                 channel at the same size (three steps each, --variants-budget seconds; --no-variants skips them).
  cpu_baseline : the CPU oracle with the reference's solver mix (ILU-BiCGStab momentum, factor-once LU pressure)
                 timed on this host at 256^2, 512^2 and 1024^2 (--cpu-size), bounded by --cpu-budget seconds.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per cell and launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this command at
# 8192^2 (tools/pmc_summary.py: FETCH_SIZE x 2 + WRITE_SIZE, fabric side, Infinity-Cache hits included)
PMC_TRAFFIC_BYTES_PER_CELL_8192 = {}       # filled from the newest profiles/r*_pmc8192_bytes_per_cell.csv of the uniform laminar run
PMC_KERNEL_OF_PROBE = {0: "k_cg_pq", 8: "k_chebk<2,4", 9: "k_chebk<1,", 7: "k_mg_up_k<true", 1: "k_mg_down_k<true,3,false",
                       10: "k_mg_down_k<true,3,true", 2: "k_correct_flux<"}


def load_pmc_traffic():
    """bytes per cell and launch of the probed kernels from the newest committed PMC summary (tools/pmc_summary.py)."""
    # the uniform laminar passes only: k-omega / graded / annulus passes carry their workload in the file name
    files = sorted(f for f in (REPO / "profiles").glob("r[0-9]*_pmc8192_bytes_per_cell.csv")
                   if not any(w in f.name for w in ("komega", "graded", "annulus")))
    if not files:
        return None, {}
    # a probe id may cover several instantiations (k_cg_pq's first / later steps, k_chebk<1,4> and <1,6>): the mean over
    # the rows that belong to it, weighted by the launches the counter pass saw
    acc = {}
    for line in files[-1].read_text().splitlines():
        if line.startswith("#") or not line.strip():
            continue
        cols = line.split(";")
        for pid, key in PMC_KERNEL_OF_PROBE.items():
            if cols[0].replace(" ", "").startswith(key.replace(" ", "")):
                try:
                    n = float(cols[8])
                    tot, cnt = acc.get(pid, (0.0, 0.0))
                    acc[pid] = (tot + float(cols[1]) * n, cnt + n)
                except (IndexError, ValueError):
                    pass
    out = {pid: tot / cnt for pid, (tot, cnt) in acc.items() if cnt > 0}
    return files[-1].name, out


def setup_annulus(nr, nt, max_iter, tol, device, tuning=None):
    """Curvilinear variant (--mesh annulus): nr x nt half annulus around a cylinder, metrics computed on the device
    from the node grid (cfd_mesh_from_nodes), single GPU."""
    from cfd_amd import _lib as L
    from cfd_amd.bc import BoundaryGeometry
    from cfd_amd.device import Device
    from cfd_amd.mesh import generate_annulus_nodes
    from cfd_amd.solver import StepControl, configure_device
    from cfd_amd.workloads import half_annulus_config

    cfg, mp = half_annulus_config(nr, nt, max_iter, tol)
    if tuning:
        cfg["solver"]["tuning"] = dict(tuning)
    node = generate_annulus_nodes(**mp)
    dev = Device(nr, nt, device)
    dev.mesh_from_nodes(node)
    ctl = StepControl(cfg)
    configure_device(dev, cfg, BoundaryGeometry.from_nodes(node), cfg["physics"]["nu"], ctl.turb_on)
    del node
    f = np.zeros((nr + 2, nt + 2))
    f[1:-1, 1:-1] = 1.0
    dev.upload(L.F_U, f)
    f[:] = 0.0
    for fid in (L.F_V, L.F_P, L.F_NUT):
        dev.upload(fid, f)
    return dev, ctl, cfg


def setup_device(n, turbulent, max_iter, tol, device, world=1, transport=None, tuning=None, lookahead=True, graded=False,
                 private=False):
    """One n x n slab per GPU of an n x (n * world) channel of height `world` (same cells, same cell
    size on every rank: weak scaling); world = 1 is the single-GPU workload of SURVEY.md 8(d).
    --strong calls this with world = 1 under an initialised process group: the n x n channel itself is
    split into one n x (n / N) slab per GPU."""
    from cfd_amd import _lib as L
    from cfd_amd.bc import BoundaryGeometry
    from cfd_amd.mesh import rect_node_axes
    from cfd_amd.slab import slab_device
    from cfd_amd.solver import StepControl, configure_device
    from cfd_amd.workloads import channel_forced_config, perturbed_parabola_rows

    ratio_j = 1.0
    if graded:
        from cfd_amd.workloads import wall_graded_ratio
        ratio_j = wall_graded_ratio(n * world, 20.0)
    cfg, mp = channel_forced_config(n, turbulent, max_iter, tol, nj=n * world, length_j=float(world), ratio_j=ratio_j)
    if tuning:
        cfg["solver"]["tuning"] = dict(tuning)
    if not lookahead:
        cfg["solver"]["flux_kernels"] = {"marching": True, "lookahead": False}
    x, y = rect_node_axes(**mp)
    if private:         # a context of this rank alone, whatever process group exists (the settle phase of main())
        from cfd_amd.device import Device
        dev = Device(n, n * world, device)
        dev.j_offset, dev.nj_global, dev.rank, dev.nranks = 0, n * world, 0, 1
    else:
        dev = slab_device(n, n * world, device, transport=transport)
    dev.mesh_from_rect(x, y)
    ctl = StepControl(cfg)
    configure_device(dev, cfg, BoundaryGeometry.from_rect_axes(x, y), cfg["physics"]["nu"], ctl.turb_on)
    shape = (n + 2, dev.nj + 2)
    f = np.zeros(shape)
    f[1:-1, 1:-1] = perturbed_parabola_rows(n, y, seed=1234, j_offset=dev.j_offset, nj_local=dev.nj)
    dev.upload(L.F_U, f)
    dev.exchange(L.F_U)            # ghost rows at the cuts = the neighbours' rows
    f[:] = 0.0
    dev.upload(L.F_V, f)
    dev.upload(L.F_P, f)
    dev.upload(L.F_NUT, f)
    if turbulent:
        f[:, :] = 1.0e-4           # uniform, ghost rows included (cut ghosts = neighbour values)
        f[[0, -1], :] = 0.0
        if dev.rank == 0:
            f[:, 0] = 0.0
        if dev.rank == dev.nranks - 1:
            f[:, -1] = 0.0
        dev.upload(L.F_K, f)
        f[f != 0.0] = 10.0
        dev.upload(L.F_OMEGA, f)
    del f
    return dev, ctl, cfg


def cpu_baselines(sizes, steps, max_iter, budget_s):
    """cpu_baseline at several sizes (256^2, 512^2, 1024^2), stopping before a size whose estimated
    cost (x8 per doubling of the side, measured 0.8 / 6.3 / 43 s per step) would exceed the time budget."""
    out, spent, last = [], 0.0, None
    for n_cpu, st in zip(sizes, steps):
        if last is not None:
            est = last[1] * (n_cpu / last[0]) ** 3 * (st + 1.5)
            if spent + est > budget_s:
                out.append({"size": n_cpu, "skipped": f"estimated {est:.0f} s would exceed --cpu-budget {budget_s:.0f} s"})
                continue
        t0 = time.perf_counter()
        r = cpu_baseline(n_cpu, st, max_iter)
        spent += time.perf_counter() - t0
        last = (n_cpu, n_cpu * n_cpu / r["value"])
        r["size"] = n_cpu
        out.append(r)
    return out


def cpu_baseline(n_cpu, steps, max_iter):
    """Oracle with the reference's solver mix, on the host cores (1 effective thread)."""
    from cfd_amd.workloads import channel_forced_config
    from oracle import cfd_oracle as orc
    cfg, mp = channel_forced_config(n_cpu, False, max_iter, 1e-8)
    mesh = orc.rect_mesh(**mp)
    y = mesh["cell_center"][:, :, 1]
    rng = np.random.default_rng(1234)
    cfg["initial"]["u"] = (1.0 - (2.0 * y - 1.0) ** 2) + 1.0e-3 * rng.standard_normal(y.shape)
    stamps = []

    def tick(fields, rec):
        stamps.append(time.perf_counter())

    cfg["time"]["max_steps"] = steps + 1
    t0 = time.perf_counter()
    orc.run_case(cfg, mesh, solver_mode="reference", freeze_pressure=True, on_step=tick)
    per_step = (stamps[-1] - stamps[0]) / max(1, len(stamps) - 1)   # first step (LU set-up) excluded
    return {"value": n_cpu * n_cpu / per_step, "unit": "cell-updates/s", "cores": 1,
            "kind": "port",
            "sample": f"{n_cpu}x{n_cpu} laminar channel, {steps} timed outer steps x {max_iter} SIMPLE iterations "
                      f"after 1 warm-up, rtol 1e-8, {per_step:.3f} s/step, host has {os.cpu_count()} cores, "
                      f"numpy/SuperLU single-threaded; total {time.perf_counter() - t0:.1f} s"}


KERNEL_NAMES = {
    0: "pressure PCG direction step: p update + SpMV + (p,q) (k_cg_pq; x is summed from the kept directions when the solve "
       "ends, k_cg_xsum)",
    1: "MG pre-smoother, finest level: 3 fused Jacobi sweeps + residual + restrict (k_mg_down_k<true,3>)",
    2: "velocity / pressure correction + ghost cells + momentum RHS + continuity sum (+ next step's time-step minima in "
       "laminar runs, + grad(2/3 k) in k-omega runs) in one marching pass (k_correct_flux, 176-184 B/cell, 32 B less on "
       "axis-aligned meshes, whose normal planes are not read; the first iteration of a run and contexts the fused pass does "
       "not cover run the Rhie-Chow kernel k_flux_march<2>, 120-128 B/cell)",
    8: "momentum Chebyshev, four fused iterations of u and v per launch (k_chebk<2,4>)",
    9: "momentum Chebyshev, four or six fused iterations of one right-hand side per launch (k_chebk<1,4> / k_chebk<1,6>: "
       "the same 64 B per cell buy six iterations where that saves a launch, at 12 instead of 8 halo lanes per wave)",
    10: "MG pre-smoother with the PCG residual update folded in (k_mg_down_k<true,3,true>)",
    7: "MG post-smoother, finest level: prolong + 3 fused Jacobi sweeps + (r,z) (k_mg_up_k<true,3>)"}
CANDIDATES = (0, 8, 9, 7, 1, 10, 2)
# ids that are instantiations of ONE kernel are one entry when the `roofline` kernel is chosen (the per-instantiation
# split is reported beside it): k_cg_pq's three forms already share id 0
GROUPS = {"k_chebk": (8, 9), "k_cg_pq": (0,), "k_mg_up_k": (7,), "k_mg_down_k": (1, 10), "k_correct_flux": (2,)}
GROUP_NAMES = {
    "k_chebk": "momentum Chebyshev, four fused iterations per launch (k_chebk<2,4> on u and v; k_chebk<1,4> / <1,6> once "
               "one of them has converged)",
    "k_cg_pq": KERNEL_NAMES[0],
    "k_mg_up_k": KERNEL_NAMES[7],
    "k_mg_down_k": "MG pre-smoother, finest level: 3 fused Jacobi sweeps + residual + restrict, with / without the PCG "
                   "residual update folded in (k_mg_down_k<true,3,*>)",
    "k_correct_flux": KERNEL_NAMES[2]}
# probed only for the byte / time accounting of the pressure solve: PCG residual update, last x update, pressure RHS
PRESSURE_IDS = (0, 1, 10, 7, 11, 12, 13)


def merge(entries):
    """(launches, mean us, mean bytes per cell) of several probe ids taken together."""
    cnt = sum(e[0] for e in entries)
    if not cnt:
        return (0, 0.0, 0.0)
    return (cnt, sum(e[0] * e[1] for e in entries) / cnt, sum(e[0] * e[2] for e in entries) / cnt)


def second_leg(args, n, world, local, barrier, torch, dist, strong):
    """N > 1, after the timed region: the other scaling mode, timed like the headline (barrier + synchronize on both
    sides, max over ranks).  Any (collective, hence symmetric) failure is reported in the JSON line instead of costing
    the headline."""
    from cfd_amd.solver import advance_one_step
    try:
        sdev, sctl, _ = setup_device(n, args.turbulent, args.max_iter, args.tol, local, 1 if strong else world, args.transport)
        sstep = 0
        for _ in range(args.warmup):
            advance_one_step(sdev, sctl, sstep)
            sstep += 1
        sdev.synchronize()
        s0 = sdev.comm_counts()
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            advance_one_step(sdev, sctl, sstep)
            sstep += 1
        sdev.synchronize()
        barrier()
        s_elapsed = time.perf_counter() - ts
        t = torch.tensor([s_elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_elapsed = float(t.item())
        sst = sdev.stats()
        cells = n * n if strong else n * n * world
        leg = {"workload": (f"the {n}x{n} channel as {world} j-slabs of {n}x{n // world} cells" if strong else
                            f"one {n}x{n} slab per GPU of a {n}x{n * world} channel of height {world}"),
               "scaling": "strong" if strong else "weak",
               "ms_per_step": 1e3 * s_elapsed / args.steps, "value": cells / (s_elapsed / args.steps),
               "unit": "cell-updates/s", "steps": args.steps, "warmup": args.warmup,
               "pressure_iters_last_step": sst["p_iters"], "momentum_iters_last_step": sst["mom_iters"],
               "collectives_per_step": dict(zip(("halo_exchanges", "all_reduces", "all_gathers"),
                                                [(a - b) / args.steps for a, b in zip(sdev.comm_counts(), s0)]))}
        sdev.close()
        return leg
    except Exception as exc:      # noqa: BLE001 - reported, not fatal
        return {"error": f"{type(exc).__name__}: {exc}"}


def variant_legs(args, local, budget_s, steps=5, warmup=8):
    """Outside the timed region, single GPU: ms per outer step of the OTHER workloads of the same size -- the wall-graded
    channel and the half annulus (stretched meshes: the reference's own cases are all wall-graded; pressure solve through
    the semi-coarsened line multigrid) and the k-omega channel -- so that the driver's line carries them.  Bounded: a
    variant is skipped once the budget is spent."""
    from cfd_amd.solver import advance_one_step
    n = args.size
    todo = (("graded", f"{n}x{n} periodic channel, wall-graded mesh (max / min dy = 20), laminar",
             lambda: setup_device(n, False, args.max_iter, args.tol, local, graded=True)),
            ("annulus", f"{n}x{n // 2} half annulus around a cylinder (r 0.5 .. 20), laminar",
             lambda: setup_annulus(n, n // 2, args.max_iter, args.tol, local)),
            ("komega", f"{n}x{n} periodic channel, uniform mesh, k-omega",
             lambda: setup_device(n, True, args.max_iter, args.tol, local)))
    out, t_start = [], time.perf_counter()
    for name, text, make in todo:
        if time.perf_counter() - t_start > budget_s:
            out.append({"variant": name, "workload": text, "skipped": "time budget of the variants leg spent"})
            continue
        dev, ctl, _ = make()
        try:
            step, its = 0, []
            for _ in range(warmup):
                advance_one_step(dev, ctl, step)
                step += 1
            dev.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                advance_one_step(dev, ctl, step)
                its.append(dev.stats()["p_iters"])
                step += 1
            dev.synchronize()
            dt = (time.perf_counter() - t0) / steps
            cells = n * (n // 2) if name == "annulus" else n * n
            out.append({"variant": name, "workload": text, "ms_per_step": 1e3 * dt, "cell_updates_per_s": cells / dt, "steps": steps,
                        "warmup": warmup, "mean_pressure_iters": float(np.mean(its)),
                        "mg_semi_levels": dev.kernel_counts()["mg_semi_levels"]})
        finally:
            dev.close()
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N child processes of this script, one rank each, with the
    rendezvous variables torch.distributed.run would set (127.0.0.1, a free port), wait for them and relay rank 0's
    JSON line as this process's only stdout line.  Runs BEFORE torch or the HIP library is imported here: the parent
    never touches the GPU and never replaces itself (no exec).  A rank that fails takes the others with it (exact
    PIDs) and the parent leaves with its code."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "CFD_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", str(Path(__file__).resolve())] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n_ranks))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code
                print(f"bench.py: rank {r} of {n_ranks} exited with code {code}", file=sys.stderr, flush=True)
                break
        time.sleep(0.05)
    for r in live:               # a failed rank leaves the others inside a collective: end them
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    reader.join(timeout=10)
    result = None
    for ln in lines:
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            result = ln                      # the last JSON object rank 0 printed
        elif ln:
            print(ln, file=sys.stderr)
    if rc == 0 and result is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    if rc == 0:
        if json.loads(result).get("n_gpus") != n_ranks:
            print(f"bench.py: rank 0 reported n_gpus = {json.loads(result).get('n_gpus')}, expected {n_ranks}",
                  file=sys.stderr, flush=True)
            return 1
        print(result, flush=True)
    return rc if rc > 0 else (1 if rc else 0)


def rehearse_launch(world, rank, backend):
    """--rehearse-launch: the launch / rendezvous / relay path alone, on the CPU (tests/test_bench_launch.py): every rank
    joins a gloo group, the ranks are counted by an all-reduce and rank 0 prints a line of the bench's shape."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "cell-updates/s per outer step", "value": None, "n_gpus": world, "launch_rehearsal": True,
                          "ranks_seen_by_all_reduce": seen,
                          "self_launched": os.environ.get("CFD_BENCH_SELF_LAUNCHED") == "1"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the workload's solver schedule settles around outer step 11 (the perturbed start needs one more four-iteration
    # momentum launch per solve until then: 43.4 against 42.1 ms per step), so the default warm-up runs past it
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--size", type=int, default=8192, help="cells per side of the channel (strong scaling: of the whole "
                    "channel, split over the GPUs; --weak: per GPU)")
    ap.add_argument("--turbulent", action="store_true", help="k-omega variant of the workload")
    ap.add_argument("--mesh", default="uniform", choices=("uniform", "graded", "annulus"),
                    help="SURVEY.md 8(d) mesh variants: uniform channel (BASELINE.json's configuration), wall-graded "
                         "channel (max / min dy = 20), --size x --size/2 half annulus (r 0.5 .. 20, single GPU)")
    ap.add_argument("--max-iter", type=int, default=2)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--cpu-size", type=str, default="256,512,1024", help="sizes of the CPU baseline legs")
    ap.add_argument("--cpu-steps", type=str, default="8,2,1", help="timed outer steps per CPU size (after one warm-up)")
    ap.add_argument("--cpu-budget", type=float, default=170.0, help="seconds the CPU baseline may take in all; sizes "
                    "whose estimated cost would exceed it are skipped and say so")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--settle", type=float, default=6.0, help="seconds of untimed GPU work on a separate small context before "
                    "the warm-up steps (the first GPU process on a freshly started machine stalls once for ~40 ms a "
                    "few seconds after it initialised the GPU; see DESIGN.md 5)")
    ap.add_argument("--probe", type=int, default=-1, help="kernel id of the live roofline probe (-1: the candidate with the "
                    "largest share of device time; 0 = PCG direction step, 8 / 9 = momentum Chebyshev with two / one "
                    "right-hand sides, 7 / 1 / 10 = multigrid up / down / down with residual update, 2 = momentum RHS)")
    ap.add_argument("--mg-sweeps", type=int, default=None, help="pre = post smoothing sweeps (2 or 3)")
    ap.add_argument("--mg-omegas", type=str, default=None, help="comma-separated damping factors, used cyclically")
    ap.add_argument("--tuning", type=str, default=None, help="name=value,... passed to cfd_set_option (A/B measurements)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--transport", default=None, choices=("rccl", "torch"),
                    help="slab collectives: rccl = enqueued by the library itself (default under the nccl backend), "
                         "torch = torch.distributed callbacks (default under gloo, host-staged)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: headline = one --size x --size slab per GPU (weak scaling); default: BASELINE.json's "
                         "configuration, the --size x --size channel split over the N GPUs (strong scaling)")
    ap.add_argument("--strong", action="store_true", help="(default for N > 1; kept for older command lines)")
    ap.add_argument("--no-second-leg", action="store_true",
                    help="N > 1: skip the leg in the other scaling mode that follows the timed region")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the untimed leg that reports ms per step of the graded / annulus / k-omega workloads (N = 1)")
    ap.add_argument("--variants-budget", type=float, default=100.0, help="seconds the variants leg may take")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="rehearsal only: all ranks use GPU 0 (with --backend gloo), numbers are not a measurement")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="test aid: run only the launch / rendezvous / relay path (gloo, no GPU) and print a stub line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N > 1 without a launcher: this process becomes the launcher (before torch / HIP are imported)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher's WORLD_SIZE and --gpus disagree: the line would claim something that did not run
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks")
    if args.rehearse_launch:
        return rehearse_launch(world, rank, args.backend)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    if not args.shared_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible "
                         f"(--shared-gpu rehearses several ranks on one GPU; its numbers are not a measurement)")
    if args.shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if args.shared_gpu:
            # functional rehearsal of the RCCL transports on a one-GPU box (loopback sockets, not a measurement)
            from cfd_amd.slab import rehearsal_env
            rehearsal_env(rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step

    n = args.size
    strong = world > 1 and not args.weak
    if strong and (n % world or (n // world) % 2 or n // world < 4):
        raise SystemExit(f"--size {n} cannot be split into {world} even j-slabs")
    tuning = None
    if args.tuning:
        tuning = {k: float(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))}
    if args.mesh == "annulus":
        if world > 1 or args.turbulent:
            raise SystemExit("--mesh annulus is a single-GPU laminar workload (cfd_mesh_from_nodes is single-rank)")
        dev, ctl, cfg = setup_annulus(n, n // 2, args.max_iter, args.tol, local, tuning=tuning)
    else:
        dev, ctl, cfg = setup_device(n, args.turbulent, args.max_iter, args.tol, local, 1 if (strong or world == 1) else world,
                                     args.transport, tuning=tuning, graded=args.mesh == "graded")
    cells_total = n * n if (strong or world == 1) else world * n * n
    if args.mesh == "annulus":
        cells_total = n * (n // 2)
    cells_gpu = cells_total // world   # cells one launch on this GPU sweeps
    if args.mg_sweeps is not None or args.mg_omegas is not None:
        nu_s = args.mg_sweeps or 3
        dev.set_multigrid(nu_pre=nu_s, nu_post=nu_s,
                          omegas=[float(v) for v in args.mg_omegas.split(",")] if args.mg_omegas else None)
    if args.settle > 0.0:
        t_settle = time.perf_counter()
        sdev, sctl, _ = setup_device(1024, args.turbulent, args.max_iter, args.tol, local, private=True)
        sstep = 0
        while time.perf_counter() - t_settle < args.settle:
            advance_one_step(sdev, sctl, sstep)
            sstep += 1
        sdev.synchronize()
        sdev.close()
    step = 0
    iters = []
    # ---- the event-pair probe runs on EVERY hot kernel inside the timed region (cfd_probe(CFD_PROBE_ALL): two event
    # records per launch, about sixty launches per step); afterwards the kernel with the largest share of the probed
    # device time is the `roofline` kernel and the others are reported beside it, all from the timed steps themselves
    shares = {}
    probe_id = args.probe
    # the event pool exists and the stream has seen its first timed event records before the timed region starts (the first
    # record on a stream costs milliseconds once): the probe is switched on for the last warm-up step and reset after it
    dev.set_option("probe_capacity", min(16384, 160 * (args.steps + 1) * max(1, args.max_iter) // 2 + 64))
    for w in range(args.warmup):
        if w == args.warmup - 1:
            dev.probe(L.PROBE_ALL if probe_id < 0 else probe_id)
        info = advance_one_step(dev, ctl, step)
        step += 1
    dev.synchronize()
    dev.probe(L.PROBE_ALL if probe_id < 0 else probe_id)
    comm0 = dev.comm_counts()
    phase0 = dev.comm_counts_by_phase()
    host0 = dev.host_counters()
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        info = advance_one_step(dev, ctl, step)
        st = dev.stats()
        iters.append((st["mom_iters"], st["p_iters"], st["k_iters"]))
        step += 1
        step_marks.append(time.perf_counter())      # host clock, no synchronisation: the host is at most one read ahead
    dev.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_kernel = {}
    group = None
    pressure_probe = None
    if probe_id < 0:
        for pid in CANDIDATES:
            per_kernel[pid] = dev.probe_read(with_bytes=True, kernel=pid)
        pressure_probe = {pid: dev.probe_read(with_bytes=True, kernel=pid) for pid in PRESSURE_IDS}
        for name, ids in GROUPS.items():
            shares[name] = sum(per_kernel[pid][0] * per_kernel[pid][1] for pid in ids)
        group = max(shares, key=shares.get)
        n_probe, probe_us, probe_bpc = merge([per_kernel[pid] for pid in GROUPS[group]])
        probe_id = GROUPS[group][0]
    else:
        n_probe, probe_us, probe_bpc = dev.probe_read(with_bytes=True)
    dev.probe(-1)
    comm1, phase1 = dev.comm_counts(), dev.comm_counts_by_phase()
    host1 = dev.host_counters()
    # ---- outside the timed region: per-phase device times of two more steps (event pairs with a host
    # sync per phase, which is why they are not collected inside the timed steps) and a plain
    # device-to-device copy of one plane as the streaming reference figure of this GPU
    dev.enable_phase_timing(True)
    dev.phase_times(reset=True)
    extra_steps, p_its = 2, []
    for _ in range(extra_steps):
        advance_one_step(dev, ctl, step)
        p_its.append(dev.stats()["p_iters"])
        step += 1
    phases = dev.phase_times(reset=True)
    dev.enable_phase_timing(False)
    # the other kernels BASELINE.json's target names (momentum RHS / Rhie-Chow, MG smoother, PCG step, Chebyshev): one
    # more step each with the event-pair probe on that kernel
    others = {}
    for pid in CANDIDATES:
        if pid == probe_id and group is None:
            continue
        if group is not None and pid in GROUPS[group]:
            continue                     # the headline kernel itself (its instantiations are listed with it)
        if pid in per_kernel:            # timed inside the timed region together with the headline kernel
            others[pid] = per_kernel[pid]
            continue
        dev.probe(pid)
        advance_one_step(dev, ctl, step)
        step += 1
        dev.synchronize()
        others[pid] = dev.probe_read(with_bytes=True)
        dev.probe(-1)
    kcounts = dev.kernel_counts()
    # ---- outside the timed region, N > 1 only: the other scaling mode
    other_leg = None
    if world > 1 and not args.no_second_leg:
        if strong or (n % world == 0 and (n // world) % 2 == 0 and n // world >= 4):
            other_leg = second_leg(args, n, world, local, barrier, torch, dist, strong=not strong)
    copy_gbs = None
    if rank == 0:
        a = torch.empty(cells_gpu, dtype=torch.float64, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * 8.0 * cells_gpu / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
    ms_per_step = 1000.0 * elapsed / args.steps
    value = cells_total / (elapsed / args.steps)

    out = None
    if rank == 0:
        pmc_file, pmc = load_pmc_traffic()
        achieved = probe_bpc * cells_gpu / (probe_us * 1e-6) / 1e9 if n_probe else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": (sum(pmc[pid] * per_kernel[pid][0] for pid in GROUPS[group]) / max(1, n_probe) * cells_gpu
                                if group is not None and n == 8192 and world == 1 and not args.turbulent and args.mesh == "uniform"
                                and all(pid in pmc for pid in GROUPS[group]) else None),
                    "traffic_unit": f"bytes per launch (PMC, profiles/{pmc_file})" if pmc_file else None,
                    "kernel": GROUP_NAMES[group] if group is not None else KERNEL_NAMES.get(probe_id, str(probe_id)),
                    "instantiations": None if group is None or len(GROUPS[group]) < 2 else [
                        {"kernel": KERNEL_NAMES[pid], "launches_timed": per_kernel[pid][0], "avg_us": per_kernel[pid][1],
                         "bytes_per_cell": per_kernel[pid][2],
                         "frac": (per_kernel[pid][2] * cells_gpu / (per_kernel[pid][1] * 1e-6) / 1e9 / HBM_PEAK_GBS
                                  if per_kernel[pid][0] else None)} for pid in GROUPS[group]],
                    "avg_us": probe_us, "launches_timed": n_probe,
                    "bytes_per_cell": probe_bpc,
                    "bytes_note": "mean ALGORITHMIC bytes per cell of the timed launches, reported by the library per launch "
                                  "(PCG direction step: 40 / 56 / 64 B for the first / second / later steps of a solve)",
                    "share_of_probed_device_time": (shares[group] / sum(shares.values()) if shares else None)}
        it = np.array(iters)
        if world == 1:
            par = "1 GPU"
        elif strong:
            par = f"{world} j-slabs of {n}x{n // world} cells of the {n}x{n} channel (strong scaling)"
        else:
            par = f"{world} j-slabs of {n}x{n} cells of one {n}x{n * world} channel of height {world} (weak scaling)"
        out = {
            "metric": "cell-updates/s per outer step", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not args.shared_gpu else "synthetic (REHEARSAL: ranks share one GPU, not a measurement)",
            "config": {"workload": (f"synthetic {n}x{n // 2} half annulus around a cylinder (r 0.5 .. 20, laminar, inlet / "
                                    f"outlet / symmetry / wall), SIMPLE " if args.mesh == "annulus" else
                                    f"synthetic {n}x{n if (strong or world == 1) else n * world} periodic channel with "
                                    f"pressure-gradient forcing ({'k-omega' if args.turbulent else 'laminar'}"
                                    f"{', wall-graded mesh with max / min dy = 20' if args.mesh == 'graded' else ''}), SIMPLE ")
                                   + f"max_iter={args.max_iter}, continuity_tol=0, cfl=cfl_diff=0.5, "
                                   f"dt_max={cfg['time']['dt_max']:.3g}, rtol={args.tol:g}",
                       "precision_note": ("fp64 arithmetic, fp64 fields, fp64 Krylov recurrences and stopping tests; the "
                                          "multigrid PRECONDITIONER of the pressure solve stores its face coefficients, its "
                                          "finest-level pre-smoothed iterate, its result z and (one rank) the vectors of its "
                                          "first coarse level in fp32 (options mg_f32 = 1, mg_c32 = 1, default): storage only, "
                                          "the converged p' is that of the fp64 system"
                                          if "mg_f32=0" not in (args.tuning or "").replace(" ", "") else
                                          "fp64 arithmetic and storage throughout (mg_f32 = 0)"),
                       "cells_per_gpu": cells_gpu,
                       "launcher": ("bench.py --gpus N started the ranks itself" if os.environ.get("CFD_BENCH_SELF_LAUNCHED") == "1"
                                    else ("external launcher (WORLD_SIZE from the environment)" if world > 1 else "single process")),
                       "rccl_ranks": dev.rccl_ranks() if world > 1 else None,
                       "process_group": None if world == 1 else {"backend": args.backend, "world_size": dist.get_world_size()},
                       "parallelism": par + ("" if world == 1 else
                                             ", halo rows + all-reduce + coarse-level all-gather "
                                             + ("enqueued by the library over RCCL" if dev.transport == "rccl" else
                                                f"through torch.distributed ({args.backend}) callbacks")
                                             + ", multigrid levels %s slab-distributed, coarser ones replicated" % (
                                                 "0..%d" % (dev.level_info()[1] - 1))
                                             + (f" [{dev.transport_note}]" if getattr(dev, "transport_note", None) else "")),
                       "collectives_per_step": None if dev.comm is None else dict(zip(
                           ("halo_exchanges", "all_reduces", "all_gathers"),
                           [(a - b) / args.steps for a, b in zip(comm1, comm0)])),
                       "collectives_per_step_by_phase": None if dev.comm is None else {
                           ph: [(a - b) / args.steps for a, b in zip(phase1[ph], phase0[ph])] for ph in phase1},
                       # host side of the timed steps: kernel launches and blocking stream synchronisations of this rank
                       # (the probe's own event records are not counted; speculative iterations -- kernel_launch_counts --
                       # end with one read each)
                       "host_per_step": {"kernel_launches": (host1[0] - host0[0]) / args.steps,
                                         "stream_synchronisations": (host1[1] - host0[1]) / args.steps},
                       "inner_iterations": args.max_iter,
                       "mean_momentum_iters": float(it[:, 0].mean()), "mean_pressure_iters": float(it[:, 1].mean()),
                       "mean_komega_iters": float(it[:, 2].mean()) if args.turbulent else 0.0,
                       "last_continuity_l2": info["cont"], "last_dt": info["dt"],
                       "kernel_launch_counts": kcounts},
            "roofline": roofline,
            "roofline_other_kernels": [
                {"kernel": KERNEL_NAMES[pid], "bytes_per_cell": bpc, "avg_us": us, "launches_timed": cnt,
                 "achieved": bpc * cells_gpu / (us * 1e-6) / 1e9 if cnt else None, "unit": "GB/s",
                 "frac": bpc * cells_gpu / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if cnt else None,
                 "traffic": pmc[pid] * cells_gpu if (n == 8192 and world == 1 and not args.turbulent and pid in pmc) else None}
                for pid, (cnt, us, bpc) in others.items()],
        }
        # second BASELINE metric: achieved HBM GB/s of the pressure solve.  Algorithmic bytes = what the launches of the
        # timed steps actually moved (the library reports the bytes of every probed launch: direction steps 36 / 52 / 60 B,
        # pre-smoother 22 B plain and 38 B with the residual update folded in, post-smoother 26 B, k_cg_r / k_cg_xlast
        # 24 B only on the iterations that run them, pressure right-hand side 112 B) + the coarser levels of each V-cycle
        # ((down + up) bytes per coarse cell x the cells of levels 1.. = 1/3 of the finest level's).
        f32 = "mg_f32=0" not in (args.tuning or "").replace(" ", "")
        coarse_bytes_per_vcycle = ((26.0 + 30.0) if f32 else (34.0 + 42.0)) / 3.0
        if f32 and world == 1 and "mg_c32=0" not in (args.tuning or "").replace(" ", ""):
            # the first coarse level keeps b and x in single precision (mg_c32): 17 + 21 B per level-1 cell
            coarse_bytes_per_vcycle = (17.0 + 21.0) / 4.0 + (26.0 + 30.0) / 12.0
        n_solves = extra_steps * args.max_iter
        press_s = phases["press"] / n_solves
        mean_p_it = float(np.mean(p_its))
        if pressure_probe is not None and pressure_probe[7][0] > 0 and args.mesh == "uniform":
            timed_solves = args.steps * args.max_iter
            fine = sum(cnt * bpc for cnt, _, bpc in pressure_probe.values()) / timed_solves
            vcycles = pressure_probe[7][0] / timed_solves
            bytes_per_solve = fine + coarse_bytes_per_vcycle * vcycles
            mix = {"direction_steps": pressure_probe[0][0] / timed_solves,
                   "presmoother_plain": pressure_probe[1][0] / timed_solves,
                   "presmoother_with_residual_update": pressure_probe[10][0] / timed_solves,
                   "postsmoother": pressure_probe[7][0] / timed_solves,
                   "residual_update_passes": pressure_probe[11][0] / timed_solves,
                   "last_x_update": pressure_probe[12][0] / timed_solves,
                   "rhs_passes": pressure_probe[13][0] / timed_solves,
                   "coarse_level_bytes_per_cell_per_vcycle": coarse_bytes_per_vcycle}
            note = ("algorithmic bytes from the launch mix of the timed steps (per solve); time = device time of the pressure "
                    "phase (right-hand side kernel, solve, host residual checks) over the two steps that follow the timed region")
        elif kcounts.get("mg_semi_levels", 0) > 0:
            # semi-coarsened line multigrid (stretched meshes): per cell of a level and V-cycle the four zebra half
            # sweeps read b (4 x 8 B / 2), the other colour's x (3 x 8 / 2), the coarse correction (8 / 2), c_j in
            # single precision (3 x 8 / 2), the line factors (4 x F / 2: l, 1 / p, and z on periodic lines, 4 B each) and
            # write x (3 x 8 / 2) and the coarse right-hand side (8 / 2): 60 + 2 F; the levels halve, so x 2 per fine
            # cell.  Krylov side per iteration: direction step 64 B (z in double here), residual update 24 B.
            # With mg_f32 (default) the levels below the finest keep b, x, coarse right-hand side and coarse correction in
            # single precision: 30 + 2 F per cell there, 56 + 2 F on the finest level.
            fac = 12.0 if args.mesh == "graded" else 8.0
            vcycle = (56.0 + 2.0 * fac) + (30.0 + 2.0 * fac) if f32 else 2.0 * (60.0 + 2.0 * fac)
            # Iterations whose |r| nobody reads leave r = r - alpha q to the finest level's way down (option semi_resid):
            # q read and r written there, 16 B instead of k_cg_r's 24.  The right-hand side pass of an axis-aligned mesh
            # does not read the normal planes (80 B), of a separable one not the area planes either (64 B).
            deferred = kcounts.get("mg_deferred_resid", 0) / max(1, kcounts.get("vcycles", 1))
            # z goes from the way up to the direction step as a float plane (option semi_z32, default on with mg_f32)
            z32 = f32 and "semi_z32=0" not in (args.tuning or "")
            if z32:
                vcycle -= 4.0
            krylov = 64.0 - (4.0 if z32 else 0.0) + 24.0 - 8.0 * deferred
            rhs_pass = {0: 112.0, 1: 80.0, 2: 64.0}.get(kcounts.get("mesh_form", 0), 112.0)
            per_it = krylov + vcycle
            bytes_per_solve = per_it * mean_p_it + rhs_pass
            mix = {"semi_coarsened_levels": kcounts["mg_semi_levels"], "vcycle_bytes_per_fine_cell": vcycle,
                   "krylov_bytes_per_iteration": krylov, "residual_updates_in_the_way_down": deferred, "rhs_pass": rhs_pass}
            note = ("semi-coarsened line multigrid: algorithmic bytes of a zebra line V(1,1) cycle over all levels + direction step "
                    "+ residual update, x iterations + right-hand side; time = device time of the pressure phase over the two steps "
                    "that follow the timed region")
        else:
            # line-smoothed / curvilinear workloads and forced single-kernel probes: nominal V(3,3)-PCG bytes per iteration
            f8 = 8.0 if f32 else 0.0
            per_it = (64 - 0.5 * f8) + 24 + (34 - 1.5 * f8) + (42 - 2.0 * f8) + ((34 - f8) + (42 - f8)) / 3.0
            bytes_per_solve = per_it * mean_p_it + 112.0
            mix = None
            note = ("nominal bytes of a point-smoothed V(3,3)-PCG iteration x iterations + right-hand side (line-smoothed levels "
                    "move more: this UNDERSTATES the traffic of --mesh graded / annulus)")
        out["pressure_solve"] = {
            "ms_per_solve": 1e3 * press_s, "iterations": mean_p_it,
            "algorithmic_bytes_per_cell_per_solve": bytes_per_solve,
            "algorithmic_bytes_per_cell_per_iteration": bytes_per_solve / mean_p_it if mean_p_it else None,
            "launches_per_solve": mix,
            "achieved_GBps": bytes_per_solve * cells_gpu / press_s / 1e9 if press_s > 0 else None,
            "frac_of_hbm_peak": bytes_per_solve * cells_gpu / press_s / 1e9 / HBM_PEAK_GBS if press_s > 0 else None,
            "note": note}
        out["weak_scaling_leg" if strong else "strong_scaling_leg"] = other_leg
        out["settle_s"] = args.settle     # untimed work on a separate 1024^2 context before the warm-up steps (--settle)
        out["phase_ms_per_step"] = {k: 1e3 * v / extra_steps for k, v in phases.items()}
        out["step_ms_host_clock"] = [round(1e3 * (b_ - a_), 2) for a_, b_ in zip([t0] + step_marks[:-1], step_marks)]
        out["device_copy_GBps"] = copy_gbs
        if world == 1 and not args.no_cpu:
            sizes = [int(v) for v in args.cpu_size.split(",") if v]
            steps_l = [int(v) for v in args.cpu_steps.split(",") if v]
            steps_l += [steps_l[-1]] * (len(sizes) - len(steps_l))
            legs = cpu_baselines(sizes, steps_l, args.max_iter, args.cpu_budget)
            done = [leg for leg in legs if "value" in leg]
            head = dict(done[0]) if done else None      # the first size is the headline baseline, the others ride along
            if head is not None:
                head["other_sizes"] = legs[1:]
            out["cpu_baseline"] = head
        else:
            out["cpu_baseline"] = None
    dev.close()
    if world == 1 and not args.no_variants and args.mesh == "uniform" and not args.turbulent and n >= 64:
        out["variants"] = variant_legs(args, local, args.variants_budget)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
