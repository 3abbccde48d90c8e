#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh run into profiles/<tag>_*: kernel stats, FETCH_SIZE / WRITE_SIZE per
kernel and HBM bytes per cell and launch.  FETCH_SIZE counts one unit per 2 KiB (MI355X_MICROARCH.md, HBM; confirmed on this
code's access patterns by tools/stream_probe2.hip under --pmc, profiles/r2_fetch_size_calibration.txt); WRITE_SIZE is exact.

    python tools/pmc_summary.py r1g [cells] [line-factor bytes: 12 periodic lines, 8 otherwise]

Stretched meshes (semi-coarsened line multigrid): the k_semi_* instantiations serve several levels each, so they are
summed per V-cycle instead of listed per launch (appended to the bytes-per-cell file).
"""
import csv
import glob
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
cells = float(sys.argv[2]) if len(sys.argv) > 2 else 8192.0 * 8192.0
# algorithmic bytes per cell and launch
ALGO = {"k_chebk<2, 4, 2>": 98.7, "k_chebk<1, 4, 2>": 64, "k_chebk<2, 4, 1>": 98.7, "k_chebk<1, 4, 1>": 64,   # <2,4>: 96 first launch of a solve, 104 later
        "k_chebk<2, 4>": 104, "k_chebk<1, 4>": 64, "k_cg_pq": 58, "k_mg_up<true>": 42, "k_mg_down<true>": 34, "k_mg_up_k<true, 3>": 42, "k_mg_down_k<true, 3>": 34, "k_cg_r": 24, "k_cg_init": 24,
        "k_cg_xlast": 24, "k_flux_march<0, false>": 104, "k_flux_march<1, false>": 112, "k_flux_march<2, false>": 120,
        "k_continuity": 104, "k_pressure_rhs": 112, "k_momentum_rhs": 120, "k_correct": 136, "k_diffusion_ap": 88,
        "k_dt": 88, "k_rowsum_bound": 24, "k_pressure_coeffs": 64, "k_flux_march<3, false>": 88,
        "k_flux_march<0, false, false>": 104, "k_flux_march<1, false, false>": 112, "k_flux_march<2, false, false>": 120,
        "k_flux_march<2, false, true>": 128, "k_flux_march<3, false, false>": 88, "k_flux_march<2, true, false>": 128,
        "k_flux_march<4, false, false>": 144,
        "k_flux_march<2, true, true>": 128, "k_dt": 16,
        "k_mg_down_k<true, 3, false>": 34, "k_mg_down_k<true, 3, true>": 50,
        # k-omega variant
        "k_flux_march<2, true>": 128, "k_turb_rhs": 144, "k_flux_march<4, false>": 144, "k_komega_coeffs<true>": 136, "k_komega_coeffs<false>": 112, "k_clip": 16,
        "k_eddy_viscosity": 24}
KNOWN_READS = {"k_cg_r": 16.0, "k_cg_xlast": 16.0, "k_cg_init": 8.0}      # calibration kernels


def algo_of(k):
    """algorithmic bytes per cell of a kernel name, template arguments of the lockstep forms ignored"""
    if k in ALGO:
        return ALGO[k]
    m = re.match(r"k_chebk<(\d), [46]", k)
    if m:       # (the six-iteration form moves the bytes of the four-iteration one)
        return 98.7 if m.group(1) == "2" else 64
    m = re.match(r"k_cg_pq<\d+, (true|false), (true|false)(?:, (true|false))?(?:, (true|false))?>", k)
    if m:       # first / second / later direction step of a solve; z in single precision; x left alone (pcg_xsum)
        if m.group(4) == "true":
            return (40 if m.group(1) == "true" else 48) - (4 if m.group(3) == "true" else 0)
        return (40 if m.group(1) == "true" else (56 if m.group(2) == "true" else 64)) - (4 if m.group(3) == "true" else 0)
    if k.startswith("k_cg_xsum"):
        return 48       # five directions read, x written (a five-iteration solve)
    if k.startswith("k_cg_pq<"):
        return 58
    # fused V-cycle kernels: the last template argument says whether the face planes (-8 B) and the pre-smoothed
    # iterate (-4 B) are single precision
    m = re.match(r"k_mg_up_k<true, 3(?:, \d+(?:, (true|false)(?:, (true|false)(?:, (\d))?)?)?)?>", k)
    if m:       # ..., single-precision faces and x, single-precision z, coarse correction read as float (VT = 1)
        return 42 - (12 if m.group(1) == "true" else 0) - (4 if m.group(2) == "true" else 0) - (1 if m.group(3) == "1" else 0)
    m = re.match(r"k_mg_down_k<true, 3, (true|false)(?:, \d+(?:, (true|false)(?:, (\d))?)?)?>", k)
    if m:       # ..., coarse right-hand side written as float (VT = 1)
        return (50 if m.group(1) == "true" else 34) - (12 if m.group(2) == "true" else 0) - (1 if m.group(3) == "1" else 0)
    rna = {"true": 32, "1": 32, "2": 48}      # axis-aligned mesh: normals not read (-32); separable areas too (-48)
    m = re.match(r"k_correct_flux<(true|false), (true|false)(?:, \d+(?:, (true|false)(?:, (true|false|\d))?)?)?", k)
    if m:       # correction + ghost cells + momentum right-hand sides: 17 planes read, 5 written (+ k, + lmin)
        return 176 + (8 if m.group(1) == "true" else 0) + (8 if (m.group(2) == "true" and m.group(1) != "true") else 0) - rna.get(m.group(4), 0)
    m = re.match(r"k_flux_march<(\d), (true|false)(?:, (true|false)(?:, \d+(?:, (true|false|\d))?)?)?", k)
    if m:
        mode, kg, dt = m.group(1), m.group(2) == "true", m.group(3) == "true"
        return {"0": 104, "1": 112, "2": 120, "3": 88, "4": 144}[mode] + (8 if (mode == "2" and (kg or dt)) else 0) - (rna.get(m.group(4), 0) if mode != "3" else 0)
    return None


def short(name):
    m = re.search(r"cfd::(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def per_kernel(pattern, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


stats, calls = {}, {}
spath = glob.glob(f"gpurun_out/{tag}_stats/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(spath, f"profiles/{tag}_bench8192_kernel_stats.csv")
for r in csv.DictReader(open(spath)):
    stats[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    calls[short(r["Name"])] = int(r["Calls"])
shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{tag}_bench8192.json")

ft, fc = per_kernel(f"gpurun_out/{tag}_fetch/**/*counter_collection.csv", "FETCH_SIZE")
wt, wc = per_kernel(f"gpurun_out/{tag}_write/**/*counter_collection.csv", "WRITE_SIZE")
for name, (t, c) in (("FETCH_SIZE", (ft, fc)), ("WRITE_SIZE", (wt, wc))):
    with open(f"profiles/{tag}_pmc8192_{name}_per_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "total_KB", "avg_KB_per_launch"])
        for k in sorted(t, key=lambda k: -t[k]):
            w.writerow([k, c[k], t[k], t[k] / c[k]])
# FETCH_SIZE unit: 1 KiB that stands for 2 KiB moved (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the
# bytes"; confirmed for THIS code's access patterns by profiles/r2_fetch_size_calibration.txt: 2048.0 B per unit for
# 8 B-per-lane marching loads, 16 B-per-lane loads and LDS-DMA loads alike, tools/stream_probe2.hip under --pmc).
# The in-run kernels with known reads are listed for comparison only: they sit above 1.0 because the fabric-side counter
# sees every 128-B line a 64-lane row touches (five for a row that starts 16 B into a line) and counts Infinity-Cache hits.
FACTOR = 2.0
known = {k: (ft[k] / fc[k] * 1024.0 * FACTOR / cells) / KNOWN_READS[k] for k in KNOWN_READS if k in ft and ft[k] > 0}
with open(f"profiles/{tag}_pmc8192_bytes_per_cell.csv", "w") as f:
    f.write("# bench.py --steps 1 --warmup 1 --no-cpu at 8192^2, one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE).\n")
    f.write("# bytes per cell and launch; fetch = FETCH_SIZE [KiB] x 2 (guide + profiles/r2_fetch_size_calibration.txt), write = WRITE_SIZE [KiB] x 1.\n")
    f.write("# fabric-side counters: Infinity-Cache hits are counted, so `total` is an upper bound of the HBM bytes.\n")
    f.write("# kernels with known reads, fetched / algorithmic read bytes: " + ", ".join(f"{k} {v:.3f}" for k, v in known.items()) + "\n")
    f.write(f"# avg_us / share from profiles/{tag}_bench8192_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 3).\n")
    f.write("kernel;total_B_per_cell;fetch_B_per_cell;write_B_per_cell;algorithmic_B_per_cell;ratio;avg_us;share_pct;launches;algorithmic_TBps\n")
    for k in sorted(ft, key=lambda k: -(stats.get(k, (0, 0))[1])):
        a = algo_of(k)
        if a is None or k not in stats:
            continue
        fr = ft[k] / fc[k] * 1024.0 * FACTOR / cells
        wr = wt.get(k, 0.0) / max(wc.get(k, 1), 1) * 1024.0 / cells
        us, share = stats[k]
        f.write(";".join([k, f"{fr + wr:.2f}", f"{fr:.2f}", f"{wr:.2f}", f"{a}", f"{(fr + wr) / a:.3f}", f"{us:.1f}", f"{share:.2f}",
                          str(fc[k]), f"{a * cells / us / 1e6:.2f}"]) + "\n")
semi = [k for k in ft if k.startswith("k_semi_") and not k.startswith(("k_semi_factor", "k_semi_to_f32"))]
if semi:
    fac = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
    coarsest = [k for k in semi if k.startswith("k_semi_coarsest")]
    vc_pmc = sum(fc[k] for k in coarsest)                  # V-cycles seen by the counter pass
    vc_stats = sum(calls.get(k, 0) for k in coarsest)      # ... by the kernel-trace pass
    fetch = sum(ft[k] for k in semi) * 1024.0 * FACTOR / vc_pmc / cells
    write = sum(wt.get(k, 0.0) for k in semi) * 1024.0 / max(1, sum(wc.get(k, 0) for k in coarsest)) / cells
    us = sum(stats[k][0] * calls[k] for k in semi if k in stats) / max(1, vc_stats)
    algo = (56.0 + 2.0 * fac) + (30.0 + 2.0 * fac)      # finest level double vectors, coarser levels single (mg_f32 = 1)
    with open(f"profiles/{tag}_pmc8192_bytes_per_cell.csv", "a") as f:
        f.write("# semi-coarsened line multigrid, all k_semi_* launches of one V-cycle (every level), per FINE cell:\n")
        f.write("semi_vcycle;total_B_per_cell;fetch_B_per_cell;write_B_per_cell;algorithmic_B_per_cell;ratio;us_per_vcycle;vcycles_timed;algorithmic_TBps\n")
        f.write(";".join(["k_semi_*", f"{fetch + write:.2f}", f"{fetch:.2f}", f"{write:.2f}", f"{algo:.1f}", f"{(fetch + write) / algo:.3f}",
                          f"{us:.1f}", str(vc_stats), f"{algo * cells / us / 1e6:.2f}"]) + "\n")
print(open(f"profiles/{tag}_pmc8192_bytes_per_cell.csv").read())
