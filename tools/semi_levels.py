#!/usr/bin/env python3
"""Per-level times of the semi-coarsened V-cycle kernels from a rocprofv3 kernel trace (s_kernel_trace.csv)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_semi" not in n or "factor" in n or "f32" in n:
        continue
    n = re.sub(r"\(.*", "", n).replace("void cfd::", "")
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    agg[(n, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot, ncyc = 0.0, 1
for k in sorted(agg, key=lambda k: (k[0], -k[1])):
    v = agg[k]
    print(f"{k[0]:28s} grid {k[1]:5d} n {len(v):5d} mean {sum(v) / len(v):8.1f} us")
    tot += sum(v)
    if "coarsest" in k[0]:
        ncyc = len(v)
print("semi kernels per V-cycle: %.1f us over %d V-cycles" % (tot / ncyc, ncyc))
