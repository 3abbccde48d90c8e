#!/usr/bin/env python3
"""Per-kernel launch counts and durations from a rocprofv3 rocpd database: python tools/db_stats.py file.db [steps]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n, t = defaultdict(int), defaultdict(float)
for name, s, e in db.execute("select name, start, end from kernels"):
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
    n[nm] += 1
    t[nm] += (e - s) / 1e3
print(f"launches/step {sum(n.values()) / steps:.1f}  kernel us/step {sum(t.values()) / steps:.1f}")
for k in sorted(t, key=lambda k: -t[k])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k[:72]:72s} n/step {n[k] / steps:7.1f} us/step {t[k] / steps:9.1f} avg {t[k] / n[k]:8.1f}")
