import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
def show(tag, x, y):
    d = np.abs(x - y); rows = d.max(axis=0)
    print(tag, "max %.3e ref %.3e" % (d.max(), np.abs(x).max()), "rows", np.argsort(rows)[-8:][::-1], np.sort(rows)[-8:][::-1])
show("single unfused vs fused", a["z0"], a["z1"])
show("unfused single vs slab", a["z0"], b["z0"])
show("fused single vs slab", a["z1"], b["z1"])
