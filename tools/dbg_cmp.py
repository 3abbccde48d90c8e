import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for k in a.files:
    d = np.abs(a[k] - b[k]); d[[0,0,-1,-1],[0,-1,0,-1]] = 0
    rows = d.max(axis=0)
    print(k, "max %.3e ref %.3e" % (d.max(), np.abs(a[k]).max()), "rows", np.argsort(rows)[-4:][::-1], np.sort(rows)[-4:][::-1])
