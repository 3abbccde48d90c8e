import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for k in a.files:
    if not k.startswith("field_"): continue
    d = np.abs(a[k] - b[k])
    d[[0,0,-1,-1],[0,-1,0,-1]] = 0
    rows = d.max(axis=0)
    print(k, "max", d.max(), "ref", np.abs(a[k]).max(), "worst rows", np.argsort(rows)[-6:][::-1], np.sort(rows)[-6:][::-1])
print(a["log"][:, :5]); print(b["log"][:, :5])
