#!/usr/bin/env python3
"""Instruction mix of the innermost loop of a gfx950 kernel (from `hipcc -S --cuda-device-only` output).
usage: isa_loop_stats.py file.s substring-of-kernel-symbol [...]
Cycle estimate per loop trip and wave: f64 VALU 4 cycles (16 lanes/clk), transcendental f64 16, other VALU 2."""
import re, sys
from collections import Counter

def kernels(path):
    cur, body, out = None, [], {}
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln.rstrip())
            if 's_endpgm' in ln:
                out[cur] = body
                cur = None
    return out

def classify(op):
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'vmem_load'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'vmem_store'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'):
        if 'f64' in op:
            if any(t in op for t in ('rcp', 'rsq', 'sqrt', 'div_scale', 'div_fmas', 'div_fixup')):
                return 'valu_f64_slow' if ('rcp' in op or 'rsq' in op or 'sqrt' in op) else 'valu_f64'
            return 'valu_f64'
        return 'valu_other'
    return 'other'

def loop_stats(body):
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', ln)
        if m: labels[m.group(1)] = i
    loops = []
    for i, ln in enumerate(body):
        m = re.match(r'\s+s_cbranch\w*\s+(\.LBB\w+)|\s+s_branch\s+(\.LBB\w+)', ln)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    if not loops: return None
    # innermost = loop that contains no other loop, pick the longest of those
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    a, b = max(inner, key=lambda l: l[1] - l[0])
    c = Counter()
    dpp = 0
    for ln in body[a:b + 1]:
        m = re.match(r'\s+([a-z_0-9]+)', ln)
        if not m or ln.strip().startswith(('.', ';')): continue
        c[classify(m.group(1))] += 1
        if 'dpp' in ln or 'row_' in ln or 'wave_sh' in ln: dpp += 1
    cyc = 4 * c['valu_f64'] + 16 * c['valu_f64_slow'] + 2 * c['valu_other'] + 1 * c['salu']
    return (b - a + 1), c, dpp, cyc

if __name__ == '__main__':
    ks = kernels(sys.argv[1])
    for pat in sys.argv[2:]:
        for name, body in ks.items():
            if pat in name:
                r = loop_stats(body)
                if r is None: continue
                n, c, dpp, cyc = r
                print(f"{name[:90]}\n   loop lines {n}: " + ", ".join(f"{k}={v}" for k, v in sorted(c.items())) + f", dpp={dpp}, est VALU cycles/trip={cyc}")
