#!/usr/bin/env python3
"""Condensed opcode stream of a kernel's largest loop, from `hipcc -S --cuda-device-only` output.
    python tools/diag/isa_loop.py file.s <substring of the mangled kernel name> [--all-loops]
M = v_mfma 32x32, m = v_mfma 16x16, E = v_exp, c = v_cvt_pk, x = v_pk_max*, w = v_permlane*, L / W = LDS read / write,
G = buffer / global load, | = s_waitcnt, B = s_barrier, v / s = other VALU / SALU, J = branch."""
import re
import sys


def short(l):
    op = l.split()[0]
    if op.startswith('v_mfma'):
        return 'M' if '32x32' in op else 'm'
    for pre, ch in (('v_exp', 'E'), ('v_cvt_pk', 'c'), ('v_pk_max', 'x'), ('v_permlane', 'w'), ('ds_read', 'L'), ('ds_load', 'L'),
                    ('ds_write', 'W'), ('ds_store', 'W'), ('buffer_load', 'G'), ('global_load', 'G'), ('s_waitcnt', '|'),
                    ('s_barrier', 'B'), ('scratch_load', '<'), ('scratch_store', '>'), ('s_cbranch', 'J'), ('s_branch', 'J'), ('s_nop', 'n')):
        if op.startswith(pre):
            return ch
    if re.match(r'^\.?LBB', l):
        return '\n' + l + ' '
    return 'v' if op.startswith('v_') else 's' if op.startswith('s_') else '?'


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):', s, flags=re.M)
    if not m:
        sys.exit("kernel not found")
    i = m.start()
    body = s[i:s.index('.Lfunc_end', i)]
    lines = [l.split(';')[0].strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(('.s', '.p', '.t', '.g', '.w'))]
    labels = {l[:-1]: n for n, l in enumerate(lines) if re.match(r'^\.?LBB\d+_\d+:$', l)}
    loops = []
    for n, l in enumerate(lines):
        mm = re.match(r's_c?branch\w* (\.?LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            loops.append((labels[mm.group(1)], n))
    print(m.group(1)[:90], 'loops', loops, 'lines', len(lines))
    # the hot loop: the innermost backward edge whose body holds matrix instructions
    hot = [t for t in loops if any(l.startswith('v_mfma') for l in lines[t[0]:t[1] + 1])]
    todo = loops if '--all-loops' in sys.argv else [min(hot or loops, key=lambda t: t[1] - t[0])]
    for a, b in todo:
        seq = lines[a:b + 1]
        print(''.join(short(l) for l in seq))
        from collections import Counter
        print(Counter(short(l) for l in seq if not short(l).startswith('\n')))


main()
