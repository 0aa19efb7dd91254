"""Canonical form of a solutions.dot.

This is the survey's normative script: it DEFINES the canonical sha256 values recorded for the
reference in BASELINE.md, so the tests apply it unchanged to the dot files this build writes.
"""
import hashlib
import re
from collections import deque


def canon(path):
    st = {}; out = {}; hdr = []
    rs = re.compile(r'^(\d+) \[shape=(\w+), label="(\d+): (S|[-\d, ]*)"\];$')
    re_ = re.compile(r'^(\d+) -> (\d+) \[label="([-\d, ]*)"\];$')
    for ln in open(path):
        ln = ln.rstrip('\n')
        if ln.startswith('#'):
            hdr.append(ln); continue
        m = rs.match(ln)
        if m:
            i = int(m.group(1)); sig = m.group(4)
            st[i] = (1 if m.group(2) == 'doublecircle' else 0, int(m.group(3)),
                     'S' if sig == 'S' else tuple(int(x) for x in sig.split(', ') if x != ''))
            continue
        m = re_.match(ln)
        if m:
            out.setdefault(int(m.group(1)), []).append((tuple(int(x) for x in m.group(3).split(', ')), int(m.group(2))))
    lines = [hdr[1] if len(hdr) > 1 else '#', hdr[2] if len(hdr) > 2 else '#']
    roots = [i for i, v in st.items() if v[2] == 'S']
    if not roots:
        return '\n'.join(lines + ['EMPTY']) + '\n', 0, 0
    num = {roots[0]: 0}; order = [roots[0]]; q = deque(order); cid = {}
    while q:
        u = q.popleft()
        for vals, v in sorted(out.get(u, [])):
            if v not in num:
                num[v] = len(num); order.append(v); q.append(v)
    ne = 0
    for u in order:
        fin, c, sig = st[u]
        if c not in cid:
            cid[c] = len(cid)
        lines.append('S %d %d %d %s' % (num[u], cid[c], fin, 'S' if sig == 'S' else ' '.join(map(str, sig))))
        for vals, v in sorted(out.get(u, [])):
            lines.append('E %d %d %s' % (num[u], num[v], ' '.join(map(str, vals)))); ne += 1
    return '\n'.join(lines) + '\n', len(order), ne


def canon_sha256(path):
    t, ns, ne = canon(path)
    return hashlib.sha256(t.encode()).hexdigest(), ns, ne
