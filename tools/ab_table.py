#!/usr/bin/env python3
"""Tabulate a tools/ab_rp2.sh log: one row per (K, N, M), one column per configuration."""
import re, sys
rows, cfgs = {}, []
for line in open(sys.argv[1]):
    m = re.match(r'\[(.*?)\] rgemm M=(\d+) K=(\d+) N=(\d+) .*?:\s+([\d.]+) us', line)
    if not m:
        continue
    cfg, M, K, N, us = m.groups()
    if cfg not in cfgs:
        cfgs.append(cfg)
    rows.setdefault((int(K), int(N), int(M)), {})[cfg] = float(us)
print("K N M : " + " | ".join(cfgs))
for k, r in rows.items():
    print(k, " ".join(f"{r.get(c, float('nan')):7.2f}" for c in cfgs))
