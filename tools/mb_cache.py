#!/usr/bin/env python3
"""Separates memory from issue limits: the same mat-vec launch with cold weights (cycled copies) and with
weights that stay in the 256 MiB Infinity Cache (one copy, re-read every launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

pkg = graft.load_package()
hb, syn = pkg.hip_backend, pkg.synth
for label, tname, k, n, mode in [("gate/up Q4_K", "Q4_K", 4096, 14336, 2), ("down Q4_K", "Q4_K", 14336, 4096, 0),
                                 ("down Q6_K", "Q6_K", 14336, 4096, 0), ("wo Q4_K", "Q4_K", 4096, 4096, 0),
                                 ("small Q4_K n=1024", "Q4_K", 4096, 1024, 0)]:
    t = syn.TYPE_IDS[tname]
    w = syn.fill_tensor("bench.weight", t, k * n, k)
    nbytes = w.nbytes * (2 if mode == 2 else 1)
    cold = hb.bench_vec_mat(t, w, k, n, mode=mode, iters=200, copies=max(2, int(600e6 // nbytes) + 1))
    warm = hb.bench_vec_mat(t, w, k, n, mode=mode, iters=200, copies=1)
    print(f"{label:22s} cold {cold:7.2f} us ({nbytes/cold/1e3:7.1f} GB/s)   cached {warm:7.2f} us ({nbytes/warm/1e3:7.1f} GB/s)", flush=True)
