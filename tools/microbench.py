#!/usr/bin/env python3
"""Kernel micro-benchmark: fused dequant mat-vec shapes of the BASELINE models, cold weights (cycled copies).

    python tools/microbench.py [--iters 40]
Prints µs per launch and algorithmic GB/s (weight bytes / time) per shape."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

SHAPES = [  # (label, type, k, n, mode)  mode 0 plain, 1 norm prologue, 2 gate/up SwiGLU pair
    ("L8 wo        Q4_K 4096->4096", "Q4_K", 4096, 4096, 0),
    ("L8 qkv-ish   Q4_K 4096->6144 +norm", "Q4_K", 4096, 6144, 1),
    ("L8 gate/up   Q4_K 4096->14336 x2", "Q4_K", 4096, 14336, 2),
    ("L8 down      Q4_K 14336->4096", "Q4_K", 14336, 4096, 0),
    ("L8 down      Q6_K 14336->4096", "Q6_K", 14336, 4096, 0),
    ("L8 output    Q6_K 4096->128256 +norm", "Q6_K", 4096, 128256, 1),
    ("MX gate/up   Q5_K 4096->14336 x2", "Q5_K", 4096, 14336, 2),
    ("T  gate/up   Q8_0 2048->5632 x2", "Q8_0", 2048, 5632, 2),
    ("T  down      Q8_0 5632->2048", "Q8_0", 5632, 2048, 0),
    ("L70 gate/up  Q4_K 8192->28672 x2", "Q4_K", 8192, 28672, 2),
    ("L70 down     Q4_K 28672->8192", "Q4_K", 28672, 8192, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--only", default="")
    ap.add_argument("--two-streams", action="store_true", help="alternate launches between two unordered streams (overlap experiment)")
    ap.add_argument("--shape", action="append", default=[], help="extra shape TYPE:K:N:MODE (e.g. Q4_K:4096:16384:2); with --only-extra nothing else runs")
    ap.add_argument("--only-extra", action="store_true")
    ap.add_argument("--copies", type=int, default=0, help="weight copies cycled through (0 = enough to stay cold; 1 = hot in the Infinity Cache)")
    args = ap.parse_args()
    pkg = graft.load_package()
    hb, syn = pkg.hip_backend, pkg.synth
    shapes = [] if args.only_extra else list(SHAPES)
    for spec in args.shape:
        tname, k, n, mode = spec.split(":")
        shapes.append((f"extra {tname} {k}->{n} mode {mode}", tname, int(k), int(n), int(mode)))
    for label, tname, k, n, mode in shapes:
        if args.only and args.only not in label:
            continue
        t = syn.TYPE_IDS[tname]
        w = syn.fill_tensor("bench.weight", t, k * n, k)
        nbytes = w.nbytes * (2 if mode == 2 else 1)
        copies = args.copies if args.copies > 0 else max(2, min(32, int(600e6 // nbytes) + 1))
        us = hb.bench_vec_mat(t, w, k, n, mode=mode | (16 if args.two_streams else 0), iters=args.iters, copies=copies)
        print(f"{label:40s} {us:9.2f} us  {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / 1e6:.1f} MB, {copies} copies)", flush=True)


if __name__ == "__main__":
    main()
