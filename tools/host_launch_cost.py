#!/usr/bin/env python3
"""How long does the HOST take to enqueue one decode token (one hipGraphLaunch of the ~196-node token graph)?
If that approaches the GPU time of the token, single-stream decode is bound by the host's graph launch, not by the GPU.
    python tools/host_launch_cost.py [--model llama-3-8b] [--n 300]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3-8b"); ap.add_argument("--mix", default="Q4_K_M"); ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
pkg = graft.load_package()
cfg = pkg.make_config(a.model, max_seq_len=a.n * 2 + 64)
eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=a.mix), a.n * 2 + 64)
eng.forward_batch([t % cfg.vocab_size for t in range(32)])
eng.decode_greedy(5, 8)
eng.synchronize()
t0 = time.perf_counter()
for _ in range(a.n):
    eng.stage_step(2)            # one token: graph launch only, no host value in or out, no synchronisation
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
print(f"{a.model} {a.mix}: host enqueue {1e6 * (t1 - t0) / a.n:.1f} us per token; GPU drained {1e6 * (t2 - t0) / a.n:.1f} us per token "
      f"({a.n} tokens; the host ran {1e3 * (t2 - t1):.1f} ms ahead of the GPU at the end)")
