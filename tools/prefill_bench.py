#!/usr/bin/env python3
"""Times lgh_prefill_batch alone (the batched f16-GEMM prompt path): `python tools/prefill_bench.py [--model M --mix X
--prompt N --iters I]`; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--mix", default="Q4_K_M")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    pkg = graft.load_package()
    cfg = pkg.make_config(a.model, max_seq_len=max(512, a.prompt + 16))
    eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=a.mix), cfg.max_seq_len)
    prompt = [i % 32000 % cfg.vocab_size for i in range(a.prompt)]
    eng.forward_batch(prompt)
    best = 1e9
    for _ in range(a.iters):
        eng.reset()
        eng.synchronize()
        t0 = time.perf_counter()
        eng.forward_batch(prompt)
        eng.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{a.model} {a.mix}: {a.prompt} prompt tokens, batched={eng.prefill_is_batched()}, best {1e3 * best:.3f} ms "
          f"= {a.prompt / best:.0f} tokens/s")
    eng.close()


if __name__ == "__main__":
    main()
