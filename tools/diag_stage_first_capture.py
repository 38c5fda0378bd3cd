#!/usr/bin/env python3
"""Reproduces what `warm_kernels` (csrc/engine.hip) guards against: in a FRESH process, a pipeline stage behind the first
one decoded from a stale position when its kernels were first launched inside a hipGraph capture (ROCm 7.0, MI355X).

    python tools/diag_stage_first_capture.py ATTN_DIRECT SPLIT N_PROMPT [FLAGS [REPS]]

builds one full context and two stage contexts split at layer SPLIT on the same GPU, feeds N_PROMPT tokens token by
token, hands the hidden vector over with a device copy, and prints max|dlogit| between the two for the next token.
Before the warm-up existed: 1.0 at rep 0 and 0.0 at every later rep of the same process, 0.0 with FLAGS=1 (no graph),
0.0 with a 1-token prompt (the captured graph's first launch was right, its replays were not).  With it: 0.0."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
ad, split, n_prompt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
cfg = pkg.make_config("test-dense-d128", max_seq_len=192)
model = pkg.SynthModel(cfg, mix="Q4_K_M")
for rep in range(reps):
    full = pkg.HipGpuInference.from_model(model, 192, attn_direct=ad, flags=fl)
    s0 = pkg.HipGpuInference.from_model(model, 192, layer_range=(0, split), attn_direct=ad, flags=fl)
    s1 = pkg.HipGpuInference.from_model(model, 192, layer_range=(split, cfg.num_layers), attn_direct=ad, flags=fl)

    def hop():
        hip.hipMemcpy(s1.stage_hidden_ptr(), s0.stage_hidden_ptr(), cfg.hidden_size * 4, 3)
        hip.hipDeviceSynchronize()

    for t in [(5 * i + 2) % cfg.vocab_size for i in range(n_prompt)]:
        full.prefill_token(t)
        s0.stage_forward(t)
        s0.synchronize()
        hop()
        s1.stage_forward(0)
        s1.synchronize()
    want = full.forward(3)
    s0.stage_forward(3)
    s0.synchronize()
    hop()
    got = s1.stage_forward(0, want_logits=True)
    print(f"rep {rep} flags {fl}: attn_direct={ad} split={split} prompt={n_prompt}: max|d|={np.abs(got - want).max():.3e}", flush=True)
    for e in (full, s0, s1):
        e.close()
