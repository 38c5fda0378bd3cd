#!/usr/bin/env python3
"""Phase profile of the persistent token kernel (diagnostic build: `make -C llama-gguf_amd stamps`, LGH_LIB_VARIANT=stamps).

    LGH_LIB_VARIANT=stamps python tools/pt_phases.py [--model llama-3-8b] [--mix Q4_K_M] [--kv 128]
Per op kind (position in the layer): mean over layers of the time between the phase stamps of one workgroup's waves, in us."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3-8b"); ap.add_argument("--mix", default="Q4_K_M"); ap.add_argument("--kv", type=int, default=128)
a = ap.parse_args()
pkg = graft.load_package()
cfg = pkg.make_config(a.model, max_seq_len=a.kv + 64)
eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=a.mix), a.kv + 64, flags=pkg.hip_backend.FLAG_PERSISTENT)
for t in range(a.kv):
    eng.prefill_token(t % cfg.vocab_size)
eng.decode_greedy(5, 8)
L = pkg.hip_backend.load_library()
nops = 5 * cfg.num_layers + 1
buf = (C.c_ulonglong * (1024 * 8 * 8))()
assert L.lgh_debug_pt_stamps(buf, len(buf)) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 8)[:nops].astype(np.int64)
names = ["qkv", "attn", "wo", "gate_up", "down"]
t0 = st[0, :, 0].min()
print(f"token span (this workgroup): {(st[nops - 1, :, 7].max() - t0) / 100:.1f} us over {nops} ops")
lay = st[:5 * cfg.num_layers].reshape(cfg.num_layers, 5, 8, 8)[2:]          # skip the first layers (cold)
for k, nm in enumerate(names):
    s = lay[:, k]                                                            # [layers, waves, stamps]
    beg = s[:, :, 0].min(axis=1, keepdims=True)
    def rel(i): return ((s[:, :, i] - beg) / 100.0)
    if nm == "attn":
        row = " ".join(f"{lbl}={rel(i).mean():5.2f}/{rel(i).max(axis=1).mean():5.2f}" for i, lbl in
                       ((1, "polled"), (2, "bar"), (3, "rows"), (4, "wavemerge+bar"), (5, "stores"), (6, "drain"), (7, "end")))
        print(f"{nm:8s} (mean/max over waves, us since op begin) {row}")
        continue
    row = " ".join(f"{lbl}={rel(i).mean():5.2f}/{rel(i).max(axis=1).mean():5.2f}" for i, lbl in
                   ((1, "waited"), (2, "x_in"), (3, "items"), (4, "bar1"), (5, "epi"), (6, "drain"), (7, "end")))
    print(f"{nm:8s} (mean/max over waves, us since op begin) {row}")
nxt = lay[:, 1:, :, 0].min(axis=2) - lay[:, :-1, :, 0].min(axis=2)
print("op-to-op begin deltas (us):", " ".join(f"{names[i]}={nxt[:, i].mean() / 100:.2f}" for i in range(4)),
      f"layer={(lay[1:, 0, :, 0].min(axis=1) - lay[:-1, 0, :, 0].min(axis=1)).mean() / 100:.2f}")
