#!/usr/bin/env python3
"""Phase stamps of the prefill GEMM (diagnostic build: `make -C llama-gguf_amd stamps`, LGH_LIB_VARIANT=stamps): per workgroup
of the last gate|up GEMM of a prompt pass — start, prologue done, and for each k-block: MFMA loop done / loads + barrier done."""
import ctypes as C
import os
import sys

os.environ.setdefault("LGH_LIB_VARIANT", "stamps")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
cfg = pkg.make_config("llama-3-8b", max_seq_len=512, num_layers=4)
eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix="Q4_K_M"), 512)
prompt = [i % 32000 for i in range(128)]
eng.forward_batch(prompt)
eng.reset()
eng.forward_batch(prompt)
eng.synchronize()
lib = pkg.hip_backend.load_library()
buf = (C.c_ulonglong * (1024 * 16))()
lib.lgh_debug_pf_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
assert lib.lgh_debug_pf_stamps(buf, 1024 * 16) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
used = st[:, 0] > 0
st = st[used]
t0 = st[:, 0].min()
names = ["start", "prologue done"] + [f"block {b} {'mfma done' if k == 0 else 'synced'}" for b in range(6) for k in range(2)] + ["stores done"]
cols = [0, 1] + list(range(2, 14)) + [14]
print(f"{len(st)} workgroups; us since the first start (min / p50 / max)")
for nm, c in zip(names, cols):
    v = (st[:, c] - t0) / 100.0
    v = v[st[:, c] > 0]
    if len(v):
        print(f"  {nm:22s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")

