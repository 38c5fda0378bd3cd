#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ with the CPU oracle (oracle/, the restatement of the reference's
CPU backend — the reference itself is Rust and cannot run here).

A fixture is DATA: seeded inputs (raw quantized block bytes from the synthetic generator, f32 vectors) and the
oracle's outputs, stored as .npz (numpy, no pickle).  Two readers:
  * tests/test_golden.py (CPU): the oracle must keep reproducing them bit-for-bit — a change of the oracle's
    arithmetic cannot slip in unnoticed once it has been pinned against the reference's known-answer tests;
  * tests/test_golden.py (-m gpu): the HIP path, through the C ABI, against the same numbers.

Regenerate (from the repo root):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

QUANT = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
FUSED = ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "Q4_0"]


def ops_fixture(pkg, orc):
    out = {}
    rng = np.random.default_rng(20260327)
    k, n = 512, 6
    x = rng.standard_normal(k).astype(np.float32)
    out["x"] = x
    for tname in QUANT:
        t = pkg.synth.TYPE_IDS[tname]
        raw = pkg.synth.fill_tensor("blk.0.golden.weight", t, k * n, k)
        out[f"{tname}.raw"] = raw
        out[f"{tname}.dequant"] = orc.dequantize(t, raw, k * n)
        out[f"{tname}.vec_mat"] = orc.vec_mat_q(t, raw, x, n)
    # the reference's own ramp (dequant.rs:1070-1303: x_i = (i - 128) * 0.1) through quantize -> dequantize
    ramp = ((np.arange(256) - 128) * 0.1).astype(np.float32)
    out["ramp"] = ramp
    for tname in QUANT:
        t = pkg.synth.TYPE_IDS[tname]
        q = orc.quantize(t, ramp)
        out[f"{tname}.ramp_blocks"] = q
        out[f"{tname}.ramp_roundtrip"] = orc.dequantize(t, q, 256)
    w = (1.0 + 0.01 * rng.uniform(-1, 1, k)).astype(np.float32)
    out["norm_w"] = w
    out["rms_norm"] = orc.rms_norm(x, w, 1e-5)
    nh, nkv, d, max_seq, kv_len = 8, 2, 64, 48, 37
    q = rng.standard_normal((nh, d)).astype(np.float32)
    kk = rng.standard_normal((nkv, d)).astype(np.float32)
    for neox in (0, 1):
        rq, rk = orc.rope(q.reshape(nh, 1, d), kk.reshape(nkv, 1, d), 17, 10000.0, 1.0, bool(neox))
        out[f"rope{neox}.q"], out[f"rope{neox}.k"] = rq.reshape(nh, d), rk.reshape(nkv, d)
    out["rope_in.q"], out["rope_in.k"] = q, kk
    kc = rng.standard_normal((nkv, max_seq, d)).astype(np.float32)
    vc = rng.standard_normal((nkv, max_seq, d)).astype(np.float32)
    out["attn.q"], out["attn.k_cache"], out["attn.v_cache"] = q, kc, vc
    out["attn.out"] = orc.attention_cached(q, kc, vc, 1.0 / np.sqrt(d), kv_len)
    out["attn.kv_len"] = np.array([kv_len], np.int64)
    g, u = rng.standard_normal(300).astype(np.float32) * 3, rng.standard_normal(300).astype(np.float32)
    out["silu.gate"], out["silu.up"], out["silu.out"] = g, u, orc.silu_mul(g, u)
    return out


def model_fixture(pkg, orc, name, mix, prompt_len, n_decode):
    cfg = pkg.synth.make_config(name)
    model = pkg.synth.SynthModel(cfg, mix)
    ref = orc.Model(cfg.as_dict())
    for tname, t, ne, data in model.tensors():
        ref.add_tensor(tname, t, ne, data)
    ref.finalize()
    prompt = [(7 * i + 3) % cfg.vocab_size for i in range(prompt_len)]
    logits = [ref.forward(prompt)]
    toks = [orc.argmax_last(logits[-1])]
    for _ in range(n_decode - 1):
        logits.append(ref.forward([toks[-1]]))
        toks.append(orc.argmax_last(logits[-1]))
    ref.close()
    lg = np.stack(logits)
    srt = np.sort(lg, axis=1)
    return {"prompt": np.array(prompt, np.int64), "tokens": np.array(toks, np.int64), "logits": lg,
            "top_gap": (srt[:, -1] - srt[:, -2]).astype(np.float32)}


def main():
    pkg, orc = graft.load_package(), graft.load_oracle()
    orc.set_isa(3)   # ORC_ISA_AVX512: the AVX-512 lane order of dot_f32 (simd.rs:80-165): the fixture must not depend on the host CPU
    np.savez_compressed(os.path.join(HERE, "ops_v1.npz"), **ops_fixture(pkg, orc))
    for name, mix, p, n in [("test-dense", "Q4_K_M", 6, 10), ("test-moe", "Q5_K_M", 5, 8), ("test-dense", "Q8_0", 6, 10)]:
        fx = model_fixture(pkg, orc, name, mix, p, n)
        np.savez_compressed(os.path.join(HERE, f"model_{name}_{mix}_v1.npz"), **fx)
        print(name, mix, fx["tokens"].tolist(), "min gap", float(fx["top_gap"].min()))


if __name__ == "__main__":
    main()
