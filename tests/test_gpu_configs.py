"""BASELINE.json's GPU configurations under `-m gpu` (shapes: `ModelConfig` presets, src/model/config.rs:239-281;
MoE layer src/model/moe.rs:321-413):

  * TinyLlama-1.1B Q4_K_M at FULL size against the CPU oracle — decode path and batched prompt path;
  * Mixtral-8x7B Q5_K_M and Llama-3-70B Q4_K_M at full WIDTH and truncated depth (real hidden / ffn / experts /
    heads / vocabulary, 2 layers) against the oracle, which at full depth would need minutes per token;
  * the same two models at FULL size on one GPU through size-independent properties: graph replay == eager launches
    bit for bit, `decode_greedy` (token fed back on device) == a host arg-max loop over `forward`, run-to-run
    determinism, and — at 70B width — a two-stage layer split on one GPU == the single context bit for bit.

(Llama-3-8B Q4_K_M at full size is in test_gpu_model.py / test_gpu_prefill.py.)  Tolerances as everywhere
(SURVEY.md §8c): logits max|d| <= 2e-3*max|logit| + 2e-3 on the exact path, 1e-2*max|logit| + 1e-2 after a batched
(f16 GEMM) prompt pass; greedy tokens identical wherever the oracle's top-1/top-2 gap exceeds 4x the error."""
import os
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tol(want, pf=False):
    return (1e-2 if pf else 2e-3) * float(np.abs(want).max()) + (1e-2 if pf else 2e-3)


def _oracle(orc, cfg, model):
    ref = orc.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors():
        ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    return ref


def _follow_oracle(orc, eng, ref, prompt, n_steps, pf=False):
    """Logits after `prompt` and over `n_steps` greedy tokens (the oracle's token is fed to both sides)."""
    got = eng.forward(prompt[-1])
    want = ref.forward(prompt)
    worst, min_gap = 0.0, np.inf
    for step in range(n_steps + 1):
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err <= _tol(want, pf), f"step {step}: max|dlogit| {err:.3e} > {_tol(want, pf):.3e}"
        srt = np.sort(want)
        gap = float(srt[-1] - srt[-2])
        min_gap = min(min_gap, gap)
        tok = orc.argmax_last(want)
        if gap > 4 * err:
            assert orc.argmax_last(got) == tok, f"step {step}: greedy token differs with gap {gap:.3e} vs err {err:.3e}"
        if step < n_steps:
            got, want = eng.forward(tok), ref.forward([tok])
    return worst, min_gap


def test_tinyllama_1_1b_q4_k_m_full_size_matches_oracle(pkg, orc):
    """configs[1]: all 22 layers, hidden 2048, ffn 5632, 32/4 heads of 64, vocab 32000, Q4_K_M mix (Q6_K output, attn_v and
    ffn_down on the `_M` layers).  Exact decode path, then the batched prompt path, both against the oracle."""
    cfg = pkg.make_config("tinyllama-1.1b", max_seq_len=192)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    ref = _oracle(orc, cfg, model)
    eng = pkg.HipGpuInference.from_model(model, 192)
    try:
        prompt = [1, 31999, 77, 4242, 9, 15000]
        for t in prompt[:-1]:
            eng.prefill_token(t)
        worst, gap = _follow_oracle(orc, eng, ref, prompt, 6)
        print(f"tinyllama exact: max|dlogit|={worst:.3e} min_gap={gap:.3e}")
        # batched prompt pass (f16 MFMA GEMMs) over 40 tokens, then decode on the exact path
        eng.reset()
        ref.reset()
        long_prompt = [(37 * i + 11) % cfg.vocab_size for i in range(41)]
        assert eng.prefill_is_batched()
        eng.forward_batch(long_prompt[:-1])
        assert eng.position() == 40
        worst, gap = _follow_oracle(orc, eng, ref, long_prompt, 3, pf=True)
        print(f"tinyllama after batched prefill: max|dlogit|={worst:.3e} min_gap={gap:.3e}")
        # device-fed greedy decode == host arg-max loop, and deterministic
        pos = eng.position()
        a = eng.decode_greedy(5, 32).tolist()
        eng.kv_truncate(pos)
        b, tok = [], 5
        for _ in range(32):
            tok = orc.argmax_last(eng.forward(tok))
            b.append(tok)
        assert a == b
    finally:
        ref.close()
        eng.close()


@pytest.mark.parametrize("name,mix,layers", [("mixtral-8x7b", "Q5_K_M", 2), ("llama-3-70b", "Q4_K_M", 2)])
def test_full_width_truncated_depth_matches_oracle(pkg, orc, name, mix, layers):
    """configs[3] and configs[4] at their real widths (Mixtral: 8 experts of 4096x14336 top-2; 70B: hidden 8192, ffn 28672,
    64/8 heads, vocab 128256) with `layers` layers: exact path and batched prompt path against the oracle."""
    cfg = pkg.make_config(name, max_seq_len=96, num_layers=layers)
    model = pkg.SynthModel(cfg, mix=mix)
    ref = _oracle(orc, cfg, model)
    eng = pkg.HipGpuInference.from_model(model, 96)
    try:
        prompt = [i % cfg.vocab_size for i in (1, 31999, 128000, 77, 5)]
        for t in prompt[:-1]:
            eng.prefill_token(t)
        worst, gap = _follow_oracle(orc, eng, ref, prompt, 4)
        print(f"{name} x{layers} exact: max|dlogit|={worst:.3e} min_gap={gap:.3e}")
        eng.reset()
        ref.reset()
        long_prompt = [(53 * i + 7) % cfg.vocab_size for i in range(34)]
        eng.forward_batch(long_prompt[:-1])
        worst, gap = _follow_oracle(orc, eng, ref, long_prompt, 2, pf=eng.prefill_is_batched())
        print(f"{name} x{layers} after prefill (batched={eng.prefill_is_batched()}): max|dlogit|={worst:.3e} min_gap={gap:.3e}")
    finally:
        ref.close()
        eng.close()


def _properties(pkg, orc, cfg, model, max_seq, n_dec):
    """Size-independent properties of a full-size model: graph == eager bitwise, device-fed greedy == host arg-max loop,
    run-to-run determinism.  Returns the engine (graph mode) positioned after the prompt, and the reference logits."""
    prompt = [(97 * i + 3) % cfg.vocab_size for i in range(6)]
    eager = pkg.HipGpuInference.from_model(model, max_seq, flags=pkg.hip_backend.FLAG_NO_GRAPH)
    for t in prompt[:-1]:
        eager.prefill_token(t)
    want = eager.forward(prompt[-1])
    want2 = eager.forward(17)
    eager.close()
    eng = pkg.HipGpuInference.from_model(model, max_seq)
    for t in prompt[:-1]:
        eng.prefill_token(t)
    got = eng.forward(prompt[-1])
    assert np.array_equal(got, want), "graph replay differs from eager launches"
    assert np.array_equal(eng.forward(17), want2)
    assert np.isfinite(got).all() and float(np.abs(got).max()) > 0
    pos = eng.position()
    a = eng.decode_greedy(23, n_dec).tolist()
    eng.kv_truncate(pos)
    b, tok = [], 23
    for _ in range(n_dec):
        tok = orc.argmax_last(eng.forward(tok))
        b.append(tok)
    assert a == b, "device-fed greedy decode differs from the host arg-max loop"
    eng.kv_truncate(pos)
    assert eng.decode_greedy(23, n_dec).tolist() == a, "decode is not deterministic run to run"
    eng.kv_truncate(pos)
    return eng, prompt


def test_mixtral_8x7b_q5_k_m_full_size_properties(pkg, orc):
    """configs[3] at full size (32 layers x 8 experts, 33 GB resident on one GPU)."""
    cfg = pkg.make_config("mixtral-8x7b", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q5_K_M")
    eng, _ = _properties(pkg, orc, cfg, model, 64, 12)
    st = eng.stats()
    assert st["weight_bytes"] > 30e9
    eng.close()


def test_llama3_70b_q4_k_m_full_size_properties_and_two_stage_split(pkg, orc):
    """configs[4] on ONE GPU (42 GB resident): the properties above, then layers 0..39 | 40..79 as two stage contexts on
    the same device with the f32[8192] hidden vector handed over == the single context, bit for bit
    (src/distributed/pipeline.rs:50-96)."""
    cfg = pkg.make_config("llama-3-70b", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    eng, prompt = _properties(pkg, orc, cfg, model, 64, 8)
    s0 = pkg.HipGpuInference.from_model(model, 64, layer_range=(0, 40))
    s1 = pkg.HipGpuInference.from_model(model, 64, layer_range=(40, 80))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    try:
        eng.reset()
        for tok in prompt[:4]:
            want = eng.forward(tok)
            s0.stage_forward(tok)
            s0.synchronize()
            assert hip.hipMemcpy(s1.stage_hidden_ptr(), s0.stage_hidden_ptr(), cfg.hidden_size * 4, 3) == 0
            assert hip.hipDeviceSynchronize() == 0
            got = s1.stage_forward(0, want_logits=True)
            assert np.array_equal(got, want)
    finally:
        for e in (eng, s0, s1):
            e.close()


def test_full_size_llama3_8b_batched_step_and_turboquant(pkg, orc):
    """Round-3 paths at BASELINE.json's FULL headline size (Llama-3-8B Q4_K_M, 32 layers, vocab 128256):
    (1) a multi-sequence step gives every sequence the single-sequence engine's logits BIT FOR BIT (3 sequences, ragged histories,
        4 steps; the device-fed greedy loop equals the single-sequence greedy decode);
    (2) the TurboQuant-3-bit KV cache (`--kv-cache-type tq3`) against the CPU oracle with the same sign vectors: logits of a short
        prompt and two greedy steps within the TurboQuant tolerance, greedy tokens identical where the gap allows."""
    cfg = pkg.make_config("llama-3-8b", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    multi = pkg.HipGpuInference.from_model(model, 64)
    single = pkg.HipGpuInference.from_model(model, 64, attn_direct=255)
    try:
        multi.batch_create(4)
        hist = [[(7919 * (s + 1) * (i + 3)) % cfg.vocab_size for i in range(3 + 2 * s)] for s in range(3)]
        slots = [2, 0, 3]
        for s in range(3):
            for t in hist[s][:-1]:
                multi.forward_multi([slots[s]], [t], want_logits=False)
        toks = [h[-1] for h in hist]
        got = []
        for step in range(4):
            logits, nxt = multi.forward_multi(slots, toks, want_logits=True, greedy=True)
            got.append(logits.copy())
            toks = [int(t) for t in nxt]
        for s in range(3):
            single.reset()
            for t in hist[s][:-1]:
                single.prefill_token(t)
            tok = hist[s][-1]
            for step in range(4):
                want = single.forward(tok)
                assert np.array_equal(got[step][s].view(np.uint32), want.view(np.uint32)), (s, step, float(np.abs(got[step][s] - want).max()))
                tok = int(np.flatnonzero(want == want.max())[-1])
    finally:
        multi.close()
        single.close()
    # ---- (2) TurboQuant at full size against the oracle
    rng = np.random.default_rng(11)
    signs = np.where(rng.integers(0, 2, cfg.num_layers * cfg.num_kv_heads * 2 * cfg.head_dim) == 1, 1.0, -1.0).astype(np.float32)
    ref = orc.Model(cfg.as_dict())
    eng = None
    try:
        for nm, t, ne, data in model.tensors():
            ref.add_tensor(nm, t, ne, data)
        ref.finalize()
        ref.set_kv_turboquant(3, signs)
        orc.set_threads(min(16, os.cpu_count() or 1))
        eng = pkg.HipGpuInference.from_model(model, 64, kv_cache_type=pkg.hip_backend.KV_TQ3, kv_rotation_signs=signs)
        prompt = [i % cfg.vocab_size for i in (1, 128000, 77, 31999)]
        for t in prompt[:-1]:
            eng.prefill_token(t)
        got, want = eng.forward(prompt[-1]), ref.forward(prompt)
        for step in range(3):
            tol = 4.0 * (2e-3 * float(np.abs(want).max()) + 2e-3)
            err = float(np.abs(got - want).max())
            assert err <= tol, f"TurboQuant step {step}: max|dlogit| {err:.3e} > {tol:.3e}"
            srt = np.sort(want)
            tok = orc.argmax_last(want)
            if float(srt[-1] - srt[-2]) > 4 * err:
                assert orc.argmax_last(got) == tok
            if step < 2:
                got, want = eng.forward(tok), ref.forward([tok])
    finally:
        ref.close()
        if eng is not None:
            eng.close()
