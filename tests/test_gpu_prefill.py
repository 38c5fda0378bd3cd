"""Batched prompt processing on the MI355X: the f16-MFMA GEMM path behind lgh_prefill_batch.

The reference has no batched prefill (gpu_only.rs:776-806 feeds the prompt token by token), so the contract is the
one SURVEY §8(a16) states: the result must equal N sequential `prefill_token`s within 1e-2 relative — this is the one
place f16 rounding enters (operands rounded to 11 significant bits, f32 accumulation).  The exact path (f32 throughout)
is the comparison target, and the CPU oracle behind it.

Stated tolerances:
  GEMM alone     max|d| <= 1e-2 * max|y|  and  rms(d) <= 2e-3 * rms(y)        (measured: ~3e-4 * rms)
  logits after   max|d| <= 1e-2 * max|logit| + 1e-2 against the exact engine and against the oracle
  greedy tokens  identical whenever the top-1/top-2 gap exceeds 4x the measured logit error"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FUSED = ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "Q4_0"]


def _weights(pkg, tname, k, n, name="blk.0.test.weight"):
    t = pkg.synth.TYPE_IDS[tname]
    return t, pkg.synth.fill_tensor(name, t, k * n, k)


def _check_gemm(got, want):
    d = got.astype(np.float64) - want
    rms_y = float(np.sqrt(np.mean(want ** 2)))
    assert np.abs(d).max() <= 1e-2 * np.abs(want).max(), f"max|d| {np.abs(d).max():.3e} vs max|y| {np.abs(want).max():.3e}"
    assert float(np.sqrt(np.mean(d ** 2))) <= 2e-3 * rms_y
    return float(np.sqrt(np.mean(d ** 2))) / rms_y


@pytest.mark.parametrize("tname", FUSED)
@pytest.mark.parametrize("k,n,m", [(256, 16, 1), (1024, 272, 37), (2048, 64, 128), (5632, 80, 100), (4096, 1040, 128)])
def test_mat_mat_matches_dequantized_product(gpu, pkg, orc, tname, k, n, m):
    """One block / ragged token counts / uneven k-splits (22 blocks) / more than one row group and several splits."""
    t, raw = _weights(pkg, tname, k, n)
    x = np.random.default_rng(k + n + m).standard_normal((m, k)).astype(np.float32)
    w = orc.dequantize(t, raw, k * n).reshape(n, k).astype(np.float64)
    want = x.astype(np.float64) @ w.T
    got = gpu.op_mat_mat(t, raw, x, n)
    rel = _check_gemm(got, want)
    print(f"{tname} k={k} n={n} m={m}: rms(d)/rms(y) = {rel:.2e}")


def test_mat_mat_full_size_rows_are_independent_and_deterministic(gpu, pkg, orc):
    """Llama-3-8B gate shape: token rows do not leak into each other, zero rows give exact zeros, and the result is
    run-to-run identical; spot columns against the oracle's dequantized weights."""
    k, n, m = 4096, 14336, 128
    t, raw = _weights(pkg, "Q4_K", k, n)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((m, k)).astype(np.float32)
    x[5] = 0.0
    y = gpu.op_mat_mat(t, raw, x, n)
    assert np.all(y[5] == 0.0)
    assert np.array_equal(y, gpu.op_mat_mat(t, raw, x, n))
    y1 = gpu.op_mat_mat(t, raw, x[40:41], n)                 # the same token alone (m = 1)
    assert np.array_equal(y1[0], y[40])
    rb = orc.nbytes_for(t, k)
    cols = rng.integers(0, n, 48)
    w = np.stack([orc.dequantize(t, raw[j * rb:(j + 1) * rb], k) for j in cols]).astype(np.float64)
    _check_gemm(y[:, cols], x.astype(np.float64) @ w.T)


def _pair(pkg, orc, name, mix, max_seq, flags_b=0):
    cfg = pkg.make_config(name, max_seq_len=max_seq)
    model = pkg.SynthModel(cfg, mix=mix)
    ref = orc.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors(keep=True):
        ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    exact = pkg.HipGpuInference.from_model(model, max_seq, flags=pkg.hip_backend.FLAG_EXACT_PREFILL)
    batched = pkg.HipGpuInference.from_model(model, max_seq, flags=flags_b)
    return cfg, ref, exact, batched


def _tol(want):
    return 1e-2 * float(np.abs(want).max()) + 1e-2


@pytest.mark.parametrize("name,mix,n_prompt", [("test-dense", "Q4_K_M", 41), ("test-dense", "Q8_0", 128), ("test-dense", "Q4_0", 17),
                                               ("test-dense", "Q5_K_M", 64), ("test-dense", "Q6_K", 33),
                                               ("test-dense-d128", "Q4_K_M", 150), ("test-dense-d128", "Q5_K_M", 129),
                                               ("test-moe", "Q5_K_M", 70), ("test-moe", "Q4_K_M", 140)])
def test_batched_prefill_matches_token_by_token_and_oracle(pkg, orc, name, mix, n_prompt):
    """The KV cache left by the batched path, seen through the logits of the following tokens.  The MoE cases group the
    block's tokens by expert; a token whose two best router logits are nearly tied may pick a different expert than the
    exact path does (the router sees f16-rounded history) — the logits tolerance covers what that does on these models."""
    cfg, ref, exact, batched = _pair(pkg, orc, name, mix, max_seq=256)
    assert batched.prefill_is_batched() and not exact.prefill_is_batched()
    prompt = [(7 * i + 3) % cfg.vocab_size for i in range(n_prompt)]
    exact.forward_batch(prompt[:-1])
    batched.forward_batch(prompt[:-1])
    assert exact.position() == batched.position() == n_prompt - 1
    want_o = ref.forward(prompt)
    errs, gaps = [], []
    tok = prompt[-1]
    for step in range(12):
        le, lb = exact.forward(tok), batched.forward(tok)
        if step == 0:
            assert np.abs(lb - want_o).max() <= _tol(want_o)      # against the CPU oracle, not only our own exact path
        errs.append(float(np.abs(lb - le).max()))
        assert errs[-1] <= _tol(le)
        srt = np.sort(le)
        gaps.append(float(srt[-1] - srt[-2]))
        te, tb = orc.argmax_last(le), orc.argmax_last(lb)
        if gaps[-1] > 4 * errs[-1]:
            assert te == tb, f"greedy token diverged with gap {gaps[-1]:.3e} vs err {errs[-1]:.3e}"
        tok = te
    print(f"{name}/{mix} prompt {n_prompt}: max|dlogit|={max(errs):.3e} tol={_tol(le):.3e} min_gap={min(gaps):.3e}")
    exact.close()
    batched.close()


def test_batched_prefill_continues_an_existing_sequence(pkg, orc):
    """forward_batch at position > 0 (a second turn): causal attention over the block sees the earlier cache rows."""
    cfg, ref, exact, batched = _pair(pkg, orc, "test-dense", "Q4_K_M", max_seq=128)
    a, b = [(5 * i + 1) % cfg.vocab_size for i in range(20)], [(11 * i + 2) % cfg.vocab_size for i in range(30)]
    for eng in (exact, batched):
        eng.forward_batch(a)
        eng.forward(77)
        eng.forward_batch(b)
        assert eng.position() == 51
    le, lb = exact.forward(9), batched.forward(9)
    assert np.abs(lb - le).max() <= _tol(le)
    batched.reset()
    exact.reset()
    exact.forward_batch(a)
    batched.forward_batch(a)                                   # after reset: rows 0.. are rewritten
    le, lb = exact.forward(9), batched.forward(9)
    assert np.abs(lb - le).max() <= _tol(le)
    exact.close()
    batched.close()


def test_models_outside_the_batched_path_prefill_exactly(pkg, orc):
    """Format mixes without tile layouts (here Q5_0: dequantized to f32 at upload) keep the token-by-token path:
    bit-identical to prefill_token."""
    cfg = pkg.make_config("test-dense", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q5_0")
    a, b = pkg.HipGpuInference.from_model(model, 64), pkg.HipGpuInference.from_model(model, 64)
    assert not a.prefill_is_batched()
    prompt = [3, 500, 41, 7, 900, 12]
    a.forward_batch(prompt)
    for t in prompt:
        b.prefill_token(t)
    assert np.array_equal(a.forward(5), b.forward(5))
    a.close()
    b.close()


def test_batched_prefill_rejects_a_prompt_beyond_the_cache(pkg):
    cfg = pkg.make_config("test-dense", max_seq_len=32)
    eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix="Q4_K_M"), 32)
    eng.forward_batch([1, 2, 3])
    with pytest.raises(pkg.BackendError) as ei:
        eng.forward_batch(list(range(40)))
    assert ei.value.variant == "InvalidArgument"
    assert eng.position() == 3                                  # nothing was written
    with pytest.raises(pkg.BackendError):
        eng.forward_batch([1, cfg.vocab_size + 5])
    assert eng.position() == 3
    eng.close()


def test_full_size_llama3_8b_batched_prefill_matches_the_exact_engine(pkg, orc):
    """BASELINE.json's headline configuration at FULL size, bench protocol: the 127 prompt tokens of `bench.py` through the
    batched path on one context, the same tokens fed one by one through the decode kernels on another (the same engine
    instance cannot hold both caches).  The logits of the last prompt token and of 8 greedy steps must agree within the
    stated tolerance; greedy tokens must be identical wherever the exact engine's top-1/top-2 gap exceeds 4x the error.
    Then the size-independent properties: a second batched pass over the same prompt is bit-identical (deterministic
    split-K order), and a 300-token prompt (3 blocks: 128 + 128 + 44) leaves position 300."""
    cfg = pkg.make_config("llama-3-8b", max_seq_len=384)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    batched = pkg.HipGpuInference.from_model(model, 384)
    try:
        assert batched.prefill_is_batched()
        prompt = [i % 32000 % cfg.vocab_size for i in range(128)]           # main.rs:1787
        batched.forward_batch(prompt[:-1])
        lb = [batched.forward(prompt[-1])]
        toks = []
        for _ in range(8):
            toks.append(orc.argmax_last(lb[-1]))
            lb.append(batched.forward(toks[-1]))
        batched.reset()
        batched.forward_batch(prompt[:-1])
        again = batched.forward(prompt[-1])
        assert np.array_equal(again, lb[0])
        batched.reset()
        batched.forward_batch([(3 * i + 1) % cfg.vocab_size for i in range(300)])
        assert batched.position() == 300
    finally:
        batched.close()
    exact = pkg.HipGpuInference.from_model(model, 384, flags=pkg.hip_backend.FLAG_EXACT_PREFILL)
    try:
        assert not exact.prefill_is_batched()
        exact.forward_batch(prompt[:-1])
        le = exact.forward(prompt[-1])
        errs, gaps = [], []
        for i in range(9):
            errs.append(float(np.abs(lb[i] - le).max()))
            assert errs[-1] <= _tol(le), f"step {i}: max|dlogit| {errs[-1]:.3e} > {_tol(le):.3e}"
            srt = np.sort(le)
            gaps.append(float(srt[-1] - srt[-2]))
            if i < 8:
                if gaps[-1] > 4 * errs[-1]:
                    assert orc.argmax_last(le) == toks[i], f"step {i}: greedy token diverged, gap {gaps[-1]:.3e} err {errs[-1]:.3e}"
                le = exact.forward(toks[i])                                 # both engines are fed the batched engine's tokens
        print(f"llama-3-8b Q4_K_M, 127-token prompt: max|dlogit|={max(errs):.3e} tol={_tol(le):.3e} min_gap={min(gaps):.3e}")
    finally:
        exact.close()


@pytest.mark.parametrize("neox", [0, 1])
def test_batched_prefill_with_biases_and_neox_rope(pkg, orc, neox):
    """Qwen2-style layers: q/k/v biases (layers.rs:438-470) and NeoX pairing (i, i + d/2) of the rotation (ops.rs:1316-1331),
    against the CPU oracle and the exact engine."""
    cfg = pkg.make_config("test-dense-d128", max_seq_len=96, use_neox_rope=neox)
    model = pkg.SynthModel(cfg, mix="Q4_K_M", with_bias=True)
    ref = orc.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors(keep=True):
        ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    exact = pkg.HipGpuInference.from_model(model, 96, flags=pkg.hip_backend.FLAG_EXACT_PREFILL)
    batched = pkg.HipGpuInference.from_model(model, 96)
    assert batched.prefill_is_batched()
    prompt = [(9 * i + 4) % cfg.vocab_size for i in range(50)]
    exact.forward_batch(prompt[:-1])
    batched.forward_batch(prompt[:-1])
    le, lb, lo = exact.forward(prompt[-1]), batched.forward(prompt[-1]), ref.forward(prompt)
    assert np.abs(le - lo).max() <= 2e-3 * np.abs(lo).max() + 2e-3          # the exact path: the decode tolerance
    assert np.abs(lb - lo).max() <= _tol(lo) and np.abs(lb - le).max() <= _tol(le)
    exact.close()
    batched.close()


@pytest.mark.parametrize("name,mix,split", [("test-dense-d128", "Q4_K_M", 2), ("test-dense-d128", "Q4_K_M", 1), ("test-moe", "Q5_K_M", 1)])
def test_stage_blocks_reproduce_the_single_context_prompt_pass(pkg, name, mix, split):
    """lgh_stage_prefill_batch: two stage contexts on one GPU, the [n][hidden] f32 block handed over device to device,
    leave the same K/V rows as one full context (bitwise: the same kernels see the same f32 block), seen through the
    logits of the decode steps that follow; 150 tokens = two blocks."""
    import ctypes as C
    cfg = pkg.make_config(name, max_seq_len=192)
    model = pkg.SynthModel(cfg, mix=mix)
    full = pkg.HipGpuInference.from_model(model, 192)
    s0 = pkg.HipGpuInference.from_model(model, 192, layer_range=(0, split))
    s1 = pkg.HipGpuInference.from_model(model, 192, layer_range=(split, cfg.num_layers))
    assert s0.prefill_is_batched() and s1.prefill_is_batched()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    prompt = [(5 * i + 2) % cfg.vocab_size for i in range(150)]
    full.forward_batch(prompt)
    for i in range(0, len(prompt), 128):
        chunk = prompt[i:i + 128]
        s0.stage_prefill_batch(chunk, len(chunk))
        s0.synchronize()
        assert hip.hipMemcpy(s1.stage_hidden_block_ptr(), s0.stage_hidden_block_ptr(), len(chunk) * cfg.hidden_size * 4, 3) == 0  # D2D
        assert hip.hipDeviceSynchronize() == 0                   # a device-to-device hipMemcpy may return before it is done
        s1.stage_prefill_batch(None, len(chunk))
        s1.synchronize()                                         # the next block overwrites s1's input block (own stream)
    assert s0.position() == s1.position() == full.position() == 150
    for tok in (3, 4, 5):
        want = full.forward(tok)
        s0.stage_forward(tok)
        s0.synchronize()
        assert hip.hipMemcpy(s1.stage_hidden_ptr(), s0.stage_hidden_ptr(), cfg.hidden_size * 4, 3) == 0
        assert hip.hipDeviceSynchronize() == 0
        got = s1.stage_forward(0, want_logits=True)
        assert np.array_equal(got, want)
    with pytest.raises(pkg.BackendError):
        s0.stage_prefill_batch(list(range(129)), 129)            # a block holds at most 128 tokens
    with pytest.raises(pkg.BackendError):
        s0.stage_prefill_batch(None, 4)                          # the first stage needs the ids
    for e in (full, s0, s1):
        e.close()
