"""Multi-sequence decode (lgh_forward_multi; the device side of the reference's BatchedEngine, src/engine_batched.rs:23-194,
200-330, 355-400) against the single-sequence engine.

The reference's batched loop calls `model.forward` for each active sequence in turn, every sequence with its own
InferenceContext — so what a sequence gets from a batched step must be exactly what the single-sequence path gives it.
Here the weights are read once for all sequences (matvec_batch.hip), and the bar is the strongest one there is: every
sequence's logits are BIT-IDENTICAL to lgh_forward's on the same token history (integer-exact matrix-core sums, the
same summation orders), for B = 1 ... 16, dense and MoE, ragged positions, slots reused.  The single-sequence engine is
pinned to the CPU oracle by tests/test_gpu_model.py; one case here repeats that check on a batched step directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engines(pkg, name, mix, max_seq=96, n_single=1, **kw):
    if name == "test-moe-e1024":
        # expert width 1024 (4 blocks of 256 -> 8-wave plans, like Mixtral's 14336): the step reads every selected expert ONCE for
        # all sequences that chose it (engine_batch.hip); "test-moe" (768: 3 blocks) stays on the sequence-by-sequence MoE path
        name, kw = "test-moe", dict(kw, expert_intermediate_size=1024)
    if name == "test-moe-e1024-k4":   # four of eight experts per token (the grouped step's entry lists hold up to 4 pairs per sequence)
        name, kw = "test-moe", dict(kw, expert_intermediate_size=1024, num_experts=8, num_experts_per_token=4)
    if name == "mixtral-8x7b-3l":
        # Mixtral's own widths (8 experts of 14336, hidden 4096, vocabulary cut to 4096 for the test's run time), three layers — one
        # Q6_K and two Q5_K down projections under Q5_K_M: the expert-grouped step on the launch plans the full model runs with
        name, kw = "mixtral-8x7b", dict(kw, num_layers=3, vocab_size=4096)
    cfg = pkg.make_config(name, max_seq_len=max_seq, **kw)
    model = pkg.SynthModel(cfg, mix=mix)
    multi = pkg.HipGpuInference.from_model(model, max_seq)
    # attn_direct=255: the single-sequence engine switches to a one-launch attention (another summation tree) for contexts of up
    # to 64 rows; the multi-sequence step always runs the split attention + merge, the structure of every longer context.  The
    # bitwise comparison is against THAT structure; against the short-context variant the logits agree to rounding (last test).
    singles = [pkg.HipGpuInference.from_model(model, max_seq, attn_direct=255) for _ in range(n_single)]
    return cfg, model, multi, singles


def _history(cfg, seq, n):
    rng = np.random.default_rng(1000 + seq)
    return [int(t) for t in rng.integers(0, cfg.vocab_size, size=n)]


@pytest.mark.parametrize("name,mix,B", [("test-dense", "Q4_K_M", 1), ("test-dense", "Q4_K_M", 2), ("test-dense", "Q4_K_M", 5),
                                        ("test-dense-d128", "Q4_K_M", 4), ("test-dense-d128", "Q4_K_M", 8), ("test-dense-d128", "Q4_K_M", 16),
                                        ("test-dense", "Q8_0", 3), ("test-dense", "Q5_K_M", 4), ("test-dense", "Q6_K", 7), ("test-dense", "Q4_0", 2),
                                        ("test-dense", "Q5_K_M", 12), ("test-dense", "Q8_0", 9), ("test-dense-d128", "Q6_K", 16),
                                        ("test-moe", "Q5_K_M", 3), ("test-moe", "Q4_K_M", 16),
                                        ("test-moe-e1024", "Q5_K_M", 3), ("test-moe-e1024", "Q4_K_M", 16), ("test-moe-e1024", "Q8_0", 7),
                                        ("test-moe-e1024-k4", "Q5_K_M", 5), ("test-moe-e1024-k4", "Q4_K_M", 16),
                                        ("mixtral-8x7b-3l", "Q5_K_M", 6), ("mixtral-8x7b-3l", "Q5_K_M", 16)])
def test_every_sequence_gets_the_single_sequence_logits_bitwise(pkg, name, mix, B):
    """B sequences with different histories and RAGGED lengths, token by token through lgh_forward_multi; each sequence's logits
    at every step equal, bit for bit, those of a single-sequence engine fed the same history."""
    cfg, model, multi, (single,) = _engines(pkg, name, mix)
    multi.batch_create(B)
    lens = [3 + (5 * s) % 11 for s in range(B)]                   # sequence s joins the batch `lens[s]` steps before the end
    T = max(lens)
    hist = [_history(cfg, s, lens[s]) for s in range(B)]
    got = [[] for _ in range(B)]
    for step in range(T):
        active = [s for s in range(B) if step >= T - lens[s]]      # ragged: sequences join at different steps (continuous batching)
        toks = [hist[s][step - (T - lens[s])] for s in active]
        logits, nxt = multi.forward_multi(active, toks, want_logits=True, greedy=True)
        for i, s in enumerate(active):
            got[s].append(logits[i].copy())
            assert int(nxt[i]) == int(np.flatnonzero(logits[i] == logits[i].max())[-1])   # greedy rule: the LAST maximal index
    for s in range(B):
        assert multi.batch_position(s) == lens[s]
        single.reset()
        for t, tok in enumerate(hist[s]):
            want = single.forward(tok)
            assert np.array_equal(got[s][t].view(np.uint32), want.view(np.uint32)), (s, t, float(np.abs(got[s][t] - want).max()))
    multi.close()
    single.close()


def test_batched_step_matches_the_cpu_oracle(pkg, orc):
    """The batched path against the CPU oracle directly (not only through the single-sequence engine): 4 sequences, logits within
    the decode tolerance of SURVEY.md 8c at every step."""
    cfg = pkg.make_config("test-dense-d128", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    eng = pkg.HipGpuInference.from_model(model, 64)
    eng.batch_create(4)
    refs = []
    for s in range(4):
        ref = orc.Model(cfg.as_dict())
        for nm, t, ne, data in model.tensors(keep=True):
            ref.add_tensor(nm, t, ne, data)
        ref.finalize()
        refs.append(ref)
    hist = [_history(cfg, 40 + s, 6) for s in range(4)]
    worst = 0.0
    for t in range(6):
        logits, _ = eng.forward_multi([0, 1, 2, 3], [hist[s][t] for s in range(4)])
        for s in range(4):
            want = refs[s].forward([hist[s][t]])
            err = float(np.abs(logits[s] - want).max())
            worst = max(worst, err)
            assert err <= 2e-3 * float(np.abs(want).max()) + 2e-3
    print(f"batched step vs oracle: max|dlogit| = {worst:.3e}")
    eng.close()


def test_slots_prefill_reset_and_device_fed_greedy_loop(pkg):
    """Slots are independent caches: a slot's prompt goes through the batched prompt path (lgh_batch_prefill) and leaves the rows
    lgh_prefill_batch would; lgh_decode_greedy_multi (tokens fed back on the device) equals a host loop over lgh_forward_multi and
    the single-sequence greedy decode; a reset slot starts over while the others keep going."""
    cfg, model, multi, singles = _engines(pkg, "test-dense-d128", "Q4_K_M", max_seq=160, n_single=1)
    single = singles[0]
    B = 3
    multi.batch_create(4)
    prompts = [_history(cfg, 70 + s, 20 + 7 * s) for s in range(B)]
    slots = [3, 0, 2]                                                # not the identity, one slot left unused
    for s in range(B):
        multi.batch_prefill(slots[s], prompts[s][:-1])
        assert multi.batch_position(slots[s]) == len(prompts[s]) - 1
    dev = multi.decode_greedy_multi(slots, [p[-1] for p in prompts], 12)          # [12, B]
    want = []
    for s in range(B):
        single.reset()
        single.forward_batch(prompts[s][:-1])
        want.append(single.decode_greedy(prompts[s][-1], 12))
    for s in range(B):
        assert np.array_equal(dev[:, s], want[s]), (s, dev[:, s], want[s])
        assert multi.batch_position(slots[s]) == len(prompts[s]) - 1 + 12
    # host loop on fresh slots == device loop
    for s in range(B):
        multi.batch_reset(slots[s])
        multi.batch_prefill(slots[s], prompts[s][:-1])
    toks = [p[-1] for p in prompts]
    for step in range(12):
        _, nxt = multi.forward_multi(slots, toks, want_logits=False, greedy=True)
        assert np.array_equal(nxt, dev[step])
        toks = [int(t) for t in nxt]
    # one slot starts over with another prompt while the others continue
    multi.batch_reset(slots[1])
    assert multi.batch_position(slots[1]) == 0
    p2 = _history(cfg, 99, 9)
    multi.batch_prefill(slots[1], p2[:-1])
    l_multi, _ = multi.forward_multi([slots[1], slots[0]], [p2[-1], toks[0]])
    single.reset()
    single.forward_batch(p2[:-1])
    assert np.array_equal(l_multi[0].view(np.uint32), single.forward(p2[-1]).view(np.uint32))
    multi.close()
    single.close()


def test_errors_leave_the_slots_unchanged(pkg):
    cfg, model, multi, (single,) = _engines(pkg, "test-dense", "Q4_K_M", max_seq=8)
    with pytest.raises(pkg.BackendError) as ei:
        multi.forward_multi([0], [1])                                 # before lgh_batch_create
    assert ei.value.variant == "InvalidArgument"
    multi.batch_create(2)
    for bad_slots, toks in (([0, 0], [1, 2]), ([2], [1]), ([0, 1, 1], [1, 2, 3])):
        with pytest.raises(pkg.BackendError) as ei:
            multi.forward_multi(bad_slots, toks)
        assert ei.value.variant == "InvalidArgument"
    with pytest.raises(pkg.BackendError):
        multi.forward_multi([0], [cfg.vocab_size])                     # llama.rs:296-302
    for t in range(8):
        multi.forward_multi([0], [t])
    with pytest.raises(pkg.BackendError) as ei:                       # the slot is full: nothing moves, the other slot is untouched
        multi.forward_multi([1, 0], [3, 4])
    assert ei.value.variant == "InvalidArgument"
    assert multi.batch_position(0) == 8 and multi.batch_position(1) == 0
    with pytest.raises(pkg.BackendError):
        multi.batch_create(3)                                         # already created with another size
    multi.close()
    single.close()


def test_short_context_variant_of_the_single_engine_agrees_to_rounding(pkg):
    """Below 64 rows the DEFAULT single-sequence engine runs its one-launch attention; the batched step (split attention + merge)
    then differs from it by summation order only."""
    cfg = pkg.make_config("test-dense-d128", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q4_K_M")
    multi = pkg.HipGpuInference.from_model(model, 64)
    single = pkg.HipGpuInference.from_model(model, 64)
    multi.batch_create(2)
    hist = [_history(cfg, 7, 20), _history(cfg, 8, 20)]
    worst = 0.0
    for t in range(20):
        logits, _ = multi.forward_multi([0, 1], [hist[0][t], hist[1][t]])
        want = single.forward(hist[0][t])
        worst = max(worst, float(np.abs(logits[0] - want).max()) / float(np.abs(want).max()))
    assert worst <= 1e-5, worst
    multi.close()
    single.close()


def test_long_context_slots_use_the_long_context_attention_and_stay_bitwise(pkg):
    """max_seq >= 2048 switches the split attention to its 8-wave variant (attention.hip), also for the multi-sequence launch: two
    slots with prompts of 2100 and 37 tokens (batched prompt path), then 6 steps; each sequence's logits equal the single-sequence
    engine's bit for bit, and the device-fed greedy loop equals its greedy decode."""
    cfg, model, multi, (single,) = _engines(pkg, "test-dense-d128", "Q4_K_M", max_seq=2304)
    multi.batch_create(2)
    prompts = [_history(cfg, 300, 2100), _history(cfg, 301, 37)]
    for s in range(2):
        multi.batch_prefill(s, prompts[s][:-1])
    toks = [p[-1] for p in prompts]
    got = []
    for step in range(6):
        logits, nxt = multi.forward_multi([0, 1], toks, want_logits=True, greedy=True)
        got.append(logits.copy())
        toks = [int(t) for t in nxt]
    for s in range(2):
        single.reset()
        single.forward_batch(prompts[s][:-1])
        tok = prompts[s][-1]
        for step in range(6):
            want = single.forward(tok)
            assert np.array_equal(got[step][s].view(np.uint32), want.view(np.uint32)), (s, step, float(np.abs(got[step][s] - want).max()))
            tok = int(np.flatnonzero(want == want.max())[-1])
        assert multi.batch_position(s) == len(prompts[s]) - 1 + 6
    multi.close()
    single.close()


@pytest.mark.parametrize("name,mix,B,kv", [("test-dense-d128", "Q4_K_M", 3, "tq3"), ("test-dense", "Q4_K_M", 5, "tq2"),
                                           ("test-dense-d128", "Q4_K_M", 4, "tq2-qjl"), ("test-moe", "Q5_K_M", 2, "tq3-qjl")])
def test_turboquant_slots_bitwise(pkg, name, mix, B, kv):
    """Multi-sequence decode over TurboQuant KV caches (kv_cache_type LGH_KV_TQ*; the reference's BatchedEngine only ever creates f32
    caches, engine_batched.rs:355-357 -> model.create_context): every slot holds its own code rows (+ QJL rows), the step's rotated
    K / V rows are compressed by the attention launch of their sequence; each sequence's logits equal the single-sequence TurboQuant
    engine's BIT FOR BIT (same kernels, the sequence as the grid's second dimension), prompts fed token by token."""
    hb = pkg.hip_backend
    kvt = {"tq2": hb.KV_TQ2, "tq3": hb.KV_TQ3, "tq2-qjl": hb.KV_TQ2_QJL, "tq3-qjl": hb.KV_TQ3_QJL}[kv]
    cfg = pkg.make_config(name, max_seq_len=64)
    model = pkg.SynthModel(cfg, mix=mix)
    rng = np.random.default_rng(23)
    signs = np.where(rng.integers(0, 2, cfg.num_layers * cfg.num_kv_heads * 2 * cfg.head_dim) == 1, 1.0, -1.0).astype(np.float32)
    qjl = rng.standard_normal(cfg.num_layers * cfg.num_kv_heads * cfg.head_dim * cfg.head_dim).astype(np.float32) if "qjl" in kv else None
    multi = pkg.HipGpuInference.from_model(model, 64, kv_cache_type=kvt, kv_rotation_signs=signs, kv_qjl_matrices=qjl)
    single = pkg.HipGpuInference.from_model(model, 64, kv_cache_type=kvt, kv_rotation_signs=signs, kv_qjl_matrices=qjl)
    try:
        multi.batch_create(B)
        lens = [4 + (3 * s) % 7 for s in range(B)]
        hist = [_history(cfg, 500 + s, lens[s]) for s in range(B)]
        for s in range(B):
            multi.batch_prefill(s, hist[s][:-1])                           # (token by token: no batched prompt path over a code cache)
            assert multi.batch_position(s) == lens[s] - 1
        toks = [h[-1] for h in hist]
        got = []
        for step in range(5):
            logits, nxt = multi.forward_multi(list(range(B)), toks, want_logits=True, greedy=True)
            got.append(logits.copy())
            toks = [int(t) for t in nxt]
        for s in range(B):
            single.reset()
            for t in hist[s][:-1]:
                single.prefill_token(t)
            tok = hist[s][-1]
            for step in range(5):
                want = single.forward(tok)
                assert np.array_equal(got[step][s].view(np.uint32), want.view(np.uint32)), (kv, s, step, float(np.abs(got[step][s] - want).max()))
                tok = int(np.flatnonzero(want == want.max())[-1])
        f32 = pkg.HipGpuInference.from_model(model, 64)
        f32.batch_create(B)
        assert multi.stats()["kv_bytes"] < f32.stats()["kv_bytes"] // 4       # the slots' caches are the compressed ones
        f32.close()
    finally:
        multi.close()
        single.close()
