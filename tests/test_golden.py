"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py with the CPU oracle).

CPU half: the oracle keeps reproducing its own pinned outputs bit-for-bit (the fixtures were generated with the
AVX-512 lane order of `dot_f32` forced, so they do not depend on the host CPU) and the synthetic generator keeps
producing the same block bytes.  GPU half (-m gpu): the HIP path, through the C ABI, against the same numbers — on the
GPU box neither /root/reference nor a regenerated expectation is involved, only the committed data.

Tolerances as in test_gpu_ops.py / test_gpu_model.py."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QUANT = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
FUSED = ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "Q4_0"]
MODELS = [("test-dense", "Q4_K_M"), ("test-moe", "Q5_K_M"), ("test-dense", "Q8_0")]


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ops_v1.npz"))      # allow_pickle stays False


@pytest.fixture()
def orc512(orc):
    prev = orc.get_isa()
    orc.set_isa(3)
    yield orc
    orc.set_isa(prev)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------------ CPU: oracle and generator
@pytest.mark.parametrize("tname", QUANT)
def test_oracle_reproduces_quant_fixtures(pkg, orc512, ops, tname):
    t = pkg.synth.TYPE_IDS[tname]
    k, n = 512, 6
    raw = pkg.synth.fill_tensor("blk.0.golden.weight", t, k * n, k)
    assert np.array_equal(raw, ops[f"{tname}.raw"]), "synthetic block generator changed"
    assert np.array_equal(_bits(orc512.dequantize(t, raw, k * n)), _bits(ops[f"{tname}.dequant"]))
    assert np.array_equal(_bits(orc512.vec_mat_q(t, raw, ops["x"], n)), _bits(ops[f"{tname}.vec_mat"]))
    q = orc512.quantize(t, ops["ramp"])
    assert np.array_equal(q, ops[f"{tname}.ramp_blocks"])
    assert np.array_equal(_bits(orc512.dequantize(t, q, 256)), _bits(ops[f"{tname}.ramp_roundtrip"]))


def test_oracle_reproduces_op_fixtures(orc512, ops):
    o = orc512
    assert np.array_equal(_bits(o.rms_norm(ops["x"], ops["norm_w"], 1e-5)), _bits(ops["rms_norm"]))
    q, k = ops["rope_in.q"], ops["rope_in.k"]
    for neox in (0, 1):
        rq, rk = o.rope(q[:, None, :], k[:, None, :], 17, 10000.0, 1.0, bool(neox))
        assert np.array_equal(_bits(rq[:, 0]), _bits(ops[f"rope{neox}.q"])) and np.array_equal(_bits(rk[:, 0]), _bits(ops[f"rope{neox}.k"]))
    d = q.shape[1]
    got = o.attention_cached(ops["attn.q"], ops["attn.k_cache"], ops["attn.v_cache"], 1.0 / np.sqrt(d), int(ops["attn.kv_len"][0]))
    assert np.array_equal(_bits(got), _bits(ops["attn.out"]))
    assert np.array_equal(_bits(o.silu_mul(ops["silu.gate"], ops["silu.up"])), _bits(ops["silu.out"]))


@pytest.mark.parametrize("name,mix", MODELS)
def test_oracle_reproduces_model_fixture(pkg, orc512, name, mix):
    fx = np.load(os.path.join(GOLD, f"model_{name}_{mix}_v1.npz"))
    cfg = pkg.make_config(name)
    model = pkg.SynthModel(cfg, mix=mix)
    ref = orc512.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors():
        ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    lg = ref.forward(fx["prompt"].tolist())
    assert np.array_equal(_bits(lg), _bits(fx["logits"][0]))
    for i, tok in enumerate(fx["tokens"][:-1]):
        assert orc512.argmax_last(lg) == int(tok)
        lg = ref.forward([int(tok)])
        assert np.array_equal(_bits(lg), _bits(fx["logits"][i + 1]))
    ref.close()


# ------------------------------------------------------------------------------------------ GPU: the HIP path vs the data
@pytest.mark.gpu
@pytest.mark.parametrize("tname", QUANT)
def test_hip_dequantize_matches_golden(gpu, pkg, ops, tname):
    t = pkg.synth.TYPE_IDS[tname]
    got = gpu.op_dequantize(t, ops[f"{tname}.raw"], 512 * 6)
    assert np.array_equal(_bits(got), _bits(ops[f"{tname}.dequant"]))
    got = gpu.op_dequantize(t, ops[f"{tname}.ramp_blocks"], 256)
    assert np.array_equal(_bits(got), _bits(ops[f"{tname}.ramp_roundtrip"]))


@pytest.mark.gpu
@pytest.mark.parametrize("tname", QUANT)
def test_hip_vec_mat_matches_golden(gpu, pkg, ops, tname):
    t = pkg.synth.TYPE_IDS[tname]
    x, w = ops["x"], ops[f"{tname}.dequant"].reshape(6, 512).astype(np.float64)
    bound = 1e-4 * (np.abs(w) @ np.abs(x.astype(np.float64))) + 1e-6
    got = gpu.op_vec_mat(t, ops[f"{tname}.raw"], x, 6)
    assert np.all(np.abs(got - ops[f"{tname}.vec_mat"]) <= bound)


@pytest.mark.gpu
def test_hip_ops_match_golden(gpu, ops):
    got = gpu.op_rms_norm(ops["x"], ops["norm_w"], 1e-5)
    assert np.abs(got - ops["rms_norm"]).max() <= 2e-6 * np.abs(ops["rms_norm"]).max()
    for neox in (0, 1):
        gq, gk = gpu.op_rope(ops["rope_in.q"], ops["rope_in.k"], 17, 10000.0, 1.0, bool(neox))
        assert np.array_equal(_bits(gq), _bits(ops[f"rope{neox}.q"])) and np.array_equal(_bits(gk), _bits(ops[f"rope{neox}.k"]))
    d = ops["attn.q"].shape[1]
    want = ops["attn.out"]
    for splits in (1, 4):
        got = gpu.op_attention_cached(ops["attn.q"], ops["attn.k_cache"], ops["attn.v_cache"], 1.0 / np.sqrt(np.float32(d)),
                                      int(ops["attn.kv_len"][0]), splits)
        assert np.abs(got - want).max() <= 2e-5 * (1 + np.abs(want).max())
    got = gpu.op_silu_mul(ops["silu.gate"], ops["silu.up"])
    assert np.abs(got - ops["silu.out"]).max() <= 4e-7 * (1 + np.abs(ops["silu.out"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name,mix", MODELS)
def test_hip_model_matches_golden(gpu, pkg, name, mix):
    """Logits within the stated tolerance at every step (the fixture's tokens are fed, so the sequences stay aligned);
    greedy token identical wherever the fixture's top-1/top-2 gap exceeds 4x the measured error."""
    fx = np.load(os.path.join(GOLD, f"model_{name}_{mix}_v1.npz"))
    cfg = pkg.make_config(name, max_seq_len=64)
    eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=mix), 64)
    wrap, ctx = pkg.GpuModelWrapper(eng), pkg.InferenceContext()
    feed = [fx["prompt"].tolist()] + [[int(t)] for t in fx["tokens"][:-1]]
    for i, toks in enumerate(feed):
        got, want = wrap.forward(toks, ctx), fx["logits"][i]
        err = float(np.abs(got - want).max())
        assert err <= 2e-3 * float(np.abs(want).max()) + 2e-3
        if float(fx["top_gap"][i]) > 4 * err:
            assert int(np.flatnonzero(got == got.max())[-1]) == int(fx["tokens"][i])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,mix", MODELS)
def test_hip_batched_prompt_pass_matches_golden(gpu, pkg, name, mix):
    """The same fixtures through `forward_batch` (the f16-GEMM prompt pass) + `forward`: the prompt's logits
    and the steps after it within the prompt-pass tolerance 1e-2 * max|logit| + 1e-2 of the committed CPU-oracle vectors."""
    fx = np.load(os.path.join(GOLD, f"model_{name}_{mix}_v1.npz"))
    cfg = pkg.make_config(name, max_seq_len=64)
    eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=mix), 64)
    prompt = fx["prompt"].tolist()
    eng.forward_batch(prompt[:-1])
    feed = [prompt[-1]] + [int(t) for t in fx["tokens"][:-1]]
    for i, tok in enumerate(feed):
        got, want = eng.forward(tok), fx["logits"][i]
        err = float(np.abs(got - want).max())
        assert err <= 1e-2 * float(np.abs(want).max()) + 1e-2
        if float(fx["top_gap"][i]) > 4 * err:
            assert int(np.flatnonzero(got == got.max())[-1]) == int(fx["tokens"][i])
    eng.close()


def test_oracle_kv_shift_left_is_a_row_move(orc, pkg):
    """KVCache::shift_left (model/mod.rs:142-172): after dropping the first `amount` rows the model behaves as if the
    remaining rows had been written at positions 0.. — checked against a fresh model whose cache rows are produced the
    same way is not possible (RoPE keeps the old rotation), so the property tested is idempotence + position arithmetic
    and that a shift followed by decoding is deterministic."""
    cfg = pkg.make_config("test-dense", max_seq_len=48)
    model = pkg.SynthModel(cfg, mix="Q8_0")
    outs = []
    for _ in range(2):
        ref = orc.Model(cfg.as_dict())
        for nm, t, ne, data in model.tensors(keep=True):
            ref.add_tensor(nm, t, ne, data)
        ref.finalize()
        ref.forward(list(range(3, 23)))
        ref.kv_shift_left(8)
        assert ref.position == 12
        outs.append(ref.forward([5]))
        ref.kv_truncate(4)
        assert ref.position == 4
        ref.kv_truncate(40)
        assert ref.position == 4
        ref.kv_shift_left(0)
        assert ref.position == 0
        ref.close()
    assert np.array_equal(outs[0], outs[1])
