"""The synthetic-model generator: valid blocks, deterministic bytes, sane statistics, and the byte totals that
BASELINE.md's roofline denominators were computed from."""
import numpy as np
import pytest


@pytest.mark.parametrize("tname", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "F32", "F16"])
def test_generated_blocks_are_valid_and_scaled(pkg, orc, tname):
    t = pkg.synth.TYPE_IDS[tname]
    k, n = 1024, 64
    raw = pkg.synth.fill_tensor("blk.0.test.weight", t, k * n, k)
    assert raw.nbytes == orc.nbytes_for(t, k * n)
    w = orc.dequantize(t, raw, k * n)
    assert np.all(np.isfinite(w))
    target = 1.0 / np.sqrt(k)
    assert 0.8 * target < w.std() < 1.25 * target, (w.std(), target)
    assert abs(w.mean()) < 0.15 * target
    # deterministic, name-keyed, thread-count independent
    again = pkg.synth.fill_tensor("blk.0.test.weight", t, k * n, k, threads=3)
    assert np.array_equal(raw, again)
    other = pkg.synth.fill_tensor("blk.1.test.weight", t, k * n, k)
    assert not np.array_equal(raw, other)


def test_norm_and_bias_kinds(pkg):
    w = pkg.synth.fill_tensor("blk.0.attn_norm.weight", pkg.synth.F32, 512, 512, kind=1).view(np.float32)
    assert np.all(np.abs(w - 1.0) <= 0.01 + 1e-7)
    b = pkg.synth.fill_tensor("blk.0.attn_q.bias", pkg.synth.F32, 512, 512, kind=2).view(np.float32)
    assert np.all(np.abs(b) <= 0.01 + 1e-7)


def test_mix_rules_and_byte_totals(pkg):
    # BASELINE.md §2 / SURVEY.md §8(d): algorithmic bytes per decoded token
    def step(name, mix, kv):
        return pkg.SynthModel(pkg.make_config(name), mix=mix).step_alg_bytes(kv) / 1e9
    assert abs(step("llama-3-8b", "Q4_K_M", 128) - (4.617 + 0.0336)) < 0.005
    assert abs(step("llama-3-8b", "Q4_K_M", 256) - (4.617 + 0.0671)) < 0.005
    assert abs(step("tinyllama-1.1b", "Q8_0", 128) - (1.099 + 0.0058)) < 0.003
    assert abs(step("tinyllama-1.1b", "Q4_K_M", 128) - (0.630 + 0.0058)) < 0.003
    assert abs(step("mixtral-8x7b", "Q5_K_M", 128) - (9.045 + 0.0336)) < 0.01
    assert abs(step("llama-3-70b", "Q4_K_M", 128) - (41.88 + 0.084)) < 0.02
    m = pkg.SynthModel(pkg.make_config("llama-3-8b"), mix="Q4_K_M")
    types = {s.name: s.ggml_type for s in m.specs()}
    assert types["output.weight"] == pkg.synth.Q6_K and types["blk.0.attn_v.weight"] == pkg.synth.Q6_K
    assert types["blk.4.attn_v.weight"] == pkg.synth.Q4_K and types["blk.6.ffn_down.weight"] == pkg.synth.Q6_K
    assert types["blk.10.ffn_gate.weight"] == pkg.synth.Q4_K


def test_moe_specs(pkg):
    m = pkg.SynthModel(pkg.make_config("test-moe"), mix="Q5_K_M")
    names = {s.name: s for s in m.specs()}
    g = names["blk.0.ffn_gate_exps.weight"]
    assert g.ne == (512, 768, 4) and names["blk.0.ffn_gate_inp.weight"].ggml_type == pkg.synth.F32
    assert names["blk.0.ffn_down_exps.weight"].ne == (768, 512, 4)
