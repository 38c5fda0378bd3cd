"""The per-op `Backend` trait surface (src/backend/mod.rs:29-265) on the MI355X: `HipBackend` over the C ABI against the
CPU oracle's restatement of src/backend/cpu/ops.rs on the same seeded inputs.

Tolerances: add / mul / scale and the f32 matmul are bit-exact (same single roundings, same ascending-k order); silu /
gelu differ by the device exp / tanh only (5e-7 of the result scale); softmax additionally sums the row as a tree
where the CPU sums sequentially (1e-5 relative); mat-vecs as in
tests/test_gpu_ops.py; attention 2e-5 absolute on O(1) values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be(pkg, gpu):
    b = pkg.HipBackend()
    assert b.name() == "hip" and b.is_available()
    yield b
    b.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("n", [1, 4, 1000, 300_001])
def test_add_mul_scale_bit_exact(be, orc, n):
    rng = np.random.default_rng(n)
    a, b = rng.standard_normal(n).astype(np.float32) * 7, rng.standard_normal(n).astype(np.float32) / 3
    assert np.array_equal(_bits(be.add(a, b)), _bits(orc.add(a, b)))
    assert np.array_equal(_bits(be.mul(a, b)), _bits(orc.mul(a, b)))
    assert np.array_equal(_bits(be.scale(a, 2.5)), _bits(orc.scale(a, 2.5)))


def test_reference_kats_through_the_device(be):
    """ops.rs:1567-1680 run on the GPU: the reference's own expected values."""
    assert np.array_equal(be.add([1, 2, 3, 4], [10, 20, 30, 40]), np.float32([11, 22, 33, 44]))
    assert np.array_equal(be.mul([1, 2, 3, 4], [2, 3, 4, 5]), np.float32([2, 6, 12, 20]))
    assert np.array_equal(be.scale([1, 2, 3, 4], 2.5), np.float32([2.5, 5, 7.5, 10]))
    assert np.array_equal(be.matmul(np.float32([[1, 2, 3], [4, 5, 6]]), np.float32([[1, 2], [3, 4], [5, 6]])), np.float32([[22, 28], [49, 64]]))
    assert np.array_equal(be.matvec(np.arange(1, 13, dtype=np.float32).reshape(3, 4), [1, 2, 3, 4]), np.float32([30, 70, 110]))
    s = be.silu([0.0, 1.0, -1.0, 2.0])
    assert abs(s[0]) < 1e-6 and abs(s[1] - 0.731) < 0.01 and abs(s[2] + 0.269) < 0.01
    p = be.softmax([1.0, 2.0, 3.0, 4.0])
    assert abs(float(p.sum()) - 1.0) < 1e-6 and p[0] < p[1] < p[2] < p[3]
    q = np.float32([1, 0, 0, 0, 0, 1, 0, 0]).reshape(1, 2, 4)
    v = np.float32([1, 2, 3, 4, 5, 6, 7, 8]).reshape(1, 2, 4)
    out = be.attention(q, q.copy(), v, 1.0 / np.sqrt(2.0))
    assert np.allclose(out[0, 0], v[0, 0], atol=1e-6)


def test_activations_and_softmax(be, orc):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 4, np.float32([0.0, -0.0, 30.0, -30.0, 88.0, -88.0])])
    for got, want in ((be.silu(x), orc.silu(x)), (be.gelu(x), orc.gelu(x))):
        assert np.all(np.abs(got - want) <= 5e-7 * np.maximum(np.abs(want), 1.0))
    rows = rng.standard_normal((37, 1000)).astype(np.float32) * 5
    got, want = be.softmax(rows), orc.softmax_rows(rows)
    assert got.shape == rows.shape and np.all(np.abs(got - want) <= 1e-5 * want + 1e-12) and np.allclose(got.sum(axis=1), 1.0, atol=5e-6)
    one = be.softmax(rng.standard_normal(70_000).astype(np.float32))                # a single long row (vocab-sized)
    assert abs(float(one.astype(np.float64).sum()) - 1.0) < 1e-5


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (5, 33, 7), (64, 300, 513), (300, 260, 280)])
def test_matmul_bit_exact(be, orc, m, k, n):
    rng = np.random.default_rng(m + k + n)
    a, b = rng.standard_normal((m, k)).astype(np.float32), rng.standard_normal((k, n)).astype(np.float32)
    assert np.array_equal(_bits(be.matmul(a, b)), _bits(orc.matmul(a, b)))


def test_matvec_and_quantized_matvec(be, pkg, orc):
    rng = np.random.default_rng(6)
    a, x = rng.standard_normal((130, 1024)).astype(np.float32), rng.standard_normal(1024).astype(np.float32)
    want = orc.matvec(a, x)
    bound = 1e-4 * (np.abs(a.astype(np.float64)) @ np.abs(x.astype(np.float64))) + 1e-6
    assert np.all(np.abs(be.matvec(a, x) - want) <= bound)
    assert np.all(np.abs(be.vec_mat(x, a, 130) - want) <= bound)                  # the same bytes read as GGUF-order [k, n]
    for tname in ("Q4_K", "Q6_K", "Q8_0"):
        t = pkg.synth.TYPE_IDS[tname]
        raw = pkg.synth.fill_tensor("blk.0.test.weight", t, 1024 * 48, 1024)
        wq = orc.vec_mat_q(t, raw, x, 48)
        w = orc.dequantize(t, raw, 1024 * 48).reshape(48, 1024).astype(np.float64)
        bq = 1e-4 * (np.abs(w) @ np.abs(x.astype(np.float64))) + 1e-6
        assert np.all(np.abs(be.matvec_q(t, raw, x, 48) - wq) <= bq)
        assert np.array_equal(be.matvec_q(t, raw, x, 48), be.vec_mat_q(x, t, raw, 48))
        assert np.array_equal(be.dequantize(t, raw, 1024 * 48).view(np.uint32), orc.dequantize(t, raw, 1024 * 48).view(np.uint32))


@pytest.mark.parametrize("nh,nkv,seq,kv,d", [(4, 2, 1, 1, 64), (8, 2, 5, 5, 64), (8, 8, 3, 40, 128), (32, 8, 6, 300, 128), (6, 2, 7, 19, 40), (3, 3, 4, 600, 16)])
def test_attention_matches_oracle(be, orc, nh, nkv, seq, kv, d):
    """Prefill-shaped (seq == kv), continuation (seq < kv) and GQA; causal mask as ops.rs:1408-1415."""
    rng = np.random.default_rng(nh + seq + kv)
    q, k, v = (rng.standard_normal(s).astype(np.float32) for s in ((nh, seq, d), (nkv, kv, d), (nkv, kv, d)))
    scale = 1.0 / np.sqrt(d)
    got, want = be.attention(q, k, v, scale), orc.attention(q, k, v, scale)
    assert np.abs(got - want).max() <= 2e-5
    assert np.array_equal(be.flash_attention(q, k, v, scale, True), got)         # backend/mod.rs:159-171: the default forwards
    last = be.attention_cached(q[:, -1, :], k, v, scale, kv)                     # the cached form sees the same rows
    assert np.abs(last - want[:, -1, :]).max() <= 2e-5


def test_named_device_resident_weights(be, pkg, orc):
    """CudaBackend::load_model_weights + the b.name() lookup of vec_mat_q (cuda/mod.rs:121-146, 511-575)."""
    t = pkg.synth.TYPE_IDS["Q4_K"]
    k, n = 2048, 96
    raw = pkg.synth.fill_tensor("blk.3.ffn_gate.weight", t, k * n, k)
    x = np.random.default_rng(8).standard_normal(k).astype(np.float32)
    assert not be.has_weight("blk.3.ffn_gate.weight")
    be.load_weight("blk.3.ffn_gate.weight", t, raw, k, n)
    assert be.has_weight("blk.3.ffn_gate.weight")
    by_name = be.vec_mat_q(x, n=n, name="blk.3.ffn_gate.weight")                  # only x crosses PCIe
    assert np.array_equal(by_name, be.vec_mat_q(x, t, raw, n))                    # same kernel, same bits as the host-tensor path
    assert np.array_equal(by_name, be.vec_mat_q(x, n=n, name="blk.3.ffn_gate.weight"))
    with pytest.raises(pkg.BackendError) as ei:
        be.vec_mat_q(x[:1024], n=n, name="blk.3.ffn_gate.weight")                  # dimension mismatch (cuda/mod.rs:528-534)
    assert ei.value.variant == "ShapeMismatch"
    with pytest.raises(pkg.BackendError):
        be.load_weight("blk.3.ffn_gate.weight", t, raw, k, n)                     # loaded twice
    with pytest.raises(pkg.BackendError):
        be.vec_mat_q(x, n=n, name="no.such.weight")                               # unknown name and no host tensor to fall back to


def test_shape_errors_follow_the_reference(be, pkg):
    with pytest.raises(pkg.BackendError) as ei:
        be.add(np.zeros(4, np.float32), np.zeros(5, np.float32))                  # check_same_shape (ops.rs:1543-1551)
    assert ei.value.variant == "ShapeMismatch"
    with pytest.raises(pkg.BackendError) as ei:
        be.matmul(np.zeros((2, 3), np.float32), np.zeros((4, 2), np.float32))     # ops.rs:443-448
    assert ei.value.variant == "ShapeMismatch"
    with pytest.raises(pkg.BackendError) as ei:
        be.matvec(np.zeros((2, 3), np.float32), np.zeros((3, 1), np.float32))     # ops.rs:536-540
    assert ei.value.variant == "InvalidArgument"
    with pytest.raises(pkg.BackendError) as ei:
        be.attention(np.zeros((4, 2, 8), np.float32), np.zeros((2, 2, 16), np.float32), np.zeros((2, 2, 16), np.float32), 1.0)
    assert ei.value.variant == "InvalidArgument"
