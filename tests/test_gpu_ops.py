"""Kernel-level parity on the MI355X: every HIP kernel on the decode path, called through the C ABI's per-op
surface, against the CPU oracle on the same seeded inputs.

Tolerances: dequantization and RoPE are bit-exact; a mat-vec may differ from the oracle's strictly
sequential f32 sum by |d| <= 1e-4 * sum_i |x_i w_i|  (measured: ~1e-6); attention / norms are stated per test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL_QUANT = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
FUSED = ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "Q4_0"]


def _weights(pkg, tname, k, n, name="blk.0.test.weight"):
    t = pkg.synth.TYPE_IDS[tname]
    return t, pkg.synth.fill_tensor(name, t, k * n, k)


def _mv_bound(orc, t, raw, x, k, n):
    w = orc.dequantize(t, raw, k * n).reshape(n, k).astype(np.float64)
    return 1e-4 * (np.abs(w) @ np.abs(x.astype(np.float64))) + 1e-6


@pytest.mark.parametrize("tname", ALL_QUANT + ["F16"])
def test_dequantize_bit_exact(gpu, pkg, orc, tname):
    t = pkg.synth.TYPE_IDS[tname]
    n = 256 * 37
    raw = pkg.synth.fill_tensor("token_embd.weight", t, n, 256)
    want = orc.dequantize(t, raw, n)
    got = gpu.op_dequantize(t, raw, n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("tname", FUSED)
@pytest.mark.parametrize("k,n", [(256, 8), (2048, 64), (4096, 130), (5632, 33), (14336, 48), (28672, 40)])
def test_fused_vec_mat(gpu, pkg, orc, tname, k, n):
    """k covers one block, even and uneven k-slices (TinyLlama ffn 5632 = 22 blocks), 7 and 14 blocks per wave (the
    70B ffn width fills 143 KB of LDS with XQ records); n covers ragged tails."""
    t, raw = _weights(pkg, tname, k, n)
    x = np.random.default_rng(k + n).standard_normal(k).astype(np.float32)
    want = orc.vec_mat_q(t, raw, x, n)
    got = gpu.op_vec_mat(t, raw, x, n)
    assert np.all(np.abs(got - want) <= _mv_bound(orc, t, raw, x, k, n))


@pytest.mark.parametrize("tname", ["Q5_0", "Q2_K", "F32", "F16"])
def test_dequantized_fallback_vec_mat(gpu, pkg, orc, tname):
    """Types without a fused kernel are expanded to f32 at upload (reference: dequant_weights.rs:211-231)."""
    k, n = 1024, 40
    t, raw = _weights(pkg, tname, k, n)
    x = np.random.default_rng(5).standard_normal(k).astype(np.float32)
    w = orc.dequantize(t, raw, k * n)
    want = orc.vec_mat_q(t, raw, x, n) if tname not in ("F32", "F16") else orc.vec_mat_f32(w, x, n)
    got = gpu.op_vec_mat(t, raw, x, n)
    bound = 1e-4 * (np.abs(w.reshape(n, k).astype(np.float64)) @ np.abs(x.astype(np.float64))) + 1e-6
    assert np.all(np.abs(got - want) <= bound)


def test_vec_mat_linearity_and_zero_input(gpu, pkg, orc):
    """Size-independent properties at a full-size shape (Llama-3-8B gate row count)."""
    k, n = 4096, 14336
    t, raw = _weights(pkg, "Q4_K", k, n)
    rng = np.random.default_rng(11)
    x1, x2 = rng.standard_normal(k).astype(np.float32), rng.standard_normal(k).astype(np.float32)
    y1, y2, y12 = gpu.op_vec_mat(t, raw, x1, n), gpu.op_vec_mat(t, raw, x2, n), gpu.op_vec_mat(t, raw, x1 + x2, n)
    assert np.abs(y12 - (y1 + y2)).max() < 5e-4 * (np.abs(y1).max() + np.abs(y2).max())
    assert np.all(gpu.op_vec_mat(t, raw, np.zeros(k, np.float32), n) == 0.0)
    assert np.array_equal(y1, gpu.op_vec_mat(t, raw, x1, n))          # run-to-run deterministic
    rows = rng.integers(0, n, 64)
    rb = orc.nbytes_for(t, k)
    for j in rows:                                                    # spot rows against the oracle
        want = orc.dot_q(t, raw[j * rb:(j + 1) * rb], x1)
        assert abs(y1[j] - want) <= 1e-4 * float(np.abs(orc.dequantize(t, raw[j * rb:(j + 1) * rb], k) * x1).sum())


@pytest.mark.parametrize("n", [64, 2048, 4096, 8192])
def test_rms_norm(gpu, orc, n):
    rng = np.random.default_rng(n)
    x, w = rng.standard_normal(n).astype(np.float32) * 3, 1 + 0.01 * rng.standard_normal(n).astype(np.float32)
    want, got = orc.rms_norm(x, w, 1e-5), gpu.op_rms_norm(x, w, 1e-5)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    assert np.allclose(gpu.op_rms_norm([1, 2, 3, 4] * 16, np.ones(64), 1e-5)[:4], [0.365, 0.730, 1.095, 1.460], atol=1e-3)


@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K", "Q8_0"])
def test_norm_prologue_vec_mat(gpu, pkg, orc, tname):
    k, n = 4096, 96
    t, raw = _weights(pkg, tname, k, n)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(k).astype(np.float32) * 2
    nw = 1 + 0.01 * rng.standard_normal(k).astype(np.float32)
    xn = orc.rms_norm(x, nw, 1e-5)
    want = orc.vec_mat_q(t, raw, xn, n)
    got = gpu.op_norm_vec_mat(t, raw, x, nw, 1e-5, n)
    assert np.all(np.abs(got - want) <= 2 * _mv_bound(orc, t, raw, xn, k, n))


@pytest.mark.parametrize("tname", ["Q4_K", "Q5_K", "Q8_0"])
def test_swiglu_epilogue(gpu, pkg, orc, tname):
    k, n = 2048, 352
    t, wg = _weights(pkg, tname, k, n, "blk.0.ffn_gate.weight")
    _, wu = _weights(pkg, tname, k, n, "blk.0.ffn_up.weight")
    rng = np.random.default_rng(9)
    x = rng.standard_normal(k).astype(np.float32)
    nw = np.ones(k, np.float32)
    xn = orc.rms_norm(x, nw, 1e-5)
    want = orc.silu_mul(orc.vec_mat_q(t, wg, xn, n), orc.vec_mat_q(t, wu, xn, n))
    got = gpu.op_swiglu_vec_mat(t, wg, wu, x, nw, 1e-5, n)
    assert np.abs(got - want).max() <= 1e-4 * (1 + np.abs(want).max())


@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("d,base", [(64, 10000.0), (128, 500000.0)])
def test_rope_bit_exact(gpu, orc, neox, d, base):
    rng = np.random.default_rng(d)
    q, k = rng.standard_normal((8, d)).astype(np.float32), rng.standard_normal((2, d)).astype(np.float32)
    for pos, scale in ((0, 1.0), (1, 1.0), (255, 1.0), (77, 4.0)):
        wq, wk = orc.rope(q[:, None, :], k[:, None, :], pos, base, scale, neox)
        gq, gk = gpu.op_rope(q, k, pos, base, scale, neox)
        assert np.array_equal(gq, wq[:, 0, :]) and np.array_equal(gk, wk[:, 0, :])
    x = np.array([[1, 2, 3, 4]], np.float32)                         # ops.rs:1729-1777 through the kernel
    gq, _ = gpu.op_rope(x, x, 1, 10000.0, 1.0, True)
    assert np.allclose(gq.ravel(), [-1.98, 1.96, 2.46, 4.02], atol=0.05)


@pytest.mark.parametrize("nh,nkv,d", [(32, 8, 128), (32, 4, 64), (64, 8, 128), (8, 8, 64), (8, 4, 128)])
@pytest.mark.parametrize("kv_len,splits", [(1, 8), (7, 1), (129, 16), (256, 16), (1000, 32)])
def test_attention_cached(gpu, orc, nh, nkv, d, kv_len, splits):
    rng = np.random.default_rng(nh * 1000 + kv_len)
    max_seq = 1024
    q = rng.standard_normal((nh, d)).astype(np.float32)
    kc = np.zeros((nkv, max_seq, d), np.float32)
    vc = np.zeros((nkv, max_seq, d), np.float32)
    kc[:, :kv_len] = rng.standard_normal((nkv, kv_len, d)).astype(np.float32)
    vc[:, :kv_len] = rng.standard_normal((nkv, kv_len, d)).astype(np.float32)
    scale = 1.0 / np.sqrt(np.float32(d))
    want = orc.attention_cached(q, kc, vc, scale, kv_len)
    got = gpu.op_attention_cached(q, kc, vc, scale, kv_len, splits)
    assert np.abs(got - want).max() <= 2e-5 * (1 + np.abs(want).max())


def test_attention_peaked_softmax(gpu, orc):
    """One key dominates (weights of the others fall under the reference's 1e-8 skip threshold, ops.rs:1529)."""
    rng = np.random.default_rng(4)
    nh, nkv, d, kv_len = 8, 2, 128, 300
    q = rng.standard_normal((nh, d)).astype(np.float32)
    kc = 0.01 * rng.standard_normal((nkv, 512, d)).astype(np.float32)
    vc = rng.standard_normal((nkv, 512, d)).astype(np.float32)
    kc[:, 123] = np.repeat(q.reshape(nkv, nh // nkv, d)[:, 0], 1, axis=0) * 4
    scale = 1.0 / np.sqrt(np.float32(d))
    want, got = orc.attention_cached(q, kc, vc, scale, kv_len), gpu.op_attention_cached(q, kc, vc, scale, kv_len, 8)
    assert np.abs(got - want).max() <= 2e-5 * (1 + np.abs(want).max())


def test_silu_mul(gpu, orc):
    rng = np.random.default_rng(6)
    g, u = rng.standard_normal(5000).astype(np.float32) * 4, rng.standard_normal(5000).astype(np.float32)
    want, got = orc.silu_mul(g, u), gpu.op_silu_mul(g, u)
    assert np.abs(got - want).max() <= 4e-7 * (1 + np.abs(want).max())


@pytest.mark.parametrize("tname", ["Q8_1", "Q8_K"])
def test_dequantize_activation_formats_bit_exact(gpu, orc, tname):
    """Q8_1 / Q8_K (blocks.rs:8-168; dequant.rs:117-123, 361-367) have no synthetic weight generator — they are the
    reference's activation formats — so the blocks come from the oracle's own quantizer."""
    t = getattr(orc, tname)
    x = (np.random.default_rng(5).standard_normal(256 * 9) * 3).astype(np.float32)
    raw = orc.quantize(t, x)
    want = orc.dequantize(t, raw, x.size)
    got = gpu.op_dequantize(t, raw, x.size)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(want - x).max() < 0.1                               # and the round trip is what the reference's bounds say


@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_kv_cache_formats_bit_exact(gpu, orc, fmt):
    """The KV cache's byte formats as the attention launch applies them (lgh_op_kv_roundtrip; kv_cache_type 1 = int8, 2 = FP8
    E4M3, 3 = FP8 E5M2) against the oracle's restatement of quantize_int8 / quantize_fp8_* and their dequantizers
    (src/model/kv_quantized.rs:385-565): bytes, scale and read-back values identical — every exactly representable value, the
    value just below the next one (the reference truncates), halves for int8's round-half-away, saturation, subnormals,
    signed zeros, and (FP8) infinities and the [480, 512) magnitudes that encode to the NaN pattern."""
    rng = np.random.default_rng(40 + fmt)
    rows = [rng.standard_normal(256).astype(np.float32) * s for s in (1.0, 0.01, 300.0)]
    rows.append(np.array([0.0, -0.0, 1e-12, -1e-12, 0.5, -1.5, 2.5, -2.5, 448.0, 500.0, -500.0, 57344.0, 70000.0, 1e9, -1e9,
                          2.0 ** -6, 2.0 ** -9, 2.0 ** -10, 2.0 ** -14, 2.0 ** -16, 2.0 ** -17, 1.52588e-5, 0.0136719], np.float32))
    if fmt != 1:
        exact = np.array([orc.kv_dequantize_fp8(fmt - 1, b) for b in range(256)], np.float32)
        exact = exact[np.isfinite(exact)]
        rows += [exact, np.nextafter(exact, np.float32(0.0)), np.nextafter(exact, np.float32(np.inf) * np.sign(exact + np.float32(1e-30)))]
        rows.append(np.array([np.inf, -np.inf], np.float32))
    else:
        rows.append(np.zeros(64, np.float32))                                   # an all-zero row: scale 1
        rows.append((np.arange(-130, 131, dtype=np.float32) + np.float32(0.5)) * np.float32(127.0 / 130.5))   # halves
    for row in rows:
        got_b, got_sc, got_back = gpu.op_kv_roundtrip(fmt, row)
        if fmt == 1:
            want_q, want_sc = orc.kv_quantize_int8(row)
            want_back = orc.kv_dequantize_int8(want_q, want_sc)
            want_b = want_q.view(np.uint8)
        else:
            want_b = np.array([orc.kv_quantize_fp8(fmt - 1, float(v)) for v in row], np.uint8)
            want_back = np.array([orc.kv_dequantize_fp8(fmt - 1, int(b)) for b in want_b], np.float32)
            want_sc = 1.0
        assert np.array_equal(got_b, want_b), (fmt, row[got_b != want_b][:4], got_b[got_b != want_b][:4], want_b[got_b != want_b][:4])
        assert got_sc == np.float32(want_sc)
        assert np.array_equal(got_back.view(np.uint32)[~np.isnan(want_back)], want_back.view(np.uint32)[~np.isnan(want_back)])
        assert np.array_equal(np.isnan(got_back), np.isnan(want_back))


@pytest.mark.parametrize("dim,bits", [(128, 2), (128, 3), (64, 2), (64, 3)])
def test_turboquant_codes_bit_exact(gpu, orc, dim, bits):
    """TurboQuantEngine::compress (src/model/turboquant/quant.rs:71-103 = rotation.rs:58-76 + codebook.rs:131-170) as the
    attention launch applies it to a new K / V row (lgh_op_tq_compress): the packed codes equal the oracle's BIT FOR BIT — the
    butterfly makes the same additions in the same order, the cell of a coordinate is the number of boundaries it is >=.
    Rows of every scale (near-unit norm, tiny, saturating), a delta, a constant, zeros, and values sitting exactly ON a cell
    boundary after rotation (a rotated delta has every coordinate at +-x / sqrt(d))."""
    rng = np.random.default_rng(500 + dim + bits)
    sign_sets = [np.where(rng.integers(0, 2, dim) == 1, 1.0, -1.0).astype(np.float32) for _ in range(3)] + [np.ones(dim, np.float32)]
    rows = [rng.standard_normal(dim).astype(np.float32) * s for s in (1.0 / np.sqrt(dim), 1.0, 1e-4, 30.0)]
    rows += [np.zeros(dim, np.float32), np.full(dim, 0.37, np.float32), (np.arange(dim, dtype=np.float32) - dim / 2) * np.float32(0.01)]
    _, bnd = orc.tq_codebook(dim, bits)
    for b in bnd:                                                    # delta * sqrt(d) * boundary: every rotated coordinate = +-boundary
        e = np.zeros(dim, np.float32)
        e[int(rng.integers(0, dim))] = np.float32(b) * np.sqrt(np.float32(dim))
        rows.append(e)
    n = 0
    for sg in sign_sets:
        for x in rows:
            got, want = gpu.op_tq_compress(x, bits, sg), orc.tq_compress(x, bits, sg)
            assert got.size == want.size == (dim // 4 if bits == 2 else dim // 8 * 3)
            assert np.array_equal(got, want), (dim, bits, np.flatnonzero(got != want)[:4])
            n += 1
    assert n == 4 * len(rows)


@pytest.mark.parametrize("dim,bits", [(64, 2), (128, 2), (128, 3), (64, 3)])
def test_turboquant_qjl_rows_bit_exact(gpu, orc, dim, bits):
    """TurboQuantEngine::compress with use_qjl (quant.rs:71-103 + QjlProjector::compress, qjl.rs:36-62) as the attention launch
    applies it to a new K row (lgh_op_tq_compress_qjl): codes, the sign bits of S r and |r| equal the oracle's BIT FOR BIT for the
    same projection matrix — residual = rotated - centroid in f32, every projection a sequential unfused dot product, the norm a
    sequential sum of squares."""
    rng = np.random.default_rng(900 + dim + bits)
    S = rng.standard_normal((dim, dim)).astype(np.float32)
    sg = np.where(rng.integers(0, 2, dim) == 1, 1.0, -1.0).astype(np.float32)
    rows = [rng.standard_normal(dim).astype(np.float32) * s for s in (1.0 / np.sqrt(dim), 1.0, 1e-4, 30.0)]
    rows += [np.zeros(dim, np.float32), np.full(dim, 0.37, np.float32), (np.arange(dim, dtype=np.float32) - dim / 2) * np.float32(0.01)]
    for x in rows:
        codes, qb, nrm = gpu.op_tq_compress_qjl(x, bits, sg, S)
        wc, wb, wn = orc.tq_compress_qjl(x, bits, sg, S)
        assert np.array_equal(codes, wc)
        assert np.array_equal(qb, wb), (dim, bits, [hex(int(v)) for v in qb], [hex(int(v)) for v in wb])
        assert np.float32(nrm).view(np.uint32) == np.float32(wn).view(np.uint32), (nrm, wn)
    # a zero projection matrix: every projection is +0.0 >= 0 -> all bits set (qjl.rs:58-60)
    codes, qb, nrm = gpu.op_tq_compress_qjl(rows[0], bits, sg, np.zeros((dim, dim), np.float32))
    assert all(int(v) == 2**64 - 1 for v in qb)
