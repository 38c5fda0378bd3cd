"""Pins the CPU oracle against every known-answer test the reference's own test-suite holds for the
decode path.  Paths are relative to /root/reference; nothing is read from it at run time."""
import struct

import numpy as np
import pytest


def test_block_sizes(orc):
    # src/tensor/dtype.rs:217-229, src/tensor/quant/blocks.rs:293-305
    want = {orc.Q4_0: (32, 18), orc.Q4_1: (32, 20), orc.Q5_0: (32, 22), orc.Q5_1: (32, 24), orc.Q8_0: (32, 34),
            orc.Q8_1: (32, 36), orc.Q2_K: (256, 84), orc.Q3_K: (256, 110), orc.Q4_K: (256, 144),
            orc.Q5_K: (256, 176), orc.Q6_K: (256, 210), orc.Q8_K: (256, 292), orc.F32: (1, 4), orc.F16: (1, 2)}
    for t, (bs, bb) in want.items():
        assert (orc.block_size(t), orc.block_bytes(t)) == (bs, bb)


def test_f16_conversion_exhaustive(orc):
    # half 2.7.1 (Cargo.lock:907) is IEEE binary16 round-to-nearest-even == numpy float16
    h = np.arange(65536, dtype=np.uint16)
    f = h.view(np.float16).astype(np.float32)
    mine = np.array([orc.lib().orc_f16_to_f32(int(v)) for v in h], dtype=np.float32)
    ok = ~np.isnan(f)
    assert np.array_equal(mine.view(np.uint32)[ok], f.view(np.uint32)[ok])
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 60000)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    mine = np.array([orc.lib().orc_f32_to_f16(float(v)) for v in xs], dtype=np.uint16)
    assert np.array_equal(mine, want)


def test_vec_mat_gguf_layout(orc):
    # src/backend/cpu/ops.rs:1813-1847 and 1850-1867
    assert orc.vec_mat_f32([1, 2, 3, 4, 5, 6], [1, 1, 1], 2).tolist() == [6.0, 15.0]
    assert orc.vec_mat_f32([1, 0, 0, 0, 1, 0], [7, 8, 9], 2).tolist() == [7.0, 8.0]


def test_rms_norm_kats(orc):
    # ops.rs:1636-1648; simd.rs:1200-1220; tests/integration_test.rs:175-214
    out = orc.rms_norm([1, 2, 3, 4], [1, 1, 1, 1], 1e-5)
    assert abs(out[0] - 0.365) < 0.01 and abs(out[3] - 1.46) < 0.01
    out = orc.rms_norm([1, 2, 3, 4], [1, 1, 1, 1], 1e-6)
    rms = np.sqrt(np.float32(30.0) / 4)
    assert np.allclose(out, np.array([1, 2, 3, 4], np.float32) / rms, atol=1e-5)


def test_rope_kats(orc):
    # ops.rs:1689-1705: position 0 is the identity (NeoX style)
    qd = np.array([1, 0, 1, 0, 0, 1, 0, 1], np.float32).reshape(2, 1, 4)
    q, k = orc.rope(qd, qd, 0, 10000.0, 1.0, True)
    assert np.allclose(q, qd, atol=1e-5) and np.allclose(k, qd, atol=1e-5)
    # ops.rs:1708-1726: interleaved pairing, pos 1 -> q[0] = cos(1)
    q, _ = orc.rope(np.array([1, 0, 0, 0], np.float32).reshape(1, 1, 4), np.array([1, 0, 0, 0], np.float32).reshape(1, 1, 4),
                    1, 10000.0, 1.0, False)
    assert abs(q.ravel()[0] - 0.54) < 0.02
    # ops.rs:1729-1777: NeoX pairing of [1,2,3,4] at pos 1
    x = np.array([1, 2, 3, 4], np.float32).reshape(1, 1, 4)
    q, _ = orc.rope(x, x, 1, 10000.0, 1.0, True)
    assert np.allclose(q.ravel(), [-1.98, 1.96, 2.46, 4.02], atol=0.05)


def test_rope_freq_scale_divides_position(orc):
    # ops.rs:1300: position = (pos + s) / freq_scale   (scripts/test_rope.py multiplies: not authoritative)
    x = np.array([1, 0, 0, 0], np.float32).reshape(1, 1, 4)
    a, _ = orc.rope(x, x, 4, 10000.0, 4.0, False)
    b, _ = orc.rope(x, x, 1, 10000.0, 1.0, False)
    assert np.array_equal(a, b)


def test_dequantize_q5_kats(orc):
    # tests/dequant_test.rs:158-212
    d = orc.lib().orc_f32_to_f16(0.1)
    blk = struct.pack("<H", d) + bytes(4) + bytes([0x88] * 16)
    out = orc.dequantize(orc.Q5_0, np.frombuffer(blk, np.uint8), 32)
    assert np.all(np.abs(out + 0.8) < 0.01)
    blk = struct.pack("<HH", d, orc.lib().orc_f32_to_f16(1.0)) + bytes(4) + bytes([0x88] * 16)
    out = orc.dequantize(orc.Q5_1, np.frombuffer(blk, np.uint8), 32)
    assert np.all(np.abs(out - 1.8) < 0.01)


ROUNDTRIP = [
    # (type, input, bound kind, bound)  — src/tensor/quant/dequant.rs:1070-1303, tests/dequant_test.rs:8-137
    ("Q4_0", lambda i: (i - 16.0) * 0.1, "abs", 0.15), ("Q4_1", lambda i: (i - 16.0) * 0.1 + 1.0, "abs", 0.15),
    ("Q4_1", lambda i: i * 0.1 + 1.0, "abs", 0.15), ("Q4_1", lambda i: i * 0.1 + 5.0, "abs", 0.15),
    ("Q5_0", lambda i: (i - 16.0) * 0.1, "abs", 0.15), ("Q5_1", lambda i: (i - 16.0) * 0.1 + 1.0, "abs", 0.15),
    ("Q8_0", lambda i: (i - 16.0) * 0.1, "abs", 0.02), ("Q8_1", lambda i: (i - 16.0) * 0.1, "abs", 0.02),
    ("Q2_K", lambda i: (i - 128.0) * 0.1, "rmse", 6.0), ("Q3_K", lambda i: (i - 128.0) * 0.1, "rmse", 2.5),
    ("Q4_K", lambda i: (i - 128.0) * 0.1, "rmse", 4.5), ("Q5_K", lambda i: (i - 128.0) * 0.1, "rmse", 4.5),
    ("Q6_K", lambda i: (i - 128.0) * 0.1, "abs", 1.0), ("Q8_K", lambda i: (i - 128.0) * 0.1, "abs", 0.1),
]


@pytest.mark.parametrize("tname,fn,kind,bound", ROUNDTRIP)
def test_quantize_roundtrip_bounds(orc, tname, fn, kind, bound):
    t = getattr(orc, tname)
    bs = orc.block_size(t)
    x = np.array([fn(np.float32(i)) for i in range(bs)], dtype=np.float32)
    y = orc.dequantize(t, orc.quantize(t, x), bs)
    if kind == "abs":
        assert np.abs(x - y).max() < bound
    else:
        assert np.sqrt(np.mean((x - y) ** 2, dtype=np.float32)) < bound


def test_quantize_zeros_and_scaling(orc):
    # tests/dequant_test.rs:66-89 (zeros), 92-121 (large values), 124-143 (precision), 216-239 (symmetry)
    for t in (orc.Q4_0, orc.Q8_0):
        assert np.all(orc.dequantize(t, orc.quantize(t, np.zeros(32, np.float32)), 32) == 0.0)
    x = (np.arange(32, dtype=np.float32) - 16.0) * 10.0
    y = orc.dequantize(orc.Q4_0, orc.quantize(orc.Q4_0, x), 32)
    assert np.abs(x - y).max() <= np.abs(x).max() / 7.0 * 1.1
    x = (np.arange(32, dtype=np.float32) - 16.0) * 0.01
    y = orc.dequantize(orc.Q8_0, orc.quantize(orc.Q8_0, x), 32)
    assert np.sqrt(np.mean((x - y) ** 2)) < 0.005
    p = np.arange(32, dtype=np.float32) * 0.1
    a = orc.dequantize(orc.Q8_0, orc.quantize(orc.Q8_0, p), 32)
    b = orc.dequantize(orc.Q8_0, orc.quantize(orc.Q8_0, -p), 32)
    assert np.abs(a + b).max() < 0.02
    # dequant.rs:1325-1345 batch dequantize
    two = np.concatenate([np.arange(32, dtype=np.float32), np.arange(32, 64, dtype=np.float32)])
    out = orc.dequantize(orc.Q4_0, orc.quantize(orc.Q4_0, two), 64)
    assert -1 <= out[0] <= 1 and out[31] >= 30 and out[32] >= 30 and out[63] >= 60


def test_simd_kats(orc):
    # simd.rs:1176-1198
    assert abs(orc.dot_f32([1, 2, 3, 4, 5, 6, 7, 8], [1] * 8) - 36.0) < 1e-6
    assert abs(orc.lib().orc_max_f32(np.array([1, 5, 3, 9, 2, 8, 4, 7, 6], np.float32).ctypes.data, 9) - 9.0) < 1e-6
    s = orc.softmax([1.0, 2.0, 3.0, 4.0])
    assert abs(s.sum() - 1.0) < 1e-6 and np.all(np.diff(s) > 0)       # ops.rs:1618-1632
    assert np.allclose(orc.silu([0.0, 1.0, -1.0]), [0.0, 0.7310586, -0.26894143], atol=1e-6)


def test_simd_kats_sum_scale_silu_mul_axpy(orc):
    """The remaining known-answer tests of src/backend/cpu/simd.rs:1184-1263 (test_sum, test_scale, test_silu_mul_inplace,
    test_axpy), on every ISA path the reference dispatches to."""
    L = orc.lib()
    for isa in (orc.ISA_SCALAR, orc.ISA_AVX2, orc.ISA_AVX512):
        orc.set_isa(isa)
        a = np.arange(1, 11, dtype=np.float32)
        assert abs(L.orc_sum_f32(a.ctypes.data, 10) - 55.0) < 1e-6                       # simd.rs:1184-1189
        a8 = np.arange(1, 9, dtype=np.float32)
        assert np.all(np.abs(orc.scale(a8, 2.0) - a8 * 2.0) < 1e-6)                      # simd.rs:1221-1231
        gate = np.array([1.0, -1.0, 2.0, 0.0, 0.5, -0.5, 3.0, -2.0], np.float32)         # simd.rs:1233-1252
        up = np.array([2.0, 3.0, 1.0, 5.0, 4.0, 2.0, 0.5, 1.0], np.float32)
        want = np.array([x / (1.0 + np.exp(np.float32(-x))) * u for x, u in zip(gate, up)], np.float32)
        assert np.all(np.abs(orc.silu_mul(gate, up) - want) < 1e-5)
        y = np.arange(10, 90, 10, dtype=np.float32)                                       # simd.rs:1254-1273
        L.orc_axpy_f32(2.0, a8.ctypes.data, y.ctypes.data, 8)
        assert np.all(np.abs(y - np.array([12, 24, 36, 48, 60, 72, 84, 96], np.float32)) < 1e-6)
    orc.set_isa(orc.ISA_AUTO)


def test_dot_f32_isa_variants_agree_to_rounding(orc):
    rng = np.random.default_rng(7)
    a, b = rng.standard_normal(4099).astype(np.float32), rng.standard_normal(4099).astype(np.float32)
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    vals = []
    for isa in (orc.ISA_SCALAR, orc.ISA_AVX2, orc.ISA_AVX512):
        orc.set_isa(isa)
        vals.append(orc.dot_f32(a, b))
    orc.set_isa(orc.ISA_AUTO)
    bound = 1e-5 * float(np.abs(a * b).sum())
    assert all(abs(v - ref) < bound for v in vals)


def test_greedy_and_argmax_rules(orc):
    # sampling/mod.rs:442-453: greedy picks index 5
    logits = [0.0, 0.1, 0.2, 0.3, 0.4, 1.0, 0.2, 0.1, 0.0, -0.1]
    assert orc.greedy_sample(logits) == 5 and orc.argmax_last(logits) == 5
    # Iterator::max_by returns the LAST maximum (main.rs:1815-1821)
    assert orc.argmax_last([1.0, 3.0, 3.0, 2.0]) == 2


def test_moe_router_kat(orc):
    # moe.rs:505-517: zero router weights, hidden 0.1 -> 2 experts, weights sum to 1; stable sort keeps [0, 1]
    idx, w = orc.moe_route(np.full(64, 0.1, np.float32), np.zeros((4, 64), np.float32), 4, 2, normalize=True)
    assert idx.tolist() == [0, 1] and abs(w.sum() - 1.0) < 0.01


def test_attention_gqa_smoke(orc):
    # ops.rs:1780-1810 (shape / GQA head mapping / finiteness), through attention_cached
    q = np.ones((4, 4), np.float32)
    kc = np.ones((2, 1, 4), np.float32)
    out = orc.attention_cached(q, kc, kc, 0.5, 1)
    assert np.all(np.isfinite(out)) and np.allclose(out, 1.0)


@pytest.mark.parametrize("tname", ["Q4_0", "Q8_0", "Q4_K", "Q5_K", "Q6_K", "Q8_K"])
def test_fused_dot_equals_dequant_dot(orc, tname):
    """Self-consistency identity for the dots the reference has no golden vector for."""
    t = getattr(orc, tname)
    rng = np.random.default_rng(hash(tname) % 2**32)
    k = 2048
    raw = orc.quantize(t, rng.standard_normal(k).astype(np.float32))
    x = rng.standard_normal(k).astype(np.float32)
    w = orc.dequantize(t, raw, k).astype(np.float64)
    want = float(np.dot(w, x.astype(np.float64)))
    got = orc.dot_q(t, raw, x)
    assert abs(got - want) <= 1e-5 * float(np.abs(w * x).sum()) + 1e-6


def test_vec_mat_q_matches_rowwise_dots_and_fallback(orc):
    rng = np.random.default_rng(3)
    k, n = 512, 6
    x = rng.standard_normal(k).astype(np.float32)
    for t in (orc.Q4_K, orc.Q6_K, orc.Q8_0, orc.Q5_0, orc.Q2_K):      # Q5_0 / Q2_K take the dequantize fallback
        raw = orc.quantize(t, rng.standard_normal(k * n).astype(np.float32))
        out = orc.vec_mat_q(t, raw, x, n)
        rb = orc.nbytes_for(t, k)
        for j in range(n):
            row = raw[j * rb:(j + 1) * rb]
            want = orc.dot_q(t, row, x) if orc.lib().orc_has_fused_dot(t) else orc.dot_f32(x, orc.dequantize(t, row, k))
            assert out[j] == np.float32(want)


def test_backend_trait_elementwise_kats(orc):
    """ops.rs:1567-1630: test_add, test_mul, test_scale (exact), test_silu, test_softmax."""
    assert np.array_equal(orc.add([1, 2, 3, 4], [10, 20, 30, 40]), np.float32([11, 22, 33, 44]))
    assert np.array_equal(orc.mul([1, 2, 3, 4], [2, 3, 4, 5]), np.float32([2, 6, 12, 20]))
    assert np.array_equal(orc.scale([1, 2, 3, 4], 2.5), np.float32([2.5, 5.0, 7.5, 10.0]))
    s = orc.silu([0.0, 1.0, -1.0, 2.0])
    assert abs(s[0]) < 1e-6 and abs(s[1] - 0.731) < 0.01 and abs(s[2] + 0.269) < 0.01
    p = orc.softmax_rows([1.0, 2.0, 3.0, 4.0])
    assert abs(float(p.sum()) - 1.0) < 1e-6 and p[0] < p[1] < p[2] < p[3]
    rows = orc.softmax_rows(np.arange(12, dtype=np.float32).reshape(3, 4))       # along the last dimension
    assert np.allclose(rows.sum(axis=1), 1.0, atol=1e-6) and np.array_equal(rows[0], rows[1])
    g = orc.gelu([0.0, 1.0, -1.0, 3.0])                                           # ops.rs:328-347: tanh approximation
    assert g[0] == 0.0 and abs(g[1] - 0.8412) < 1e-3 and abs(g[2] + 0.1588) < 1e-3 and abs(g[3] - 2.9964) < 1e-3


def test_backend_trait_matmul_matvec_kats(orc):
    """ops.rs:1651-1680: test_matmul ([[22, 28], [49, 64]]) and test_matvec ([30, 70, 110]), both exact."""
    a, b = np.float32([[1, 2, 3], [4, 5, 6]]), np.float32([[1, 2], [3, 4], [5, 6]])
    assert np.array_equal(orc.matmul(a, b), np.float32([[22, 28], [49, 64]]))
    assert np.array_equal(orc.matvec(np.arange(1, 13, dtype=np.float32).reshape(3, 4), [1, 2, 3, 4]), np.float32([30, 70, 110]))
    rng = np.random.default_rng(1)                                               # large enough for the tiled variant (m*k*n >= 256^3)
    a, b = rng.standard_normal((300, 260)).astype(np.float32), rng.standard_normal((260, 280)).astype(np.float32)
    assert np.allclose(orc.matmul(a, b), a.astype(np.float64) @ b.astype(np.float64), atol=2e-4)


def test_backend_trait_attention_kats(orc):
    """ops.rs:1779-1810: test_attention_simple (position 0 sees only itself) and test_attention_gqa."""
    q = np.float32([1, 0, 0, 0, 0, 1, 0, 0]).reshape(1, 2, 4)
    v = np.float32([1, 2, 3, 4, 5, 6, 7, 8]).reshape(1, 2, 4)
    out = orc.attention(q, q.copy(), v, 1.0 / np.sqrt(2.0))
    assert np.array_equal(out[0, 0], v[0, 0])                                    # causal: a softmax over one element
    assert np.all(out[0, 1] > v[0, 0]) and np.all(out[0, 1] < v[0, 1])           # a proper mixture at position 1
    ones = orc.attention(np.ones((4, 1, 4), np.float32), np.ones((2, 1, 4), np.float32), np.ones((2, 1, 4), np.float32), 0.5)
    assert np.all(np.isfinite(ones)) and np.array_equal(ones, np.ones((4, 1, 4), np.float32))
    # the cached form on the same data agrees (ops.rs:1479-1537 vs 1353-1472): last query position of a 5-row context
    rng = np.random.default_rng(2)
    qq, kk, vv = (rng.standard_normal(s).astype(np.float32) for s in ((4, 3, 8), (2, 5, 8), (2, 5, 8)))
    full = orc.attention(qq, kk, vv, 0.35)
    cached = orc.attention_cached(qq[:, 2, :], kk, vv, 0.35, 5)
    assert np.allclose(full[:, 2, :], cached, atol=1e-6)


def test_int8_kv_cache_kats(orc):
    """The reference's int8 KV format (src/model/kv_quantized.rs): test_int8_roundtrip (:547-560), the Int8 leg of
    test_quantized_kv_cache_basic (:612-665), test_shift_left's constant rows (:688-727: a row of equal values survives exactly)."""
    data = np.arange(128, dtype=np.float32) * np.float32(0.1) - np.float32(6.4)
    q, sc = orc.kv_quantize_int8(data)
    assert abs(sc - 6.4 / 127.0) < 1e-7 and q[0] == -127 and q.min() >= -128 and q.max() <= 127
    dec = orc.kv_dequantize_int8(q, sc)
    rel = np.where(np.abs(data) > 1e-6, np.abs(data - dec) / np.maximum(np.abs(data), 1e-30), np.abs(data - dec))
    assert rel.max() < 0.02
    k = np.arange(4 * 64, dtype=np.float32) * np.float32(0.01) - np.float32(1.0)      # head 0 = the first 64 values
    v = np.arange(4 * 64, dtype=np.float32) * np.float32(0.02) - np.float32(0.5)
    for row in (k[:64], v[:64]):
        q, sc = orc.kv_quantize_int8(row)
        dec = orc.kv_dequantize_int8(q, sc)
        rel = np.where(np.abs(row) > 1e-6, np.abs(row - dec) / np.maximum(np.abs(row), 1e-30), np.abs(row - dec))
        assert rel.max() < 0.15
    for pos in range(5):
        q, sc = orc.kv_quantize_int8(np.full(4, float(pos), np.float32))
        assert np.abs(orc.kv_dequantize_int8(q, sc) - pos).max() < 0.01
    q, sc = orc.kv_quantize_int8(np.zeros(8, np.float32))       # an all-zero row: scale 1, values 0
    assert sc == 1.0 and not q.any()
    q, sc = orc.kv_quantize_int8(np.array([0.5, -1.5, 2.5, -2.5], np.float32) * np.float32(2.5 / 127 * 127 / 2.5))
    assert list(q) == [25, -76, 127, -127]                       # f32::round: halves away from zero (25.4 -> 25, -76.2 -> -76)


def _fp8_value(fmt, b):
    """Independent decode of an FP8 byte (OCP E4M3 'fn' / E5M2): what the reference's dequantize_fp8_* must produce."""
    ebits, mbits, bias = (4, 3, 7) if fmt == 1 else (5, 2, 15)
    s = -1.0 if b & 0x80 else 1.0
    e, m = (b >> mbits) & ((1 << ebits) - 1), b & ((1 << mbits) - 1)
    if fmt == 1 and (b & 0x7F) == 0x7F:
        return float("nan")
    if fmt == 2 and e == 31:
        return s * float("inf") if m == 0 else float("nan")
    if e == 0:
        return 0.0 if m == 0 else s * m * 2.0 ** (1 - bias - mbits)      # (the reference returns +0 for both zeros)
    return s * (1.0 + m / (1 << mbits)) * 2.0 ** (e - bias)


def test_fp8_kv_cache_kats(orc):
    """The reference's FP8 KV formats (src/model/kv_quantized.rs:413-565): test_fp8_e4m3_roundtrip (:563-585),
    test_fp8_e5m2_roundtrip (:588-609) and the FP8 legs of test_quantized_kv_cache_basic (:612-665) with their tolerances;
    then every byte against an independent decode, the truncating (not rounding) encode, saturation, and the quirk that an
    E4M3 magnitude in [480, 512) encodes to the NaN pattern 0x7F."""
    for fmt, values, tol in ((orc.FP8_E4M3, [0.0, 1.0, -1.0, 0.5, 0.0136719, 448.0, 2.0 ** -6, 2.0 ** -9], 0.05),
                             (orc.FP8_E5M2, [0.0, 1.0, -1.0, 0.5, 57344.0, 2.0 ** -14, 1.52588e-5], 0.1)):
        for val in values:
            d = orc.kv_dequantize_fp8(fmt, orc.kv_quantize_fp8(fmt, val))
            if val == 0.0:
                assert d == 0.0
            elif abs(val) < 1e-5:
                assert abs(d) < 0.01
            else:
                assert abs(val - d) / abs(val) < tol, (fmt, val, d)
        k = np.arange(4 * 64, dtype=np.float32) * np.float32(0.01) - np.float32(1.0)
        v = np.arange(4 * 64, dtype=np.float32) * np.float32(0.02) - np.float32(0.5)
        for row in (k[:64], v[:64]):
            for a in row:
                b = orc.kv_dequantize_fp8(fmt, orc.kv_quantize_fp8(fmt, float(a)))
                rel = abs(a - b) / abs(a) if abs(a) > 1e-6 else abs(a - b)
                assert rel < 0.25, (fmt, a, b)
        for byte in range(256):
            want, got = _fp8_value(fmt, byte), orc.kv_dequantize_fp8(fmt, byte)
            assert (np.isnan(want) and np.isnan(got)) or want == got, (fmt, byte, want, got)
            if not np.isnan(want) and not np.isinf(want) and want != 0.0:
                assert orc.kv_quantize_fp8(fmt, got) == byte                         # exact values encode to themselves
                below_next = np.nextafter(np.float32(_fp8_value(fmt, byte + 1)), np.float32(0.0)).item() \
                    if (byte & 0x7F) < (0x7E if fmt == orc.FP8_E4M3 else 0x7B) else None
                if below_next is not None and not np.isnan(below_next):
                    assert orc.kv_quantize_fp8(fmt, below_next) == byte, (fmt, byte)   # truncation: just below the next value stays here
    assert orc.kv_quantize_fp8(orc.FP8_E4M3, 1e6) == 0x7E and orc.kv_quantize_fp8(orc.FP8_E4M3, -1e6) == 0xFE   # saturation to 448
    assert orc.kv_quantize_fp8(orc.FP8_E5M2, 1e9) == 0x7C                                                        # ... and to inf
    assert orc.kv_quantize_fp8(orc.FP8_E4M3, 500.0) == 0x7F and np.isnan(orc.kv_dequantize_fp8(orc.FP8_E4M3, 0x7F))
    assert orc.kv_quantize_fp8(orc.FP8_E4M3, 2.0 ** -10) == 0x00 and orc.kv_quantize_fp8(orc.FP8_E5M2, 2.0 ** -17) == 0x00
    assert orc.kv_quantize_fp8(orc.FP8_E4M3, float("nan")) == 0xFF and orc.kv_quantize_fp8(orc.FP8_E5M2, float("inf")) == 0x7C


# ---------------------------------------------------------------------------------------------------------------------
# TurboQuant KV cache (what `--kv-cache-type tq2 | tq3` selects): the reference's own unit tests of
# src/model/turboquant/{codebook,rotation}.rs and src/model/kv_turboquant.rs, restated.  The rotation's sign vector is an
# input of the oracle (the reference draws it from rand's StdRng, which is not restated): the tests that hold for ANY sign
# vector run on deterministic +-1 patterns; test_deterministic / test_different_seeds_differ (rotation.rs:168-190) are about
# the RNG stream itself and have no counterpart.
# ---------------------------------------------------------------------------------------------------------------------
def _signs(n, salt=0):
    i = np.arange(n, dtype=np.uint64)
    h = (i + np.uint64(salt) * np.uint64(7919)) * np.uint64(0x9E3779B97F4A7C15)
    return np.where((h >> np.uint64(40)) & np.uint64(1), 1.0, -1.0).astype(np.float32)


def test_turboquant_codebook_kats(orc):
    """codebook.rs:277-348: 1-bit round trip, 2-bit ordering, vector quantize round trip, dot_with_packed consistency,
    3-bit packing; plus the table constants themselves (codebook.rs:20-49) and packed_bytes (255-262)."""
    c1, b1 = orc.tq_codebook(128, 1)                                                   # test_codebook_1bit_roundtrip
    assert len(c1) == 2 and orc.tq_quantize(128, 1, -0.1) == 0 and orc.tq_quantize(128, 1, 0.1) == 1
    assert c1[0] < 0 < c1[1] and abs(c1[0] + c1[1]) < 1e-6
    c2, b2 = orc.tq_codebook(128, 2)                                                   # test_codebook_2bit_ordering
    assert len(c2) == 4 and all(c2[i] < c2[i + 1] for i in range(3))
    inv = np.float32(1.0) / np.sqrt(np.float32(128.0))
    assert np.array_equal(c2, np.array([-1.5102326, -0.4528427, 0.4528427, 1.5102326], np.float32) * inv)
    assert np.array_equal(b2, np.array([-0.98153765, 0.0, 0.98153765], np.float32) * inv)
    c3, b3 = orc.tq_codebook(64, 3)
    inv64 = np.float32(1.0) / np.sqrt(np.float32(64.0))
    assert np.array_equal(c3, np.array([-2.1521645, -1.3441838, -0.7561303, -0.2453404, 0.2453404, 0.7561303, 1.3441838, 2.1521645], np.float32) * inv64)
    assert np.array_equal(b3, np.array([-1.74817415, -1.05015705, -0.50073535, 0.0, 0.50073535, 1.05015705, 1.74817415], np.float32) * inv64)
    # quantize = the number of boundaries the value is >= (boundaries ascending), the boundary itself goes up
    for val, want in ((-1.0, 0), (float(b2[0]), 1), (np.nextafter(b2[0], np.float32(-1)).item(), 0), (0.0, 2), (1.0, 3)):
        assert orc.tq_quantize(128, 2, val) == want
    data = (np.arange(128, dtype=np.float32) - np.float32(64.0)) * np.float32(0.001)   # test_vector_quantize_roundtrip
    packed = orc.tq_quantize_vector(128, 2, data)
    assert packed.size == 32
    deq = orc.tq_dequantize_vector(128, 2, packed, 128)
    assert deq.size == 128 and np.all(np.abs(data - deq) < 0.1)
    data = (np.arange(64, dtype=np.float32) - np.float32(32.0)) * np.float32(0.005)    # test_dot_with_packed_consistency
    query = np.arange(64, dtype=np.float32) * np.float32(0.01)
    packed = orc.tq_quantize_vector(64, 2, data)
    deq = orc.tq_dequantize_vector(64, 2, packed, 64)
    direct = np.float32(0.0)
    for a, b in zip(query, deq):
        direct = np.float32(direct + np.float32(a * b))
    assert abs(float(direct) - orc.tq_dot_with_packed(64, 2, query, packed, 64)) < 1e-5
    data = (np.arange(16, dtype=np.float32) - np.float32(8.0)) * np.float32(0.02)      # test_3bit_packing
    packed = orc.tq_quantize_vector(16, 3, data)
    assert packed.size == orc.tq_packed_bytes(3, 16) == 6
    assert orc.tq_dequantize_vector(16, 3, packed, 16).size == 16
    # the 3-bit stream: index i sits at bits [3i, 3i + 3) of its group's 24-bit little-endian word
    idx = [orc.tq_quantize(16, 3, float(v)) for v in data]
    for g in range(2):
        word = int(packed[3 * g]) | int(packed[3 * g + 1]) << 8 | int(packed[3 * g + 2]) << 16
        assert [(word >> (3 * i)) & 7 for i in range(8)] == idx[8 * g: 8 * g + 8]
    assert [orc.tq_packed_bytes(b, n) for b, n in ((1, 128), (1, 9), (2, 128), (2, 5), (3, 128), (3, 9))] == [16, 2, 32, 2, 48, 6]


def test_turboquant_rotation_kats(orc):
    """rotation.rs:141-229: round trip (128), norm preservation (64), non-power-of-two round trips (3 ... 100) and norm
    preservation (80), with the reference's inputs and tolerances; plus the butterfly itself on a delta (every output +-1/sqrt(d))."""
    x = np.arange(128, dtype=np.float32) * np.float32(0.01) - np.float32(0.64)         # test_roundtrip
    s = _signs(128, 42)
    rot = orc.tq_rotate(x, s)
    assert rot.size == 128 and np.all(np.abs(orc.tq_rotate_inverse(rot, 128, s) - x) < 1e-4)
    x = np.arange(64, dtype=np.float32) * np.float32(0.02) - np.float32(0.64)          # test_norm_preservation
    rot = orc.tq_rotate(x, _signs(64, 123))
    assert abs(float(np.sqrt(np.sum(x * x))) - float(np.sqrt(np.sum(rot * rot)))) < 1e-3
    for dim in (3, 5, 7, 10, 13, 17, 33, 65, 80, 96, 100):                             # test_non_power_of_two_roundtrip
        pd = orc.tq_padded_dim(dim)
        assert pd >= dim and pd & (pd - 1) == 0 and (pd == 1 or pd // 2 < dim)
        x = np.arange(dim, dtype=np.float32) * np.float32(0.01) - np.float32(0.5)
        s = _signs(pd, dim)
        rot = orc.tq_rotate(x, s)
        assert rot.size == pd and np.all(np.abs(orc.tq_rotate_inverse(rot, dim, s) - x) < 1e-3), dim
    x = np.arange(80, dtype=np.float32) * np.float32(0.02) - np.float32(0.8)           # test_non_power_of_two_norm_preservation
    rot = orc.tq_rotate(x, _signs(128, 123))
    assert abs(float(np.sqrt(np.sum(x * x))) - float(np.sqrt(np.sum(rot * rot)))) < 1e-3
    e = np.zeros(8, np.float32); e[3] = 1.0                                            # H D e_3: column 3 of the Hadamard matrix, signed
    s = np.array([1, -1, 1, 1, -1, 1, -1, 1], np.float32)
    want = np.array([(-1) ** bin(i & 3).count("1") for i in range(8)], np.float32) * s[3] / np.sqrt(np.float32(8))
    assert np.allclose(orc.tq_rotate(e, s), want, atol=0, rtol=0)


def test_turboquant_kv_cache_kats(orc):
    """kv_turboquant.rs:289-428 on the cache the oracle's model path uses: attention over compressed K / V is finite and points
    the right way (test_attention_direction, test_attention_layer_multi_head), the 2-bit cache is < 25 % of f32
    (test_memory_savings); write / truncate / shift_left bookkeeping is covered at model level (tests/test_gpu_model.py)."""
    dim, bits = 64, 2
    sk, sv = _signs(64, 1), _signs(64, 2)
    k_sim = np.arange(64, dtype=np.float32) * np.float32(0.01)                         # head 0 of the reference's 4 x 64 rows
    k_opp = -k_sim
    v1, v2 = np.full(64, 1.0, np.float32), np.full(64, 2.0, np.float32)
    kc = np.concatenate([orc.tq_compress(k_sim, bits, sk), orc.tq_compress(k_opp, bits, sk)])
    vc = np.concatenate([orc.tq_compress(v1, bits, sv), orc.tq_compress(v2, bits, sv)])
    scale = float(np.float32(1.0) / np.sqrt(np.float32(64.0)))
    out = orc.tq_attention_head(k_sim, kc, vc, 2, bits, sk, sv, scale)
    assert out.size == 64 and np.all(np.isfinite(out))
    # (the reference asserts no more than the length here: its rows are far from unit norm, where a 2-bit codebook saturates)
    # the softmax over the two positions: the output is the convex combination of the two decompressed value rows
    rq = orc.tq_rotate(k_sim, sk)
    s0, s1 = (np.float32(orc.tq_dot_with_packed(64, bits, rq, kc[16 * i: 16 * i + 16], 64)) * np.float32(scale) for i in range(2))
    assert s0 > s1                                                                      # the query IS the first key
    d = [orc.tq_rotate_inverse(orc.tq_dequantize_vector(64, bits, vc[16 * i: 16 * i + 16], 64), 64, sv) for i in range(2)]
    e = np.exp(np.array([s0, s1], np.float32) - max(s0, s1))
    w = e / np.float32(e[0] + e[1])
    assert np.allclose(out, w[0] * d[0] + w[1] * d[1], rtol=1e-6, atol=1e-7)
    k = np.full(64, 0.5, np.float32)                                                   # test_attention_layer_multi_head: one position
    out = orc.tq_attention_head(k, orc.tq_compress(k, bits, sk), orc.tq_compress(np.full(64, 1.0, np.float32), bits, sv), 1, bits, sk, sv, scale)
    assert np.all(np.isfinite(out))
    # one position: softmax weight 1 -> the output is the decompressed value row itself
    v = np.linspace(-1, 1, 64).astype(np.float32)
    vcodes = orc.tq_compress(v, bits, sv)
    want = orc.tq_rotate_inverse(orc.tq_dequantize_vector(64, bits, vcodes, 64), 64, sv)
    assert np.array_equal(orc.tq_attention_head(k, orc.tq_compress(k, bits, sk), vcodes, 1, bits, sk, sv, scale), want)
    assert orc.tq_packed_bytes(2, 128) * 2 * 4 * 2 < (128 * 4 * 2 * 4 * 2) // 4                                  # test_memory_savings
    # compress = rotate + quantize_vector (quant.rs:71-103)
    x = (np.arange(64, dtype=np.float32) - np.float32(32.0)) * np.float32(0.01)         # test_mse_compress_decompress input
    assert np.array_equal(orc.tq_compress(x, 2, sk), orc.tq_quantize_vector(64, 2, orc.tq_rotate(x, sk)))
    assert orc.tq_compress(x, 3, sk).size == 24


def _gauss(dim, seed):
    """A QJL projection matrix S[dim][dim] ~ N(0, 1) (an INPUT of the oracle and of the library: the reference draws it from
    rand's StdRng + rand_distr::StandardNormal, qjl.rs:44-52, which is not restated)."""
    return np.random.default_rng(seed).standard_normal((dim, dim)).astype(np.float32)


def test_turboquant_qjl_kats(orc):
    """TurboQuantProd (tq2-qjl / tq3-qjl): the reference's own deterministic unit tests of the sign-bit dot product (qjl.rs:236-250,
    backend/cpu/simd.rs:1286-1350) and of the entry size (quant.rs:340-352); the seeded-RNG tests (qjl.rs:186-234) as properties
    over matrices drawn here."""
    # test_dot_with_sign_bits_all_positive / _alternating
    assert orc.tq_dot_with_sign_bits([1, 2, 3, 4], [0xF], 4) == 10.0
    assert orc.tq_dot_with_sign_bits([1, 2, 3, 4], [0b0101], 4) == 1.0 - 2.0 + 3.0 - 4.0
    # simd.rs: test_dot_sign_bits_fast_basic / all_ones / all_zeros / odd_count / (1286-1300) alternating over 128
    assert orc.tq_dot_with_sign_bits([1, 2, 3, 4], [0b1010], 4) == 2.0
    v = np.arange(1, 129, dtype=np.float32)
    assert abs(orc.tq_dot_with_sign_bits(v, [2**64 - 1, 2**64 - 1], 128) - float(v.sum())) < 1e-2
    assert orc.tq_dot_with_sign_bits(np.ones(64, np.float32), [0], 64) == -64.0
    assert orc.tq_dot_with_sign_bits([1, 2, 3, 4, 5, 6, 7], [0b1111111], 7) == 28.0
    alt = int("01" * 32, 2)                                                             # even bits set: +, -, +, - ...
    want = np.float32(0.0)
    for i in range(128):
        want = np.float32(want + v[i] * np.float32(1.0 if i % 2 == 0 else -1.0))
    assert orc.tq_dot_with_sign_bits(v, [alt, alt], 128) == float(want)
    # test_bytes_per_entry
    assert orc.tq_bytes_per_entry(128, 2, False) == 128 // 4
    assert orc.tq_bytes_per_entry(128, 2, True) == 128 // 4 + (128 + 63) // 64 * 8 + 4
    assert orc.tq_bytes_per_entry(80, 3, True) == 48 + 16 + 4                            # padded to 128
    # test_compress_produces_correct_shape: 128 -> 2 words, norm > 0; the bits are the signs of S x, the norm is |x|
    S = _gauss(128, 42)
    x = (np.arange(128, dtype=np.float32) - np.float32(64.0)) * np.float32(0.01)
    bits, norm = orc.tq_qjl_compress(S, x)
    assert bits.size == 2 and norm > 0.0 and abs(norm - float(np.linalg.norm(x.astype(np.float64)))) < 1e-5
    z = S.astype(np.float64) @ x.astype(np.float64)
    got = np.array([(int(bits[i // 64]) >> (i % 64)) & 1 for i in range(128)])
    safe = np.abs(z) > 1e-4                                                             # away from the f32 / f64 rounding boundary
    assert np.array_equal(got[safe], (z >= 0)[safe].astype(int)) and safe.sum() > 120
    # test_fast_matches_slow: inner_product (projects the query row by row) == project_query + inner_product_fast
    S32 = _gauss(32, 99)
    key = np.arange(32, dtype=np.float32) * np.float32(0.02)
    query = (np.arange(32, dtype=np.float32) - np.float32(16.0)) * np.float32(0.01)
    kb, kn = orc.tq_qjl_compress(S32, key)
    pq = orc.tq_qjl_project(S32, query)
    fast = orc.tq_qjl_inner_product_fast(pq, kb, kn)
    slow = np.float32(0.0)
    for i in range(32):
        slow = np.float32(slow + pq[i] * np.float32(1.0 if (int(kb[0]) >> i) & 1 else -1.0))
    slow = np.float32(np.float32(np.sqrt(np.float32(np.pi / 2)) / np.float32(32)) * np.float32(kn)) * slow
    assert abs(float(slow) - fast) < 1e-6
    # test_inner_product_unbiased / test_inner_product_right_ballpark as a property: over many matrices the estimate's mean
    # converges to the true inner product (the estimator is unbiased: E[sqrt(pi/2)/d * |k| * sum_i (S q)_i sign((S k)_i)] = <q, k>)
    dim = 64
    xk = (np.arange(dim, dtype=np.float32) - np.float32(32.0)) * np.float32(0.01)
    yq = np.arange(dim, dtype=np.float32) * np.float32(0.01)
    true = float(xk.astype(np.float64) @ yq.astype(np.float64))
    est = []
    for seed in range(200):
        Sm = _gauss(dim, 1000 + seed)
        b, n = orc.tq_qjl_compress(Sm, xk)
        est.append(orc.tq_qjl_inner_product_fast(orc.tq_qjl_project(Sm, yq), b, n))
    est = np.array(est)
    assert abs(est.mean() - true) < 4 * est.std() / np.sqrt(len(est)) + 1e-3, (est.mean(), true, est.std())
    assert np.all(np.abs(est - true) / abs(true) < 2.0)                                  # "in the right ballpark", every draw


def test_turboquant_prod_engine_kats(orc):
    """TurboQuantEngine with use_qjl (quant.rs:71-168): compress = MSE codes + QJL of the residual; attention_score = polar +
    correction; the correction shrinks the score error of the codes on average (what TurboQuant_prod is for)."""
    dim, bits = 64, 2
    sk, sv = _signs(64, 1), _signs(64, 2)
    S = _gauss(64, 7)
    x = (np.arange(64, dtype=np.float32) - np.float32(32.0)) * np.float32(0.01)         # test_prod_compress_has_qjl input
    codes, qb, rn = orc.tq_compress_qjl(x, bits, sk, S)
    assert np.array_equal(codes, orc.tq_compress(x, bits, sk))                          # the MSE part is unchanged
    rot = orc.tq_rotate(x, sk)
    res = rot - orc.tq_dequantize_vector(64, bits, codes, 64)                           # f32 subtraction, as quant.rs:85-89
    b2, n2 = orc.tq_qjl_compress(S, res)
    assert np.array_equal(qb, b2) and rn == n2 and rn > 0.0
    # one cached position: attention output = that position's decompressed V row, whatever the score
    v = np.linspace(-1, 1, 64).astype(np.float32)
    vcodes = orc.tq_compress(v, bits, sv)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(64.0)))
    out = orc.tq_attention_head_qjl(x, codes, qb, [rn], vcodes, 1, bits, sk, sv, S, scale)
    assert np.array_equal(out, orc.tq_rotate_inverse(orc.tq_dequantize_vector(64, bits, vcodes, 64), 64, sv))
    # scores: polar + correction (quant.rs:152-166) against the true inner product, unit-norm rows (the codebook's regime)
    rng = np.random.default_rng(5)
    err_mse, err_prod = [], []
    for t in range(64):
        k = rng.standard_normal(64).astype(np.float32); k /= np.float32(np.linalg.norm(k))
        q = rng.standard_normal(64).astype(np.float32); q /= np.float32(np.linalg.norm(q))
        Sm = _gauss(64, 100 + t)
        c, b, n = orc.tq_compress_qjl(k, bits, sk, Sm)
        rq = orc.tq_rotate(q, sk)
        polar = orc.tq_dot_with_packed(64, bits, rq, c, 64)
        corr = orc.tq_qjl_inner_product_fast(orc.tq_qjl_project(Sm, rq), b, n)
        true = float(k.astype(np.float64) @ q.astype(np.float64))
        err_mse.append(polar - true)
        err_prod.append(polar + corr - true)
    # the MSE codes shrink inner products (biased towards 0); the correction removes the bias: mean error closer to 0
    assert abs(np.mean(err_prod)) <= abs(np.mean(err_mse)) + 0.02
    # two positions: softmax over (polar + correction) * scale, output = convex combination of the decompressed V rows
    k0 = rng.standard_normal(64).astype(np.float32); k0 /= np.float32(np.linalg.norm(k0))
    k1 = -k0
    c0, b0, n0 = orc.tq_compress_qjl(k0, bits, sk, S)
    c1, b1, n1 = orc.tq_compress_qjl(k1, bits, sk, S)
    v0, v1 = np.full(64, 1.0, np.float32), np.full(64, 2.0, np.float32)
    vc = np.concatenate([orc.tq_compress(v0, bits, sv), orc.tq_compress(v1, bits, sv)])
    out = orc.tq_attention_head_qjl(k0, np.concatenate([c0, c1]), np.concatenate([b0, b1]), [n0, n1], vc, 2, bits, sk, sv, S, 1.0)
    rq = orc.tq_rotate(k0, sk)
    pq = orc.tq_qjl_project(S, rq)
    s = [np.float32(np.float32(orc.tq_dot_with_packed(64, bits, rq, c, 64)) + np.float32(orc.tq_qjl_inner_product_fast(pq, b, n)))
         for c, b, n in ((c0, b0, n0), (c1, b1, n1))]
    assert s[0] > s[1]
    e = np.exp(np.array(s, np.float32) - max(s))
    w = e / np.float32(e[0] + e[1])
    d = [orc.tq_rotate_inverse(orc.tq_dequantize_vector(64, bits, vc[16 * i: 16 * i + 16], 64), 64, sv) for i in range(2)]
    assert np.allclose(out, w[0] * d[0] + w[1] * d[1], rtol=1e-6, atol=1e-7)
