"""The arithmetic model behind the split float16 conv stack (adapted_amd/csrc/cnn_conv_split.h), checked without a GPU:
a float32 value carried as hi = RN16(v), lo = RN16((v - hi) * 2^11) keeps 22 significant bits whatever its magnitude, and
a 448-term dot product from split operands -- hi hi + (hi lo + lo hi) 2^-11, float16 products exact in float32 accumulators --
lands as close to the exact sum as a float32 fmaf chain does.  (The device kernels themselves are compared with torch's float32
conv and with the exact-float32 MFMA kernels in tests/test_gpu_cnn.py.)"""
import numpy as np


def split(v):
    v = np.asarray(v, dtype=np.float32)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def join(hi, lo):
    return hi.astype(np.float64) + lo.astype(np.float64) / 2048.0


def test_split_keeps_22_bits_over_the_stored_range():
    rng = np.random.default_rng(1)
    # magnitudes from 1e-6 (float16 subnormals of the hi part) to 3e4 (the top of the stored range)
    v = (rng.uniform(0.5, 1.0, 200000) * 10.0 ** rng.uniform(-6, 4.45, 200000)).astype(np.float32)
    hi, lo = split(v)
    assert np.isfinite(hi.astype(np.float32)).all() and np.isfinite(lo.astype(np.float32)).all()
    rel = np.abs(join(hi, lo) - v.astype(np.float64)) / v.astype(np.float64)
    normal = v >= 2.0 ** -14  # hi is a normal float16 there: 11 + 11 bits
    assert rel[normal].max() <= 2.0 ** -22
    # below, the hi part is a float16 subnormal (absolute step 2^-24), the scaled residual still recovers the value to 2^-24 * 2^-11
    assert np.abs(join(hi, lo) - v.astype(np.float64))[~normal].max() <= 2.0 ** -35
    # powers of two scale through exactly (the per-layer weight scale, the 2^-4 of the stored activations)
    h2, l2 = split(v * np.float32(0.0625))
    ok = v * np.float32(0.0625) >= 2.0 ** -14
    assert np.array_equal(h2[ok].astype(np.float32) * 16.0, hi[ok].astype(np.float32))


def test_split_dot_product_is_as_close_as_a_float32_chain():
    rng = np.random.default_rng(2)
    K, N = 448, 4000
    a = np.maximum(rng.normal(0.3, 1.0, (N, K)), 0.0).astype(np.float32)          # activations behind a ReLU
    w = (rng.normal(0.0, 0.05, (N, K)) * (rng.random((N, K)) < 0.9)).astype(np.float32)
    exact = (a.astype(np.float64) * w.astype(np.float64)).sum(axis=1)
    # the float32 matrix instruction: one fmaf chain in k order
    chain = np.zeros(N, dtype=np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)  # (one rounding per step)
    # split operands: weights scaled by a power of two, three float16 products per term (exact in float32), two float32 accumulators
    sw = np.float32(2.0 ** 17)
    ah, al = split(a)
    wh, wl = split(w * sw)
    f32 = np.float32
    main = np.zeros(N, dtype=np.float32)
    cross = np.zeros(N, dtype=np.float32)
    for k0 in range(0, K, 16):  # a 32x32x16 instruction adds 16 products to the accumulator
        s = slice(k0, k0 + 16)
        main = (main.astype(np.float64) + (ah[:, s].astype(np.float64) * wh[:, s].astype(np.float64)).sum(axis=1)).astype(f32)
        cross = (cross.astype(np.float64) + (ah[:, s].astype(np.float64) * wl[:, s].astype(np.float64)).sum(axis=1)).astype(f32)
        cross = (cross.astype(np.float64) + (al[:, s].astype(np.float64) * wh[:, s].astype(np.float64)).sum(axis=1)).astype(f32)
    got = (main.astype(np.float64) + cross.astype(np.float64) / 2048.0) / float(sw)
    scale = (np.abs(a.astype(np.float64) * w.astype(np.float64))).sum(axis=1)
    err_split = np.abs(got - exact) / scale
    err_chain = np.abs(chain.astype(np.float64) - exact) / scale
    # both far below the 2e-5 of the score scale the conv-stack tests allow; the split sum within a small factor of the chain
    assert err_chain.max() < 2e-7 and err_split.max() < 4e-7
    assert np.median(err_split) < 4 * np.median(err_chain) + 1e-9
