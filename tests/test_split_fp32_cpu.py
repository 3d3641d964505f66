"""Split fp32 (csrc/gemm.hip.h, PREC_X3), the arithmetic fact it rests on, checked on the CPU: an fp32 number is the EXACT sum of three
bf16 numbers, and the three cross products the kernel drops are below fp32 rounding."""
import numpy as np


def bf16_round(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)



def test_three_bf16_terms_are_an_exact_split():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.uniform(-6, 6, 200000), [0.0, 1.0, -1.0, 3.0e38, 1e-30]]).astype(np.float32)
    p0 = bf16_round(x)
    r1 = x - p0
    p1 = bf16_round(r1)
    r2 = r1 - p1
    p2 = bf16_round(r2)
    assert np.array_equal(p2, r2), "the third term must be representable in bf16"
    assert np.array_equal((p0.astype(np.float64) + p1 + p2).astype(np.float32), x)
    assert np.array_equal(p0.astype(np.float64) + p1 + p2, x.astype(np.float64)), "x = p0 + p1 + p2 exactly"
    # the three dropped cross terms (p1 w2 + p2 w1 + p2 w2) against |x w|
    w = rng.standard_normal(x.size).astype(np.float32)
    q0 = bf16_round(w); q1 = bf16_round(w - q0); q2 = bf16_round(w - q0 - q1)
    dropped = p1.astype(np.float64) * q2 + p2.astype(np.float64) * q1 + p2.astype(np.float64) * q2
    ok = (np.abs(x) > 1e-25) & (np.abs(x) < 1e30) & (w != 0)
    rel = dropped[ok] / np.abs(x[ok].astype(np.float64) * w[ok])
    half_ulp = 2.0 ** -24
    assert np.abs(rel).max() <= 2 * half_ulp             # worst case: one fp32 ulp of the product
    assert np.sqrt((rel ** 2).mean()) <= 0.15 * half_ulp  # rms: ~0.1 of half an ulp (a single fp32 rounding of the product has 0.43)
    assert abs(rel.mean()) <= 1e-3 * half_ulp             # and no bias: the remainders of round-to-nearest are zero-mean


