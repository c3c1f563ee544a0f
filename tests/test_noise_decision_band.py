"""CPU job: the error budget of the step kernel's fp32 accept / reject decision (afe_kernels.hip,
three_accepted).  The kernel evaluates r2~ = x~^2 + y~^2 in fp32 from the HIGH engine word of each uniform
and trusts it outside a +-1e-5 band around 1 (and above 1e-5); inside the band the exact libstdc++
arithmetic decides.  This test re-enacts that fp32 arithmetic with numpy (one rounding per fma, as the
instructions do) on millions of random engine-word quadruples and checks it against the exact double
computation: the error must stay far inside the band, and no sure decision may ever disagree with libstdc++'s
`!(r2 > 1.0 || r2 == 0.0)`.  (The GPU test test_device_normals_match_libstdcxx_known_answers checks the
same thing end to end: engine words after the draws are bit-identical for 200 000 seeds.)"""
import numpy as np

R = 2147483646.0


def _canonical(lo, hi):
    """std::generate_canonical<double, 53>(minstd_rand0): (lo - 1) + (hi - 1) * R, divided by R^2"""
    return ((lo - 1).astype(np.float64) + (hi - 1).astype(np.float64) * R) / (R * R)


def _fp32_estimate(hx, hy):
    k = np.float32(2.0) * np.float32(1.0 / 2147483646.0)              # 2.0f * kInvR, folded in float

    def coord(h):
        hf = (h - 1).astype(np.float32)                               # v_cvt_f32_u32
        return (hf.astype(np.float64) * np.float64(k) - 1.0).astype(np.float32)    # v_fma_f32: one rounding
    xf, yf = coord(hx), coord(hy)
    yy = (yf * yf).astype(np.float32)                                 # v_mul_f32
    return (xf.astype(np.float64) * xf.astype(np.float64) + yy.astype(np.float64)).astype(np.float32)   # v_fma_f32


def test_fp32_decision_never_contradicts_the_exact_one():
    rng = np.random.default_rng(20261003)
    worst, n_unsure, n = 0.0, 0, 0
    for _ in range(4):
        w = rng.integers(1, 2147483647, size=(4, 1_000_000), dtype=np.int64)   # lo_x, hi_x, lo_y, hi_y
        x = 2.0 * _canonical(w[0], w[1]) - 1.0
        y = 2.0 * _canonical(w[2], w[3]) - 1.0
        r2 = x * x + y * y
        r2f = _fp32_estimate(w[1], w[3])
        worst = max(worst, float(np.abs(r2f.astype(np.float64) - r2).max()))
        sure_accept = (r2f < np.float32(1.0) - np.float32(1e-5)) & (r2f > np.float32(1e-5))
        sure_reject = r2f > np.float32(1.0) + np.float32(1e-5)
        exact_accept = ~((r2 > 1.0) | (r2 == 0.0))
        assert exact_accept[sure_accept].all()
        assert not exact_accept[sure_reject].any()
        n_unsure += int((~sure_accept & ~sure_reject).sum())
        n += w.shape[1]
    assert worst < 1e-6, worst                 # measured 3.1e-7: a 30-fold margin inside the 1e-5 band
    assert n_unsure / n < 1e-4                 # the exact path runs for ~2e-5 of the candidates


def test_worst_case_words():
    """corners of the word range, and r2 pushed against 1 from both sides along the axes"""
    m = 2147483646
    h = np.array([1, 2, m // 2, m // 2 + 1, m - 1, m, m // 2 + 7, 3], np.int64)
    hx, hy = np.meshgrid(h, h)
    hx, hy = hx.ravel(), hy.ravel()
    for lo in (1, m):
        los = np.full_like(hx, lo)
        x = 2.0 * _canonical(los, hx) - 1.0
        y = 2.0 * _canonical(los, hy) - 1.0
        r2 = x * x + y * y
        r2f = _fp32_estimate(hx, hy)
        assert np.abs(r2f.astype(np.float64) - r2).max() < 1e-6
