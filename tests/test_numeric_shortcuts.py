"""Host-side proofs-by-enumeration of the arithmetic shortcuts the HIP kernel
uses in place of the reference's literal operations (csrc/rvo3d_math.hpp, rvo3d_pairs.hpp).
Each must be EXACTLY equivalent, not just close."""
import ctypes
import math

import numpy as np

libm = ctypes.CDLL("libm.so.6")
libm.fma.restype = ctypes.c_double
libm.fma.argtypes = [ctypes.c_double] * 3


def sq_threshold(tau):
    x = tau * tau
    while math.sqrt(np.nextafter(x, np.inf)) <= tau:
        x = float(np.nextafter(x, np.inf))
    while math.sqrt(x) > tau:
        x = float(np.nextafter(x, -np.inf))
    return x


def test_squared_thresholds_equal_sqrt_compare():
    """norm <= tau  <=>  norm^2 <= T(tau)   (gate 10, building gate 5, goal 0.4)."""
    rng = np.random.default_rng(0)
    for tau in (10.0, 5.0, 0.4):
        T = sq_threshold(tau)
        x = T
        for _ in range(2000):  # walk 2000 ulps either side of the boundary
            x = float(np.nextafter(x, np.inf))
            assert math.sqrt(x) > tau
        x = T
        for _ in range(2000):
            assert math.sqrt(x) <= tau
            x = float(np.nextafter(x, -np.inf))
        xs = rng.uniform(0, 4 * tau * tau, 200000)
        assert np.array_equal(np.sqrt(xs) <= tau, xs <= T)


def test_k_over_1000_is_correctly_rounded():
    ks = np.concatenate([np.arange(-70000, 70001), np.random.default_rng(1).integers(-2**40, 2**40, 200000)])
    for k in ks.astype(np.float64):
        q = k * 0.001
        r = libm.fma(-q, 1000.0, k)
        assert q + r * 0.001 == k / 1000.0, k


def test_float32_of_k_over_100_needs_no_division():
    """float32(rint(x*100) * 0.01) == float32(rint(x*100) / 100) for |k| < 2^24."""
    ks = np.concatenate([np.arange(-300000, 300001),
                         np.random.default_rng(2).integers(-2**24 + 1, 2**24, 3000000)]).astype(np.float64)
    assert np.array_equal((ks * 0.01).astype(np.float32), (ks / 100.0).astype(np.float32))


def test_python_round_vs_numpy_round_integer_compare():
    """alpha > beta on k/100 doubles is the integer compare alpha_c > beta_c."""
    ks = np.arange(0, 400, dtype=np.float64)
    v = ks / 100.0
    assert np.all(np.diff(v) > 0)
    for k in range(0, 400):
        assert round(k / 100.0, 2) == k / 100.0 and float(np.round(np.float64(k / 100.0), 2)) == k / 100.0


def test_angle_bin_cosines():
    """Bin edges of ir_gym.py:91-100 as cosines (device compares c, not acos(c))."""
    assert math.cos(math.pi / 18) == 0.984807753012208
    assert math.cos(math.pi / 6) == 0.8660254037844387
    assert math.cos(math.pi / 3) == 0.5000000000000001
    assert math.cos(math.pi / 2) == 6.123233995736766e-17
    assert math.acos(0.0) == math.pi / 2
    rng = np.random.default_rng(3)
    c = rng.uniform(-1, 1, 200000)
    edges = [math.cos(math.pi / 18), math.cos(math.pi / 6), math.cos(math.pi / 3), math.cos(math.pi / 2)]
    firm = np.all(np.abs(c[:, None] - np.array(edges)[None]) > 1e-12, axis=1)
    ang = np.arccos(c)
    by_ang = np.select([ang < math.pi / 18, ang < math.pi / 6, ang < math.pi / 3, ang < math.pi / 2], [3, 1, .5, 0], -4)
    by_cos = np.select([c > edges[0], c > edges[1], c > edges[2], c > edges[3]], [3, 1, .5, 0], -4)
    assert np.array_equal(by_ang[firm], by_cos[firm])


def test_cone_prefilter_is_conservative():
    """Pairs the device pre-filter rejects (cos beta < cos(alpha + 1e-4)) are never
    inside the cone by the reference's rounded test alpha > beta."""
    rng = np.random.default_rng(4)
    n = 400000
    d2 = rng.uniform(0.17, 100, n)
    R = 0.4
    w2 = rng.uniform(0.01, 9, n)
    nab, nw = np.sqrt(d2), np.sqrt(w2)
    alpha = np.arcsin(R / nab)
    beta = np.where(rng.random(n) < 0.7, alpha + rng.normal(0, 3e-4, n), rng.uniform(0, np.pi / 2, n)).clip(1e-9, np.pi / 2)
    dp = np.cos(beta) * nab * nw
    K = 0.999999995 * np.sqrt(d2 - R * R) - 1.0e-4 * R
    rejected = (K > 0) & (dp * dp < (w2 * (K * K)) * (1 - 1e-9))
    alpha_c = np.array([round(float(x), 2) for x in np.arcsin(R / nab)])
    beta_c = np.round(np.arccos(dp / (nab * nw)), 2)
    inside = alpha_c > beta_c
    assert rejected.sum() > 1000 and inside.sum() > 1000
    assert not (rejected & inside).any()


def test_fmod_shortcut_for_one_period_either_side():
    """np_mod's fast path (rvo3d_math.hpp): fmod(a, b) is a for |a| < b and a - b, exactly, for
    b <= a < 2b; the remainder logic of npy_divmod follows unchanged."""
    rng = np.random.default_rng(5)
    b = 360.0
    a = np.concatenate([rng.uniform(-360, 720, 200000), np.array([0.0, -0.0, 360.0, 359.99999999999994, 360.00000000000006,
                                                                   719.9999999999999, -359.99999999999994, 90.0, 450.0, -90.0]),
                        np.round(rng.uniform(-360, 720, 50000), 2)])
    a = a[(a > -b) & (a < 2 * b)]
    fast = np.where(a >= b, a - b, a)
    ref = np.fmod(a, b)
    assert np.array_equal(fast, ref) and np.array_equal(np.signbit(fast), np.signbit(ref))
    # and the whole remainder, against numpy's own
    m = fast.copy()
    nz = m != 0
    m[nz & (m < 0)] += b
    m[~nz] = 0.0
    assert np.array_equal(m, np.mod(a, b))


def test_doubling_w_is_exact_in_float32():
    """Stage X1 works with 2 w = 4 a - (v_i + v_j) instead of w = 2 a - (v_i + v_j) / 2
    (rvo3d_pairs.hpp): the two are exactly a factor 2 apart in float32, operation by operation, so
    every comparison of the cone filter - homogeneous of degree 2 in w - decides the same."""
    rng = np.random.default_rng(6)
    n = 500000
    f = np.float32
    a = np.round(rng.uniform(-3, 3, n), 2).astype(f)
    vi = rng.normal(0, 1.2, n).astype(f)
    vj = rng.normal(0, 1.2, n).astype(f)
    vi[rng.random(n) < 0.1] = 0
    w = f(2) * a - f(0.5) * (vi + vj)
    w2 = f(4) * a - (vi + vj)
    assert w.dtype == np.float32 and w2.dtype == np.float32
    assert np.array_equal(f(2) * w, w2)
    d = rng.uniform(-10, 10, n).astype(f)
    assert np.array_equal(f(2) * (d * w), d * w2) and np.array_equal(f(4) * (w * w), w2 * w2)
