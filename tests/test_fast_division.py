"""The sort passes divide by run-time constants (bin edge, rows per partition, bins per column
block) with one multiply-high: n / d == (n * m) >> 32, m = floor(2^32 / d) + 1, valid while
n * d < 2^32 (DsmParams::mul_*, make_dsm_params; div_by() in amhip_sort.hip).  The identity itself,
checked on the host over every divisor the plan can produce."""
import numpy as np


def test_multiply_high_division_is_exact_below_the_bound():
    rng = np.random.default_rng(1)
    for d in list(range(2, 300)) + [511, 512, 513, 1000, 2048, 4095, 4096, 65535, 65536, 1 << 20]:
        m = (1 << 32) // d + 1
        assert m < (1 << 32)
        n_max = ((1 << 32) - 1) // d            # n * d < 2^32
        ns = np.unique(np.concatenate([
            np.arange(0, min(n_max, 5000) + 1, dtype=np.uint64),
            np.array([n_max, max(n_max - 1, 0), n_max // 2], dtype=np.uint64),
            rng.integers(0, n_max + 1, 20000, dtype=np.uint64),
            (np.arange(1, min(n_max // d, 3000) + 1, dtype=np.uint64) * np.uint64(d)),          # multiples
            (np.arange(1, min(n_max // d, 3000) + 1, dtype=np.uint64) * np.uint64(d) - np.uint64(1)),
        ]))
        ns = ns[ns <= n_max]
        q = (ns * np.uint64(m)) >> np.uint64(32)
        assert np.array_equal(q, ns // np.uint64(d)), d
