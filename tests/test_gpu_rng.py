"""-m gpu: the device RNG layer against the oracle (and through it against real libstdc++)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pcg32_shuffle_uniform_bit_exact(oracle):
    import alphazero as az
    assert np.array_equal(az.rng_probe("pcg32", 12345, 4), [1411482639, 3165192603, 3360792183, 2433038347])
    assert np.array_equal(az.rng_probe("pcg32", 99, 10000), oracle.pcg32_outputs(99, 10000))
    for n in (1, 2, 3, 6, 7, 8, 33, 112, 113, 250):
        assert np.array_equal(az.rng_probe("shuffle", 17 + n, n, reps=50), oracle.shuffle_iota(17 + n, n, 50)), n
    assert np.array_equal(az.rng_probe("shuffle", 12345, 7)[0], [1, 2, 3, 0, 6, 4, 5])
    assert np.array_equal(az.rng_probe("uniform", 5, 100000), oracle.uniform01(5, 100000))


@pytest.mark.parametrize("alpha", [10.83 / 7, 10.83 / 2, 0.3, 0.77, 10.83e-6, 5.0])
def test_gamma_bit_exact_vs_oracle(oracle, alpha):
    import alphazero as az
    for fresh in (False, True):
        dev = az.rng_probe("gamma_fresh" if fresh else "gamma", 77, 50000, param=alpha)
        ref = oracle.gamma(77, alpha, 50000, fresh_each=fresh)
        bad = np.flatnonzero(dev != ref)
        assert bad.size == 0, (alpha, fresh, bad[:5], dev[bad[:5]], ref[bad[:5]])
