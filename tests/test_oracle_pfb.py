"""CPU: the channelizer oracle (oracle/oracle_pfb.c) against an independent float64 numpy statement
of the same definition, and the properties that make it a channelizer."""
import numpy as np
import pytest


def _direct(x, h, M):
    D, L = M // 2, M * 16
    n_out = (x.size - L) // D + 1
    y = np.zeros((M, n_out), dtype=np.complex128)
    r = np.arange(M)
    for m in range(n_out):
        seg = x[m * D:m * D + L].astype(np.complex128) * h
        u = seg.reshape(16, M).sum(axis=0)                      # u_m[r] = sum_p h[r+pM] x[mD+r+pM]
        X = np.fft.fft(u)                                       # sum_r u[r] e^{-2 pi i k r / M}
        y[:, m] = X * ((-1.0) ** (r * m))
    return y


@pytest.mark.parametrize("M", [40, 16])
def test_oracle_matches_float64_definition(oracle, M):
    rng = np.random.default_rng(M)
    n = M * 16 + (M // 2) * 200 + 7
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    h = oracle.pfb_proto(M).astype(np.float64)
    got = oracle.pfb(x, M)
    want = _direct(x, h, M)
    assert got.shape == want.shape == (M, oracle.pfb_nout(n, M))
    scale = np.abs(want).max()
    assert np.max(np.abs(got - want)) <= 2e-6 * scale          # f32 arithmetic, ~700 terms per output


@pytest.mark.parametrize("M", [40, 16])
def test_tone_lands_in_its_bin(oracle, M):
    n = M * 16 + (M // 2) * 400
    t = np.arange(n)
    for k in (0, 3, M // 2, M - 1):
        x = np.exp(2j * np.pi * k * t / M).astype(np.complex64)
        p = (np.abs(oracle.pfb(x, M)) ** 2).mean(axis=1)
        assert np.argmax(p) == k
        others = np.delete(p, [k, (k - 1) % M, (k + 1) % M])
        assert others.max() < 1e-6 * p[k]                        # > 60 dB to non-adjacent bins


def test_short_input_has_no_output(oracle):
    assert oracle.pfb_nout(639, 40) == 0 and oracle.pfb_nout(640, 40) == 1 and oracle.pfb_nout(660, 40) == 2
    assert oracle.pfb(np.zeros(100, np.complex64), 16).shape == (16, 0)


def test_block_order_entry_differs_only_where_a_structural_zero_meets_a_non_finite_sample(oracle):
    """`oracle_pfb_block_order` (what the experimental matrix-pipe FIR equals bit for bit) runs the FIR as banded-Toeplitz
    matrix blocks: the same fmaf chain per output on finite input -- bit for bit --, but the blocks' structural zeros
    multiply every sample of the block, so a NaN / Inf reaches outputs whose 16-tap window does not hold it."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(40 * 16 + 20 * 300) + 1j * rng.standard_normal(40 * 16 + 20 * 300)).astype(np.complex64)
    a, b = oracle.pfb(x, 40), oracle.pfb(x, 40, block_order=True)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x[3000] = np.nan
    a, b = oracle.pfb(x, 40), oracle.pfb(x, 40, block_order=True)
    na, nb = np.isnan(a.view(np.float32)), np.isnan(b.view(np.float32))
    assert na.any() and (nb | ~na).all() and nb.sum() > na.sum()          # the plain chain's NaNs are a strict subset
    assert np.array_equal(a.view(np.uint32)[~nb], b.view(np.uint32)[~nb])


def test_the_prime_factor_fft_and_the_legacy_cooley_tukey_entry(oracle):
    """Since round 4 the M = 40 DFT is specified as a prime-factor 8 x 5 decomposition (no twiddle products); the
    Cooley-Tukey form of rounds 1-3 stays behind `legacy_fft` as the specification of the A/B partner kernels.  Both are the
    same DFT: they differ from each other and from a float64 DFT in the last bits only, the switch is per call (it does not
    stick), and M = 16 has one form."""
    rng = np.random.default_rng(11)
    n = 40 * 16 + 20 * 500 + 3
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    want = _direct(x, oracle.pfb_proto(40).astype(np.float64), 40)
    scale = np.abs(want).max()
    pfa, ct, again = oracle.pfb(x, 40), oracle.pfb(x, 40, legacy_fft=True), oracle.pfb(x, 40)
    assert np.array_equal(pfa.view(np.uint32), again.view(np.uint32))
    assert not np.array_equal(pfa.view(np.uint32), ct.view(np.uint32))
    assert np.max(np.abs(pfa - ct)) <= 5e-7 * scale
    assert np.max(np.abs(pfa - want)) <= 2e-6 * scale and np.max(np.abs(ct - want)) <= 2e-6 * scale
    x16 = x[:16 * 16 + 8 * 300]
    assert np.array_equal(oracle.pfb(x16, 16).view(np.uint32), oracle.pfb(x16, 16, legacy_fft=True).view(np.uint32))
