"""Per-kernel parity: each HIP kernel vs its oracle twin, bit-exact on seeded inputs."""
import ctypes
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu
GEN = 14293326489335486720


@pytest.mark.parametrize("form", [0, 1])
def test_poseidon_permute_batch(form):
    """form 0: literal partial rounds; 1: the fast form (one S-box, one 11-term dot product, 11 multiply-adds per round) the leaf-hash kernels
    use where the chip is full. Inputs include the extremes of the lazy arithmetic (0, p-1, 2^32-1 patterns)."""
    rng = np.random.default_rng(1)
    st = rand_felts(rng, (1000, 12))
    st[1] = 0
    st[2] = np.arange(12, dtype=np.uint64)
    P = 0xFFFFFFFF00000001
    st[3] = P - 1
    st[4] = 0xFFFFFFFF
    st[5] = 0xFFFFFFFF00000000
    st[6, ::2] = P - 1; st[6, 1::2] = 0
    st[7, 0] = P - 1; st[7, 1:] = 0
    st[8, 0] = 0; st[8, 1:] = P - 1
    exp = st.copy()
    O = oracle()
    for i in range(exp.shape[0]):
        O.orc_permute(ptr(exp[i]))
    got = st.copy()
    gpu_check(qpgpu().qpgpu_test_permute_form(ptr(got), got.shape[0], form))
    assert np.array_equal(got, exp)
    assert got[1, 0] == 0x3c18a9786cb0b359 and got[2, 0] == 0xd64e1e3efc5b8e9e   # SURVEY.md App. A.2 vectors


@pytest.mark.parametrize("log_n", [1, 3, 6, 10, 12, 13, 14, 17])
def test_ntt_forward_dif_and_inverse_dit(log_n):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    cols = 3
    a = rand_felts(rng, (cols, n))
    rev = bitrev_indices(log_n)
    O = oracle()
    exp = a.copy()
    for c in range(cols):
        O.orc_ntt(ptr(exp[c]), n, 0)
    got = a.copy()
    gpu_check(qpgpu().qpgpu_test_ntt(ptr(got), cols, log_n, 0))          # DIF: natural -> bit-reversed
    assert np.array_equal(got, exp[:, rev])
    # inverse DIT takes the bit-reversed values back to natural-order n * coefficients
    back = got.copy()
    gpu_check(qpgpu().qpgpu_test_ntt(ptr(back), cols, log_n, 3))
    ninv = O.orc_inv(n)
    exp_back = np.array([[O.orc_mul(int(x), ninv) for x in row] for row in back[:, :min(n, 64)]], dtype=np.uint64)
    assert np.array_equal(exp_back, a[:, :min(n, 64)])


@pytest.mark.parametrize("log_n", [2, 6, 12, 13, 14])
def test_ntt_dit_forward_and_dif_inverse(log_n):
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    a = rand_felts(rng, (2, n))
    rev = bitrev_indices(log_n)
    O = oracle()
    exp = a.copy()
    for c in range(2):
        O.orc_ntt(ptr(exp[c]), n, 0)
    got = np.ascontiguousarray(a[:, rev])
    gpu_check(qpgpu().qpgpu_test_ntt(ptr(got), 2, log_n, 1))              # DIT: bit-reversed -> natural
    assert np.array_equal(got, exp)
    inv = a.copy()
    for c in range(2):
        O.orc_ntt(ptr(inv[c]), n, 1)
    got = a.copy()
    gpu_check(qpgpu().qpgpu_test_ntt(ptr(got), 2, log_n, 2))              # inverse DIF, unscaled, bit-reversed out
    ninv = O.orc_inv(n)
    got_nat = got[:, rev]
    chk = np.array([[O.orc_mul(int(x), ninv) for x in row[:64]] for row in got_nat], dtype=np.uint64)
    assert np.array_equal(chk, inv[:, :64])


@pytest.mark.parametrize("log_n", [4, 8, 9, 10, 11, 12, 13, 14, 15])
def test_intt_values_to_natural_coefficients(log_n):
    rng = np.random.default_rng(300 + log_n)
    n = 1 << log_n
    vals = rand_felts(rng, (3, n))
    exp = vals.copy()
    for c in range(3):
        oracle().orc_ntt(ptr(exp[c]), n, 1)
    got = np.zeros_like(vals)
    gpu_check(qpgpu().qpgpu_test_intt_natural(ptr(vals), 3, log_n, ptr(got)))
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("log_n,shift", [(3, GEN), (6, GEN), (9, GEN), (10, 7), (11, GEN), (12, GEN), (13, GEN), (14, GEN), (15, GEN)])
def test_lde_leaf_order(log_n, shift):
    rng = np.random.default_rng(200 + log_n)
    n = 1 << log_n
    cols, rate = 3, 3
    N = n << rate
    coeffs = rand_felts(rng, (cols, n))
    exp = np.zeros((cols, N), dtype=np.uint64)
    O = oracle()
    for c in range(cols):
        O.orc_coset_lde(ptr(coeffs[c]), n, rate, shift, ptr(exp[c]))
    got = np.zeros((cols, N), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_lde(ptr(coeffs), cols, log_n, rate, shift, ptr(got)))
    rev = bitrev_indices(log_n + rate)
    assert np.array_equal(got, exp[:, rev])            # leaf L holds the value at natural index rev(L)


@pytest.mark.parametrize("width,leaves,cap", [(135, 512, 4), (139, 2048, 4), (24, 8192, 4), (135, 4096, 4), (84, 256, 4), (20, 64, 4), (3, 64, 2), (4, 32, 0), (32, 16, 4), (9, 4096, 4)])
def test_merkle_tree(width, leaves, cap):
    rng = np.random.default_rng(width * 7 + leaves)
    cols = rand_felts(rng, (width, leaves))
    rows = np.ascontiguousarray(cols.T)
    O = oracle()
    tot = 0
    m = leaves
    while True:
        tot += m
        if m <= (1 << cap):
            break
        m >>= 1
    exp = np.zeros((tot, 4), dtype=np.uint64)
    k = O.orc_merkle(ptr(rows), leaves, width, cap, ptr(exp))
    assert k == tot
    got = np.zeros((tot, 4), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_merkle(ptr(cols), width, leaves, cap, ptr(got)))
    assert np.array_equal(got, exp)


def test_bitrev_permute():
    rng = np.random.default_rng(5)
    a = rand_felts(rng, (2, 1 << 9))
    out = np.zeros_like(a)
    gpu_check(qpgpu().qpgpu_test_bitrev(ptr(a), 2, 9, 1, ptr(out)))
    assert np.array_equal(out, a[:, bitrev_indices(9)])


@pytest.mark.parametrize("log_n", [13, 14, 15, 16, 17, 18])
def test_lde_strided_radix_path(log_n):
    """>= 64 transforms and n >= 2^13 take the strided radix passes (register radix-2^M for 13..16, LDS radix-256 + ntt16<9/10> for
    17/18) instead of the tiled kernel."""
    rng = np.random.default_rng(400 + log_n)
    n = 1 << log_n
    cols, rate = 8, 3
    N = n << rate
    coeffs = rand_felts(rng, (cols, n))
    got = np.zeros((cols, N), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_lde(ptr(coeffs), cols, log_n, rate, GEN, ptr(got)))
    rev = bitrev_indices(log_n + rate)
    exp = np.zeros(N, dtype=np.uint64)
    for c in (0, cols - 1):
        oracle().orc_coset_lde(ptr(coeffs[c]), n, rate, GEN, ptr(exp))
        assert np.array_equal(got[c], exp[rev])
