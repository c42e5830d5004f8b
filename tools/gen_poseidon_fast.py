#!/usr/bin/env python3
"""Derives the "fast partial round" form of Poseidon-Goldilocks' 22 partial rounds and writes
zk-circuits_amd/csrc/poseidon_fast_constants.h (used by the HOST transcript permutation only).

Derivation (own; the tables are functions of the MDS matrix and the round constants alone, SURVEY.md App. A.2 notes they are
"derivable, not needed for correctness"). With M the MDS matrix, S0 the S-box on coordinate 0 and c_t the round constants,
the partial rounds are s <- M*S0(s + c_t), t = 0..21. Going backwards from D_21 = M, split D_t = [[a, w^T], [v, B]] as
D_t = S_t * P_t with P_t = diag(1, B) (commutes with S0) and the sparse S_t = [[a, w^T B^-1], [v, I]], and let D_{t-1} = P_t*M.
The constants P_t*c_t are pushed backwards through S_{t-1}^-1: the part on coordinates 1..11 commutes with S0 and joins the
previous round's constants, the part on coordinate 0 becomes a scalar added after the previous S-box. What is left:
    s += FIRST;  s[1:] = INIT * s[1:];  for t: s0 = sbox(s0) + Q[t];  (s0, rest) = (a*s0 + WHAT[t].rest, rest + s0*V[t])
The script checks the result against the plain round-by-round permutation on random states before writing anything.
"""
import os, random, sys
sys.path.insert(0, os.path.dirname(__file__))
from gen_poseidon_constants import constants, P

CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11
W = 12

def mat_mds():
    return [[(CIRC[(c - r) % W] + (DIAG[r] if r == c else 0)) % P for c in range(W)] for r in range(W)]

def matmul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][t] * B[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]

def matvec(A, x):
    return [sum(a * b for a, b in zip(row, x)) % P for row in A]

def inv(A):
    n = len(A); M = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if M[r][c] % P)
        M[c], M[piv] = M[piv], M[c]
        iv = pow(M[c][c], P - 2, P)
        M[c] = [x * iv % P for x in M[c]]
        for r in range(n):
            if r != c and M[r][c]:
                f = M[r][c]
                M[r] = [(x - f * y) % P for x, y in zip(M[r], M[c])]
    return [row[n:] for row in M]

def sbox(x): return pow(x, 7, P)

def plain_permute(st, rc):
    M = mat_mds(); st = st[:]
    for r in range(30):
        st = [(x + rc[12 * r + i]) % P for i, x in enumerate(st)]
        if r < 4 or r >= 26: st = [sbox(x) for x in st]
        else: st[0] = sbox(st[0])
        st = matvec(M, st)
    return st

def derive(rc):
    M = mat_mds()
    D = M; S = [None] * 22; Pm = [None] * 22
    for t in range(21, -1, -1):
        a = D[0][0]; w = D[0][1:]; v = [D[i][0] for i in range(1, W)]; B = [row[1:] for row in D[1:]]
        Binv = inv(B)
        what = [sum(w[k] * Binv[k][j] for k in range(W - 1)) % P for j in range(W - 1)]
        S[t] = (a, what, v)
        Pt = [[1] + [0] * (W - 1)] + [[0] + B[i] for i in range(W - 1)]
        Pm[t] = Pt
        D = matmul(Pt, M)
    def S_full(t):
        a, what, v = S[t]
        return [[a] + what] + [[v[i]] + [1 if j == i else 0 for j in range(W - 1)] for i in range(W - 1)]
    k = [matvec(Pm[t], [rc[12 * (4 + t) + i] for i in range(W)]) for t in range(22)]
    Q = [0] * 22
    K = k[21]
    for t in range(21, 0, -1):
        z = matvec(inv(S_full(t - 1)), K)
        Q[t - 1] = z[0]
        K = [k[t - 1][0]] + [(k[t - 1][i] + z[i]) % P for i in range(1, W)]
    first = matvec(inv(Pm[0]), K)
    init = [row[1:] for row in Pm[0][1:]]
    return first, init, Q, S

def fast_permute(st, rc, first, init, Q, S):
    M = mat_mds(); st = st[:]
    for r in range(4):
        st = [(x + rc[12 * r + i]) % P for i, x in enumerate(st)]
        st = matvec(M, [sbox(x) for x in st])
    st = [(x + f) % P for x, f in zip(st, first)]
    st = [st[0]] + matvec(init, st[1:])
    for t in range(22):
        s0 = (sbox(st[0]) + Q[t]) % P
        a, what, v = S[t]
        d = (a * s0 + sum(x * y for x, y in zip(what, st[1:]))) % P
        st = [d] + [(st[1 + i] + s0 * v[i]) % P for i in range(W - 1)]
    for r in range(26, 30):
        st = [(x + rc[12 * r + i]) % P for i, x in enumerate(st)]
        st = matvec(M, [sbox(x) for x in st])
    return st

if __name__ == "__main__":
    rc = constants()
    first, init, Q, S = derive(rc)
    rnd = random.Random(7)
    for _ in range(20):
        st = [rnd.randrange(P) for _ in range(W)]
        assert plain_permute(st, rc) == fast_permute(st, rc, first, init, Q, S)
    assert plain_permute([0] * 12, rc)[:2] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3]
    assert all(S[t][0] == 25 for t in range(22)) and Q[21] == 0
    def arr(xs): return ", ".join("0x%016xULL" % x for x in xs)
    # the product's host transcript and the oracle (CPU baseline) each get their own copy: neither includes the other's tree
    for path in (os.path.join(os.path.dirname(__file__), "..", "zk-circuits_amd", "csrc", "poseidon_fast_constants.h"),
                 os.path.join(os.path.dirname(__file__), "..", "oracle", "poseidon_fast_constants.h")):
      with open(path, "w") as f:
          f.write("// GENERATED by tools/gen_poseidon_fast.py (derivation and self-check there); do not edit.\n")
          f.write("// Fast form of the 22 partial rounds: s += FIRST; s[1:] = INIT*s[1:]; per round s0 = sbox(s0) + Q[t];\n")
          f.write("// (s0, rest) = (25*s0 + WHAT[t].rest, rest + s0*V[t]).\n#pragma once\n")
          f.write("#define QP_POSEIDON_FAST_FIRST " + arr(first) + "\n")
          f.write("#define QP_POSEIDON_FAST_Q " + arr(Q) + "\n")
          f.write("#define QP_POSEIDON_FAST_INIT \\\n" + ", \\\n".join("  {" + arr(row) + "}" for row in init) + "\n")
          f.write("#define QP_POSEIDON_FAST_WHAT \\\n" + ", \\\n".join("  {" + arr(S[t][1]) + "}" for t in range(22)) + "\n")
          f.write("#define QP_POSEIDON_FAST_V \\\n" + ", \\\n".join("  {" + arr(S[t][2]) + "}" for t in range(22)) + "\n")
      print("wrote", path)
    # Device tables (csrc only): the dot products of the fast form take each constant a as 27/27/10-bit limbs of a and of a' = a*2^32 mod p,
    # so that a*x = a*x_lo + a'*x_hi is six 32x32 multiply-adds whose column sums over 22 terms stay below 2^64 (poseidon_dev.h, dot11_dev).
    def limbs(a):
        out = []
        for v in (a, a * (1 << 32) % P):
            l = [v & 0x7FFFFFF, (v >> 27) & 0x7FFFFFF, v >> 54]
            assert l[0] >= 1 and l[2] < 1024                      # dot11_dev's X0 - X2 >= 0 argument needs a nonzero low limb
            out += l
        return out
    def arr32(xs): return ", ".join("0x%08xu" % x for x in xs)
    path = os.path.join(os.path.dirname(__file__), "..", "zk-circuits_amd", "csrc", "poseidon_fast_dev_constants.h")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_poseidon_fast.py; do not edit. Per constant a: limbs (27, 27, 10 bits) of a, then of a*2^32 mod p.\n#pragma once\n")
        f.write("// all 360 round constants followed by a row of zeros (what the last round's MDS product has to add), so that 'the constants after round r' is a row for every r\n")
        f.write("#define QP_POSEIDON_FASTDEV_RC_PAD " + ", ".join("0x%016xULL" % x for x in list(rc) + [0] * 12) + "\n")
        f.write("// round constants of rounds 0-3 followed by FIRST (what the fourth full round's MDS folds in instead of round 4's constants)\n")
        f.write("#define QP_POSEIDON_FASTDEV_RC_HEAD " + ", ".join("0x%016xULL" % x for x in list(rc[:48]) + list(first)) + "\n")
        f.write("#define QP_POSEIDON_FASTDEV_INIT \\\n" + ", \\\n".join("  " + arr32(limbs(x)) for row in init for x in row) + "\n")
        f.write("// per round t: V[t][0..10] followed by Q[t+1] (0 after the last round): what one round's multiply-adds and the next S-box need\n")
        f.write("#define QP_POSEIDON_FASTDEV_VQ \\\n" + ", \\\n".join("  " + ", ".join("0x%016xULL" % x for x in list(S[t][2]) + [Q[t + 1] if t < 21 else 0]) for t in range(22)) + "\n")
        f.write("#define QP_POSEIDON_FASTDEV_WHAT \\\n" + ", \\\n".join("  " + arr32(limbs(x)) for t in range(22) for x in S[t][1]) + "\n")
    print("wrote", path)
