#!/usr/bin/env python3
"""Generates the committed golden fixtures from the independent pure-Python restatement
(oracle/bn254_pyref.py).  Run in the build container:  python tests/golden/gen_golden.py

The reference itself holds no golden vectors and cannot be run here (Rust), so these are
vectors of the RESTATEMENT, pinned by the reference's algebraic identities
(tests/test_oracle.py); they guard the C oracle and the HIP path against regressions and
make the GPU-box parity tests independent of any big-int recomputation.
All values are canonical integers (hex); tests convert to Montgomery limbs.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bn254_pyref as R  # noqa: E402

SEED = 0xB2540001


def hx(x):
    return hex(x)


def pt1(p):
    return [hx(p[0]), hx(p[1])]


def pt2(q):
    return [hx(q[0][0]), hx(q[0][1]), hx(q[1][0]), hx(q[1][1])]


def main():
    st = SEED
    n = 12
    s, t = [], []
    for _ in range(n):
        st, x = R.rand_scalar(st)
        s.append(x)
    for _ in range(n):
        st, x = R.rand_scalar(st)
        t.append(x)
    P = [R.G1_GEN] + [R.g1_mul(R.G1_GEN, x) for x in s]
    Q = [R.G2_GEN] + [R.g2_mul(R.G2_GEN, x) for x in t]
    vec = {"comment": "index 0 is e(G1gen, G2gen) (BASELINE.json configs[0]); others [s_i]G1, [t_i]G2, SplitMix64 seed 0xB2540001",
           "g1": [pt1(p) for p in P], "g2": [pt2(q) for q in Q], "miller": [], "pairing": []}
    for p, q in zip(P, Q):
        m = R.miller_loop_native(q, p)
        vec["miller"].append([hx(c) for c in m])
        vec["pairing"].append([hx(c) for c in R.final_exp_native(m)])
    # multi-Miller groups (T1 / T3 shapes and the Groth16 k = 4 shape)
    groups = []
    for k, idx in ((2, [1, 2]), (2, [3, 4]), (3, [5, 6, 7]), (4, [8, 9, 10, 11])):
        pairs = [(P[i], Q[i]) for i in idx]
        m = R.multi_miller_loop_native(pairs)
        groups.append({"k": k, "idx": idx, "miller": [hx(c) for c in m], "pairing": [hx(c) for c in R.final_exp_native(m)]})
    # T3 (final_exp_native.rs:240-264): P0 = 5 G1, Q0 = 6 G2, P1 = 30 G1, Q1 = -G2
    P0, Q0 = R.g1_mul(R.G1_GEN, 5), R.g2_mul(R.G2_GEN, 6)
    P1, Q1 = R.g1_mul(R.G1_GEN, 30), R.g2_neg(R.G2_GEN)
    m = R.multi_miller_loop_native([(P0, Q0), (P1, Q1)])
    vec["t3"] = {"g1": [pt1(P0), pt1(P1)], "g2": [pt2(Q0), pt2(Q1)], "miller": [hx(c) for c in m],
                 "pairing": [hx(c) for c in R.final_exp_native(m)]}
    vec["groups"] = groups
    # final exponentiation / pow / frobenius on arbitrary Fq12 (T4, T7 shapes)
    rng = random.Random(SEED)
    xs = [[rng.randrange(R.P) for _ in range(12)] for _ in range(4)]
    xs.append([1] + [0] * 11)
    vec["fq12_in"] = [[hx(c) for c in x] for x in xs]
    vec["final_exp"] = [[hx(c) for c in R.final_exp_native(x)] for x in xs]
    vec["pow_x"] = [[hx(c) for c in R.pow_native(x, [R.BN_X])] for x in xs]
    vec["frobenius"] = {str(k): [[hx(c) for c in R.frobenius_map_native(x, k)] for x in xs] for k in (0, 1, 2, 3, 6, 11, 13)}
    vec["fq12_mul"] = [[hx(c) for c in R.fq12_mul(xs[i], xs[(i + 1) % len(xs)])] for i in range(len(xs))]
    vec["naf"] = {"bn_x": R.get_naf([R.BN_X]), "six_u_plus_2_value": sum(d << i for i, d in enumerate(R.SIX_U_PLUS_2_NAF)),
                  "two_limbs": {"exp": [hx(0xFFFFFFFFFFFFFFFF), hx(0x1234)], "naf": R.get_naf([0xFFFFFFFFFFFFFFFF, 0x1234])}}
    vec["consts"] = {"c2": [hx(c) for c in R._end_constants()[0]], "c3": [hx(c) for c in R._end_constants()[1]],
                     "frob_coeffs": {str(k): [hx(c) for c in R.frob_coeffs(k)] for k in range(12)}}
    with open(os.path.join(HERE, "bn254_vectors.json"), "w") as f:
        json.dump(vec, f, indent=0)
    print("wrote bn254_vectors.json")


if __name__ == "__main__":
    main()
