"""Shared test plumbing: oracle loader (C restatement via ctypes), the pure-Python restatement,
point/field conversions, deterministic inputs.  Test infrastructure only."""
import ctypes
import importlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254_pyref as R  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
_U64P = ctypes.POINTER(ctypes.c_uint64)


def pkg():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("plonky2-bn254-pairing_amd")


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_oracle = None


def oracle():
    """ctypes handle on oracle/libbn254_oracle.so (built on demand with gcc)."""
    global _oracle
    if _oracle is None:
        so = os.environ.get("BN254_ORACLE_SO", os.path.join(ROOT, "oracle", "libbn254_oracle.so"))
        src = os.path.join(ROOT, "oracle", "bn254_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.oracle_get_naf.restype = ctypes.c_long
        _oracle = lib
    return _oracle


# ---------------------------------------------------------------- conversions (canonical ints <-> Montgomery u64 limbs)
def fq_words(x):
    return R.limbs4(R.to_mont(x))


def g1_aos(points):
    return np.array([w for p in points for c in p for w in fq_words(c)], dtype=np.uint64)


def g2_aos(points):
    return np.array([w for q in points for c2 in q for c in c2 for w in fq_words(c)], dtype=np.uint64)


def fq12_aos(elems):
    return np.array([w for e in elems for c in e for w in fq_words(c)], dtype=np.uint64)


def fq12_from_aos(a, n):
    a = np.asarray(a, dtype=np.uint64).reshape(n, 12, 4)
    return [[R.from_mont(sum(int(a[i, c, l]) << (64 * l) for l in range(4))) for c in range(12)] for i in range(n)]


def to_soa(aos, words):
    return pkg().layout.to_soa(aos, words)


def to_aos(soa, words):
    return pkg().layout.to_aos(soa, words)


# ---------------------------------------------------------------- deterministic inputs (seed 0xB2540001)
SEED = 0xB2540001


def scalars(n, seed=SEED):
    st = seed
    out = []
    for _ in range(2 * n):
        st, s = R.rand_scalar(st)
        out.append(s)
    return out[:n], out[n:]


def subgroup_points(n, seed=SEED):
    """P_i = [s_i]G1, Q_i = [t_i]G2 (canonical affine ints)."""
    s, t = scalars(n, seed)
    return [R.g1_mul(R.G1_GEN, x) for x in s], [R.g2_mul(R.G2_GEN, x) for x in t]


def rand_fq12(n, seed=1):
    import random
    rng = random.Random(seed)
    return [[rng.randrange(R.P) for _ in range(12)] for _ in range(n)]


# ---------------------------------------------------------------- oracle wrappers (AoS numpy in/out)
def oracle_pairing(g1, g2, n, threads=1):
    out = np.zeros(48 * n, dtype=np.uint64)
    rc = oracle().oracle_pairing_mt(ptr(g1), ptr(g2), ptr(out), ctypes.c_size_t(n), ctypes.c_int(threads))
    assert rc == 0
    return out


def oracle_miller(g1, g2, n):
    out = np.zeros(48 * n, dtype=np.uint64)
    assert oracle().oracle_miller_loop(ptr(g1), ptr(g2), ptr(out), ctypes.c_size_t(n)) == 0
    return out


def oracle_multi_miller(g1, g2, n_groups, k):
    out = np.zeros(48 * n_groups, dtype=np.uint64)
    assert oracle().oracle_multi_miller_loop(ptr(g1), ptr(g2), ptr(out), ctypes.c_size_t(n_groups), ctypes.c_size_t(k)) == 0
    return out


def oracle_multi_pairing(g1, g2, n_groups, k):
    out = np.zeros(48 * n_groups, dtype=np.uint64)
    assert oracle().oracle_multi_pairing(ptr(g1), ptr(g2), ptr(out), ctypes.c_size_t(n_groups), ctypes.c_size_t(k)) == 0
    return out


def oracle_final_exp(f, n):
    out = np.zeros(48 * n, dtype=np.uint64)
    rc = oracle().oracle_final_exp(ptr(f), ptr(out), ctypes.c_size_t(n))
    return rc, out


def oracle_fq12_mul(a, b, n):
    out = np.zeros(48 * n, dtype=np.uint64)
    assert oracle().oracle_fq12_mul(ptr(a), ptr(b), ptr(out), ctypes.c_size_t(n)) == 0
    return out


def oracle_fq12_pow(a, exp_limbs, n):
    e = np.array(exp_limbs, dtype=np.uint64)
    out = np.zeros(48 * n, dtype=np.uint64)
    assert oracle().oracle_fq12_pow(ptr(a), ptr(e), ctypes.c_size_t(e.size), ptr(out), ctypes.c_size_t(n)) == 0
    return out


def oracle_pow_native(a, exp_limbs, n):
    e = np.array(exp_limbs, dtype=np.uint64)
    out = np.zeros(48 * n, dtype=np.uint64)
    rc = oracle().oracle_pow_native(ptr(a), ptr(e), ctypes.c_size_t(e.size), ptr(out), ctypes.c_size_t(n))
    return rc, out


def oracle_frobenius(a, power, n):
    out = np.zeros(48 * n, dtype=np.uint64)
    assert oracle().oracle_frobenius_map(ptr(a), ctypes.c_size_t(power), ptr(out), ctypes.c_size_t(n)) == 0
    return out


def oracle_get_naf(exp_limbs):
    e = np.array(exp_limbs, dtype=np.uint64)
    naf = np.zeros(64 * e.size + 1, dtype=np.int8)
    n = oracle().oracle_get_naf(ptr(e), ctypes.c_size_t(e.size), ptr(naf))
    return n, naf[:max(n, 0)].tolist()


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
