"""The C-ABI library loads without a GPU and exports every symbol include/bn254_pairing.h declares;
host-only entry points (get_naf, index map, constants) behave like the reference; compute entry
points fail loudly (no CPU fallback) when no device is present."""
import os
import re

import numpy as np
import pytest

import helpers as H
from helpers import R

HEADER = os.path.join(H.ROOT, "include", "bn254_pairing.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bn254_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def pk():
    p = H.pkg()
    if not os.path.exists(p.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return p


def test_header_symbols_exported(pk):
    lib = pk.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bn254_pairing.h but not exported"
    assert set(pk.ABI_SYMBOLS) == set(declared)


def test_host_logic_get_naf_and_constants(pk):
    assert pk.get_naf([R.BN_X]) == R.get_naf([R.BN_X])
    assert pk.get_naf([0xFFFFFFFFFFFFFFFF, 0x1234]) == R.get_naf([0xFFFFFFFFFFFFFFFF, 0x1234])
    assert pk.get_naf([0]) == [0] * 64
    with pytest.raises(pk.Bn254Error) as ei:       # reference panics (final_exp_native.rs:123)
        pk.get_naf([0xFFFFFFFFFFFFFFFF])
    assert ei.value.status == pk.ERR_NAF_CARRY
    lib = pk.load_library()
    assert lib.bn254_bn_x() == R.BN_X == pk.BN_X
    naf = lib.bn254_six_u_plus_2_naf()
    assert [naf[i] for i in range(65)] == R.SIX_U_PLUS_2_NAF == pk.SIX_U_PLUS_2_NAF
    x = list(range(12))
    assert [x[lib.bn254_myfq12_to_ark_index(j)] for j in range(12)] == R.myfq12_to_ark(x)
    assert lib.bn254_myfq12_to_ark_index(12) == -1


def test_frob_coeffs_host_table(pk):
    """frob_coeffs(index) (final_exp_native.rs:183-192) is served from a host-side table: no device needed."""
    vec = H.load_golden("bn254_vectors.json")
    for k in range(12):
        got = pk.frob_coeffs(k)
        want = [int(x, 16) for x in vec["consts"]["frob_coeffs"][str(k)]]
        assert [R.from_mont(pk.layout.limbs_to_int(got[:4])), R.from_mont(pk.layout.limbs_to_int(got[4:]))] == want
        assert want == list(R.frob_coeffs(k))
    assert list(pk.frob_coeffs(13)) == list(pk.frob_coeffs(1))
    out = np.zeros(8, dtype=np.uint64)
    assert pk.load_library().bn254_frob_coeffs(12, out.ctypes.data_as(__import__("ctypes").c_void_p)) == pk.ERR_INVALID_ARG


def test_conjugates(pk):
    x = (123456789, R.P - 5)
    arr = np.array(H.fq_words(x[0]) + H.fq_words(x[1]), dtype=np.uint64)
    c = pk.conjugate_fp2(arr)
    want = R.conjugate_fp2(x)
    assert list(c) == H.fq_words(want[0]) + H.fq_words(want[1])
    c = pk.neg_conjugate_fp2(arr)
    want = R.neg_conjugate_fp2(x)
    assert list(c) == H.fq_words(want[0]) + H.fq_words(want[1])


def test_no_cpu_fallback_without_gpu(pk):
    if pk.device_count() > 0:
        pytest.skip("GPU present")
    g1 = np.zeros(8, dtype=np.uint64)
    g2 = np.zeros(16, dtype=np.uint64)
    with pytest.raises(pk.Bn254Error) as ei:
        pk.pairing_batch(g1, g2, 1)
    assert ei.value.status == pk.ERR_NO_DEVICE
    with pytest.raises(pk.Bn254Error):
        pk.final_exp_batch(np.zeros(48, dtype=np.uint64), 1)


def test_missing_library_fails_loudly(pk, tmp_path):
    with pytest.raises(pk.Bn254Error):
        pk.load_library(str(tmp_path / "nope.so"))


def test_bad_arguments(pk):
    with pytest.raises(pk.Bn254Error):
        pk.pairing_batch(np.zeros(7, dtype=np.uint64), np.zeros(16, dtype=np.uint64), 1)
    with pytest.raises(pk.Bn254Error):
        pk.multi_miller_loop_native([])


def test_cpp_host_header_compiles():
    """include/bn254_pairing.hpp (C++ mirror of the reference's function names) is valid C++17."""
    import subprocess
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(H.ROOT, "include"), "-x", "c++",
                           os.path.join(H.ROOT, "include", "bn254_pairing.hpp")])
