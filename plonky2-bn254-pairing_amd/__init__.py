"""plonky2-bn254-pairing_amd -- MI355X-native batched BN254 pairing engine (host-side plumbing).

This package is a thin ctypes binding over the C ABI in include/bn254_pairing.h
(libbn254_pairing_hip.so, built in-tree by __graft_entry__.build()).  It mirrors the
names of the reference's public native functions
(/root/reference/src/{pairing,miller_loop_native,final_exp_native}.rs):

    pairing, miller_loop_native, multi_miller_loop_native, final_exp_native,
    frobenius_map_native, pow_native, get_naf, frob_coeffs,
    conjugate_fp2, neg_conjugate_fp2, SIX_U_PLUS_2_NAF, BN_X

plus `*_batch` forms over struct-of-arrays numpy / torch buffers.  All field data is
u64 little-endian limbs in Montgomery form (R = 2^256), i.e. ark-ff's in-memory
representation; see `layout` below.  There is NO CPU fallback: every compute entry point
raises if the HIP library is missing or no GPU is visible.

The directory name contains hyphens, so import it with
    importlib.import_module("plonky2-bn254-pairing_amd")
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbn254_pairing_hip.so")

BN254_OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_ZERO_DIVISOR = -4
ERR_NAF_CARRY = -5
ERR_ALLOC = -6
ERR_INFINITY = -7
ERR_NOT_ON_CURVE = -8
ERR_NOT_IN_SUBGROUP = -9
CHECK_INFINITY, CHECK_ON_CURVE, CHECK_SUBGROUP = 1, 2, 4       # bn254_check_points_ex flags
CHECK_SUBGROUP_PORTABLE = 8                                    # ... the subgroup check on the portable HIP C++ kernel (cross-check of the generated one)
PT_INFINITY, PT_NOT_ON_CURVE, PT_NOT_IN_SUBGROUP = 2, 4, 8     # bits of its per-point verdict byte
LATENCY_INHERIT = (1 << (8 * ctypes.sizeof(ctypes.c_size_t))) - 1

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN_X = 4965661367192848881  # src/final_exp_native.rs:15
# src/miller_loop_native.rs:314-318
SIX_U_PLUS_2_NAF = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
    1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
    0, 1, 0, 1, 1,
]

G1_WORDS, G2_WORDS, FQ12_WORDS = 8, 16, 48
FQ12_MYFQ12, FQ12_ARK = 0, 1          # coefficient order of element-major Fq12 data (include/bn254_pairing.h)


class Bn254Error(RuntimeError):
    """Raised where the reference panics (or the device/library is unusable)."""

    def __init__(self, status, what=""):
        self.status = status
        msg = _strerror(status)
        super().__init__(f"{what}: {msg} (status {status})" if what else f"{msg} (status {status})")


_lib = None

_PROTOS = {
    "bn254_device_count": (ctypes.c_int, []),
    "bn254_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "bn254_last_status": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "bn254_scratch_bytes": (ctypes.c_size_t, [ctypes.c_size_t, ctypes.c_size_t]),
    "bn254_pairing_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_miller_loop_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_miller_loop_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_final_exp_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_final_exp_batch": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_check_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_check_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_sharded": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int]),
    "bn254_multi_pairing_sharded": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "bn254_pairing_sharded_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_sharded_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                       ctypes.c_void_p]),
    "bn254_release_stream": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "bn254_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "bn254_host_unregister": (ctypes.c_int, [ctypes.c_void_p]),
    "bn254_alloc_pinned": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "bn254_free_pinned": (ctypes.c_int, [ctypes.c_void_p]),
    "bn254_host_is_pinned": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "bn254_check_points_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_check_points": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_check_points_ex_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "bn254_check_points_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "bn254_set_stream_latency": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]),
    "bn254_last_kernel": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "bn254_reserve": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]),
    "bn254_set_latency_threshold": (None, [ctypes.c_size_t]),
    "bn254_get_latency_threshold": (ctypes.c_size_t, []),
    "bn254_set_latency_lanes": (None, [ctypes.c_int]),
    "bn254_get_latency_lanes": (ctypes.c_int, []),
    "bn254_fq12_mul_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_fq12_mul_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_frobenius_map_batch_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_frobenius_map_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pow_batch_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pow_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_get_naf": (ctypes.c_long, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "bn254_frob_coeffs": (ctypes.c_int, [ctypes.c_size_t, ctypes.c_void_p]),
    "bn254_six_u_plus_2_naf": (ctypes.POINTER(ctypes.c_int8), []),
    "bn254_bn_x": (ctypes.c_uint64, []),
    "bn254_myfq12_to_ark_index": (ctypes.c_int, [ctypes.c_int]),
    "bn254_generate_pairs_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_soa_from_elems_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_soa_to_elems_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_batch_elems_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_batch_elems_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_g2_lines_bytes": (ctypes.c_size_t, [ctypes.c_size_t]),
    "bn254_set_wide_groups": (None, [ctypes.c_size_t]),
    "bn254_get_wide_groups": (ctypes.c_size_t, []),
    "bn254_g2_lines_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_batch_elems_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_check_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_check_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_check_sharded_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]),
    "bn254_pairing_fixed_g2_check_target_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_check_target_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_fixed_g2_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_miller_loop_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_multi_pairing_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "bn254_pairing_sharded_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "bn254_multi_pairing_sharded_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "bn254_multi_pairing_check_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "bn254_final_exp_batch_elems": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
}
# symbols include/bn254_pairing.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = sorted(_PROTOS)


def load_library(path=None):
    """Loads the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    try:  # PyTorch (device memory / streams plumbing) bundles its own HIP runtime: let it load first so
        import torch  # noqa: F401  -- the process ends up with exactly one libamdhip64
    except Exception:
        pass
    if not os.path.exists(p):
        raise Bn254Error(ERR_NO_DEVICE, f"HIP extension {p} not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(p)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _strerror(status):
    try:
        return load_library().bn254_strerror(status).decode()
    except Exception:  # library unavailable: still produce a message
        return {ERR_NO_DEVICE: "no HIP device / library"}.get(status, "error")


def device_count():
    return load_library().bn254_device_count()


def _check(rc, what):
    if rc != BN254_OK:
        raise Bn254Error(rc, what)


# ----------------------------------------------------------------------------- layout helpers
class layout:
    """SoA (limb-major) <-> AoS conversions.  SoA: elem(c, l, i) = buf[(c*4 + l)*n + i]."""

    @staticmethod
    def to_soa(aos, words):
        a = np.ascontiguousarray(aos, dtype=np.uint64).reshape(-1, words)
        return np.ascontiguousarray(a.T).reshape(-1)

    @staticmethod
    def to_aos(soa, words):
        s = np.ascontiguousarray(soa, dtype=np.uint64).reshape(words, -1)
        return np.ascontiguousarray(s.T).reshape(-1)

    @staticmethod
    def fq_to_limbs(mont_int):
        return [(mont_int >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]

    @staticmethod
    def limbs_to_int(limbs):
        return sum(int(w) << (64 * i) for i, w in enumerate(limbs))

    @staticmethod
    def to_mont(x):
        return (x << 256) % P

    @staticmethod
    def from_mont(x):
        return (x * pow(1 << 256, -1, P)) % P


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _np_in(a, words, n):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    if a.size != words * n:
        raise Bn254Error(ERR_INVALID_ARG, f"expected {words * n} u64 words, got {a.size}")
    return a


def _np_out(out, size, dtype=np.uint64):
    """the result array of a host-pointer call: a fresh one, or the caller's (e.g. page-locked: `alloc_pinned`)"""
    if out is None:
        return np.empty(size, dtype=dtype)
    if not isinstance(out, np.ndarray) or out.dtype != dtype or out.size != size or not out.flags.c_contiguous or not out.flags.writeable:
        raise Bn254Error(ERR_INVALID_ARG, f"out: expected a writable contiguous {np.dtype(dtype).name} array of {size} entries")
    return out


# ----------------------------------------------------------------------------- page-locked host memory
def alloc_pinned(words):
    """A page-locked uint64 array of `words` entries (bn254_alloc_pinned = hipHostMalloc): host-pointer calls on such arrays (inputs AND
    `out=`) copy by DMA at the link rate instead of through the runtime's staging buffers.  Free it with `free_pinned`."""
    ptr = ctypes.c_void_p()
    _check(load_library().bn254_alloc_pinned(8 * max(int(words), 1), ctypes.byref(ptr)), "alloc_pinned")
    buf = (ctypes.c_uint64 * int(words)).from_address(ptr.value)
    a = np.frombuffer(buf, dtype=np.uint64)
    _PINNED[a.ctypes.data] = ptr.value
    return a


def free_pinned(a):
    ptr = _PINNED.pop(a.ctypes.data, None)
    if ptr is None:
        raise Bn254Error(ERR_INVALID_ARG, "free_pinned: not an array from alloc_pinned")
    _check(load_library().bn254_free_pinned(ptr), "free_pinned")


_PINNED = {}


def host_register(a):
    """Page-locks the memory of an existing contiguous numpy array in place (bn254_host_register = hipHostRegister)."""
    _check(load_library().bn254_host_register(_ptr(a), a.nbytes), "host_register")


def host_unregister(a):
    _check(load_library().bn254_host_unregister(_ptr(a)), "host_unregister")


def host_is_pinned(a):
    return bool(load_library().bn254_host_is_pinned(_ptr(a), a.nbytes))


# ----------------------------------------------------------------------------- batch API (host numpy buffers, SoA)
def pairing_batch(g1, g2, n, device=0, out=None):
    """n x pairing(p, q)  (src/pairing.rs:20-22), MyFq12 coefficient order, SoA."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = _np_out(out, FQ12_WORDS * n)
    _check(lib.bn254_pairing_batch(_ptr(g1), _ptr(g2), _ptr(out), n, device, None), "pairing")
    return out


def miller_loop_batch(g1, g2, n, device=0):
    """n x miller_loop_native(Q, P)  (src/miller_loop_native.rs:320-322)."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_miller_loop_batch(_ptr(g1), _ptr(g2), _ptr(out), n, device, None), "miller_loop_native")
    return out


def final_exp_batch(f, n, device=0):
    """n x final_exp_native(a)  (src/final_exp_native.rs:209-213)."""
    lib = load_library()
    f = _np_in(f, FQ12_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_final_exp_batch(_ptr(f), _ptr(out), n, device, None), "final_exp_native")
    return out


def multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=True, device=0, out=None):
    """n_groups x multi_miller_loop_native(k pairs) [+ final_exp_native]  (:324-326)."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = _np_out(out, FQ12_WORDS * n_groups)
    _check(lib.bn254_multi_pairing_batch(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, 1 if do_final_exp else 0, device, None),
           "multi_miller_loop_native")
    return out


def multi_pairing_check_batch(g1, g2, n_groups, k, device=0):
    """verdict[g] = (final_exp_native(multi_miller_loop_native(group g)) == MyFq12::one), the product check of
    final_exp_native.rs:245-263 / a Groth16 verifier: uint8 array of n_groups entries."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = np.empty(n_groups, dtype=np.uint8)
    _check(lib.bn254_multi_pairing_check_batch(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, device, None), "multi-pairing check")
    return out


def pairing_sharded(g1, g2, n, n_devices):
    """pairing_batch over devices 0..n_devices-1 of this process (contiguous slices, no exchange step)."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_pairing_sharded(_ptr(g1), _ptr(g2), _ptr(out), n, n_devices), "pairing (sharded)")
    return out


def multi_pairing_sharded(g1, g2, n_groups, k, n_devices, do_final_exp=True):
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = np.empty(FQ12_WORDS * n_groups, dtype=np.uint64)
    _check(lib.bn254_multi_pairing_sharded(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, 1 if do_final_exp else 0, n_devices),
           "multi_miller_loop_native (sharded)")
    return out


def fq12_mul_batch(a, b, n, device=0):
    lib = load_library()
    a, b = _np_in(a, FQ12_WORDS, n), _np_in(b, FQ12_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_fq12_mul_batch(_ptr(a), _ptr(b), _ptr(out), n, device, None), "MyFq12 mul")
    return out


def frobenius_map_batch(a, power, n, device=0):
    lib = load_library()
    a = _np_in(a, FQ12_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_frobenius_map_batch(_ptr(a), power, _ptr(out), n, device, None), "frobenius_map_native")
    return out


def pow_batch(a, exp, n, device=0):
    lib = load_library()
    a = _np_in(a, FQ12_WORDS, n)
    e = np.ascontiguousarray(exp, dtype=np.uint64)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_pow_batch(_ptr(a), _ptr(e), e.size, _ptr(out), n, device, None), "pow_native")
    return out


# ----------------------------------------------------------------------------- element-major API (host numpy buffers)
# elems[i*W + w]: the order the reference's callers hold &[G1Affine] / Vec<MyFq12> / Vec<Fq12> in; the planes are made on the device.
def pairing_batch_elems(g1, g2, n, out_order=FQ12_MYFQ12, device=0, out=None):
    """n x pairing(p, q); out_order = FQ12_ARK gives ark `Fq12` words, the value src/pairing.rs:20-22 returns."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = _np_out(out, FQ12_WORDS * n)
    _check(lib.bn254_pairing_batch_elems(_ptr(g1), _ptr(g2), _ptr(out), n, out_order, device, None), "pairing")
    return out


def miller_loop_batch_elems(g1, g2, n, device=0):
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_miller_loop_batch_elems(_ptr(g1), _ptr(g2), _ptr(out), n, device, None), "miller_loop_native")
    return out


def multi_pairing_batch_elems(g1, g2, n_groups, k, do_final_exp=True, out_order=FQ12_MYFQ12, device=0, out=None):
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = _np_out(out, FQ12_WORDS * n_groups)
    _check(lib.bn254_multi_pairing_batch_elems(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, 1 if do_final_exp else 0, out_order, device, None),
           "multi_miller_loop_native")
    return out


def pairing_sharded_elems(g1, g2, n, n_devices, out_order=FQ12_MYFQ12):
    """pairing_batch_elems over devices 0..n_devices-1 of this process (contiguous slices, no exchange step)."""
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_pairing_sharded_elems(_ptr(g1), _ptr(g2), _ptr(out), n, out_order, n_devices), "pairing (sharded)")
    return out


def multi_pairing_sharded_elems(g1, g2, n_groups, k, n_devices, do_final_exp=True, out_order=FQ12_MYFQ12):
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = np.empty(FQ12_WORDS * n_groups, dtype=np.uint64)
    _check(lib.bn254_multi_pairing_sharded_elems(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, 1 if do_final_exp else 0, out_order, n_devices),
           "multi_miller_loop_native (sharded)")
    return out


def multi_pairing_check_batch_elems(g1, g2, n_groups, k, device=0):
    lib = load_library()
    g1, g2 = _np_in(g1, G1_WORDS, n_groups * k), _np_in(g2, G2_WORDS, n_groups * k)
    out = np.empty(n_groups, dtype=np.uint8)
    _check(lib.bn254_multi_pairing_check_batch_elems(_ptr(g1), _ptr(g2), _ptr(out), n_groups, k, device, None), "multi-pairing check")
    return out


def final_exp_batch_elems(f, n, in_order=FQ12_MYFQ12, out_order=FQ12_MYFQ12, device=0):
    lib = load_library()
    f = _np_in(f, FQ12_WORDS, n)
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    _check(lib.bn254_final_exp_batch_elems(_ptr(f), _ptr(out), n, in_order, out_order, device, None), "final_exp_native")
    return out


# ----------------------------------------------------------------------------- device-pointer API (torch tensors / raw pointers)
def soa_from_elems_dev(elems, soa, words, n, fq12_order=FQ12_MYFQ12, device=0, stream=None):
    _check(load_library().bn254_soa_from_elems_dev(_dev(elems), _dev(soa), words, n, fq12_order, device, _stream(stream)), "soa_from_elems")


def soa_to_elems_dev(soa, elems, words, n, fq12_order=FQ12_MYFQ12, device=0, stream=None):
    _check(load_library().bn254_soa_to_elems_dev(_dev(soa), _dev(elems), words, n, fq12_order, device, _stream(stream)), "soa_to_elems")



def _dev(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr()) if hasattr(t, "data_ptr") else ctypes.c_void_p(int(t))


def _stream(stream):
    if stream is None:
        return None
    return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))


def pairing_batch_dev(g1, g2, out, n, device=0, stream=None):
    _check(load_library().bn254_pairing_batch_dev(_dev(g1), _dev(g2), _dev(out), n, device, _stream(stream)), "pairing")


def miller_loop_batch_dev(g1, g2, out, n, device=0, stream=None):
    _check(load_library().bn254_miller_loop_batch_dev(_dev(g1), _dev(g2), _dev(out), n, device, _stream(stream)), "miller_loop_native")


def final_exp_batch_dev(f, out, n, device=0, stream=None):
    _check(load_library().bn254_final_exp_batch_dev(_dev(f), _dev(out), n, device, _stream(stream)), "final_exp_native")


def multi_pairing_batch_dev(g1, g2, out, n_groups, k, do_final_exp=True, device=0, stream=None):
    _check(load_library().bn254_multi_pairing_batch_dev(_dev(g1), _dev(g2), _dev(out), n_groups, k, 1 if do_final_exp else 0, device,
                                                        _stream(stream)), "multi_miller_loop_native")


def multi_pairing_batch_elems_dev(g1, g2, out, n_groups, k, do_final_exp=True, out_order=FQ12_MYFQ12, device=0, stream=None):
    """device-resident ELEMENT-major arrays in and out (one G1Affine / G2Affine / Fq12 after the other): a plain launch for the throughput kernels"""
    _check(load_library().bn254_multi_pairing_batch_elems_dev(_dev(g1), _dev(g2), _dev(out), n_groups, k, 1 if do_final_exp else 0, out_order, device,
                                                              _stream(stream)), "multi_miller_loop_native")


def pairing_batch_elems_dev(g1, g2, out, n, out_order=FQ12_MYFQ12, device=0, stream=None):
    _check(load_library().bn254_pairing_batch_elems_dev(_dev(g1), _dev(g2), _dev(out), n, out_order, device, _stream(stream)), "pairing")


def pairing_fixed_g2_batch(g1, g2_var, g2_fixed, k_fixed, n, device=0, elems=False, out_order=FQ12_MYFQ12):
    """host arrays: n groups of (own pair + k_fixed pairs whose G2 points, g2_fixed, are the same for every group); limb-major, or everything element-major.
    g2_var=None: groups WITHOUT a pair of their own (k_fixed G1 points each: every G2 point is one of the fixed ones)"""
    lib = load_library()
    own = 0 if g2_var is None else 1
    g1, g2_fixed = _np_in(g1, G1_WORDS, n * (own + k_fixed)), _np_in(g2_fixed, G2_WORDS, k_fixed)
    g2_var = _np_in(g2_var, G2_WORDS, n) if own else None
    out = np.empty(FQ12_WORDS * n, dtype=np.uint64)
    pv = _ptr(g2_var) if own else None
    if elems:
        _check(lib.bn254_pairing_fixed_g2_batch_elems(_ptr(g1), pv, _ptr(g2_fixed), k_fixed, _ptr(out), n, out_order, device, None), "pairing (fixed G2)")
    else:
        _check(lib.bn254_pairing_fixed_g2_batch(_ptr(g1), pv, _ptr(g2_fixed), k_fixed, _ptr(out), n, device, None), "pairing (fixed G2)")
    return out


def pairing_fixed_g2_check_batch_elems(g1, g2_var, g2_fixed, k_fixed, n, target=None, device=0):
    """HOST element-major arrays (n x (1 + k_fixed) G1 structs, n G2, k_fixed fixed G2) -> n verdict bytes: product == target (48 words; None = one)"""
    own = 0 if g2_var is None else 1                         # (g2_var=None: groups without a pair of their own)
    g1, g2_fixed = _np_in(g1, 8, n * (own + k_fixed)), _np_in(g2_fixed, 16, k_fixed)
    g2_var = _np_in(g2_var, 16, n) if own else None
    verdict = np.zeros(n, dtype=np.uint8)
    keep, tp = _target_words(target)
    _check(load_library().bn254_pairing_fixed_g2_check_batch_elems(_ptr(g1), _ptr(g2_var) if own else None, _ptr(g2_fixed), k_fixed, tp, _ptr(verdict), n, device, None),
           "pairing check (fixed G2)")
    return verdict


def pairing_fixed_g2_check_sharded_elems(g1, g2_var, g2_fixed, k_fixed, n, n_devices, target=None):
    """pairing_fixed_g2_check_batch_elems over the first n_devices GPUs of this process (contiguous slices of the groups per device)"""
    own = 0 if g2_var is None else 1
    g1, g2_fixed = _np_in(g1, 8, n * (own + k_fixed)), _np_in(g2_fixed, 16, k_fixed)
    g2_var = _np_in(g2_var, 16, n) if own else None
    verdict = np.zeros(n, dtype=np.uint8)
    keep, tp = _target_words(target)
    _check(load_library().bn254_pairing_fixed_g2_check_sharded_elems(_ptr(g1), _ptr(g2_var) if own else None, _ptr(g2_fixed), k_fixed, tp, _ptr(verdict), n, n_devices),
           "pairing check (fixed G2, sharded)")
    return verdict


def g2_lines_bytes(k_fixed):
    return load_library().bn254_g2_lines_bytes(k_fixed)


def g2_lines_dev(g2_fixed, k_fixed, table, device=0, stream=None):
    """table (device tensor of g2_lines_bytes(k_fixed) bytes) <- every step's line coefficients of the k_fixed G2 points (limb-major device tensor)"""
    _check(load_library().bn254_g2_lines_dev(_dev(g2_fixed), k_fixed, _dev(table), device, _stream(stream)), "g2_lines")


def pairing_fixed_g2_batch_dev(g1, g2_var, table, k_fixed, out, n, device=0, stream=None):
    """n groups: final_exp_native(multi_miller_loop_native([(P0, Q0)] + [(P_j, Qfix_j)])) with the Qfix_j of `table`; g1: n x (1 + k_fixed) points"""
    _check(load_library().bn254_pairing_fixed_g2_batch_dev(_dev(g1), _dev(g2_var), _dev(table), k_fixed, _dev(out), n, device, _stream(stream)), "pairing (fixed G2)")


def pairing_fixed_g2_batch_elems_dev(g1, g2_var, table, k_fixed, out, n, out_order=FQ12_MYFQ12, device=0, stream=None):
    _check(load_library().bn254_pairing_fixed_g2_batch_elems_dev(_dev(g1), _dev(g2_var), _dev(table), k_fixed, _dev(out), n, out_order, device, _stream(stream)),
           "pairing (fixed G2)")


def pairing_fixed_g2_check_batch_dev(g1, g2_var, table, k_fixed, verdict, n, device=0, stream=None):
    _check(load_library().bn254_pairing_fixed_g2_check_batch_dev(_dev(g1), _dev(g2_var), _dev(table), k_fixed, _dev(verdict), n, device, _stream(stream)),
           "pairing check (fixed G2)")


def multi_pairing_check_batch_dev(g1, g2, verdict, n_groups, k, device=0, stream=None):
    _check(load_library().bn254_multi_pairing_check_batch_dev(_dev(g1), _dev(g2), _dev(verdict), n_groups, k, device, _stream(stream)),
           "multi-pairing check")


def _target_words(target):
    """48 host words (MyFq12 order, canonical Montgomery limbs) of a comparison target, or None = MyFq12::one"""
    if target is None:
        return None, None
    t = np.ascontiguousarray(np.asarray(target, dtype=np.uint64).reshape(-1))
    if t.size != 48:
        raise ValueError("target: one Fq12 = 48 words")
    return t, _ptr(t)


def pairing_fixed_g2_check_target_batch_dev(g1, g2_var, table, k_fixed, target, verdict, n, device=0, stream=None):
    """verdict[g] = 1 iff the group's product equals `target` (48 host words as the pairing calls return them; None = one): a Groth16 verifier's
    e(A, B) e(-L, gamma) e(-C, delta) == e(alpha, beta) with gamma, delta in the table"""
    keep, tp = _target_words(target)
    _check(load_library().bn254_pairing_fixed_g2_check_target_batch_dev(_dev(g1), _dev(g2_var), _dev(table), k_fixed, tp, _dev(verdict), n, device,
                                                                         _stream(stream)), "pairing check (fixed G2)")


def multi_pairing_check_target_batch_dev(g1, g2, target, verdict, n_groups, k, device=0, stream=None):
    keep, tp = _target_words(target)
    _check(load_library().bn254_multi_pairing_check_target_batch_dev(_dev(g1), _dev(g2), tp, _dev(verdict), n_groups, k, device, _stream(stream)),
           "multi-pairing check")


def fq12_mul_batch_dev(a, b, out, n, device=0, stream=None):
    """MyFq12 `Mul` on device-resident SoA batches (call sites miller_loop_native.rs:153,239,345)."""
    _check(load_library().bn254_fq12_mul_batch_dev(_dev(a), _dev(b), _dev(out), n, device, _stream(stream)), "MyFq12 mul")


def frobenius_map_batch_dev(a, power, out, n, device=0, stream=None):
    """frobenius_map_native(a, power) on a device-resident SoA batch (src/final_exp_native.rs:17-54)."""
    _check(load_library().bn254_frobenius_map_batch_dev(_dev(a), power, _dev(out), n, device, _stream(stream)), "frobenius_map_native")


def pow_batch_dev(a, exp, out, n, device=0, stream=None):
    """pow_native(a, exp) on a device-resident SoA batch; exp = u64 limbs, least significant first, shared by the batch
    (src/final_exp_native.rs:56-84)."""
    e = np.ascontiguousarray(exp, dtype=np.uint64)
    _check(load_library().bn254_pow_batch_dev(_dev(a), _ptr(e), e.size, _dev(out), n, device, _stream(stream)), "pow_native")


def _devlist(devices):
    arr = (ctypes.c_int * len(devices))(*devices)
    return arr, len(devices)


def pairing_sharded_dev(g1, g2, out, n, devices, stream=None):
    """n pairings resident on devices[0] (device tensors / pointers), slice i computed on devices[i]; synchronous."""
    arr, nd = _devlist(devices)
    _check(load_library().bn254_pairing_sharded_dev(_dev(g1), _dev(g2), _dev(out), n, arr, nd, _stream(stream)), "pairing (sharded, device)")


def multi_pairing_sharded_dev(g1, g2, out, n_groups, k, devices, do_final_exp=True, stream=None):
    arr, nd = _devlist(devices)
    _check(load_library().bn254_multi_pairing_sharded_dev(_dev(g1), _dev(g2), _dev(out), n_groups, k, 1 if do_final_exp else 0, arr, nd,
                                                          _stream(stream)), "multi_miller_loop_native (sharded, device)")


def release_stream(device=0, stream=None):
    _check(load_library().bn254_release_stream(device, _stream(stream)), "release_stream")


def check_points(g1, g2, n, device=0):
    """Raises Bn254Error(ERR_INFINITY) when a pair of the limb-major host batch holds the point at infinity (x = y = 0) in G1 or G2."""
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    _check(load_library().bn254_check_points(_ptr(g1), _ptr(g2), n, device, None), "check_points")


def check_points_dev(g1, g2, n, device=0, stream=None):
    """device batches: sets the stream's sticky status (last_status then raises ERR_INFINITY)"""
    _check(load_library().bn254_check_points_dev(_dev(g1), _dev(g2), n, device, _stream(stream)), "check_points")


def check_points_ex(g1, g2, n, flags=CHECK_INFINITY | CHECK_SUBGROUP, device=0, want_per_point=False):
    """The reference's implicit input contract on limb-major host batches (ark's `Affine::new`: on the curve and, for G2, in the
    r-torsion -- /root/reference/src/miller_loop_native.rs:303,311).  Raises Bn254Error with ERR_INFINITY / ERR_NOT_ON_CURVE /
    ERR_NOT_IN_SUBGROUP; with want_per_point returns (status, uint8 verdicts) instead of raising."""
    g1, g2 = _np_in(g1, G1_WORDS, n), _np_in(g2, G2_WORDS, n)
    per = np.zeros(n, dtype=np.uint8) if want_per_point else None
    rc = load_library().bn254_check_points_ex(_ptr(g1), _ptr(g2), n, flags, _ptr(per) if want_per_point else None, device, None)
    if want_per_point:
        return rc, per
    _check(rc, "check_points_ex")


def check_points_ex_dev(g1, g2, n, flags=CHECK_INFINITY | CHECK_SUBGROUP, per_point=None, device=0, stream=None):
    """device batches: sets the stream's sticky point-check status (last_status then raises); per_point: optional uint8 device tensor"""
    _check(load_library().bn254_check_points_ex_dev(_dev(g1), _dev(g2), n, flags, _dev(per_point) if per_point is not None else None, device,
                                                    _stream(stream)), "check_points_ex")


def set_stream_latency(threshold=LATENCY_INHERIT, lanes=-1, device=0, stream=None):
    """This (device, stream)'s own kernel selection: batches of at most `threshold` items take the lane-cooperative kernel (0: never),
    `lanes` = 0 / 16 / 32 / 64 picks its program family; LATENCY_INHERIT / -1 return to the process-wide defaults."""
    _check(load_library().bn254_set_stream_latency(device, _stream(stream), threshold, lanes), "set_stream_latency")


def last_kernel(device=0, stream=None):
    """1 = the throughput kernel, 16 / 32 / 64 = lanes per item of the lane-cooperative kernel, 0 = no compute launch on this stream yet"""
    return load_library().bn254_last_kernel(device, _stream(stream))


def reserve(n, k=1, device=0, stream=None):
    """Sizes the per-(device, stream) buffers for `_dev` calls of up to n lanes x k pairs: no later call of that size allocates."""
    _check(load_library().bn254_reserve(device, _stream(stream), n, k), "reserve")


def set_wide_groups(max_groups):
    """groups of more than 64 pairs: batches of fewer than max_groups groups spread each group over several lanes (default 65536; 0: never)"""
    load_library().bn254_set_wide_groups(max_groups)


def get_wide_groups():
    return load_library().bn254_get_wide_groups()


def set_latency_threshold(n):
    """Batches of at most n pairings take the lane-cooperative (latency) kernel; 0 turns it off (include/bn254_pairing.h)."""
    load_library().bn254_set_latency_threshold(n)


def get_latency_threshold():
    return load_library().bn254_get_latency_threshold()


def set_latency_lanes(lanes):
    """0: the lane-cooperative program family by launch size (sixty-four / thirty-two lanes per item for the smallest launches); 16 / 32 / 64: fixed."""
    load_library().bn254_set_latency_lanes(lanes)


def get_latency_lanes():
    return load_library().bn254_get_latency_lanes()


def generate_pairs_dev(seed, g1_out, g2_out, n, device=0, stream=None):
    _check(load_library().bn254_generate_pairs_dev(seed, _dev(g1_out), _dev(g2_out), n, device, _stream(stream)), "generate_pairs")


def generator_scalars(seed, i):
    """The scalars (s_i, t_i) bn254_generate_pairs_dev uses for pair i: P_i = [s_i] G1, Q_i = [t_i] G2.  Four SplitMix64
    draws from the state seed ^ (0xD1B54A32D192ED03 * (i + 1)) give two 128-bit strings (low word first); each is read as 32
    radix-16 digits, a zero digit counting as 1, so s = sum d_k 16^k with every d_k in 1..15 (the fixed-base window method
    of the generator kernel then performs exactly one table addition per digit).  Never zero mod r; NOT uniform mod r:
    synthetic bench / test inputs -- the pairing kernels have no data-dependent control flow."""
    m64 = (1 << 64) - 1
    st = (seed ^ (0xD1B54A32D192ED03 * (i + 1))) & m64
    draws = []
    for _ in range(4):
        st = (st + 0x9E3779B97F4A7C15) & m64
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
        draws.append(z ^ (z >> 31))

    def digits(v):
        return sum(max((v >> (4 * k)) & 15, 1) << (4 * k) for k in range(32))

    return digits(draws[0] | (draws[1] << 64)), digits(draws[2] | (draws[3] << 64))


def last_status(device=0, stream=None):
    """Synchronises the stream and raises where the reference would have panicked."""
    _check(load_library().bn254_last_status(device, _stream(stream)), "device status")


# ----------------------------------------------------------------------------- scalar mirrors of the reference's functions
# Points / field elements are flat numpy uint64 arrays in the AoS order of include/bn254_pairing.h
# (for n = 1 AoS == SoA): G1 = x,y (8 words); G2 = x.c0,x.c1,y.c0,y.c1 (16); MyFq12 = coeffs[0..12] (48).
def miller_loop_native(Q, P_):
    """miller_loop_native(Q: &G2Affine, P: &G1Affine) -> MyFq12   (src/miller_loop_native.rs:320)"""
    return miller_loop_batch(P_, Q, 1)


def multi_miller_loop_native(pairs):
    """multi_miller_loop_native(pairs: Vec<(&G1Affine, &G2Affine)>) -> MyFq12   (:324)"""
    k = len(pairs)
    if k == 0:
        raise Bn254Error(ERR_INVALID_ARG, "multi_miller_loop_native: empty pair list (the reference panics: pairs[0])")
    g1 = layout.to_soa(np.concatenate([np.asarray(a, dtype=np.uint64) for a, _ in pairs]), G1_WORDS)
    g2 = layout.to_soa(np.concatenate([np.asarray(b, dtype=np.uint64) for _, b in pairs]), G2_WORDS)
    return multi_pairing_batch(g1, g2, 1, k, do_final_exp=False)


def final_exp_native(a):
    """final_exp_native(a: MyFq12) -> MyFq12   (src/final_exp_native.rs:209)"""
    return final_exp_batch(a, 1)


def pairing(p, q):
    """pairing(p: G1Affine, q: G2Affine) -> Fq12 in ark's flat order (src/pairing.rs:20-22)"""
    my = pairing_batch(p, q, 1).reshape(12, 4)
    lib = load_library()
    return np.concatenate([my[lib.bn254_myfq12_to_ark_index(j)] for j in range(12)])


def frobenius_map_native(a, power):
    """frobenius_map_native(a: MyFq12, power: usize)   (src/final_exp_native.rs:17)"""
    return frobenius_map_batch(a, power, 1)


def pow_native(a, exp):
    """pow_native(a: MyFq12, exp: Vec<u64>)   (src/final_exp_native.rs:56)"""
    return pow_batch(a, exp, 1)


def get_naf(exp):
    """get_naf(exp: Vec<u64>) -> Vec<i8>   (src/final_exp_native.rs:86) -- host logic in the C++ library."""
    lib = load_library()
    e = np.ascontiguousarray(exp, dtype=np.uint64)
    naf = np.zeros(64 * max(e.size, 1) + 1, dtype=np.int8)
    n = lib.bn254_get_naf(_ptr(e), e.size, _ptr(naf))
    if n < 0:
        raise Bn254Error(int(n), "get_naf")
    return naf[:n].tolist()


def frob_coeffs(index):
    """frob_coeffs(index: usize) -> Fq2 as 8 u64 (c0, c1)   (src/final_exp_native.rs:183)"""
    out = np.zeros(8, dtype=np.uint64)
    _check(load_library().bn254_frob_coeffs(index % 12 if index >= 12 else index, _ptr(out)), "frob_coeffs")
    return out


def _fq_neg_limbs(l4):
    v = layout.limbs_to_int(l4)
    return np.array(layout.fq_to_limbs((P - v) % P), dtype=np.uint64)


def conjugate_fp2(x):
    """conjugate_fp2(x: Fq2) -> Fq2 = (c0, -c1)   (src/miller_loop_native.rs:284)"""
    x = np.asarray(x, dtype=np.uint64)
    return np.concatenate([x[:4], _fq_neg_limbs(x[4:8])])


def neg_conjugate_fp2(x):
    """neg_conjugate_fp2(x: Fq2) -> Fq2 = (-c0, c1)   (src/miller_loop_native.rs:291)"""
    x = np.asarray(x, dtype=np.uint64)
    return np.concatenate([_fq_neg_limbs(x[:4]), x[4:8]])
