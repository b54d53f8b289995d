//! Rust shim over `include/bn254_pairing.h`: the reference's native functions with their original
//! signatures (src/pairing.rs:20, src/miller_loop_native.rs:320,324, src/final_exp_native.rs:17,56,86,183,209),
//! executed by the MI355X engine.  `repr(Rust)` structs are never transmuted: fields are copied limb by
//! limb (`Fp.0.0` is the Montgomery representation the C ABI uses, so no conversion happens).
//! NOT compiled in the build image (no Rust toolchain there); kept as the binding a maintainer adds.
#![allow(non_snake_case)]
use ark_bn254::{Fq, Fq12, Fq2, G1Affine, G2Affine};
use ark_ff::{BigInt, Fp};
use plonky2_bn254::fields::native::MyFq12;
use std::os::raw::{c_int, c_long, c_void};

extern "C" {
    fn bn254_pairing_batch(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_miller_loop_batch(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_final_exp_batch(f: *const u64, out: *mut u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_multi_pairing_batch(g1: *const u64, g2: *const u64, out: *mut u64, n_groups: usize, k: usize, do_final_exp: c_int,
                                 device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_multi_pairing_check_batch(g1: *const u64, g2: *const u64, verdict: *mut u8, n_groups: usize, k: usize, device: c_int,
                                       stream: *mut c_void) -> c_int;
    fn bn254_reserve(device: c_int, stream: *mut c_void, n: usize, k: usize) -> c_int;
    fn bn254_alloc_pinned(bytes: usize, out: *mut *mut c_void) -> c_int;
    fn bn254_free_pinned(ptr: *mut c_void) -> c_int;
    fn bn254_host_register(ptr: *mut c_void, bytes: usize) -> c_int;
    fn bn254_host_unregister(ptr: *mut c_void) -> c_int;
    fn bn254_set_latency_threshold(n: usize);
    fn bn254_get_latency_threshold() -> usize;
    fn bn254_set_latency_lanes(lanes: c_int);
    fn bn254_get_latency_lanes() -> c_int;
    fn bn254_check_points(g1: *const u64, g2: *const u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_check_points_ex(g1: *const u64, g2: *const u64, n: usize, flags: c_int, per_point: *mut u8, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_set_stream_latency(device: c_int, stream: *mut c_void, threshold: usize, lanes: c_int) -> c_int;
    fn bn254_pairing_sharded(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, n_devices: c_int) -> c_int;
    // device pointers on devices[0] (NULL: devices 0..n_devices-1); shard i runs on devices[i]; synchronous
    #[allow(dead_code)]
    fn bn254_pairing_sharded_dev(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, devices: *const c_int, n_devices: c_int,
                                 stream: *mut c_void) -> c_int;
    fn bn254_frobenius_map_batch(a: *const u64, power: usize, out: *mut u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_pow_batch(a: *const u64, exp: *const u64, exp_limbs: usize, out: *mut u64, n: usize, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_get_naf(exp: *const u64, exp_limbs: usize, naf: *mut i8) -> c_long;
    fn bn254_frob_coeffs(index: usize, out8: *mut u64) -> c_int;
    fn bn254_myfq12_to_ark_index(j: c_int) -> c_int;
    // element-major entry points: the order a `&[G1Affine]` / `Vec<Fq12>` is copied out in (fields one after the other)
    fn bn254_pairing_batch_elems(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, out_order: c_int, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_multi_pairing_batch_elems(g1: *const u64, g2: *const u64, out: *mut u64, n_groups: usize, k: usize, do_final_exp: c_int,
                                       out_order: c_int, device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_pairing_sharded_elems(g1: *const u64, g2: *const u64, out: *mut u64, n: usize, out_order: c_int, n_devices: c_int) -> c_int;
    fn bn254_multi_pairing_check_batch_elems(g1: *const u64, g2: *const u64, verdict: *mut u8, n_groups: usize, k: usize, device: c_int,
                                             stream: *mut c_void) -> c_int;
    fn bn254_pairing_fixed_g2_batch_elems(g1: *const u64, g2_var: *const u64, g2_fixed: *const u64, k_fixed: usize, out: *mut u64, n: usize, out_order: c_int,
                                          device: c_int, stream: *mut c_void) -> c_int;
    fn bn254_pairing_fixed_g2_check_batch_elems(g1: *const u64, g2_var: *const u64, g2_fixed: *const u64, k_fixed: usize, target: *const u64, verdict: *mut u8,
                                                n: usize, device: c_int, stream: *mut c_void) -> c_int;
}
const FQ12_MYFQ12: c_int = 0;
const FQ12_ARK: c_int = 1;

pub const BN_X: u64 = 4965661367192848881;
pub const SIX_U_PLUS_2_NAF: [i8; 65] = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0,
    -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1,
];

fn limbs(x: &Fq) -> [u64; 4] { (x.0).0 }
fn from_limbs(l: &[u64]) -> Fq { Fp(BigInt([l[0], l[1], l[2], l[3]]), core::marker::PhantomData) }
fn ok(rc: c_int) { if rc != 0 { panic!("bn254 engine status {}", rc) } } // the reference panics in the same places

/// SoA batch writers (elem(c, l, i) = buf[(c*4 + l)*n + i]).
pub fn pack_g1(ps: &[G1Affine]) -> Vec<u64> {
    let n = ps.len(); let mut b = vec![0u64; 8 * n];
    for (i, p) in ps.iter().enumerate() { for l in 0..4 { b[l * n + i] = limbs(&p.x)[l]; b[(4 + l) * n + i] = limbs(&p.y)[l]; } }
    b
}
pub fn pack_g2(qs: &[G2Affine]) -> Vec<u64> {
    let n = qs.len(); let mut b = vec![0u64; 16 * n];
    for (i, q) in qs.iter().enumerate() {
        for l in 0..4 {
            b[l * n + i] = limbs(&q.x.c0)[l]; b[(4 + l) * n + i] = limbs(&q.x.c1)[l];
            b[(8 + l) * n + i] = limbs(&q.y.c0)[l]; b[(12 + l) * n + i] = limbs(&q.y.c1)[l];
        }
    }
    b
}
fn pack_fq12(a: &MyFq12) -> [u64; 48] { let mut b = [0u64; 48]; for c in 0..12 { b[4 * c..4 * c + 4].copy_from_slice(&limbs(&a.coeffs[c])); } b }
fn unpack_fq12(b: &[u64]) -> MyFq12 { let mut c = [Fq::from(0u64); 12]; for i in 0..12 { c[i] = from_limbs(&b[4 * i..4 * i + 4]); } MyFq12 { coeffs: c } }

pub fn miller_loop_native(Q: &G2Affine, P: &G1Affine) -> MyFq12 {
    let (g1, g2) = (pack_g1(&[*P]), pack_g2(&[*Q])); let mut out = [0u64; 48];
    ok(unsafe { bn254_miller_loop_batch(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), 1, 0, core::ptr::null_mut()) });
    unpack_fq12(&out)
}
pub fn multi_miller_loop_native(pairs: Vec<(&G1Affine, &G2Affine)>) -> MyFq12 {
    let ps: Vec<G1Affine> = pairs.iter().map(|p| *p.0).collect(); let qs: Vec<G2Affine> = pairs.iter().map(|p| *p.1).collect();
    let (g1, g2) = (pack_g1(&ps), pack_g2(&qs)); let mut out = [0u64; 48];
    ok(unsafe { bn254_multi_pairing_batch(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), 1, pairs.len(), 0, 0, core::ptr::null_mut()) });
    unpack_fq12(&out)
}
pub fn final_exp_native(a: MyFq12) -> MyFq12 {
    let inp = pack_fq12(&a); let mut out = [0u64; 48];
    ok(unsafe { bn254_final_exp_batch(inp.as_ptr(), out.as_mut_ptr(), 1, 0, core::ptr::null_mut()) });
    unpack_fq12(&out)
}
pub fn pairing(p: G1Affine, q: G2Affine) -> Fq12 {
    let (g1, g2) = (pack_g1(&[p]), pack_g2(&[q])); let mut out = [0u64; 48];
    ok(unsafe { bn254_pairing_batch(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), 1, 0, core::ptr::null_mut()) });
    unpack_fq12(&out).into() // MyFq12 -> Fq12, as at src/pairing.rs:21 (or use bn254_myfq12_to_ark_index)
}
pub fn frobenius_map_native(a: MyFq12, power: usize) -> MyFq12 {
    let inp = pack_fq12(&a); let mut out = [0u64; 48];
    ok(unsafe { bn254_frobenius_map_batch(inp.as_ptr(), power, out.as_mut_ptr(), 1, 0, core::ptr::null_mut()) });
    unpack_fq12(&out)
}
pub fn pow_native(a: MyFq12, exp: Vec<u64>) -> MyFq12 {
    let inp = pack_fq12(&a); let mut out = [0u64; 48];
    ok(unsafe { bn254_pow_batch(inp.as_ptr(), exp.as_ptr(), exp.len(), out.as_mut_ptr(), 1, 0, core::ptr::null_mut()) });
    unpack_fq12(&out)
}
pub fn get_naf(exp: Vec<u64>) -> Vec<i8> {
    let mut naf = vec![0i8; 64 * exp.len() + 1];
    let n = unsafe { bn254_get_naf(exp.as_ptr(), exp.len(), naf.as_mut_ptr()) };
    if n < 0 { panic!("get_naf: carry out of the top limb") }
    naf.truncate(n as usize); naf
}
pub fn frob_coeffs(index: usize) -> Fq2 {
    let mut o = [0u64; 8]; ok(unsafe { bn254_frob_coeffs(index % 12, o.as_mut_ptr()) });
    Fq2::new(from_limbs(&o[0..4]), from_limbs(&o[4..8]))
}
pub fn conjugate_fp2(x: Fq2) -> Fq2 { Fq2::new(x.c0, -x.c1) }
pub fn neg_conjugate_fp2(x: Fq2) -> Fq2 { Fq2::new(-x.c0, x.c1) }

/// Element-major writers: fields copied out one after the other (no transposition on the host; `repr(Rust)` structs are
/// still never transmuted).  The engine makes its limb-major planes on the device.
pub fn elems_g1(ps: &[G1Affine]) -> Vec<u64> { let mut b = Vec::with_capacity(8 * ps.len()); for p in ps { b.extend_from_slice(&limbs(&p.x)); b.extend_from_slice(&limbs(&p.y)); } b }
pub fn elems_g2(qs: &[G2Affine]) -> Vec<u64> {
    let mut b = Vec::with_capacity(16 * qs.len());
    for q in qs { for f in [&q.x.c0, &q.x.c1, &q.y.c0, &q.y.c1] { b.extend_from_slice(&limbs(f)); } }
    b
}
fn fq12_from_ark_words(w: &[u64]) -> Fq12 {
    let f = |j: usize| from_limbs(&w[4 * j..4 * j + 4]);
    let f2 = |j: usize| Fq2::new(f(2 * j), f(2 * j + 1));
    Fq12::new(ark_bn254::Fq6::new(f2(0), f2(1), f2(2)), ark_bn254::Fq6::new(f2(3), f2(4), f2(5)))
}
/// Sizes the library's per-(device, stream) buffers for calls of up to `n` lanes x `k` pairs on device 0 / the NULL stream
/// (the ones every function of this shim uses): no later call of that size allocates device memory.
pub fn reserve(n: usize, k: usize) { ok(unsafe { bn254_reserve(0, core::ptr::null_mut(), n, k) }) }

/// Batches of at most `n` items run on the lane-cooperative (latency) kernel -- what the scalar functions of this shim (`pairing`,
/// `miller_loop_native`, `multi_miller_loop_native`, `final_exp_native`: one element per call) get by default; 0 turns it off.
pub fn set_latency_threshold(n: usize) { unsafe { bn254_set_latency_threshold(n) } }
pub fn latency_threshold() -> usize { unsafe { bn254_get_latency_threshold() } }
/// 0 (default): sixty-four / thirty-two lanes per item while the launch is at most one wave per SIMD, sixteen beyond; 16 / 32 / 64: fixed.
pub fn set_latency_lanes(lanes: i32) { unsafe { bn254_set_latency_lanes(lanes) } }
pub fn latency_lanes() -> i32 { unsafe { bn254_get_latency_lanes() } }

/// The reference never looks at `infinity` (its line functions read raw x / y, miller_loop_native.rs:10-44): an infinite input is outside
/// its contract.  Callers that want it REPORTED use this: the flags of the structs first (no device work), then the engine's check of the
/// coordinates (ark's affine identity is x = y = 0); `Err(-7)` = BN254_ERR_INFINITY.
pub fn check_points(ps: &[G1Affine], qs: &[G2Affine]) -> Result<(), i32> {
    assert_eq!(ps.len(), qs.len());       // the planes of both arrays are packed with stride n: a shorter `qs` would be read past its end
    if ps.is_empty() { return Ok(()); }
    if ps.iter().any(|p| p.infinity) || qs.iter().any(|q| q.infinity) { return Err(-7); }
    let (g1, g2) = (pack_g1(ps), pack_g2(qs));
    match unsafe { bn254_check_points(g1.as_ptr(), g2.as_ptr(), ps.len(), 0, core::ptr::null_mut()) } { 0 => Ok(()), rc => Err(rc) }
}

/// The rest of ark's implicit contract, for untrusted points: `G2Affine::new` -- which the reference itself calls on the Frobenius images
/// of Q (miller_loop_native.rs:303,311) -- panics unless the point is on the curve and in the r-torsion.  The engine computes a value for
/// any coordinates; this runs its optional check (flags 1 | 2 | 4: infinity, on-curve, G2 subgroup) and panics like the reference would:
/// `Err(-7 / -8 / -9)` = infinity / not on the curve / G2 not in the subgroup.
pub fn check_points_full(ps: &[G1Affine], qs: &[G2Affine]) -> Result<(), i32> {
    assert_eq!(ps.len(), qs.len());
    if ps.is_empty() { return Ok(()); }
    if ps.iter().any(|p| p.infinity) || qs.iter().any(|q| q.infinity) { return Err(-7); }
    let (g1, g2) = (pack_g1(ps), pack_g2(qs));
    match unsafe { bn254_check_points_ex(g1.as_ptr(), g2.as_ptr(), ps.len(), 7, core::ptr::null_mut(), 0, core::ptr::null_mut()) } { 0 => Ok(()), rc => Err(rc) }
}
/// Kernel selection of the NULL stream of device 0 (the stream this shim uses) alone, whatever other users of the library set
/// process-wide: `threshold = usize::MAX` / `lanes = -1` return to the defaults.
pub fn set_stream_latency(threshold: usize, lanes: i32) { ok(unsafe { bn254_set_stream_latency(0, core::ptr::null_mut(), threshold, lanes) }) }

/// Page-locked staging for the batch functions: the reference's callers hold `repr(Rust)` structs, so a batch is copied limb by limb into
/// u64 words anyway -- into page-locked words (`bn254_alloc_pinned`) the engine then copies by DMA underneath its kernels, and a large batch
/// runs at the resident-data rate instead of the pageable-copy rate (include/bn254_pairing.h, PAGE-LOCKED HOST MEMORY).  Keep one
/// `PinnedWords` per array and reuse it across calls: allocating page-locked memory is slow.
pub struct PinnedWords { ptr: *mut u64, cap: usize }
impl PinnedWords {
    pub fn new() -> Self { PinnedWords { ptr: core::ptr::null_mut(), cap: 0 } }
    /// at least `words` u64 (contents unspecified after growth)
    pub fn reserve(&mut self, words: usize) -> &mut [u64] {
        if words > self.cap {
            if !self.ptr.is_null() { ok(unsafe { bn254_free_pinned(self.ptr as *mut c_void) }); self.ptr = core::ptr::null_mut(); self.cap = 0; }
            let mut p: *mut c_void = core::ptr::null_mut();
            ok(unsafe { bn254_alloc_pinned(8 * words.max(1), &mut p) });
            self.ptr = p as *mut u64; self.cap = words;
        }
        unsafe { core::slice::from_raw_parts_mut(self.ptr, words) }
    }
}
impl Drop for PinnedWords { fn drop(&mut self) { if !self.ptr.is_null() { unsafe { bn254_free_pinned(self.ptr as *mut c_void); } } } }
/// Page-locks a buffer the caller owns for the lifetime of the guard (`bn254_host_register`): for `Vec<u64>` staging that is reused.
pub struct HostRegistration { ptr: *mut c_void }
impl HostRegistration {
    pub fn new(buf: &mut [u64]) -> Self { ok(unsafe { bn254_host_register(buf.as_mut_ptr() as *mut c_void, 8 * buf.len()) }); HostRegistration { ptr: buf.as_mut_ptr() as *mut c_void } }
}
impl Drop for HostRegistration { fn drop(&mut self) { unsafe { bn254_host_unregister(self.ptr); } } }
/// Reusable page-locked staging of one caller (three arrays: G1 words, G2 words, Fq12 words).
pub struct BatchStaging { g1: PinnedWords, g2: PinnedWords, out: PinnedWords }
impl BatchStaging { pub fn new() -> Self { BatchStaging { g1: PinnedWords::new(), g2: PinnedWords::new(), out: PinnedWords::new() } } }
/// `pairing_batch_fq12` through page-locked staging: n x `pairing(p, q)` as `Fq12` (src/pairing.rs:20-22) at the engine's resident-data rate.
pub fn pairing_batch_fq12_pinned(st: &mut BatchStaging, ps: &[G1Affine], qs: &[G2Affine]) -> Vec<Fq12> {
    assert_eq!(ps.len(), qs.len()); let n = ps.len();
    if n == 0 { return Vec::new(); }
    let g1 = st.g1.reserve(8 * n);
    for (i, p) in ps.iter().enumerate() { g1[8 * i..8 * i + 4].copy_from_slice(&limbs(&p.x)); g1[8 * i + 4..8 * i + 8].copy_from_slice(&limbs(&p.y)); }
    let g2 = st.g2.reserve(16 * n);
    for (i, q) in qs.iter().enumerate() {
        g2[16 * i..16 * i + 4].copy_from_slice(&limbs(&q.x.c0)); g2[16 * i + 4..16 * i + 8].copy_from_slice(&limbs(&q.x.c1));
        g2[16 * i + 8..16 * i + 12].copy_from_slice(&limbs(&q.y.c0)); g2[16 * i + 12..16 * i + 16].copy_from_slice(&limbs(&q.y.c1));
    }
    let (p1, p2) = (st.g1.ptr as *const u64, st.g2.ptr as *const u64);
    let out = st.out.reserve(48 * n);
    ok(unsafe { bn254_pairing_batch_elems(p1, p2, out.as_mut_ptr(), n, FQ12_ARK, 0, core::ptr::null_mut()) });
    out.chunks_exact(48).map(fq12_from_ark_words).collect()
}

/// New: whole batches in one launch (what the engine is for).  Output: MyFq12 per pairing.
pub fn pairing_batch(ps: &[G1Affine], qs: &[G2Affine]) -> Vec<MyFq12> {
    assert_eq!(ps.len(), qs.len()); let n = ps.len();
    let (g1, g2) = (elems_g1(ps), elems_g2(qs)); let mut out = vec![0u64; 48 * n];
    ok(unsafe { bn254_pairing_batch_elems(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), n, FQ12_MYFQ12, 0, core::ptr::null_mut()) });
    out.chunks_exact(48).map(unpack_fq12).collect()
}
/// New: n x `pairing(p, q)` as `Fq12`, exactly what src/pairing.rs:20-22 returns (the `.into()` runs on the device).
pub fn pairing_batch_fq12(ps: &[G1Affine], qs: &[G2Affine]) -> Vec<Fq12> {
    assert_eq!(ps.len(), qs.len()); let n = ps.len();
    let (g1, g2) = (elems_g1(ps), elems_g2(qs)); let mut out = vec![0u64; 48 * n];
    ok(unsafe { bn254_pairing_batch_elems(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), n, FQ12_ARK, 0, core::ptr::null_mut()) });
    out.chunks_exact(48).map(fq12_from_ark_words).collect()
}
/// New: groups of k pairs, `multi_miller_loop_native` per group (+ `final_exp_native` when asked), one launch.
pub fn multi_pairing_batch(ps: &[G1Affine], qs: &[G2Affine], k: usize, do_final_exp: bool) -> Vec<MyFq12> {
    assert!(k > 0 && ps.len() == qs.len() && ps.len() % k == 0); let n_groups = ps.len() / k;
    let (g1, g2) = (elems_g1(ps), elems_g2(qs)); let mut out = vec![0u64; 48 * n_groups];
    ok(unsafe { bn254_multi_pairing_batch_elems(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), n_groups, k, do_final_exp as c_int, FQ12_MYFQ12, 0,
                                                core::ptr::null_mut()) });
    out.chunks_exact(48).map(unpack_fq12).collect()
}
/// New: the same batch spread over the first `n_devices` GPUs of this process (contiguous slices, no exchange).
pub fn pairing_sharded(ps: &[G1Affine], qs: &[G2Affine], n_devices: i32) -> Vec<MyFq12> {
    assert_eq!(ps.len(), qs.len()); let n = ps.len();
    let (g1, g2) = (elems_g1(ps), elems_g2(qs)); let mut out = vec![0u64; 48 * n];
    ok(unsafe { bn254_pairing_sharded_elems(g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), n, FQ12_MYFQ12, n_devices) });
    out.chunks_exact(48).map(unpack_fq12).collect()
}
/// New: `final_exp_native(multi_miller_loop_native(group)) == MyFq12::one` for every group of k pairs
/// (the check of final_exp_native.rs:245-263), one verdict per group instead of 384 bytes.
pub fn multi_pairing_check_batch(ps: &[G1Affine], qs: &[G2Affine], k: usize) -> Vec<bool> {
    assert!(k > 0 && ps.len() == qs.len() && ps.len() % k == 0); let n_groups = ps.len() / k;
    let (g1, g2) = (elems_g1(ps), elems_g2(qs)); let mut v = vec![0u8; n_groups];
    ok(unsafe { bn254_multi_pairing_check_batch_elems(g1.as_ptr(), g2.as_ptr(), v.as_mut_ptr(), n_groups, k, 0, core::ptr::null_mut()) });
    v.into_iter().map(|b| b != 0).collect()
}
/// New: groups whose last `fixed.len()` G2 points are the same for the whole batch (a Groth16 verifier's beta, gamma, delta): `ps` holds
/// 1 + fixed.len() G1 points per group (the group's own first), `qs` the groups' own G2 points.  `final_exp_native(multi_miller_loop_native(..))`
/// per group -- the same value as `multi_pairing_batch` on the expanded pairs -- with the fixed pairs' point steps done once for the batch.
pub fn pairing_fixed_g2_batch(ps: &[G1Affine], qs: &[G2Affine], fixed: &[G2Affine]) -> Vec<MyFq12> {
    let (n, kf) = (qs.len(), fixed.len());
    assert!(kf > 0 && kf <= 4 && ps.len() == n * (kf + 1));
    let (g1, g2, gf) = (elems_g1(ps), elems_g2(qs), elems_g2(fixed)); let mut out = vec![0u64; 48 * n];
    ok(unsafe { bn254_pairing_fixed_g2_batch_elems(g1.as_ptr(), g2.as_ptr(), gf.as_ptr(), kf, out.as_mut_ptr(), n, FQ12_MYFQ12, 0, core::ptr::null_mut()) });
    out.chunks_exact(48).map(unpack_fq12).collect()
}
/// New: a Groth16 verifier's pairing check for a batch of proofs: `product of the group's 1 + fixed.len() pairings == target` (`None`: `MyFq12::one`).
/// With gamma, delta as `fixed` and `target = pairing(alpha, beta)` a proof costs 1 + 2 pairs.
/// `qs` empty: the groups have NO pair of their own -- `ps` holds `fixed.len()` points per group, every G2 point is one of `fixed` (a KZG / PLONK opening
/// check `e(P_1, [tau] G2) e(P_2, G2) == 1`: two pairings for the price of one with a free G2 point).
pub fn pairing_fixed_g2_check_batch(ps: &[G1Affine], qs: &[G2Affine], fixed: &[G2Affine], target: Option<&MyFq12>) -> Vec<bool> {
    let kf = fixed.len();
    assert!(kf > 0 && kf <= 4);
    let own = if qs.is_empty() { 0 } else { 1 };
    let n = if own == 1 { qs.len() } else { ps.len() / kf };
    assert!(ps.len() == n * (kf + own));
    if n == 0 { return Vec::new(); }
    let (g1, g2, gf) = (elems_g1(ps), elems_g2(qs), elems_g2(fixed)); let mut v = vec![0u8; n];
    let t: Option<[u64; 48]> = target.map(pack_fq12);
    let tp = t.as_ref().map_or(core::ptr::null(), |w| w.as_ptr());
    let qp = if own == 1 { g2.as_ptr() } else { core::ptr::null() };
    ok(unsafe { bn254_pairing_fixed_g2_check_batch_elems(g1.as_ptr(), qp, gf.as_ptr(), kf, tp, v.as_mut_ptr(), n, 0, core::ptr::null_mut()) });
    v.into_iter().map(|b| b != 0).collect()
}
#[allow(dead_code)] fn _ark_index(j: i32) -> i32 { unsafe { bn254_myfq12_to_ark_index(j) } }
