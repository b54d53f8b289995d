/*
 * bn254_pairing.h -- C ABI of the MI355X-native batched BN254 pairing engine.
 *
 * Drop-in boundary for the native witness path of qope/plonky2-bn254-pairing.  The
 * reference has no FFI of its own: its boundary is a set of plain Rust `pub fn`s.  Each
 * entry point below names the reference function it replaces (file:line relative to the
 * reference repository); rust-shim/ and INTEGRATION.md show the Rust side that keeps the
 * original signatures on top of this header.
 *
 * DATA FORMAT (all entry points)
 *   Fq      4 x u64 little-endian limbs, Montgomery form, R = 2^256 -- bit-identical to
 *           ark-ff's `Fp.0.0` for ark_bn254::Fq, so the shim does no conversion.
 *   Batches are struct-of-arrays, limb-major:   elem(c, l, i) = buf[(c*4 + l)*n + i]
 *     G1      c in {x, y}                                  ->  8*n u64
 *     G2      c in {x.c0, x.c1, y.c0, y.c1}                -> 16*n u64
 *     MyFq12  c = MyFq12.coeffs index 0..11                -> 48*n u64
 *             (coeffs[i] + coeffs[i+6] u is the Fq2 coefficient of w^i, w^6 = 9+u)
 *   For n = 1 SoA and AoS coincide, which is what the scalar Rust signatures use.
 *
 * POINTERS   `*_dev` entry points take DEVICE pointers (HBM-resident inputs/outputs; the
 *            timed path).  The un-suffixed entry points take HOST pointers and stage
 *            through the device (PCIe inclusive).
 * STREAM     `stream` is a hipStream_t (NULL = default stream).  `_dev` calls are enqueued on it and return after the
 *            launch: none of them waits for the stream.  What a `_dev` call may do on the calling thread is ALLOCATE: the
 *            library keeps scratch, a status word and a few work buffers per (device, stream) and grows them when a call
 *            needs more (larger n or k, the first group of more than 64 pairs, the first `== one` verdict, the first
 *            pow_native); an outgrown buffer is retired -- work queued on the stream may still use it -- and freed inside the
 *            next bn254_last_status / bn254_release_stream.  bn254_reserve(device, stream, n, k) sizes everything for calls of
 *            up to n lanes x k pairs up front: after it no `_dev` call of that size or smaller allocates.  bn254_pow_batch_dev
 *            copies its digits through pinned staging slots (four after bn254_reserve, more are added while all are in
 *            flight; with 64 pow calls queued on one stream the next one waits for the oldest copy).  Status words are read,
 *            on that same stream, only inside bn254_last_status().  Host-pointer calls synchronise before returning.
 * ERRORS     0 = success, negative = failure (bn254_strerror).  The reference panics on
 *            the same conditions (division by zero in ark's `/`); the Rust shim turns a
 *            negative status back into a panic.  Points at infinity are outside the
 *            reference's contract (it reads raw x/y) and outside the hot path's; bn254_check_points
 *            gives them a distinct status (BN254_ERR_INFINITY) for callers that ask.
 * THREADING  Re-entrant.  The only state is what the library keeps per (device, stream):
 *            scratch, the status word, the staging buffers of the host-pointer calls.  Each
 *            such context has a mutex: a `_dev` call holds it from the look-up of its
 *            buffers to the launch, a host-pointer call from staging to the read-back of
 *            its results, so host threads may share a stream (the scalar Rust / C++
 *            signatures all use device 0 / the NULL stream: concurrent callers are
 *            serialised, not corrupted); calls on different streams run independently.
 * LIFETIME   A stream's context is created by the first call that names the stream and lives
 *            until bn254_release_stream(device, stream) -- call it before destroying a stream
 *            you used, otherwise its buffers stay allocated (a later stream that reuses the
 *            handle value simply inherits them).  A call that races a release on the same
 *            (device, stream) keeps the context alive until it returns (the context is
 *            reference-counted); its buffers are freed by whichever of the two finishes last.
 *            The private streams of the host-pointer pipeline and of the `_sharded_dev` calls
 *            live as long as the library, with their buffers: per stream one scratch area of
 *            grid x pitch bytes (bn254_scratch_bytes: 0.5 GiB for single pairings at a full
 *            grid, 2.5 GiB for 64 pairs per lane) plus the staging buffers of the largest chunk.
 */
#ifndef BN254_PAIRING_H
#define BN254_PAIRING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN254_OK 0
#define BN254_ERR_INVALID_ARG (-1)
#define BN254_ERR_NO_DEVICE (-2)
#define BN254_ERR_HIP (-3)
#define BN254_ERR_ZERO_DIVISOR (-4) /* reference: ark `/` panics (final_exp_native.rs:74,200) */
#define BN254_ERR_NAF_CARRY (-5)    /* reference: get_naf assert at final_exp_native.rs:123 */
#define BN254_ERR_ALLOC (-6)
#define BN254_ERR_INFINITY (-7)     /* bn254_check_points: a point at infinity (ark affine form x = y = 0) in the batch */
#define BN254_ERR_NOT_ON_CURVE (-8)    /* bn254_check_points_ex: a point is not on its curve / a coordinate is not below p */
#define BN254_ERR_NOT_IN_SUBGROUP (-9) /* bn254_check_points_ex: a G2 point outside the r-torsion; reference: `G2Affine::new` panics
                                        * (src/miller_loop_native.rs:303,311) */

/* number of visible HIP devices (0 when none / no driver) */
int bn254_device_count(void);
const char* bn254_strerror(int status);
/* Synchronises `stream` on `device` and returns the sticky status of the device-side
 * checks accumulated since the last call; clears it.  Two sticky words are kept per stream: the zero-divisor flag of the
 * compute kernels and the flags of the point checks; the point checks win (infinity, then not-on-curve, then not-in-subgroup),
 * whatever ran in between: a pairing on an invalid point that also divides by zero still reports the invalid point. */
int bn254_last_status(int device, void* stream);
/* Bytes of device scratch a call over n lanes (k pairs each) will use (informational; never touches the HIP runtime:
 * the persistent grid is min(work items, CUs), CUs = those of a device the library already runs on, else 256). */
size_t bn254_scratch_bytes(size_t n, size_t k);
/* Sizes everything the library keeps for (device, stream) -- scratch for k pairs per lane, the status word, the verdict's
 * Fq12 buffer, the sub-group buffers of groups of more than 64 pairs, the pow_native digit buffer, four pinned staging
 * slots and the device copies of the latency path's round programs (thirty-five programs, 28 MB, once per device; without bn254_reserve the first
 * small call of each kind uploads its program with a blocking copy) -- for calls of up to n lanes (units) x k pairs.  Afterwards `_dev` calls of that size or smaller neither allocate
 * nor wait.  Does not wait for the stream either (buffers that must grow are retired, see STREAM). */
int bn254_reserve(int device, void* stream, size_t n, size_t k);
/* LATENCY PATH.  The reference's functions are scalar: pairing(p, q) (src/pairing.rs:20-22), miller_loop_native(q, p)
 * (miller_loop_native.rs:320-322), multi_miller_loop_native(pairs) (:324-326), final_exp_native(a) (final_exp_native.rs:209-213).
 * The throughput kernels put one item on one lane: 3.6 M dependent instructions for a pairing, 6.5 ms however small the batch.
 * Behind the same entry points -- bn254_pairing_batch, bn254_miller_loop_batch, bn254_final_exp_batch, bn254_multi_pairing_batch
 * with k <= 4 pairs (both values of do_final_exp), bn254_multi_pairing_check_batch, their `_dev` and `_elems` forms -- sits a second,
 * lane-cooperative kernel: one item on sixteen lanes, four items per wave (pairing: 0.49 M instructions deep, 1.01 ms -- 0.53 ms on
 * more lanes, below; a four-pair product check 0.74 ms instead of 13.4), the same values bit for bit.  Batches of at most `n` items take it, per function scaled by
 * its measured crossover against the throughput kernel (round 5: x1.5 pairing, miller_loop_native, final_exp_native and the products of two and three
 * pairings, x1.75 the product of four, x1.75 / x1 / x1.5 the exact values of two / three / four pairs: with the default, pairing() up to 24 576 items -- above 4 096 as seven launches of
 * programs small enough for eight waves per CU: the Miller loop, then the final exponentiation in six pieces through per-stream buffers); 0 turns it off.  Process-wide DEFAULT
 * (a stream may carry its own: bn254_set_stream_latency); default 16384. */
void bn254_set_latency_threshold(size_t n);
size_t bn254_get_latency_threshold(void);
/* The lane-cooperative programs exist for sixteen lanes per item (four items per wave; every function), for thirty-two (two items per
 * wave: fewer, fuller rounds -- pairing 0.70 ms instead of 1.01 -- for twice the lanes) and for sixty-four
 * (one item per wave: pairing 0.53 ms -- the accumulator's chain runs as single products there, the sums in the combinations that follow;
 * the lines of a step of three or four pairs are multiplied with each other off the accumulator's chain).
 * 0 (default): sixty-four / thirty-two while the launch is at most one wave per SIMD (1024 / 2048 items on MI355X), sixteen beyond;
 * 16 / 32 / 64: that family whatever the size (measurements, tests; a function without a program of the family takes the next
 * smaller one). */
void bn254_set_latency_lanes(int lanes);
int bn254_get_latency_lanes(void);
/* The two settings above are process-wide DEFAULTS.  A (device, stream) may carry its own: calls on that stream then select their
 * kernel by `threshold` / `lanes` whatever other threads set for theirs (two host threads with different needs -- one serving scalar
 * calls, one timing the throughput kernel -- use two streams).  threshold = BN254_LATENCY_INHERIT / lanes = -1 return the stream to
 * the process-wide default.  Never touches the HIP runtime; the setting lives until bn254_release_stream. */
#define BN254_LATENCY_INHERIT ((size_t)-1)
int bn254_set_stream_latency(int device, void* stream, size_t threshold, int lanes);
/* Which kernel the most recent pairing / Miller / final-exponentiation launch on (device, stream) took: 1 = the throughput kernel (one
 * item per lane), 16 / 32 / 64 = the lane-cooperative kernel with that many lanes per item, 0 = none yet (diagnostic; no HIP call). */
int bn254_last_kernel(int device, void* stream);
/* Scratch and the status word are kept per (device, stream), so calls on different streams are independent;
 * this frees what the library holds for `stream` (call it before destroying a stream you used). */
int bn254_release_stream(int device, void* stream);

/* PAGE-LOCKED HOST MEMORY.  The reference's callers hold their values in host memory (`pairing(p: G1Affine, q: G2Affine)`, src/pairing.rs:20;
 * `Vec<(&G1Affine, &G2Affine)>`, miller_loop_native.rs:324), so a binding's batch calls go through the host-pointer entry points.  From pageable
 * memory every byte passes the HIP runtime's staging buffers (a host memcpy on the issuing thread); from page-locked memory the copy engines move it
 * at the link rate underneath the kernels, and the chunked pipeline of the large-batch calls (bn254_pairing_batch, bn254_multi_pairing_batch, their
 * `_elems` and `_sharded` forms above 65 536 units) then delivers the resident-data rate minus the first chunk's copy-in and the last chunk's copy-out.
 * The entry points detect page-locked buffers themselves (hipPointerGetAttributes on all three arrays): nothing else changes for the caller.
 *   bn254_alloc_pinned / bn254_free_pinned: page-locked memory from the runtime (hipHostMalloc, portable across devices);
 *   bn254_host_register / bn254_host_unregister: page-lock memory the caller already owns, in place (hipHostRegister; a Rust `Vec`'s buffer,
 *     a numpy array) -- registering costs about as much as one pageable copy of the range, so do it once for buffers that are reused;
 *   bn254_host_is_pinned: 1 when [ptr, ptr + bytes) is page-locked memory the runtime knows, else 0 (diagnostic).
 * Memory the caller page-locked itself through HIP is recognised the same way. */
int bn254_alloc_pinned(size_t bytes, void** out);
int bn254_free_pinned(void* ptr);
int bn254_host_register(void* ptr, size_t bytes);
int bn254_host_unregister(void* ptr);
int bn254_host_is_pinned(const void* ptr, size_t bytes);

/* ---- the hot path -------------------------------------------------------------------- */

/* pairing(p, q) = final_exp_native(miller_loop_native(&q, &p))      src/pairing.rs:20-22
 * out: MyFq12 coefficient order (apply bn254_myfq12_to_ark_index for ark Fq12 order). */
int bn254_pairing_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int device, void* stream);
int bn254_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int device, void* stream);

/* miller_loop_native(Q, P)                                   src/miller_loop_native.rs:320-322
 * bit-exact un-normalised affine-line value of the reference. */
int bn254_miller_loop_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream);
int bn254_miller_loop_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream);

/* final_exp_native(a)                                        src/final_exp_native.rs:209-213
 * arbitrary non-zero Fq12 input (T4, final_exp_native.rs:274-285). */
int bn254_final_exp_batch_dev(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream);
int bn254_final_exp_batch(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream);

/* multi_miller_loop_native(pairs)                            src/miller_loop_native.rs:324-326
 * n_groups independent groups of k pairs (G1 then G2, the reference's argument order);
 * pair j of group g is element g*k + j of the g1/g2 batches (batch length n_groups*k).
 * do_final_exp != 0 applies final_exp_native to each group's shared-f Miller value
 * (the Groth16 shape).  Any k >= 1, as the reference's Vec: one kernel holds up to 64 pairs
 * per lane; larger groups are walked in sub-groups of 64 whose Miller values are multiplied
 * (MyFq12 Mul) -- the same field element, limb for limb (the reference's own test asserts
 * multi == product, :336-348). */
int bn254_multi_pairing_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k,
                                  int do_final_exp, int device, void* stream);
/* Groups of more than 64 pairs (multi_miller_loop_native takes any Vec): the kernels hold 64 pairs' state per lane, larger groups are composed of
 * sub-groups (the Miller value of a group is the product of the Miller values of any partition of its pairs: the same limbs).  A batch of MANY groups walks
 * every group on its own lane, sub-group after sub-group; a batch of FEWER than `max_groups` groups (default 65 536: a full grid) -- one aggregated check over
 * thousands of pairs -- spreads each group over k / C lanes of C pairs (C: a divisor of k up to 64), one launch of
 * the Miller kernel over all of them, a multiplication tree per group, the final exponentiation of n_groups values: one group of 131 072 pairs 7.5 ms
 * instead of minutes (2^20 pairs: 43 ms).  Groups of 5 .. 64 pairs take the same route when its estimated time (passes over the grid x the cost of a lane of C pairs, the tree,
 * the final exponentiation) beats one launch of the k-pair kernel, which on a partial grid is latency-bound: one group of 64 pairs 145 ms on one lane, 1 ms
 * spread; 10 000 groups of 64 pairs 34 ms instead of 146; 16 384 groups of 8 pairs 9 ms instead of 22.  C is the divisor of k with the smallest estimate.
 * 0: never. */
void bn254_set_wide_groups(size_t max_groups);
size_t bn254_get_wide_groups(void);
int bn254_multi_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k,
                              int do_final_exp, int device, void* stream);

/* The check the reference's tests apply to a product of pairings (final_exp_native.rs:245-263,
 * `final_exp_native(multi_miller_loop_native(pairs)) == MyFq12::one`), i.e. the Groth16 verifier shape:
 * verdict[g] = 1 iff group g's product of k pairings is exactly one, else 0.  One byte per group leaves the
 * device instead of 384. */
int bn254_multi_pairing_check_batch_dev(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k,
                                        int device, void* stream);
int bn254_multi_pairing_check_batch(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k,
                                    int device, void* stream);
/* The same check against a value the caller holds: verdict[g] = 1 iff group g's product equals `target` -- 48 words in HOST memory, MyFq12
 * coefficient order, the canonical limbs the pairing entry points return (it travels in the kernel's arguments: no upload, capturable); NULL =
 * MyFq12::one.  A Groth16 verifier keeps e(alpha, beta) with its verifying key and compares e(A, B) e(-L, gamma) e(-C, delta) with it: one pair
 * less per proof than the product == one form. */
int bn254_multi_pairing_check_target_batch_dev(const uint64_t* g1, const uint64_t* g2, const uint64_t* target, uint8_t* verdict, size_t n_groups,
                                               size_t k, int device, void* stream);

/* ---- one process, several GPUs: host pointers, contiguous slices of the batch per device
 * 0..n_devices-1, no exchange step; every device runs the same kernels on a private stream.  Independent
 * pairings (src/pairing.rs:20-22) or k-pair groups (src/miller_loop_native.rs:324-326, groups stay whole). */
int bn254_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int n_devices);
int bn254_multi_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k,
                                int do_final_exp, int n_devices);

/* Device-pointer form of the above: the whole batch (limb-major planes) is resident in HBM of devices[0]
 * (devices == NULL: devices 0..n_devices-1) and `stream` is a stream of that device behind which the inputs are
 * valid.  Shard i -- contiguous slice i of the units -- runs on devices[i]; its slices move plane by plane with
 * hipMemcpyPeerAsync (copy engines over xGMI: no CU on either side, so transfers run under the other shards'
 * kernels) and the results land in `out` on devices[0].  A device may be named more than once (its shards take
 * turns).  Returns when all results are in `out` (synchronous), with the first failure of any shard. */
int bn254_pairing_sharded_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, const int* devices, int n_devices,
                              void* stream);
int bn254_multi_pairing_sharded_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                    const int* devices, int n_devices, void* stream);

/* ---- batched public helpers of the reference ------------------------------------------ */

/* MyFq12 `Mul` (plonky2-bn254 fields::native::MyFq12; call sites miller_loop_native.rs:153,239,345) */
int bn254_fq12_mul_batch_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int device, void* stream);
int bn254_fq12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int device, void* stream);
/* frobenius_map_native(a, power)                             src/final_exp_native.rs:17-54 */
int bn254_frobenius_map_batch_dev(const uint64_t* a, size_t power, uint64_t* out, size_t n, int device, void* stream);
int bn254_frobenius_map_batch(const uint64_t* a, size_t power, uint64_t* out, size_t n, int device, void* stream);
/* pow_native(a, exp)                                         src/final_exp_native.rs:56-84
 * exp: exp_limbs u64 limbs, least significant first (the reference's Vec<u64>), shared by the batch. */
int bn254_pow_batch_dev(const uint64_t* a, const uint64_t* exp, size_t exp_limbs, uint64_t* out, size_t n, int device, void* stream);
int bn254_pow_batch(const uint64_t* a, const uint64_t* exp, size_t exp_limbs, uint64_t* out, size_t n, int device, void* stream);
/* get_naf(exp)                                               src/final_exp_native.rs:86-128
 * host-side; naf must hold 64*exp_limbs + 1 entries; returns the length or BN254_ERR_NAF_CARRY. */
long bn254_get_naf(const uint64_t* exp, size_t exp_limbs, int8_t* naf);
/* frob_coeffs(index) -> Fq2 as 8 u64 (c0 limbs, c1 limbs)     src/final_exp_native.rs:183-192
 * index 0..11 (the only values frobenius_map_native uses, :22). */
int bn254_frob_coeffs(size_t index, uint64_t* out8);
/* SIX_U_PLUS_2_NAF (miller_loop_native.rs:314-318) and BN_X (final_exp_native.rs:15).
 * bn254_miller_loop_batch / bn254_multi_pairing_batch(do_final_exp = 0) walk this table digit for digit -- their values are the
 * reference's.  The entry points whose Miller value goes straight into the final exponentiation (bn254_pairing_batch,
 * bn254_multi_pairing_batch(do_final_exp = 1), bn254_multi_pairing_check_batch) walk a minimal-weight signed binary form of the same
 * number 6x + 2 instead (65 digits, 22 non-zero: the same doublings, four additions less): the chain changes the Miller value by
 * factors from proper subfields only, which the final exponentiation removes -- their results are the reference's limb for limb. */
const int8_t* bn254_six_u_plus_2_naf(void); /* 65 entries */
uint64_t bn254_bn_x(void);
/* index map MyFq12 -> ark Fq12 flat order (`.into()` at src/pairing.rs:21):
 * ark_flat[j] = myfq12.coeffs[bn254_myfq12_to_ark_index(j)], j = 0..11 */
int bn254_myfq12_to_ark_index(int j);

/* ---- element-major data ("elems") ------------------------------------------------------
 * The reference's callers hold `G1Affine` / `G2Affine` / `MyFq12` / `Fq12` values one after the other
 * (pairing.rs:20 takes them by value, miller_loop_native.rs:324 as `Vec<(&G1Affine, &G2Affine)>`):
 *   elems[i*W + w], W = 8 (G1: x, y), 16 (G2: x.c0, x.c1, y.c0, y.c1), 48 (Fq12), 4 limbs per Fq as above.
 * These entry points take / return that order, so a binding copies fields sequentially and never transposes on
 * the host; the limb-major planes are made on the device (one HBM pass).  For W = 48, `fq12_order` selects the
 * coefficient order of the element-major side:
 *   BN254_FQ12_MYFQ12  MyFq12.coeffs[0..12]
 *   BN254_FQ12_ARK     ark Fq12 (c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1): MyFq12 `From`/`Into<Fq12>`,
 *                      the `.into()` of src/pairing.rs:21 -- what pairing() returns. */
#define BN254_FQ12_MYFQ12 0
#define BN254_FQ12_ARK 1
/* device pointers; src != dst; words = 8, 16 or 48 (fq12_order is ignored unless words = 48) */
int bn254_soa_from_elems_dev(const uint64_t* elems, uint64_t* soa, size_t words, size_t n, int fq12_order, int device, void* stream);
int bn254_soa_to_elems_dev(const uint64_t* soa, uint64_t* elems, size_t words, size_t n, int fq12_order, int device, void* stream);
/* host pointers, element-major in and out: pairing() (src/pairing.rs:20-22), miller_loop_native (:320),
 * multi_miller_loop_native (:324; pair j of group g = element g*k + j) and final_exp_native (final_exp_native.rs:209) */
int bn254_pairing_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int device, void* stream);
/* ... on DEVICE-resident element-major arrays (another device library's `G1Affine` / `G2Affine` arrays, a `Vec<Fq12>` to be filled): the
 * throughput kernels read and write element-major arrays themselves (a launch, nothing else); batches small enough for the lane-cooperative
 * programs, which read planes, pass the transposition kernels and per-stream staging buffers (allocated on first use). */
int bn254_pairing_batch_elems_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int device, void* stream);
int bn254_multi_pairing_batch_elems_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                        int out_order, int device, void* stream);
int bn254_miller_loop_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream);
int bn254_multi_pairing_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                    int out_order, int device, void* stream);
int bn254_final_exp_batch_elems(const uint64_t* f_in, uint64_t* out, size_t n, int in_order, int out_order, int device, void* stream);
/* one process, several GPUs (as bn254_pairing_sharded / bn254_multi_pairing_sharded), element-major host arrays */
int bn254_pairing_sharded_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int n_devices);
int bn254_multi_pairing_sharded_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                      int out_order, int n_devices);
/* the product-of-pairings check (final_exp_native.rs:245-263) on element-major pairs: one byte per group */
int bn254_multi_pairing_check_batch_elems(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device,
                                          void* stream);

/* ---- fixed G2 points --------------------------------------------------------------------
 * multi_miller_loop_native (src/miller_loop_native.rs:192-282, :324-326) takes any pairs; its real consumer, a Groth16 verifier, calls it with
 * three of its four G2 points -- beta, gamma, delta of the verifying key -- THE SAME for every proof.  The point steps of such a pair do not depend on
 * the proof at all.  bn254_g2_lines_dev walks them once per fixed point and leaves every step's line coefficients in `table` (device memory,
 * bn254_g2_lines_bytes(k_fixed) bytes, in the engine's internal limb form: opaque, valid for the library build that made it); the batch calls then compute,
 * per group g,
 *     final_exp_native(multi_miller_loop_native([(P[g][0], Q0[g]), (P[g][1], Qfix_1), ..., (P[g][k], Qfix_k)]))          (k = k_fixed <= 4)
 * with the fixed pairs reduced to one scaling of a table line by (Px, Py) and one sparse multiplication per step (every table line is kept with the constant
 * coefficient ONE, which makes that multiplication six two-product passes): 5.3 M instructions per group of 1 + 3 pairs against 7.11 M for four free pairs
 * (2^18 groups: 38.8 ms against 52.2 on one MI355X), and no per-lane point state but the group's own.  The same limbs as
 * bn254_multi_pairing_batch_dev(do_final_exp = 1) on the expanded pairs (the value after the final exponentiation does not see how the Miller value was
 * reached).  g1: n x (1 + k_fixed) G1 points, group-major like every multi-pairing batch (limb-major planes of n (1 + k) points; or element-major for
 * the `_elems` form); g2_var: n G2 points; out: n Fq12.  The `_check` form gives the `== MyFq12::one` verdict byte per group instead
 * (final_exp_native.rs:245-263).  The G2 points of the table must be in the r-torsion like any other (bn254_check_points_ex).  One launch:
 * n (1 + k_fixed) <= 2^23 points per call.  g2_var == NULL (every form below): the groups have NO pair of their own -- g1 holds k_fixed points per group,
 * every G2 point is one of the table's (a KZG / PLONK opening check e(P_1, [tau] G2) e(P_2, G2): two pairings for 3.45 M instructions, what ONE costs with a
 * free G2 point; 2^18 such checks: 25.4 ms = 10.3 M/s against 34.8 ms for two free pairs; k_fixed = 1, pairing(P_i, Q) with one Q for the whole batch:
 * 12.6 M pairings/s).  Small batches (below the latency threshold, bn254_set_latency_threshold; k_fixed <= 3) are a fraction of one
 * grid of that kernel (8 ms whatever n is): the table carries the fixed points behind its lines, and such a call expands the pairs and runs the
 * lane-cooperative k-pair program instead (one group of 1 + 3 pairs: 0.8 ms; the same limbs; its buffers are allocated on first use, or by
 * bn254_reserve(device, stream, n, 1 + k_fixed): the call is then a sequence of plain launches, capturable like the others). */
size_t bn254_g2_lines_bytes(size_t k_fixed);
int bn254_g2_lines_dev(const uint64_t* g2_fixed, size_t k_fixed, uint64_t* table, int device, void* stream);
int bn254_pairing_fixed_g2_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint64_t* out, size_t n, int device,
                                     void* stream);
int bn254_pairing_fixed_g2_batch_elems_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint64_t* out, size_t n,
                                           int out_order, int device, void* stream);
int bn254_pairing_fixed_g2_check_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint8_t* verdict, size_t n,
                                           int device, void* stream);
/* ... against `target` (48 host words, NULL = one; see bn254_multi_pairing_check_target_batch_dev): with gamma, delta fixed and e(alpha, beta) as the
 * target a Groth16 proof costs 1 + 2 pairs, 4.66 M instructions instead of 5.3 M (2^18 proofs: 34.3 ms). */
int bn254_pairing_fixed_g2_check_target_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, const uint64_t* target,
                                                  uint8_t* verdict, size_t n, int device, void* stream);
/* host-pointer forms (what a binding uses): g2_fixed = the k_fixed fixed points themselves; the table is made inside the call (2.1 ms) -- unless one of the
 * stream's last four distinct sets of fixed points is the same (a verifier's keys do not change between its calls): their tables are kept.  One proof's
 * check (1 + 2 pairs against a target) from host structs, verdict read back: 0.71 ms; 65 536: 9.2 ms.
 * `_elems`: every array element-major.  More than 65 536 groups go through the two-worker chunked pipeline of the other host-pointer calls (copies under
 * the kernels; any n with n (1 + k_fixed) < 2^29); the pipeline keeps the table of its last call and makes none when the fixed points are the same
 * again.  2^18 Groth16 checks (1 + 2 pairs against a target) from host structs: 36.4 ms = 7.2 M proofs/s, copies included (resident: 34.4 ms). */
int bn254_pairing_fixed_g2_batch(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, uint64_t* out, size_t n, int device,
                                 void* stream);
int bn254_pairing_fixed_g2_batch_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, uint64_t* out, size_t n,
                                       int out_order, int device, void* stream);
/* a Groth16 verifier's whole pairing check on host structs (element-major as above): verdict[g] = 1 iff the group's product equals `target` (48 host words,
 * MyFq12 order; NULL = MyFq12::one); one byte per group comes back instead of 384 */
int bn254_pairing_fixed_g2_check_batch_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, const uint64_t* target,
                                             uint8_t* verdict, size_t n, int device, void* stream);
/* ... spread over the first n_devices GPUs of this process (contiguous slices of the groups, the two-worker pipeline and its own line table on each): what
 * bn254_pairing_sharded_elems is to bn254_pairing_batch_elems */
int bn254_pairing_fixed_g2_check_sharded_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, const uint64_t* target,
                                               uint8_t* verdict, size_t n, int n_devices);

/* ---- input validation (optional) ----------------------------------------------------------
 * The reference never checks for the point at infinity: its line functions read raw x / y (src/miller_loop_native.rs:10-44) and
 * ignore `G1Affine::infinity` / `G2Affine::infinity`, so an infinite input is outside its contract and outside the hot path's.
 * A caller that wants a DISTINCT status for it runs this check on the same limb-major batches: a pair whose G1 or G2
 * coordinates are all zero (ark's affine identity) makes the `_dev` form set the stream's sticky status -- the next
 * bn254_last_status returns BN254_ERR_INFINITY -- and the host-pointer form return BN254_ERR_INFINITY.  One HBM pass over the
 * inputs; nothing is computed from them.  (The Rust shim tests the `infinity` flag of the structs it is handed instead.) */
int bn254_check_points_dev(const uint64_t* g1, const uint64_t* g2, size_t n, int device, void* stream);
int bn254_check_points(const uint64_t* g1, const uint64_t* g2, size_t n, int device, void* stream);
/* The rest of the reference's IMPLICIT input contract.  ark-ec's `Affine::new` asserts on-curve and in-subgroup, and the reference
 * builds the Frobenius images of Q with it (`G2Affine::new(out_x, out_y)`, src/miller_loop_native.rs:303,311): a G2 point outside the
 * r-torsion makes the reference PANIC at the end of its Miller loop, and `G1Affine::rand` / `G2Affine::rand` (src/pairing.rs:65-66)
 * never produce one.  The pairing kernels do not look -- they return a value for any coordinates (and not the value a from-scratch
 * affine computation would give for off-curve input: their projective steps use the curve equation).  A caller with untrusted points
 * runs this check first; `flags` selects
 *   BN254_CHECK_INFINITY   all-zero G1 or G2 coordinates (as bn254_check_points)                          -> BN254_ERR_INFINITY
 *   BN254_CHECK_ON_CURVE   y^2 = x^3 + 3 (G1), y^2 = x^3 + 3/(9+u) (G2), every coordinate below p           -> BN254_ERR_NOT_ON_CURVE
 *   BN254_CHECK_SUBGROUP   on-curve, and Q in the r-torsion: [x+1]Q + psi([x]Q) + psi^2([x]Q) == psi^3([2x]Q)
 *                          (ePrint 2022/348; psi = the reference's twisted_frobenius; G1 has cofactor one) -> BN254_ERR_NOT_IN_SUBGROUP
 * `per_point` (may be NULL): one byte per pair, the OR of 2 (infinity), 4 (not on curve), 8 (not in subgroup) -- which pair it was.
 * The `_dev` form sets the stream's sticky point-check status (reported by the next bn254_last_status in the order above, whatever
 * ran in between); the host-pointer form returns it.  COST: not free and not part of the hot path -- the subgroup check is a 63-bit
 * scalar multiplication on the twist per pair (63 doublings, 24 mixed and 2 general additions in Jacobian coordinates: about 0.75 k Fq2
 * products, an eighth of a pairing's field work).  It runs on a generated gfx950 kernel like the pairings (k_subcheck: 0.50 M instructions
 * per point); infinity and on-curve stay plain HIP C++ (HBM-bound).  Measured per 2^20 resident pairs on MI355X: 0.07 ms (infinity) /
 * 0.29 ms (+ on-curve) / 14.4 ms (+ subgroup: 72.8 M pairs/s, 0.14 of the pairings' own time; the portable C++ form of the same criterion: 54.5 ms).  HBM: the 192 input bytes per pair
 * twice (+ 5 bytes of verdicts).
 *   BN254_CHECK_SUBGROUP_PORTABLE   the subgroup check on the compiler-scheduled HIP C++ kernel instead (csrc/bn254_point_checks.h: the same
 *                          criterion with every exceptional case of the group law spelled out) -- the generated kernel's cross-check */
#define BN254_CHECK_INFINITY 1
#define BN254_CHECK_ON_CURVE 2
#define BN254_CHECK_SUBGROUP 4
#define BN254_CHECK_SUBGROUP_PORTABLE 8
int bn254_check_points_ex_dev(const uint64_t* g1, const uint64_t* g2, size_t n, int flags, uint8_t* per_point, int device, void* stream);
int bn254_check_points_ex(const uint64_t* g1, const uint64_t* g2, size_t n, int flags, uint8_t* per_point, int device, void* stream);

/* ---- synthetic inputs (bench / tests): on-device subgroup points ------------------------ */
/* P_i = [s_i] G1, Q_i = [t_i] G2 with s_i, t_i from SplitMix64(seed, i) (non-zero, < 2^128);
 * affine, Montgomery, SoA.  Device pointers. */
int bn254_generate_pairs_dev(uint64_t seed, uint64_t* g1_out, uint64_t* g2_out, size_t n, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BN254_PAIRING_H */
