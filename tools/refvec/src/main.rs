//! refvec: the reference's native functions (qope/plonky2-bn254-pairing, src/{pairing,miller_loop_native,final_exp_native}.rs)
//! on the exact inputs of tests/golden/bn254_vectors.json.  Every field element is printed TWICE: as the canonical integer
//! (hex) and as ark's raw in-memory limbs `Fp.0.0` (4 x u64, Montgomery form) -- the second is what pins the conventions nothing
//! inside this repository can check: ark's Montgomery R = 2^256 limb format, the flat coefficient order of ark's Fq12 behind
//! `MyFq12 -> Fq12` (src/pairing.rs:21), and ark's G2 generator.
//! NOT compiled in the build image (no Rust there); see Cargo.toml for how to run it.
use ark_bn254::{Fq, Fq12, Fq2, G1Affine, G2Affine};
use ark_ec::AffineRepr;
use ark_ff::{BigInteger, PrimeField};
use num_bigint::BigUint;
use plonky2_bn254::fields::native::MyFq12;
use plonky2_bn254_pairing::final_exp_native::{final_exp_native, frob_coeffs, frobenius_map_native, get_naf, pow_native, BN_X};
use plonky2_bn254_pairing::miller_loop_native::{
    conjugate_fp2, miller_loop_native, multi_miller_loop_native, neg_conjugate_fp2, SIX_U_PLUS_2_NAF,
};
use plonky2_bn254_pairing::pairing::pairing;
use serde_json::{json, Value};

fn fq_from_hex(s: &str) -> Fq {
    let s = s.trim_start_matches("0x");
    Fq::from(BigUint::parse_bytes(s.as_bytes(), 16).expect("hex"))
}
fn fq_out(x: &Fq) -> Value {
    let canon: BigUint = x.into_bigint().into();
    // `x.0` is the Montgomery representation, `(x.0).0` its four u64 limbs: the words the C ABI exchanges
    json!({ "int": format!("0x{:x}", canon), "mont_limbs": (x.0).0.iter().map(|w| format!("0x{:016x}", w)).collect::<Vec<_>>() })
}
fn fq2_out(x: &Fq2) -> Value { json!([fq_out(&x.c0), fq_out(&x.c1)]) }
fn myfq12_out(a: &MyFq12) -> Value { Value::Array(a.coeffs.iter().map(fq_out).collect()) }
/// ark's Fq12 in its flat order c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1 (what `bn254_myfq12_to_ark_index` claims)
fn arkfq12_out(f: &Fq12) -> Value {
    let mut v = Vec::new();
    for h in [&f.c0, &f.c1] { for k in [&h.c0, &h.c1, &h.c2] { v.push(fq_out(&k.c0)); v.push(fq_out(&k.c1)); } }
    Value::Array(v)
}
fn g1_in(v: &Value) -> G1Affine { G1Affine::new(fq_from_hex(v[0].as_str().unwrap()), fq_from_hex(v[1].as_str().unwrap())) }
fn g2_in(v: &Value) -> G2Affine {
    let f = |i: usize| fq_from_hex(v[i].as_str().unwrap());
    G2Affine::new(Fq2::new(f(0), f(1)), Fq2::new(f(2), f(3)))
}
fn myfq12_in(v: &Value) -> MyFq12 {
    let c: Vec<Fq> = v.as_array().unwrap().iter().map(|s| fq_from_hex(s.as_str().unwrap())).collect();
    MyFq12 { coeffs: c.try_into().unwrap() }
}

fn main() {
    let path = std::env::args().nth(1).expect("usage: refvec tests/golden/bn254_vectors.json");
    let vec: Value = serde_json::from_str(&std::fs::read_to_string(path).unwrap()).unwrap();
    let g1: Vec<G1Affine> = vec["g1"].as_array().unwrap().iter().map(g1_in).collect();
    let g2: Vec<G2Affine> = vec["g2"].as_array().unwrap().iter().map(g2_in).collect();
    let mut out = serde_json::Map::new();
    out.insert("source".into(), json!("qope/plonky2-bn254-pairing native functions on tests/golden/bn254_vectors.json (tools/refvec)"));
    // conventions: generators as ark holds them, the constants of the two source files
    out.insert("g1_generator".into(), json!([fq_out(&G1Affine::generator().x), fq_out(&G1Affine::generator().y)]));
    out.insert("g2_generator".into(), json!([fq2_out(&G2Affine::generator().x), fq2_out(&G2Affine::generator().y)]));
    out.insert("bn_x".into(), json!(BN_X));
    out.insert("six_u_plus_2_naf".into(), json!(SIX_U_PLUS_2_NAF.to_vec()));
    // miller_loop_native (miller_loop_native.rs:320), final_exp_native (final_exp_native.rs:209), pairing (pairing.rs:20)
    let mut miller = Vec::new(); let mut fexp = Vec::new(); let mut pair = Vec::new();
    for (p, q) in g1.iter().zip(g2.iter()) {
        let m = miller_loop_native(q, p);
        miller.push(myfq12_out(&m));
        fexp.push(myfq12_out(&final_exp_native(m)));
        pair.push(arkfq12_out(&pairing(*p, *q)));
    }
    out.insert("miller".into(), Value::Array(miller));
    out.insert("final_exp_of_miller".into(), Value::Array(fexp));
    out.insert("pairing_ark_order".into(), Value::Array(pair));
    // multi_miller_loop_native (:324) on the groups of the fixture (and T3's pairs)
    let mut groups = Vec::new();
    for g in vec["groups"].as_array().unwrap() {
        let idx: Vec<usize> = g["idx"].as_array().unwrap().iter().map(|i| i.as_u64().unwrap() as usize).collect();
        let pairs: Vec<(&G1Affine, &G2Affine)> = idx.iter().map(|&i| (&g1[i], &g2[i])).collect();
        let m = multi_miller_loop_native(pairs);
        groups.push(json!({ "idx": idx, "miller": myfq12_out(&m), "pairing": myfq12_out(&final_exp_native(m)) }));
    }
    out.insert("groups".into(), Value::Array(groups));
    let t3p: Vec<G1Affine> = vec["t3"]["g1"].as_array().unwrap().iter().map(g1_in).collect();
    let t3q: Vec<G2Affine> = vec["t3"]["g2"].as_array().unwrap().iter().map(g2_in).collect();
    let m = multi_miller_loop_native(t3p.iter().zip(t3q.iter()).collect());
    out.insert("t3".into(), json!({ "miller": myfq12_out(&m), "pairing": myfq12_out(&final_exp_native(m)) }));
    // final_exp_native / pow_native / frobenius_map_native / MyFq12 Mul on arbitrary Fq12 (T4, T7 shapes)
    let xs: Vec<MyFq12> = vec["fq12_in"].as_array().unwrap().iter().map(myfq12_in).collect();
    out.insert("final_exp".into(), Value::Array(xs.iter().map(|x| myfq12_out(&final_exp_native(*x))).collect()));
    out.insert("pow_x".into(), Value::Array(xs.iter().map(|x| myfq12_out(&pow_native(*x, vec![BN_X]))).collect()));
    let mut frob = serde_json::Map::new();
    for k in [0usize, 1, 2, 3, 6, 11, 13] {
        frob.insert(k.to_string(), Value::Array(xs.iter().map(|x| myfq12_out(&frobenius_map_native(*x, k))).collect()));
    }
    out.insert("frobenius".into(), Value::Object(frob));
    out.insert("fq12_mul".into(), Value::Array((0..xs.len()).map(|i| myfq12_out(&(xs[i] * xs[(i + 1) % xs.len()]))).collect()));
    // MyFq12 -> Fq12 (`.into()`, pairing.rs:21): the flat order of ark's tower for a value whose coefficients are 0..11
    let probe = MyFq12 { coeffs: (0u64..12).map(Fq::from).collect::<Vec<_>>().try_into().unwrap() };
    let probe_ark: Fq12 = probe.into();
    out.insert("myfq12_0_to_11_as_ark".into(), arkfq12_out(&probe_ark));
    // get_naf (:86), frob_coeffs (:183), conjugate_fp2 / neg_conjugate_fp2 (miller_loop_native.rs:284,291)
    out.insert("naf_bn_x".into(), json!(get_naf(vec![BN_X])));
    out.insert("naf_two_limbs".into(), json!(get_naf(vec![0xFFFFFFFFFFFFFFFFu64, 0x1234])));
    out.insert("frob_coeffs".into(), Value::Array((0..12).map(|k| fq2_out(&frob_coeffs(k))).collect()));
    let z = Fq2::new(Fq::from(5u64), Fq::from(7u64));
    out.insert("conjugate_fp2_5_7".into(), fq2_out(&conjugate_fp2(z)));
    out.insert("neg_conjugate_fp2_5_7".into(), fq2_out(&neg_conjugate_fp2(z)));
    println!("{}", serde_json::to_string_pretty(&Value::Object(out)).unwrap());
}
