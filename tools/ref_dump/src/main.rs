//! ref_dump: runs the REFERENCE (halo2_proofs + pairing_bn256 at the pinned revisions) on small fixed inputs and prints
//! one JSON object with everything the C oracle / the HIP build had to assume about byte-level conventions:
//!
//!   * constants: Fr::ZETA, Fr::DELTA, Fr::root_of_unity(), Fr::ROOT_OF_UNITY_INV, S, R = Fr::one() memory image
//!   * memory images (raw bytes of the in-memory structs) of an Fr, a G1Affine, a G1 -- the layout behind the
//!     transmutes of arithmetic.rs:351-352,364-365 (size_of is dumped too: a 72-byte G1Affine would show here)
//!   * to_bytes() of the generator, 2G, -G and the identity (compressed point flags, poly/commitment.rs:241-294)
//!   * from_bytes_wide of a fixed 64-byte input (transcript.rs:282-291)
//!   * best_fft of x_i = (i + 1)^3 + 7 for log_n = 3 and 10; EvaluationDomain constants for (j = 3, k = 4)
//!   * coeff_to_extended / extended_to_coeff of the same vector (the coset convention, poly/domain.rs:270-350)
//!   * best_multiexp of s_i = (i + 2)^5, P_i = [i + 1]G for n = 8 and 300
//!   * the ordering used by find_max_scalar_bits (plonk/prover.rs:252-254): max of {1, r - 1, 2^200}
//!
//! Inputs are closed-form so the consumer (tests/test_ref_pinning.py) regenerates them without any shared PRNG.
//! Scalars are printed as canonical little-endian hex (`to_repr`) AND, where layout matters, as raw memory bytes.
use ff::{Field, PrimeField};
use group::{Curve, Group};
use halo2_proofs::arithmetic::{best_fft, best_multiexp, FieldExt};
use halo2_proofs::poly::EvaluationDomain;
use pairing::bn256::{Fr, G1Affine, G1};

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{:02x}", b)).collect()
}
fn raw<T>(v: &T) -> String {
    let p = v as *const T as *const u8;
    hex(unsafe { std::slice::from_raw_parts(p, std::mem::size_of::<T>()) })
}
fn repr(v: &Fr) -> String {
    hex(v.to_repr().as_ref())
}
fn list(v: &[Fr]) -> String {
    format!("[{}]", v.iter().map(|x| format!("\"{}\"", repr(x))).collect::<Vec<_>>().join(","))
}
fn input(n: usize) -> Vec<Fr> {
    (0..n).map(|i| Fr::from(i as u64 + 1).pow_vartime(&[3, 0, 0, 0]) + Fr::from(7)).collect()
}

fn main() {
    let mut out: Vec<String> = vec![];
    let mut put = |k: &str, v: String| out.push(format!("\"{}\": {}", k, v));
    let q = |s: String| format!("\"{}\"", s);

    // ---- constants
    put("S", format!("{}", Fr::S));
    put("zeta", q(repr(&Fr::ZETA)));
    put("delta", q(repr(&Fr::DELTA)));
    put("root_of_unity", q(repr(&Fr::root_of_unity())));
    put("root_of_unity_inv", q(repr(&Fr::ROOT_OF_UNITY_INV)));
    put("multiplicative_generator", q(repr(&Fr::multiplicative_generator())));

    // ---- memory images
    let g = G1Affine::generator();
    let gp = G1::generator();
    let two_g = (gp + gp).to_affine();
    let neg_g = -g;
    let id = G1Affine::identity();
    put("sizeof", format!("{{\"Fr\": {}, \"G1Affine\": {}, \"G1\": {}}}", std::mem::size_of::<Fr>(),
                          std::mem::size_of::<G1Affine>(), std::mem::size_of::<G1>()));
    put("mem_fr_one", q(raw(&Fr::one())));
    put("mem_fr_seven", q(raw(&Fr::from(7))));
    put("mem_g1affine_generator", q(raw(&g)));
    put("mem_g1affine_two_g", q(raw(&two_g)));
    put("mem_g1affine_identity", q(raw(&id)));
    put("mem_g1_generator", q(raw(&gp)));
    put("mem_g1_identity", q(raw(&G1::identity())));
    put("mem_g1_two_g_projective", q(raw(&(gp + gp))));

    // ---- encodings
    put("bytes_generator", q(hex(g.to_bytes().as_ref())));
    put("bytes_two_g", q(hex(two_g.to_bytes().as_ref())));
    put("bytes_neg_g", q(hex(neg_g.to_bytes().as_ref())));
    put("bytes_identity", q(hex(id.to_bytes().as_ref())));
    let mut wide = [0u8; 64];
    for (i, b) in wide.iter_mut().enumerate() {
        *b = (i as u8).wrapping_mul(37).wrapping_add(11);
    }
    put("from_bytes_wide_input", q(hex(&wide)));
    put("from_bytes_wide", q(repr(&Fr::from_bytes_wide(&wide))));

    // ---- best_fft (arithmetic.rs:546-705)
    for log_n in [3u32, 10u32] {
        let n = 1usize << log_n;
        let mut a = input(n);
        let omega = Fr::root_of_unity().pow_vartime(&[1u64 << (Fr::S - log_n), 0, 0, 0]);
        put(&format!("fft_{}_omega", log_n), q(repr(&omega)));
        best_fft(&mut a, omega, log_n);
        put(&format!("fft_{}_output", log_n), list(&a));
    }

    // ---- EvaluationDomain (poly/domain.rs:44-149) and the coset transforms
    let (j, k) = (3u32, 4u32);
    let dom = EvaluationDomain::<Fr>::new(j, k);
    put("domain_j3_k4_extended_k", format!("{}", dom.extended_k()));
    put("domain_j3_k4_omega", q(repr(&dom.get_omega())));
    put("domain_j3_k4_extended_omega", q(repr(&dom.get_extended_omega())));
    let mut poly = dom.empty_coeff();
    for (c, v) in poly.iter_mut().zip(input(1 << k)) {
        *c = v;
    }
    let ext = dom.coeff_to_extended(poly.clone());
    put("coset_j3_k4_extended", list(&ext[..]));
    let back = dom.extended_to_coeff(ext);
    put("coset_j3_k4_back", list(&back[..]));
    let lag = dom.coeff_to_lagrange(poly.clone());
    put("lagrange_j3_k4", list(&lag[..]));

    // ---- best_multiexp (arithmetic.rs:465-492)
    for n in [8usize, 300usize] {
        let scalars: Vec<Fr> = (0..n).map(|i| Fr::from(i as u64 + 2).pow_vartime(&[5, 0, 0, 0])).collect();
        let mut acc = G1::identity();
        let bases: Vec<G1Affine> = (0..n)
            .map(|_| {
                acc = acc + gp;
                acc.to_affine()
            })
            .collect();
        let r = best_multiexp(&scalars, &bases).to_affine();
        put(&format!("msm_{}_bytes", n), q(hex(r.to_bytes().as_ref())));
        put(&format!("msm_{}_mem", n), q(raw(&r)));
    }

    // ---- Ord on Fr as used by find_max_scalar_bits (plonk/prover.rs:252-254)
    let big = Fr::from(2).pow_vartime(&[200, 0, 0, 0]);
    let cands = [Fr::one(), -Fr::one(), big];
    let mx = cands.iter().fold(Fr::zero(), |a, b| if a < *b { *b } else { a });
    put("max_of_one_minus_one_2p200", q(repr(&mx)));

    // ---- verifying-key digest framing (plonk.rs:91-109): Blake2b-512 "Halo2-Verify-Key" over len || text
    let text = "ref_dump";
    let mut h = blake2b_simd::Params::new().hash_length(64).personal(b"Halo2-Verify-Key").to_state();
    h.update(&(text.len() as u64).to_le_bytes());
    h.update(text.as_bytes());
    put("vk_framing_digest", q(repr(&Fr::from_bytes_wide(h.finalize().as_array()))));

    println!("{{\n  {}\n}}", out.join(",\n  "));
}
