#!/usr/bin/env python3
"""tools/make_hip_patch.py -- builds integration/halo2_proofs_hip.patch: the `hip` cargo feature of halo2_proofs.

Applies a fixed list of textual edits to a scratch copy of the reference crate and writes the unified diff with ONE line
of context, plus integration/hip.rs as the new file halo2_proofs/src/hip.rs and integration/evaluation_hip.rs as
halo2_proofs/src/plonk/evaluation_hip.rs.  Re-targeted (every row of SURVEY.md 8(b)):
  * arithmetic.rs: best_fft, gpu_ifft, gpu_multiexp*, gpu_multiexp_bound_and_fft; commitment.rs: commit_lagrange_and_ifft,
    commit_lagrange_with_bound, Params::{unsafe_setup, read} -> register_params (+ Drop);
  * poly/domain.rs: ifft / lagrange_to_coeff_st, coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly;
  * plonk/evaluation.rs: Evaluator::evaluate_h (the cuda signature: coefficient forms), evaluate / evaluate_with_theta;
  * poly/multiopen/gwc/prover.rs: the batching loop;
  * the struct shapes and data flow that `cuda` switches (plonk.rs:226-240, keygen.rs, permutation{,/keygen,/prover}.rs,
    prover.rs: no extended cosets in the proving key / per-proof state) are switched by `hip` too: `any(cuda, hip)`;
  * plonk/prover.rs:56-74 N_GPU default = the library's device pool.
tests/test_integration_patch.py checks that the committed patch still applies (`git apply --check`) to a fresh copy.

    python tools/make_hip_patch.py [/root/reference]
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["halo2_proofs/Cargo.toml", "halo2_proofs/build.rs", "halo2_proofs/src/lib.rs", "halo2_proofs/src/arithmetic.rs",
         "halo2_proofs/src/poly/commitment.rs", "halo2_proofs/src/poly/domain.rs", "halo2_proofs/src/plonk.rs",
         "halo2_proofs/src/plonk/keygen.rs", "halo2_proofs/src/plonk/permutation.rs",
         "halo2_proofs/src/plonk/permutation/keygen.rs", "halo2_proofs/src/plonk/permutation/prover.rs",
         "halo2_proofs/src/plonk/prover.rs", "halo2_proofs/src/plonk/evaluation.rs",
         "halo2_proofs/src/poly/multiopen/gwc/prover.rs"]
NEW_FILES = {"halo2_proofs/src/hip.rs": "hip.rs", "halo2_proofs/src/plonk/evaluation_hip.rs": "evaluation_hip.rs",
             "halo2_proofs/src/hip_resident.rs": "hip_resident.rs"}

PERMUTATION_HIP = '''        #[cfg(feature = "hip")]
        {
            let mut last_z = C::Scalar::one();
            for (i, (columns, permutations)) in self
                .columns
                .chunks(chunk_len)
                .zip(pkey.permutations.chunks(chunk_len))
                .enumerate()
            {
                let values: Vec<&[C::Scalar]> = columns
                    .iter()
                    .map(|&column| {
                        let values = match column.column_type() {
                            Any::Advice => advice,
                            Any::Fixed => fixed,
                            Any::Instance => instance,
                        };
                        &values[column.index()][..]
                    })
                    .collect();
                let sigmas: Vec<&[C::Scalar]> = permutations.iter().map(|p| &p[..]).collect();
                let z = crate::hip::permutation_product(
                    &values,
                    &sigmas,
                    &*beta,
                    &*gamma,
                    &C::Scalar::DELTA.pow(&[i as u64 * chunk_len as u64, 0, 0, 0]),
                    &C::Scalar::DELTA,
                    &domain.get_omega(),
                    &last_z,
                );
                let mut z = domain.lagrange_from_vec(z);
                // Set blinding factors
                for z in &mut z[params.n as usize - blinding_factors..] {
                    *z = C::Scalar::random(&mut rng);
                }
                // Set new last_z
                last_z = z[params.n as usize - (blinding_factors + 1)];
                sets.push(z);
            }
            return Ok(sets);
        }

'''

HIP_FUNCTIONS = '''#[cfg(feature = "hip")]
pub fn gpu_multiexp_single_gpu_with_bound<C: CurveAffine>(
    coeffs: &[C::Scalar],
    bases: &[C],
    max_bits: usize,
) -> C::Curve {
    if max_bits == 0 || coeffs.len() == 0 {
        C::Curve::identity()
    } else {
        crate::hip::msm(coeffs, bases, max_bits, false)
    }
}

#[cfg(feature = "hip")]
pub fn gpu_multiexp_bound<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C], max_bits: usize) -> C::Curve {
    if max_bits == 0 || coeffs.len() == 0 {
        C::Curve::identity()
    } else {
        // the N_GPU split and the host fold of the partial points live in the library (h2_msm_multi)
        crate::hip::msm(coeffs, bases, max_bits, true)
    }
}

#[cfg(feature = "hip")]
pub fn gpu_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
    gpu_multiexp_bound(coeffs, bases, C::Scalar::NUM_BITS as usize)
}

#[cfg(feature = "hip")]
pub fn gpu_multiexp_bound_and_fft<C: CurveAffine>(
    coeffs: &mut [C::Scalar],
    bases: &[C],
    max_bits: usize,
    omega: &C::Scalar,
    divisor: &C::Scalar,
    log_n: u32,
) -> C::Curve {
    crate::hip::msm_intt(coeffs, bases, max_bits, omega, divisor, log_n)
}

#[cfg(feature = "hip")]
pub fn gpu_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {
    crate::hip::ntt(a, &omega, log_n)
}

#[cfg(feature = "hip")]
pub fn gpu_ifft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32, divisor: G::Scalar) {
    crate::hip::intt(a, &omega, &divisor, log_n)
}

'''

BUILD_RS = '''    #[cfg(feature = "hip")]
    {
        // libhalo2_hip.so: `make -C halo2-gpu-specific_amd/csrc` (hipcc --offload-arch=gfx950)
        println!("cargo:rustc-link-search=native={}", std::env::var("HALO2_HIP_LIB_DIR").expect("HALO2_HIP_LIB_DIR"));
        println!("cargo:rustc-link-lib=dylib=halo2_hip");
    }
'''

ANY = 'any(feature = "cuda", feature = "hip")'


def replace_once(text, old, new, what):
    assert text.count(old) >= 1, "anchor not found: " + what
    return text.replace(old, new, 1)


def switch_shape(text, expect):
    """the struct-shape / data-flow switches: whatever `cuda` selects, `hip` selects too"""
    n = text.count('#[cfg(feature = "cuda")]') + text.count('#[cfg(not(feature = "cuda"))]')
    assert n == expect, "expected %d cuda cfg sites, found %d" % (expect, n)
    text = text.replace('#[cfg(feature = "cuda")]', '#[cfg(%s)]' % ANY)
    return text.replace('#[cfg(not(feature = "cuda"))]', '#[cfg(not(%s))]' % ANY)


N_GPU_CUDA = '''            #[cfg(feature = "cuda")]
            {
                ec_gpu_gen::rust_gpu_tools::Device::all().len().to_string()
            }
            #[cfg(not(feature = "cuda"))]
            {
                "1".to_owned()
            }
'''
N_GPU_HIP = '''            #[cfg(CUDA_ONLY)]
            {
                ec_gpu_gen::rust_gpu_tools::Device::all().len().to_string()
            }
            #[cfg(feature = "hip")]
            {
                // the library's device pool (HALO2_PROOFS_N_GPU is read there too)
                crate::hip::device_count().to_string()
            }
            #[cfg(not(CUDA_OR_HIP))]
            {
                "1".to_owned()
            }
'''

COEFF_TO_EXTENDED_OLD = '''        assert_eq!(a.values.len(), 1 << self.k);

        //let timer = start_timer!(|| format!("prepare {}", self.k));
        self.distribute_powers_zeta(&mut a.values, true);
        //end_timer!(timer);

        a.values.resize(self.extended_len(), G::group_zero());
        best_fft(&mut a.values, self.extended_omega, self.extended_k);

        Polynomial {
            values: a.values,
            _marker: PhantomData,
        }
'''
COEFF_TO_EXTENDED_NEW = '''        assert_eq!(a.values.len(), 1 << self.k);

        // zeta-power pre-scale, zero padding and the extended NTT: one fused device pass over the 2^k inputs
        #[cfg(feature = "hip")]
        {
            let values = crate::hip::coeff_to_extended(
                &a.values,
                self.k,
                self.extended_k,
                &self.g_coset,
                &self.g_coset_inv,
                &self.extended_omega,
            );
            a.values = values;
        }

        #[cfg(not(feature = "hip"))]
        {
            //let timer = start_timer!(|| format!("prepare {}", self.k));
            self.distribute_powers_zeta(&mut a.values, true);
            //end_timer!(timer);

            a.values.resize(self.extended_len(), G::group_zero());
            best_fft(&mut a.values, self.extended_omega, self.extended_k);
        }

        Polynomial {
            values: a.values,
            _marker: PhantomData,
        }
'''
EXTENDED_TO_COEFF_OLD = '''        assert_eq!(a.values.len(), self.extended_len());

        // Inverse FFT
        Self::ifft(
            &mut a.values,
            self.extended_omega_inv,
            self.extended_k,
            self.extended_ifft_divisor,
        );

        // Distribute powers to move from coset; opposite from the
        // transformation we performed earlier.
        self.distribute_powers_zeta(&mut a.values, false);

        // Truncate it to match the size of the quotient polynomial; the
        // evaluation domain might be slightly larger than necessary because
        // it always lies on a power-of-two boundary.
        a.values
            .truncate((&self.n * self.quotient_poly_degree) as usize);

        a.values
'''
EXTENDED_TO_COEFF_NEW = '''        assert_eq!(a.values.len(), self.extended_len());

        // inverse extended NTT with 1 / n_ext and the zeta^-1 powers folded into its last pass; only the
        // n * quotient_poly_degree coefficients the caller keeps come back over PCIe
        #[cfg(feature = "hip")]
        {
            a.values = crate::hip::extended_to_coeff(
                &a.values,
                (&self.n * self.quotient_poly_degree) as usize,
                self.extended_k,
                &self.g_coset,
                &self.g_coset_inv,
                &self.extended_omega_inv,
                &self.extended_ifft_divisor,
            );
        }

        #[cfg(not(feature = "hip"))]
        {
            // Inverse FFT
            Self::ifft(
                &mut a.values,
                self.extended_omega_inv,
                self.extended_k,
                self.extended_ifft_divisor,
            );

            // Distribute powers to move from coset; opposite from the
            // transformation we performed earlier.
            self.distribute_powers_zeta(&mut a.values, false);

            // Truncate it to match the size of the quotient polynomial; the
            // evaluation domain might be slightly larger than necessary because
            // it always lies on a power-of-two boundary.
            a.values
                .truncate((&self.n * self.quotient_poly_degree) as usize);
        }

        a.values
'''
DIVIDE_OLD = '''        // Divide to obtain the quotient polynomial in the coset evaluation
        // domain.
        parallelize(&mut a.values, |h, mut index| {
            for h in h {
                h.group_scale(&self.t_evaluations[index % self.t_evaluations.len()]);
                index += 1;
            }
        });
'''
DIVIDE_NEW = '''        // Divide to obtain the quotient polynomial in the coset evaluation
        // domain.
        #[cfg(feature = "hip")]
        crate::hip::divide_by_vanishing_poly(&mut a.values, &self.t_evaluations);

        #[cfg(not(feature = "hip"))]
        parallelize(&mut a.values, |h, mut index| {
            for h in h {
                h.group_scale(&self.t_evaluations[index % self.t_evaluations.len()]);
                index += 1;
            }
        });
'''

EVALUATE_HIP = '''    #[cfg(feature = "hip")]
    {
        return super::evaluation_hip::evaluate_lc(
            &[expression.clone()],
            size,
            rot_scale,
            fixed,
            advice,
            instance,
            _theta,
        );
    }

'''
EVALUATE_THETA_HIP = '''        #[cfg(feature = "hip")]
        {
            return super::evaluation_hip::evaluate_lc(expressions, size, rot_scale, fixed, advice, instance, theta);
        }

'''

GWC_HIP = '''            #[cfg(feature = "hip")]
            let poly_batch = {
                let queries = &commitment_at_a_point.queries;
                let m = queries.len();
                if m <= 4 {
                    let mut poly_batch = zero();
                    for query in queries.iter() {
                        assert_eq!(query.get_point(), z);

                        let poly = query.get_commitment().poly;
                        poly_batch = poly_batch * *v + poly;
                    }
                    poly_batch
                } else {
                    // poly_batch = sum_i v^(m-1-i) p_i: one upload per operand and one fused pass (the cuda branch below
                    // re-uploads every p_i for an eval_mul_c / eval_sum pair)
                    let mut coeffs = vec![C::Scalar::one(); m];
                    for i in (0..m - 1).rev() {
                        coeffs[i] = coeffs[i + 1] * *v;
                    }
                    let polys: Vec<&[C::Scalar]> = queries
                        .iter()
                        .map(|query| {
                            assert_eq!(query.get_point(), z);
                            &query.get_commitment().poly.values[..]
                        })
                        .collect();
                    Polynomial {
                        values: crate::hip::lincomb(&polys, &coeffs),
                        _marker: PhantomData,
                    }
                }
            };

'''

PARAMS_DROP = '''/// `hip`: the device copies (and shifted-base tables) of `g` / `g_lagrange` registered by `unsafe_setup` / `read`
/// go with the parameters.
#[cfg(feature = "hip")]
impl<C: CurveAffine> Drop for Params<C> {
    fn drop(&mut self) {
        crate::hip::unregister_params(&self.g, &self.g_lagrange);
    }
}

'''


PK_DROP = '''/// `hip`: the device copies of `fixed_polys` / `permutation.polys` / `l0` / `l_last` registered by `keygen_pk` go with the key.
#[cfg(feature = "hip")]
impl<C: CurveAffine> Drop for ProvingKey<C> {
    fn drop(&mut self) {
        crate::hip::unregister_proving_key(self);
    }
}

'''


def edit(rel, text):
    if rel.endswith("Cargo.toml"):
        return replace_once(text, 'cuda = ["ec-gpu-gen/cuda", "pairing/gpu"]\n',
                            'cuda = ["ec-gpu-gen/cuda", "pairing/gpu"]\nhip = []\n', "cuda feature line")
    if rel.endswith("build.rs"):
        return replace_once(text, "fn main() {\n", "fn main() {\n" + BUILD_RS, "build.rs main")
    if rel.endswith("lib.rs"):
        return replace_once(text, "pub mod arithmetic;\n", 'pub mod arithmetic;\n#[cfg(feature = "hip")]\npub mod hip;\n#[cfg(feature = "hip")]\npub mod hip_resident;\n', "mod arithmetic")
    if rel.endswith("arithmetic.rs"):
        text = replace_once(text, "pub fn best_multiexp_gpu_cond<", HIP_FUNCTIONS + "pub fn best_multiexp_gpu_cond<", "best_multiexp_gpu_cond")
        # the two dispatchers: best_multiexp_gpu_cond (:442-458) and best_fft (:546-554)
        text = replace_once(text, '                if #[cfg(feature = "cuda")] {\n                    gpu_multiexp(coeffs, bases)',
                            '                if #[cfg(%s)] {\n                    gpu_multiexp(coeffs, bases)' % ANY, "multiexp dispatch")
        text = replace_once(text, '        if #[cfg(feature = "cuda")]{\n            return gpu_fft(a, omega, log_n);',
                            '        if #[cfg(%s)]{\n            return gpu_fft(a, omega, log_n);' % ANY, "fft dispatch")
        return text
    if rel.endswith("commitment.rs"):
        # Params::unsafe_setup (:56-124) and Params::read (:256-294): the SRS is registered with the library once
        text = replace_once(text, """        let additional_data = Vec::from(s_g2.to_bytes().as_ref());
        Params {
            k,
            n,
            g,
            g_lagrange,
            additional_data,
        }
""", """        let additional_data = Vec::from(s_g2.to_bytes().as_ref());
        let params = Params {
            k,
            n,
            g,
            g_lagrange,
            additional_data,
        };
        #[cfg(feature = "hip")]
        crate::hip::register_params(&params.g, &params.g_lagrange);
        params
""", "unsafe_setup tail")
        text = replace_once(text, """        Ok(Params {
            k,
            n: n as u64,
            g,
            g_lagrange,
            additional_data,
        })
""", """        let params = Params {
            k,
            n: n as u64,
            g,
            g_lagrange,
            additional_data,
        };
        #[cfg(feature = "hip")]
        crate::hip::register_params(&params.g, &params.g_lagrange);
        Ok(params)
""", "Params::read tail")
        text = replace_once(text, "/// These are the verifier parameters for the polynomial commitment scheme.\n",
                            PARAMS_DROP + "/// These are the verifier parameters for the polynomial commitment scheme.\n", "Params Drop")
        text = replace_once(text, '    #[cfg(feature = "cuda")]\n    /// This commits to a polynomial using its evaluations over the $2^k$ size',
                            '    #[cfg(%s)]\n    /// This commits to a polynomial using its evaluations over the $2^k$ size' % ANY, "commit_lagrange_and_ifft gpu")
        text = replace_once(text, '    #[cfg(not(feature = "cuda"))]\n    /// This commits to a polynomial using its evaluations over the $2^k$ size',
                            '    #[cfg(not(%s))]\n    /// This commits to a polynomial using its evaluations over the $2^k$ size' % ANY, "commit_lagrange_and_ifft cpu")
        text = replace_once(text, '        #[cfg(feature = "cuda")]\n        let res =\n            crate::arithmetic::gpu_multiexp_single_gpu_with_bound',
                            '        #[cfg(%s)]\n        let res =\n            crate::arithmetic::gpu_multiexp_single_gpu_with_bound' % ANY, "with_bound gpu")
        text = replace_once(text, '        #[cfg(not(feature = "cuda"))]\n        let res = best_multiexp_gpu_cond(&scalars, &bases[..]);',
                            '        #[cfg(not(%s))]\n        let res = best_multiexp_gpu_cond(&scalars, &bases[..]);' % ANY, "with_bound cpu")
        return text
    if rel.endswith("domain.rs"):
        text = replace_once(text, '        #[cfg(feature = "cuda")]\n        // Perform inverse FFT to obtain the polynomial in coefficient form\n        crate::arithmetic::gpu_ifft',
                            '        #[cfg(%s)]\n        // Perform inverse FFT to obtain the polynomial in coefficient form\n        crate::arithmetic::gpu_ifft' % ANY, "lagrange_to_coeff_st gpu")
        text = replace_once(text, '        #[cfg(not(feature = "cuda"))]\n        Self::ifft_st(', '        #[cfg(not(%s))]\n        Self::ifft_st(' % ANY, "lagrange_to_coeff_st cpu")
        text = replace_once(text, '        #[cfg(not(feature = "cuda"))]\n        {\n            best_fft(a, omega_inv, log_n);',
                            '        #[cfg(not(%s))]\n        {\n            best_fft(a, omega_inv, log_n);' % ANY, "ifft cpu")
        text = replace_once(text, '        #[cfg(feature = "cuda")]\n        crate::arithmetic::gpu_ifft(a, omega_inv, log_n, divisor)',
                            '        #[cfg(%s)]\n        crate::arithmetic::gpu_ifft(a, omega_inv, log_n, divisor)' % ANY, "ifft gpu")
        text = replace_once(text, '    #[cfg(not(feature = "cuda"))]\n    fn ifft_st(', '    #[cfg(not(%s))]\n    fn ifft_st(' % ANY, "ifft_st")
        text = replace_once(text, COEFF_TO_EXTENDED_OLD, COEFF_TO_EXTENDED_NEW, "coeff_to_extended body")
        text = replace_once(text, EXTENDED_TO_COEFF_OLD, EXTENDED_TO_COEFF_NEW, "extended_to_coeff body")
        text = replace_once(text, DIVIDE_OLD, DIVIDE_NEW, "divide_by_vanishing_poly body")
        return text
    if rel.endswith("src/plonk.rs"):
        text = replace_once(text, "mod evaluation;\n", 'mod evaluation;\n#[cfg(feature = "hip")]\nmod evaluation_hip;\n', "mod evaluation")
        text = replace_once(text, "impl<C: CurveAffine> ProvingKey<C> {\n    /// Get the underlying [`VerifyingKey`].\n",
                            PK_DROP + "impl<C: CurveAffine> ProvingKey<C> {\n    /// Get the underlying [`VerifyingKey`].\n", "ProvingKey Drop")
        return switch_shape(text, 5)
    if rel.endswith("plonk/keygen.rs"):
        # keygen_pk (:442-455) and keygen_pk_from_info (:540-553): the key's coefficient forms are registered with the library once
        for head, what in (("    Ok(ProvingKey {\n        vk,\n", "keygen_pk tail"), ("    Ok(ProvingKey {\n        vk: vk.clone(),\n", "keygen_pk_from_info tail")):
            at = text.index(head)
            end = text.index("    })\n}\n", at)
            body = text[at + len("    Ok("):end] + "    };\n"
            text = (text[:at] + "    let pk = " + body + '    #[cfg(feature = "hip")]\n    crate::hip::register_proving_key(&pk);\n    Ok(pk)\n}\n'
                    + text[end + len("    })\n}\n"):])
            assert text.count("crate::hip::register_proving_key(&pk);") >= 1, what
        assert text.count("crate::hip::register_proving_key(&pk);") == 2
        return switch_shape(text, 8)
    if rel.endswith("plonk/permutation.rs"):
        return switch_shape(text, 1)
    if rel.endswith("permutation/keygen.rs"):
        return switch_shape(text, 2)
    if rel.endswith("permutation/prover.rs"):
        # the grand products of the permutation argument: one device call per set of columns instead of three rayon passes over n
        # rows, a batch inversion and a serial running product on the host (the sets stay sequential: each starts from the last)
        text = replace_once(text, "        let mut sets = vec![];\n\n        let raw_zs = self\n", "        let mut sets = vec![];\n\n" + PERMUTATION_HIP +
                            "        let raw_zs = self\n", "permutation commit")
        return switch_shape(text, 1)
    if rel.endswith("plonk/prover.rs"):
        text = replace_once(text, N_GPU_CUDA, N_GPU_HIP, "N_GPU default")
        # the advice columns' coefficient forms are final from here to the end of the proof (evaluator, evaluations at x, the
        # multiopen folds read them): registered for that long -- a guard, so that every way out of the function unregisters
        site = ("                advice_cosets,\n            }\n        })\n        .collect::<Vec<_>>();\n\n"
                "    #[cfg(feature = \"cuda\")]\n    let h_poly = pk.ev.evaluate_h(\n")
        guard = ('    #[cfg(feature = "hip")]\n'
                 "    let _hip_advice_polys = crate::hip::RegisteredPolys::new(\n"
                 "        advice.iter().flat_map(|a| a.advice_polys.iter().map(|p| &p.values[..])),\n"
                 "    );\n\n")
        assert text.count(site) == 2, "advice_polys sites of create_proof_ext / create_proof_from_witness"
        text = text.replace(site, site.replace('    #[cfg(feature = "cuda")]\n    let h_poly', guard + '    #[cfg(feature = "cuda")]\n    let h_poly'))
        text = switch_shape(text, 19 - 2)          # the two sites of the N_GPU block became cuda / hip / neither
        text = text.replace("#[cfg(CUDA_ONLY)]", '#[cfg(feature = "cuda")]')
        return text.replace("#[cfg(not(CUDA_OR_HIP))]", "#[cfg(not(%s))]" % ANY)
    if rel.endswith("plonk/evaluation.rs"):
        # the CPU evaluate_h leaves when either GPU feature supplies one (the hip body is plonk/evaluation_hip.rs)
        text = replace_once(text, '    #[cfg(not(feature = "cuda"))]\n    pub(in crate::plonk) fn evaluate_h(',
                            '    #[cfg(not(%s))]\n    pub(in crate::plonk) fn evaluate_h(' % ANY, "cpu evaluate_h")
        cpu_eval = '    #[cfg(not(feature = "cuda"))]\n    {\n        let mut values = vec![F::zero(); size];\n        let isize = size as i32;\n'
        text = replace_once(text, cpu_eval, EVALUATE_HIP + cpu_eval.replace('not(feature = "cuda")', "not(%s)" % ANY), "evaluate cpu block")
        cpu_theta = '        #[cfg(not(feature = "cuda"))]\n        {\n            let mut values = vec![F::zero(); size];\n            let isize = size as i32;\n'
        text = replace_once(text, cpu_theta, EVALUATE_THETA_HIP + cpu_theta.replace('not(feature = "cuda")', "not(%s)" % ANY), "evaluate_with_theta cpu block")
        return text
    if rel.endswith("gwc/prover.rs"):
        cpu = '            #[cfg(not(feature = "cuda"))]\n            let poly_batch = {\n'
        text = replace_once(text, cpu, GWC_HIP + cpu.replace('not(feature = "cuda")', "not(%s)" % ANY), "gwc cpu batch")
        return text
    raise AssertionError(rel)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    with tempfile.TemporaryDirectory() as tmp:
        for side in ("a", "b"):
            for rel in FILES:
                dst = os.path.join(tmp, side, rel)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copy(os.path.join(ref, rel), dst)
        for rel in FILES:
            path = os.path.join(tmp, "b", rel)
            with open(path) as f:
                text = f.read()
            with open(path, "w") as f:
                f.write(edit(rel, text))
        for rel, src in NEW_FILES.items():
            shutil.copy(os.path.join(ROOT, "integration", src), os.path.join(tmp, "b", rel))
        res = subprocess.run(["diff", "-U1", "-r", "-N", "a", "b"], cwd=tmp, capture_output=True, text=True)
        assert res.returncode == 1, res.stderr
        lines = [l for l in res.stdout.splitlines(keepends=True) if not l.startswith("diff -U1")]
        # drop the timestamps of the ---/+++ lines
        out = []
        for l in lines:
            if l.startswith("--- ") or l.startswith("+++ "):
                l = l.split("\t")[0] + "\n"
            out.append(l)
    with open(os.path.join(ROOT, "integration", "halo2_proofs_hip.patch"), "w") as f:
        f.write("".join(out))
    print("wrote integration/halo2_proofs_hip.patch (%d lines)" % len(out))


if __name__ == "__main__":
    main()
