#!/usr/bin/env python3
"""tools/make_hip_patch.py -- builds integration/halo2_proofs_hip.patch: the `hip` cargo feature of halo2_proofs.

Applies a fixed list of textual edits to a scratch copy of the reference crate (re-targeting the host-buffer boundary
of SURVEY.md 8(b): best_fft, gpu_ifft, gpu_multiexp*, commit_lagrange_and_ifft, commit_lagrange_with_bound) and writes
the unified diff with ONE line of context, plus integration/hip.rs as the new file halo2_proofs/src/hip.rs.
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
         "halo2_proofs/src/poly/commitment.rs", "halo2_proofs/src/poly/domain.rs"]

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


def edit(rel, text):
    if rel.endswith("Cargo.toml"):
        return replace_once(text, 'cuda = ["ec-gpu-gen/cuda", "pairing/gpu"]\n',
                            'cuda = ["ec-gpu-gen/cuda", "pairing/gpu"]\nhip = []\n', "cuda feature line")
    if rel.endswith("build.rs"):
        return replace_once(text, "fn main() {\n", "fn main() {\n" + BUILD_RS, "build.rs main")
    if rel.endswith("lib.rs"):
        return replace_once(text, "pub mod arithmetic;\n", 'pub mod arithmetic;\n#[cfg(feature = "hip")]\npub mod hip;\n', "mod arithmetic")
    if rel.endswith("arithmetic.rs"):
        text = replace_once(text, "pub fn best_multiexp_gpu_cond<", HIP_FUNCTIONS + "pub fn best_multiexp_gpu_cond<", "best_multiexp_gpu_cond")
        # the two dispatchers: best_multiexp_gpu_cond (:442-458) and best_fft (:546-554)
        text = replace_once(text, '                if #[cfg(feature = "cuda")] {\n                    gpu_multiexp(coeffs, bases)',
                            '                if #[cfg(%s)] {\n                    gpu_multiexp(coeffs, bases)' % ANY, "multiexp dispatch")
        text = replace_once(text, '        if #[cfg(feature = "cuda")]{\n            return gpu_fft(a, omega, log_n);',
                            '        if #[cfg(%s)]{\n            return gpu_fft(a, omega, log_n);' % ANY, "fft dispatch")
        return text
    if rel.endswith("commitment.rs"):
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
        shutil.copy(os.path.join(ROOT, "integration", "hip.rs"), os.path.join(tmp, "b", "halo2_proofs", "src", "hip.rs"))
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
