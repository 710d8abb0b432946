"""The Rust side of the drop-in as verifiable source (VERDICT r1 #8 / missing #4): integration/hip.rs (the `extern "C"`
shim) and integration/halo2_proofs_hip.patch (the `hip` cargo feature re-targeting the reference's call sites).
No Rust toolchain exists here, so "verifiable" means: the patch applies cleanly to a scratch copy of the reference
(`git apply --check`), every `extern "C"` name in hip.rs is an exported symbol of libhalo2_hip.so with the header's
argument count, and the patch is what tools/make_hip_patch.py generates from the reference today."""
import os
import re
import shutil
import subprocess

import pytest

from h2util import ROOT

REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "halo2_proofs_hip.patch")
HIP_RS = os.path.join(ROOT, "integration", "hip.rs")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "halo2_proofs")), reason="the reference tree is not on this machine")


def _files_in_patch():
    with open(PATCH) as f:
        return sorted({l[6:].strip() for l in f if l.startswith("--- a/")})


@needs_ref
def test_patch_applies_to_the_reference(tmp_path):
    for rel in _files_in_patch():
        src = os.path.join(REF, rel)
        if os.path.exists(src):
            dst = tmp_path / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copy(src, dst)
    res = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    res = subprocess.run(["git", "apply", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    # the new module is the committed shim, byte for byte, and the feature is wired
    assert (tmp_path / "halo2_proofs/src/hip.rs").read_text() == open(HIP_RS).read()
    assert "hip = []" in (tmp_path / "halo2_proofs/Cargo.toml").read_text()
    arith = (tmp_path / "halo2_proofs/src/arithmetic.rs").read_text()
    for fn in ("gpu_multiexp_single_gpu_with_bound", "gpu_multiexp_bound", "gpu_multiexp_bound_and_fft", "gpu_fft", "gpu_ifft"):
        assert re.search(r'#\[cfg\(feature = "hip"\)\]\npub fn %s<' % fn, arith), fn
    assert arith.count('any(feature = "cuda", feature = "hip")') == 2          # both dispatchers


@needs_ref
def test_patch_is_what_the_generator_emits(tmp_path):
    want = open(PATCH).read()
    res = subprocess.run(["python3", os.path.join(ROOT, "tools", "make_hip_patch.py"), REF], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(PATCH).read() == want, "integration/halo2_proofs_hip.patch is stale: re-run tools/make_hip_patch.py"


def test_shim_declares_only_exported_symbols_with_the_header_arity():
    """every `pub fn h2_*` of hip.rs exists in include/halo2_hip.h with the same number of parameters and is exported by
    the built library"""
    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    header = open(os.path.join(ROOT, "include", "halo2_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rs = open(HIP_RS).read()
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    block = re.sub(r"//[^\n]*", "", block)
    decls = re.findall(r"pub fn (h2_\w+)\s*\((.*?)\)\s*(?:->\s*[\w\s\*]+)?;", block, flags=re.S)
    assert len(decls) >= 19
    for name, params in decls:
        assert hasattr(L, name), name
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, header, flags=re.S)
        assert m, "not in the header: " + name
        c_params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        rs_params = [p for p in params.split(",") if p.strip()]
        assert len(c_params) == len(rs_params), (name, c_params, rs_params)
