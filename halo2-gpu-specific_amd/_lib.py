"""ctypes loader for libhalo2_hip.so.  Fails loudly when the extension is missing."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class H2Error(RuntimeError):
    pass


def lib_path():
    """the in-tree library; H2_LIB names another build of the same ABI (tools/gen_sanitize.sh: the generator's host code
    under ASan / UBSan) -- a library all the same: nothing is loaded in its place when the file is missing"""
    return os.environ.get("H2_LIB") or os.path.join(_HERE, "libhalo2_hip.so")


def build(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ -> libhalo2_hip.so (in-tree)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return lib_path()


# name -> (restype, argtypes); every symbol include/halo2_hip.h declares
_vp, _sz, _i32, _u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_uint32
_fp = ctypes.POINTER(ctypes.c_float)
SYMBOLS = {
    "h2_version": (ctypes.c_int, []),
    "h2_device_count": (ctypes.c_int, []),
    "h2_last_error": (ctypes.c_char_p, []),
    "h2_synchronize": (ctypes.c_int, []),
    "h2_set_device": (ctypes.c_int, [ctypes.c_int]),
    "h2_dev_alloc": (ctypes.c_int, [_sz, _vp]),
    "h2_dev_free": (ctypes.c_int, [_vp]),
    "h2_host_alloc_pinned": (ctypes.c_int, [_sz, _vp]),
    "h2_host_free_pinned": (ctypes.c_int, [_vp]),
    "h2_stream_create": (ctypes.c_int, [_vp]),
    "h2_stream_destroy": (ctypes.c_int, [_vp]),
    "h2_stream_synchronize": (ctypes.c_int, [_vp]),
    "h2_dev_upload": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "h2_dev_download": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "h2_release_plans": (ctypes.c_int, []),
    "h2_set_table_budget": (ctypes.c_int, [_sz]),
    "h2_library_memory_bytes": (_sz, []),
    "h2_ntt": (ctypes.c_int, [_vp, _vp, _u32]),
    "h2_intt": (ctypes.c_int, [_vp, _vp, _vp, _u32]),
    "h2_intt_to": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32]),
    "h2_coeff_to_extended": (ctypes.c_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "h2_extended_to_coeff": (ctypes.c_int, [_vp, _vp, _sz, _u32, _vp, _vp, _vp, _vp]),
    "h2_msm": (ctypes.c_int, [_vp, _vp, _sz, _u32, _vp]),
    "h2_msm_multi": (ctypes.c_int, [_vp, _vp, _sz, _u32, _vp]),
    "h2_bases_register": (ctypes.c_int, [_vp, _sz]),
    "h2_bases_unregister": (ctypes.c_int, [_vp]),
    "h2_poly_register": (ctypes.c_int, [_vp, _sz]),
    "h2_poly_unregister": (ctypes.c_int, [_vp]),
    "h2_g1_sum": (ctypes.c_int, [_vp, _sz, _vp]),
    "h2_msm_intt": (ctypes.c_int, [_vp, _vp, _sz, _u32, _vp, _vp, _u32, _vp]),
    "h2_batch_mont": (ctypes.c_int, [_vp, _sz]),
    "h2_batch_unmont": (ctypes.c_int, [_vp, _sz]),
    "h2_eval_op": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, _i32, _i32, _sz, _vp]),
    "h2_divide_by_vanishing_poly": (ctypes.c_int, [_vp, _sz, _vp, _sz]),
    "h2_dev_ntt": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp]),
    "h2_dev_coset_ntt": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "h2_dev_coset_intt": (ctypes.c_int, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "h2_dev_ntt_batch": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, _u32, _vp]),
    "h2_dev_intt_batch": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, _vp, _u32, _vp]),
    "h2_dev_coset_ntt_batch": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _u32, _vp, _vp, _vp]),
    "h2_dev_coeff_to_extended_batch": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "h2_dev_intt": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "h2_dev_coeff_to_extended": (ctypes.c_int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "h2_dev_extended_to_coeff": (ctypes.c_int, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "h2_msm_scratch_bytes": (_sz, [_sz, _u32]),
    "h2_msm_batch_scratch_bytes": (_sz, [_sz, _u32, _sz]),
    "h2_msm_shape": (ctypes.c_int, [_sz, _u32, _vp, _vp, _vp]),
    "h2_dev_msm": (ctypes.c_int, [_vp, _vp, _sz, _u32, _vp, _sz, _vp, _vp]),
    "h2_dev_msm_batch": (ctypes.c_int, [_vp, _sz, _vp, _sz, _u32, _vp, _sz, _vp, _vp]),
    "h2_dev_msm_batch_ex": (ctypes.c_int, [_vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "h2_dev_eval_op": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, _i32, _i32, _sz, _vp, _vp]),
    "h2_dev_divide_by_vanishing_poly": (ctypes.c_int, [_vp, _sz, _vp, _sz, _vp]),
    "h2_dev_batch_mont": (ctypes.c_int, [_vp, _sz, _vp]),
    "h2_dev_batch_unmont": (ctypes.c_int, [_vp, _sz, _vp]),
    "h2_dev_widen_u64": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_max_scalar_bits": (ctypes.c_int, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "h2_dev_random_points": (ctypes.c_int, [ctypes.c_uint64, _sz, _vp, _vp]),
    "h2_dev_bases_precompute": (ctypes.c_int, [_vp, _sz, ctypes.c_uint32, _vp]),
    "h2_dev_bases_forget": (ctypes.c_int, [_vp]),
    "h2_dev_bases_precompute_bytes": (_sz, [_sz, ctypes.c_uint32]),
    "h2_eval_polynomial": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_eval_polynomial": (ctypes.c_int, [_vp, _sz, _vp, _vp, _vp]),
    "h2_dev_eval_polynomial_batch": (ctypes.c_int, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "h2_batch_invert": (ctypes.c_int, [_vp, _sz]),
    "h2_dev_batch_invert": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "h2_kate_division": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_kate_division": (ctypes.c_int, [_vp, _sz, _vp, _vp, _vp]),
    "h2_prefix_product": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_prefix_product": (ctypes.c_int, [_vp, _sz, _vp, _vp, _vp]),
    "h2_dev_lincomb": (ctypes.c_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "h2_dev_permutation_sigma": (ctypes.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "h2_dev_permutation_terms": (ctypes.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "h2_permutation_terms": (ctypes.c_int, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "h2_permutation_product": (ctypes.c_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "h2_eval_polynomial_batch": (ctypes.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "h2_logup_grand_sum": (ctypes.c_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    "h2_prefix_sum": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_distribute_powers": (ctypes.c_int, [_vp, _sz, _vp]),
    "h2_permutation_sigma": (ctypes.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "h2_logup_multiplicity": (ctypes.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "h2_quotient_sum": (ctypes.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "h2_dev_random_fr": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_random_fr": (ctypes.c_int, [_vp, _sz, _vp]),
    "h2_dev_distribute_powers": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_g1_fold": (ctypes.c_int, [_vp, _u32, _u32, _vp, _vp]),
    "h2_dev_prefix_sum": (ctypes.c_int, [_vp, _sz, _vp, _vp, _vp]),
    "h2_logup_scratch_bytes": (_sz, [_sz]),
    "h2_dev_logup_multiplicity": (ctypes.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp]),
    "h2_dev_logup_multiplicity_bits": (ctypes.c_int, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp, _vp]),
    "h2_dev_logup_counts": (ctypes.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp, _vp, _sz, _vp]),
    "h2_dev_logup_emit": (ctypes.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "h2_dev_fixed_base_mul": (ctypes.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "h2_dev_points_decompress": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_dev_points_compress": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "h2_evalh_prepare": (ctypes.c_int, [_vp, _vp]),
    "h2_evalh_compile": (ctypes.c_int, [_vp, _vp]),
    "h2_evalh_source": (ctypes.c_int, [_vp, _u32, _vp, _sz, _vp]),
    "h2_evalh_stage_args": (ctypes.c_int, [_vp, _u32, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp, _sz, _vp]),
    "h2_evalh_generated_launches": (ctypes.c_uint64, []),
    "h2_evaluate_h": (ctypes.c_int, [_vp, _vp]),
    "h2_evaluate_h_coeff": (ctypes.c_int, [_vp, _vp]),
    "h2_quotient_poly_coeff": (ctypes.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz]),
    "h2_lincomb": (ctypes.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "h2_dev_evaluate_h": (ctypes.c_int, [_vp, _vp, _vp]),
    "h2_timer_start": (ctypes.c_int, [_vp]),
    "h2_timer_stop": (ctypes.c_int, [_vp, _fp]),
}


def lib():
    """The loaded C-ABI library.  Raises H2Error if libhalo2_hip.so has not been built:
    the product path never falls back to a CPU implementation."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise H2Error(
                "libhalo2_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C halo2-gpu-specific_amd/csrc`; there is no CPU fallback" % path
            )
        # torch ships its own HIP runtime; when both live in one process torch has to initialise first, then the
        # library binds to the same runtime instance (the other order leaves one of them without a device)
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().h2_last_error()
        raise H2Error("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else ""))
