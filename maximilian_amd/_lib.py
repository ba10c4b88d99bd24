"""ctypes binding of libmaxigpu.so (the C-ABI declared in include/maxigpu.h).

There is deliberately no fallback: if the HIP library has not been built, or no HIP
device is visible when a compute entry point is called, this raises.  Nothing in this
package imports or calls the CPU oracle under oracle/.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MXG_LIB") or os.path.join(_HERE, "libmaxigpu.so")  # (MXG_LIB: an A/B build of the same ABI, tools only)

_lib = None

# name -> (restype, argtypes); mirrors include/maxigpu.h one to one.
SIGNATURES = {
    "mxg_init": (c_int, [c_int]),
    "mxg_last_error": (c_char_p, []),
    "mxg_version": (c_char_p, []),
    "mxg_settings": (c_int, [c_size_t, c_size_t, c_size_t]),
    "mxg_sample_rate": (c_size_t, []),
    "mxg_malloc": (c_void_p, [c_size_t]),
    "mxg_free": (c_int, [c_void_p]),
    "mxg_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_memcpy_d2d_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_host_alloc": (c_void_p, [c_size_t]),
    "mxg_host_free": (c_int, [c_void_p]),
    "mxg_event_sync": (c_int, [c_void_p]),
    "mxg_event_query": (c_int, [c_void_p]),
    "mxg_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "mxg_host_render": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "mxg_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "mxg_stream_create": (c_void_p, []),
    "mxg_stream_destroy": (c_int, [c_void_p]),
    "mxg_stream_sync": (c_int, [c_void_p]),
    "mxg_sync": (c_int, []),
    "mxg_event_create": (c_void_p, []),
    "mxg_event_destroy": (c_int, [c_void_p]),
    "mxg_event_record": (c_int, [c_void_p, c_void_p]),
    "mxg_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "mxg_prof_enable": (c_int, [c_int]),
    "mxg_prof_reset": (c_int, []),
    "mxg_prof_count": (c_int, []),
    "mxg_prof_read": (c_int, [c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_size_t)]),
    "mxg_prof_overhead_ms": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "mxg_tune": (c_int, [c_char_p, c_int]),
    "mxg_last_async_error": (c_int, []),
    "mxg_osc_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_osc_render_pitch": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_osc_render_mix": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_osc_mix_groups": (c_size_t, [c_size_t]),
    "mxg_osc_render_mix_rows": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_mix_rows_sum": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxg_osc_tables_groups": (c_size_t, [c_size_t]),
    "mxg_osc_render_tables": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_osc_render_tables_ex": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p]),
    "mxg_filter_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_int, c_void_p,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_filter_render_coefs": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_filter_coeffs_host": (c_int, [c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxg_env_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_env_coeff_host": (c_double, [c_int, c_double]),
    "mxg_mtof_host": (c_double, [c_int]),
    "mxg_voice_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "mxg_voice_render_mix_rows": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_voice_render_mix": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_envgen_stages_host": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_envgen_render": (c_int, [c_size_t, c_size_t, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "mxg_filter2_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_svf_coeffs_host": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxg_biquad_coeffs_host": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_mix_stereo": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_mix_bus": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "mxg_osc_noise": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_sample_render_frompos": (c_int, [c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "mxg_sample_render_trig": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "mxg_delay_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mxg_sample_upload": (c_void_p, [c_void_p, c_size_t]),
    "mxg_sample_free": (c_int, [c_void_p]),
    "mxg_sample_render": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_fft_plan_create": (c_void_p, [c_int, c_int, c_int]),
    "mxg_fft_plan_destroy": (c_int, [c_void_p]),
    "mxg_fft_plan_bins": (c_int, [c_void_p]),
    "mxg_sample_load_wav": (c_void_p, [c_char_p, c_int, c_void_p, c_void_p]),
    "mxg_sample_save_wav": (c_int, [c_char_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mxg_ifft_plan_create": (c_void_p, [c_int, c_int, c_int]),
    "mxg_ifft_plan_destroy": (c_int, [c_void_p]),
    "mxg_ifft_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_ifft_batch_complex": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_convolve_create": (c_void_p, [c_void_p, c_size_t, c_double, c_int, c_int]),
    "mxg_convolve_destroy": (c_int, [c_void_p]),
    "mxg_convolve_frames": (c_int, [c_void_p]),
    "mxg_convolve_impulse": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mxg_convolve_play": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "mxg_convolve_reset": (c_int, [c_void_p]),
    "mxg_convolve_output": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "mxg_convolve_input": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mxg_sampler_freq_host": (c_int, [c_size_t, c_void_p, c_size_t, c_void_p]),
    "mxg_sampler_render": (c_int, [c_size_t, c_size_t, c_int, c_int, c_void_p, c_size_t] + [c_void_p] * 12),
    "mxg_fft_features": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mxg_fft_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "mxg_mfcc_plan_create": (c_void_p, [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, c_double, c_double]),
    "mxg_mfcc_plan_destroy": (c_int, [c_void_p]),
    "mxg_mfcc_plan_tables": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mxg_mfcc_plan_matrix_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_mfcc_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p,
                               c_int, c_void_p]),
    "mxg_fft_mfcc_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "mxg_grain_plan_create": (c_void_p, [c_int, c_double, c_int]),
    "mxg_grain_plan_destroy": (c_int, [c_void_p]),
    "mxg_grain_plan_window": (c_int, [c_void_p, c_void_p]),
    "mxg_granular_render": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "mxg_granular_render_mix": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "mxg_comm_unique_id": (c_int, [c_void_p]),
    "mxg_comm_create": (c_void_p, [c_void_p, c_int, c_int]),
    "mxg_comm_destroy": (c_int, [c_void_p]),
    "mxg_comm_rank": (c_int, [c_void_p]),
    "mxg_comm_size": (c_int, [c_void_p]),
    "mxg_comm_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mxg_mix_reduce": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_void_p]),
    "mxg_granular_retries": (c_int, []),
    "mxg_mixq_create": (c_void_p, [c_void_p, c_size_t, c_int, c_int]),
    "mxg_mixq_create_grouped": (c_void_p, [c_void_p, c_size_t, c_int, c_int, c_size_t]),
    "mxg_mixq_destroy": (c_int, [c_void_p]),
    "mxg_mixq_set_sink": (c_int, [c_void_p, c_void_p, c_size_t]),
    "mxg_mixq_slot": (c_void_p, [c_void_p, c_void_p]),
    "mxg_mixq_push": (c_int, [c_void_p, c_void_p]),
    "mxg_mixq_flush": (c_int, [c_void_p, c_void_p]),
    "mxg_mixq_result": (c_void_p, [c_void_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "mxg_mixq_release": (c_int, [c_void_p, c_void_p]),
    "mxg_i64_from_i32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mxg_i32_from_i64": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
}


# measurement only (include/maxicalib.h, libmaxicalib.so): the bandwidth probes of bench.py and tools/*_ceiling.py
CALIB_PATH = os.path.join(_HERE, "libmaxicalib.so")
CALIB_SIGNATURES = {
    "mxg_calib_fill": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "mxg_calib_fill_ex": (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mxg_calib_read_ex": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}
_calib = None


def calib():
    """Load libmaxicalib.so (after libmaxigpu.so, whose runtime it links against)."""
    global _calib
    if _calib is None:
        lib()
        C = ctypes.CDLL(CALIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in CALIB_SIGNATURES.items():
            fn = getattr(C, name)
            fn.restype = res
            fn.argtypes = args
        _calib = C
    return _calib


class MaxiGpuError(RuntimeError):
    pass


def lib():
    """Load libmaxigpu.so once (importing torch first, when present, so that both share one
    HIP runtime -- the library resolves libamdhip64.so.7 by SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaxiGpuError(
            "libmaxigpu.so is not built (%s). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C maximilian_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (shared HIP runtime; optional)
    except Exception:  # pragma: no cover
        pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status, what=""):
    if status < 0:
        msg = lib().mxg_last_error().decode("utf-8", "replace")
        raise MaxiGpuError("%s failed (%d): %s" % (what or "libmaxigpu call", status, msg))
    return status
