"""ctypes binding of libgrafx_amd.so (the C ABI declared in include/grafx_amd.h).

Loading is strict: if the library is missing the import raises with the build
command — there is no Python/CPU fallback for the hot path.
"""
import ctypes
import os

from .build import LIB

i64, f32p, vp, sz = ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t


class RowMap(ctypes.Structure):
    _fields_ = [("inner", i64), ("stride_outer", i64), ("stride_inner", i64), ("stride_ch", i64)]


# name -> (restype, argtypes); must list every symbol of include/grafx_amd.h (tests/test_abi.py checks)
SIGNATURES = {
    "gfx_abi_version": (ctypes.c_int, []),
    "gfx_device_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(sz)]),
    "gfx_fftconv_nparts": (i64, [i64]),
    "gfx_fir_spectrum_bytes": (sz, [i64, i64]),
    "gfx_fftconv_workspace_bytes": (sz, [i64, i64, i64, i64, i64, i64]),
    "gfx_fir_spectrum_f32": (ctypes.c_int, [f32p, f32p, i64, vp, i64, i64, vp]),
    "gfx_fftconv_f32": (ctypes.c_int, [f32p, RowMap, vp, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp, sz, vp]),
    "gfx_fftconv_ex_f32": (ctypes.c_int, [f32p, RowMap, vp, i64, i64, f32p, RowMap, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp, sz, vp]),
    "gfx_fftconv_sched_f32": (ctypes.c_int, [f32p, RowMap, vp, i64, i64, f32p, RowMap, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp, sz, ctypes.c_int, vp]),
    "gfx_fftconv_rowmax_f32": (ctypes.c_int, [f32p, RowMap, vp, i64, i64, f32p, RowMap, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp, sz, vp, ctypes.POINTER(ctypes.c_int), vp]),
    "gfx_fftconv_last_kernel": (ctypes.c_char_p, []),
    "gfx_fir_direct_max_taps": (i64, []),
    "gfx_fir_direct_f32": (ctypes.c_int, [f32p, RowMap, f32p, i64, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp]),
    "gfx_fftconv_part_len": (i64, [i64, i64]),
    "gfx_fir_spectrum_bytes_ex": (sz, [i64, i64, i64]),
    "gfx_fftconv_workspace_bytes_ex": (sz, [i64, i64, i64, i64, i64, i64, i64]),
    "gfx_fir_spectrum_ex_f32": (ctypes.c_int, [f32p, f32p, i64, vp, i64, i64, i64, vp]),
    "gfx_fir_grad_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, i64, i64, i64, i64, i64, i64, i64, vp]),
    "gfx_fir_spectrum_rev_f32": (ctypes.c_int, [f32p, RowMap, i64, i64, i64, i64, vp, vp]),
    "gfx_fftconv_tee_f32": (ctypes.c_int, [f32p, RowMap, vp, f32p, RowMap, f32p, RowMap, i64, i64, i64, i64, i64, i64, i64, vp, sz, vp]),
    "gfx_odd_alias_plan_bytes": (sz, [i64]),
    "gfx_odd_alias_workspace_bytes": (sz, [i64, i64]),
    "gfx_odd_alias_plan_f32": (ctypes.c_int, [vp, i64, vp, sz, vp]),
    "gfx_odd_alias_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_rows_f32": (ctypes.c_int, [f32p, f32p, RowMap, i64, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_adjoint_f32": (ctypes.c_int, [f32p, i64, i64, i64, f32p, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_precise_plan_bytes": (sz, [i64]),
    "gfx_odd_alias_precise_workspace_bytes": (sz, [i64, i64]),
    "gfx_odd_alias_precise_plan_f32": (ctypes.c_int, [vp, i64, vp, sz, vp]),
    "gfx_odd_alias_precise_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_precise_adjoint_f32": (ctypes.c_int, [f32p, i64, i64, i64, f32p, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_pair_plan_bytes": (sz, [i64]),
    "gfx_odd_alias_pair_workspace_bytes": (sz, [i64, i64]),
    "gfx_odd_alias_pair_plan_f32": (ctypes.c_int, [vp, i64, vp, sz, vp]),
    "gfx_odd_alias_pair_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_pair_rows_f32": (ctypes.c_int, [f32p, f32p, RowMap, i64, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_odd_alias_pair_max_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp, vp]),
    "gfx_odd_alias_pair_precise_max_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp, ctypes.c_int, vp]),
    "gfx_onepole_energy_f32": (ctypes.c_int, [f32p, RowMap, i64, f32p, f32p, i64, i64, i64, i64, ctypes.c_int, vp, vp]),
    "gfx_odd_alias_pair_rows_max_f32": (ctypes.c_int, [f32p, f32p, RowMap, i64, i64, i64, i64, i64, i64, vp, vp, sz, vp, vp]),
    "gfx_odd_alias_pair_precise_plan_bytes": (sz, [i64]),
    "gfx_odd_alias_pair_precise_workspace_bytes": (sz, [i64, i64]),
    "gfx_odd_alias_pair_precise_plan_f32": (ctypes.c_int, [vp, i64, vp, sz, vp]),
    "gfx_odd_alias_pair_precise_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "gfx_irdft_f32": (ctypes.c_int, [f32p, ctypes.c_int, f32p, i64, i64, i64, i64, f32p, vp]),
    "gfx_rdft_f32": (ctypes.c_int, [f32p, f32p, i64, i64, i64, vp]),
    "gfx_iir_fsm_native": (ctypes.c_int, [i64]),
    "gfx_iir_fsm_plan_bytes": (sz, [i64]),
    "gfx_iir_fsm_plan_f32": (ctypes.c_int, [vp, i64, vp]),
    "gfx_iir_fsm_fir_f32": (ctypes.c_int, [f32p, f32p, vp, f32p, i64, i64, i64, vp]),
    "gfx_iir_fsm_fir_f64c_f32": (ctypes.c_int, [vp, vp, vp, f32p, i64, i64, i64, vp]),
    "gfx_iir_fsm_bwd_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, i64, vp]),
    "gfx_peq_coeffs_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, i64, i64, ctypes.c_int, vp]),
    "gfx_peq_coeffs_bwd_f32": (ctypes.c_int, [f32p] * 8 + [i64, i64, ctypes.c_int, vp]),
    "gfx_biquad_coeffs_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, f32p, i64, vp]),
    "gfx_dynamics_fused_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64,
                                              ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, vp]),
    "gfx_dynamics_fused_ex_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                 ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, vp]),
    "gfx_dynamics_fused_u1_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                 ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, f32p, vp]),
    "gfx_dynamics_bwd_u1_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                               ctypes.c_int, ctypes.c_int, f32p, RowMap, f32p, f32p, f32p, vp]),
    "gfx_dynamics_bwd_u1_ws_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                  ctypes.c_int, ctypes.c_int, f32p, RowMap, f32p, f32p, f32p, vp, sz, vp]),
    "gfx_dynamics_bwd_rescan_ws_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                      ctypes.c_int, ctypes.c_int, f32p, RowMap, f32p, f32p, f32p, vp, sz, vp]),
    "gfx_stft_f32": (ctypes.c_int, [f32p, f32p, f32p, i64, i64, i64, i64, vp]),
    "gfx_dynamics_ws_bytes": (sz, [i64]),
    "gfx_dynamics_ws_bytes_ex": (sz, [i64, i64, i64]),
    "gfx_dynamics_last_kernel": (ctypes.c_char_p, []),
    "gfx_dynamics_bwd_ws_bytes": (sz, [i64, i64]),
    "gfx_dynamics_fused_ws_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                 ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, f32p, vp, sz, vp]),
    "gfx_dynamics_fused_mix_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                  ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, f32p, vp, sz, vp, i64, i64,
                                                  f32p, i64, i64, i64, vp, i64, i64, vp]),
    "gfx_dynamics_fused_mix_flags_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                        ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, f32p, vp, sz, vp, i64, i64,
                                                        f32p, i64, i64, i64, vp, i64, i64, ctypes.c_int, vp]),
    "gfx_ballistics_bwd_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, vp]),
    "gfx_ballistics_bwd_ws_bytes": (sz, [i64, i64]),
    "gfx_ballistics_bwd_ws_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, vp, sz, vp]),
    "gfx_dyn_gain_bwd_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, ctypes.c_int,
                                            ctypes.c_int, f32p, f32p, f32p, vp]),
    "gfx_dyn_dx_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, i64, i64, i64, vp]),
    "gfx_dynamics_bwd_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                            ctypes.c_int, ctypes.c_int, f32p, RowMap, f32p, f32p, f32p, f32p, vp]),
    "gfx_onepole_dz_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, i64, i64, i64, vp]),
    "gfx_energy_f32": (ctypes.c_int, [f32p, RowMap, f32p, i64, i64, i64, vp]),
    "gfx_onepole_f32": (ctypes.c_int, [f32p, f32p, f32p, i64, i64, i64, i64, ctypes.c_int, vp]),
    "gfx_onepole_fir_f32": (ctypes.c_int, [f32p, f32p, i64, i64, vp]),
    "gfx_ballistics_f32": (ctypes.c_int, [f32p, f32p, f32p, i64, i64, vp]),
    "gfx_dyn_gain_apply_f32": (ctypes.c_int, [f32p, RowMap, f32p, f32p, RowMap, f32p, f32p, f32p, i64, i64, i64, i64,
                                              ctypes.c_int, ctypes.c_int, vp]),
    "gfx_dynamics_ballistics_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                   ctypes.c_int, ctypes.c_int, vp, sz, vp]),
    "gfx_ballistics_ws_bytes": (sz, [i64]),
    "gfx_ballistics_ws_f32": (ctypes.c_int, [f32p, f32p, ctypes.c_int, f32p, i64, i64, vp, sz, vp]),
    "gfx_ballistics_energy_f32": (ctypes.c_int, [f32p, RowMap, i64, f32p, ctypes.c_int, f32p, i64, i64, vp, sz, vp]),
    "gfx_dyn_gain_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "gfx_apply_gain_f32": (ctypes.c_int, [f32p, RowMap, f32p, f32p, RowMap, i64, i64, i64, ctypes.c_int, vp]),
    "gfx_stereo_gain_f32": (ctypes.c_int, [f32p, RowMap, f32p, f32p, RowMap, i64, i64, i64, vp]),
    "gfx_stereo_gain_mix_f32": (ctypes.c_int, [f32p, RowMap, f32p, f32p, RowMap, i64, i64, i64, vp, i64, i64, f32p, i64, i64,
                                               i64, vp, i64, i64, vp]),
    "gfx_biquad_cascade_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, f32p, f32p, i64, i64, i64, i64, i64, ctypes.c_int, vp]),
    "gfx_noise_shaping_ir_f32": (ctypes.c_int, [f32p, i64, f32p, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                ctypes.c_float, ctypes.c_float, vp]),
    "gfx_row_mean_f32": (ctypes.c_int, [f32p, RowMap, f32p, i64, i64, i64, vp]),
    "gfx_waveshaper_f32": (ctypes.c_int, [f32p, RowMap, f32p, RowMap, i64, i64, i64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, f32p, f32p, f32p, f32p, i64, f32p, vp]),
    "gfx_gather_sum_f32": (ctypes.c_int, [f32p, i64, i64, i64, vp, vp, f32p, i64, i64, i64, i64, i64, i64, i64, vp]),
    "gfx_gather_sum_fanout_f32": (ctypes.c_int, [f32p, i64, i64, i64, vp, vp, i64, f32p, i64, i64, i64, i64, i64, i64, i64, vp]),
    "gfx_istft_basis_bytes": (sz, [i64]),
    "gfx_istft_basis_f32": (ctypes.c_int, [f32p, f32p, i64, vp]),
    "gfx_stft_reverb_workspace_bytes": (sz, [i64, i64, i64]),
    "gfx_stft_reverb_ir_ex_f32": (ctypes.c_int, [f32p, i64, f32p, f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, i64, i64, i64,
                                                 ctypes.c_int, vp, sz, vp]),
    "gfx_stft_reverb_workspace_bytes_sched": (sz, [i64, i64, i64, i64, i64, ctypes.c_int]),
    "gfx_stft_reverb_ir_sched_f32": (ctypes.c_int, [f32p, i64, f32p, f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, i64, i64,
                                                    i64, ctypes.c_int, vp, sz, ctypes.c_int, vp]),
    "gfx_stft_reverb_ir_f32": (ctypes.c_int, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, i64, i64, i64, i64, i64,
                                              ctypes.c_int, vp, sz, vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError(
                f"{LIB} not found: the HIP extension is not built. Run `python -m grafx_amd.build` "
                "(or __graft_entry__.build()). grafx_amd has no CPU fallback for the processors."
            )
        handle = ctypes.CDLL(LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = ABI drift, fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


class GfxError(RuntimeError):
    pass


_CODES = {-1: "GFX_EINVAL (bad argument)", -2: "GFX_ENOSPC (workspace too small)", -3: "GFX_ELAUNCH (HIP launch failed)"}


def check(code, what):
    if code != 0:
        raise GfxError(f"{what} failed: {_CODES.get(code, code)}")
