"""ctypes binding of librvip_hip.so (the C ABI declared in include/rvip_hip.h).

The product path has NO CPU fallback: ``lib()`` raises if the shared library is missing, and every
compute call raises ``RvipError`` on a non-zero return code.  Loading the library and resolving its
symbols needs no GPU (the CPU test-suite checks exactly that); launching needs one.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RVIP_LIB') or os.path.join(_HERE, 'librvip_hip.so')     # RVIP_LIB: A/B another build of the same ABI

EXPECTED_ABI = 8          # RVIP_ABI_VERSION of include/rvip_hip.h (tests/test_host_cpu.py holds the two together)
F32, BF16, F16 = 0, 1, 2
ACT = {None: 0, 'linear': 0, 'relu': 1, 'elu': 2, 'sigmoid': 3}
LOSS_MSE, LOSS_BCE_DICE = 0, 1
STATE_STEP, STATE_LR, STATE_SEED, STATE_WORDS = 0, 1, 2, 8
ERRORS = {-1: 'RVIP_EINVAL (bad shape / alignment / null pointer)', -2: 'RVIP_EUNSUPPORTED',
          -3: 'RVIP_EWORKSPACE (workspace too small)', -4: 'RVIP_ELAUNCH (HIP launch failed)'}

vp = C.c_void_p


class RvipError(RuntimeError):
    pass


class Conv3x3Desc(C.Structure):
    _fields_ = [('x0', vp), ('c0', C.c_int32), ('up0', C.c_int32),
                ('x1', vp), ('c1', C.c_int32),
                ('w_packed', vp), ('bias', vp),
                ('y', vp), ('y1', vp), ('csplit', C.c_int32),
                ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('cout', C.c_int32),
                ('act', C.c_int32), ('dtype', C.c_int32),
                ('depth', C.c_int32), ('kd', C.c_int32), ('down2', C.c_int32), ('subpix', C.c_int32), ('stream_in', C.c_int32),
                ('mask_bits', vp), ('mask_channels', C.c_int32), ('mask_scale', C.c_float), ('sign_bits', vp), ('sums_from', C.c_int32), ('cu_limit', C.c_int32)]


class PackEntry(C.Structure):
    _fields_ = [('w_off', C.c_longlong), ('f_off', C.c_longlong), ('d_off', C.c_longlong), ('cin', C.c_int32), ('cout', C.c_int32),
                ('taps', C.c_int32), ('mode', C.c_int32)]


class Wgrad3x3Desc(C.Structure):
    _fields_ = [('x0', vp), ('c0', C.c_int32), ('up0', C.c_int32),
                ('x1', vp), ('c1', C.c_int32),
                ('dy', vp), ('dw', vp),
                ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('cout', C.c_int32),
                ('dtype', C.c_int32),
                ('workspace', vp), ('workspace_bytes', C.c_size_t),
                ('depth', C.c_int32), ('kd', C.c_int32), ('defer_fold', C.c_int32),
                ('w_master', vp), ('dot_rows', vp), ('dot_rows_bytes', C.c_size_t), ('w_phase', vp), ('cu_limit', C.c_int32)]


class ApplyDesc(C.Structure):
    _fields_ = [('z', vp), ('y', vp), ('pooled', vp),
                ('scale', vp), ('shift', vp),
                ('act', C.c_int32),
                ('drop_rate', C.c_float), ('mask', vp), ('state', vp), ('layer_id', C.c_int32),
                ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('c', C.c_int32),
                ('dtype', C.c_int32),
                ('argmax', vp), ('keep_bits', vp)]


class BnBwdDesc(C.Structure):
    _fields_ = [('dy', vp), ('z', vp), ('dz', vp),
                ('gamma', vp), ('mean', vp), ('invstd', vp),
                ('scale', vp), ('shift', vp),
                ('dgamma', vp), ('dbeta', vp), ('dbias', vp),
                ('coef', vp),
                ('act', C.c_int32), ('act_after_bn', C.c_int32),
                ('drop_rate', C.c_float), ('mask', vp), ('state', vp), ('layer_id', C.c_int32),
                ('rows', C.c_longlong), ('c', C.c_int32),
                ('dtype', C.c_int32),
                ('workspace', vp), ('workspace_bytes', C.c_size_t),
                ('bias_rows', vp), ('bias_rows_bytes', C.c_size_t),
                ('dpooled', vp), ('argmax', vp), ('h', C.c_int32), ('w', C.c_int32)]


class BnCoefSrc(C.Structure):
    _fields_ = [('rows', vp), ('nrows', C.c_int32), ('stride', C.c_int32), ('offset', C.c_int32), ('reserved', C.c_int32)]


class BnCoefDesc(C.Structure):
    _fields_ = [('t1', BnCoefSrc * 2), ('t2', BnCoefSrc * 2),
                ('gamma', vp), ('beta', vp), ('mean', vp), ('invstd', vp),
                ('dgamma', vp), ('dbeta', vp), ('coef', vp),
                ('flags', vp), ('fallback', C.POINTER(BnBwdDesc)),
                ('count', C.c_longlong), ('c', C.c_int32),
                ('min_gamma', C.c_float), ('max_beta_ratio', C.c_float)]


class HeadCoefDesc(C.Structure):
    _fields_ = [('bn', C.POINTER(BnBwdDesc)), ('beta', vp),
                ('head_w', vp), ('dlogit', vp), ('k', C.c_int32), ('nrows', C.c_int32),
                ('mse_rows', vp), ('head_dw', vp), ('head_db', vp),
                ('sums', vp), ('loss_out', vp), ('inv_count', C.c_float),
                ('flags', vp), ('min_gamma', C.c_float), ('max_beta_ratio', C.c_float),
                ('loss_kind', C.c_int32), ('w_bce', C.c_float), ('w_dice', C.c_float), ('local_over_global', C.c_float), ('dscale', C.c_float),
                ('pred', vp), ('y_true', vp), ('dcoef', vp)]


class FoldEntry(C.Structure):
    _fields_ = [('src', vp), ('dst', vp), ('nrows', C.c_int32), ('stride', C.c_int32), ('width', C.c_longlong)]


# name -> (restype, argtypes); every symbol include/rvip_hip.h declares
SIGNATURES = {
    'rvip_abi_version': (C.c_int, []),
    'rvip_build_info': (C.c_char_p, []),
    'rvip_last_hip_error': (C.c_int, []),
    'rvip_device_check': (C.c_int, []),
    'rvip_conv3x3_fwd': (C.c_int, [C.POINTER(Conv3x3Desc), vp]),
    'rvip_conv3x3_sign_bits_ok': (C.c_int, [C.POINTER(Conv3x3Desc)]),
    'rvip_conv3x3_fwd_stats_rows': (C.c_int, [C.POINTER(Conv3x3Desc)]),
    'rvip_conv3x3_fwd_stats': (C.c_int, [C.POINTER(Conv3x3Desc), vp, C.c_size_t, vp]),
    'rvip_conv3x3_fwd_sums_rows': (C.c_int, [C.POINTER(Conv3x3Desc)]),
    'rvip_conv3x3_fwd_sums': (C.c_int, [C.POINTER(Conv3x3Desc), vp, C.c_size_t, vp]),
    'rvip_bn_stats_finalize': (C.c_int, [vp, C.c_int, C.c_longlong, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp, vp]),
    'rvip_pack_conv3x3_weights': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    'rvip_pack_subpixel_weights': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'rvip_pack_subpixel_dgrad_weights': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'rvip_pack_all_conv3x3_weights': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    'rvip_pack_table_check': (C.c_int, [vp, C.c_int, C.c_int]),
    'rvip_pack_all_conv3x3_weights_tick': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    'rvip_conv3x3_wgrad_workspace': (C.c_size_t, [C.c_int] * 5),
    'rvip_conv3x3_wgrad': (C.c_int, [C.POINTER(Wgrad3x3Desc), vp]),
    'rvip_conv3x3_wgrad_splits': (C.c_int, [C.POINTER(Wgrad3x3Desc)]),
    'rvip_conv3x3_wgrad_dot_rows': (C.c_int, [C.POINTER(Wgrad3x3Desc)]),
    'rvip_conv3x3_wgrad_form': (C.c_int, [C.POINTER(Wgrad3x3Desc)]),
    'rvip_conv3x3_wgrad_dgrad_ok': (C.c_int, [C.POINTER(Wgrad3x3Desc), C.POINTER(Conv3x3Desc)]),
    'rvip_conv3x3_wgrad_dgrad': (C.c_int, [C.POINTER(Wgrad3x3Desc), C.POINTER(Conv3x3Desc), vp, C.c_size_t, vp]),
    'rvip_fold_rows_batch': (C.c_int, [vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'rvip_bn_bwd_rows': (C.c_int, [C.c_longlong, C.c_int, C.c_int]),
    'rvip_bn_bwd_apply_head_rows': (C.c_int, [C.c_longlong, C.c_int, C.c_int, C.c_int]),
    'rvip_conv3x3_c1_fwd': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_conv3x3_c1_fwd_stats_rows': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'rvip_conv3x3_c1_fwd_stats': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_conv3x3_c1_wgrad': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_conv3x3_c1_wgrad_rows': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'rvip_conv3d_c1_fwd': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_conv3d_c1_wgrad': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_conv3x3_cn_fwd': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_conv3x3_cn_wgrad': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_reduce_workspace': (C.c_size_t, [C.c_longlong, C.c_int]),
    'rvip_bn_train_stats': (C.c_int, [vp, C.c_longlong, C.c_int, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int,
                                      vp, vp, vp, vp, vp, C.c_size_t, vp]),
    'rvip_bn_infer_coeffs': (C.c_int, [vp, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp]),
    'rvip_bn_apply': (C.c_int, [C.POINTER(ApplyDesc), vp]),
    'rvip_bn_apply_argmax_ok': (C.c_int, [C.c_int, C.c_int]),
    'rvip_bn_bwd_reduce': (C.c_int, [C.POINTER(BnBwdDesc), vp]),
    'rvip_bn_bwd_apply': (C.c_int, [C.POINTER(BnBwdDesc), vp]),
    'rvip_bn_bwd_coef': (C.c_int, [C.POINTER(BnCoefDesc), vp]),
    'rvip_maxpool2x2_bwd': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_subsample_odd': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_upsample2x_fwd': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_upsample2x_bwd': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rvip_head_fwd': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_longlong, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_bn_apply_head': (C.c_int, [C.POINTER(ApplyDesc), vp, vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp]),
    'rvip_bn_apply_head_mse_rows': (C.c_int, [C.c_longlong, C.c_int, C.c_int, C.c_int]),
    'rvip_bn_apply_head_mse': (C.c_int, [C.POINTER(ApplyDesc), vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float,
                                         vp, C.c_size_t, vp, C.c_size_t, vp]),
    'rvip_head_mse_coef': (C.c_int, [C.POINTER(HeadCoefDesc), vp]),
    'rvip_bn_apply_head_bcedice': (C.c_int, [C.POINTER(ApplyDesc), vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp]),
    'rvip_bn_bwd_apply_head_lazy': (C.c_int, [C.POINTER(BnBwdDesc), vp, vp, vp, vp, C.c_int, vp]),
    'rvip_bn_bwd_reduce_head': (C.c_int, [C.POINTER(BnBwdDesc), vp, vp, C.c_int, vp, vp, vp]),
    'rvip_bn_bwd_apply_head': (C.c_int, [C.POINTER(BnBwdDesc), vp, vp, C.c_int, vp]),
    'rvip_head_grad': (C.c_int, [vp, vp, vp, vp, vp, C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    'rvip_head_bwd': (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_longlong, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    'rvip_landmarks': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    'rvip_postprocess_workspace': (C.c_size_t, [C.c_int] * 4),
    'rvip_postprocess': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, C.c_size_t, vp]),
    'rvip_adam_step': (C.c_int, [vp, vp, vp, vp, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]),
    'rvip_state_tick': (C.c_int, [vp, vp]),
    'rvip_scale_f32': (C.c_int, [vp, C.c_longlong, C.c_float, vp]),
    'rvip_convert': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_longlong, vp]),
}

_lib = None


def lib():
    """The loaded library with argtypes set; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvipError('%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). '
                            'There is no CPU fallback for the product path.' % LIB_PATH)
        try:
            # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be
            # mapped BEFORE this library so that our NEEDED libamdhip64.so.7 binds to the same runtime; loaded the
            # other way round the process ends up with two HIP runtimes and launches fail with hipErrorNoDevice.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        got = L.rvip_abi_version()
        if got != EXPECTED_ABI:        # also for RVIP_LIB overrides: descriptors of another ABI would be misread silently
            raise RvipError('%s has ABI version %d, this package binds version %d: rebuild it (__graft_entry__.build())'
                            % (LIB_PATH, got, EXPECTED_ABI))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        L = lib()
        raise RvipError('%s failed: %s (hipError %d)' % (what, ERRORS.get(rc, rc), L.rvip_last_hip_error()))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
