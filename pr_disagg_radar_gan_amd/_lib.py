"""ctypes binding of librdgan_hip.so (C ABI: include/rdgan.h).

The product path has NO CPU fallback: if the library is missing or no MI355X is visible,
every compute entry point raises.  (The CPU oracle lives in /oracle and is test-only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librdgan_hip.so")

c_f32p = ctypes.c_void_p      # device pointers travel as integers
c_stream = ctypes.c_void_p

# every symbol include/rdgan.h declares: name -> (restype, argtypes)
class LaunchStat(ctypes.Structure):
    """rdgan_launch_stat (include/rdgan.h)"""
    _fields_ = [("plan", ctypes.c_int), ("kind", ctypes.c_int), ("batch", ctypes.c_int), ("launches", ctypes.c_int),
                ("gflop", ctypes.c_double), ("ms", ctypes.c_double), ("name", ctypes.c_char * 48), ("kernel", ctypes.c_char * 48)]


SIGNATURES = {
    "rdgan_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "rdgan_destroy": (None, [ctypes.c_void_p]),
    "rdgan_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "rdgan_gen_param_count": (ctypes.c_long, [ctypes.c_void_p]),
    "rdgan_critic_param_count": (ctypes.c_long, [ctypes.c_void_p]),
    "rdgan_workspace_bytes": (ctypes.c_long, [ctypes.c_void_p]),
    "rdgan_gen_forward": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_stream]),
    "rdgan_check_numerics": (ctypes.c_int, [ctypes.c_void_p, c_stream]),
    "rdgan_critic_forward": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int,
                                            ctypes.c_uint64, c_stream]),
    "rdgan_critic_grad": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_uint64,
                                         c_f32p, ctypes.c_int, c_stream]),
    "rdgan_gen_grad": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_uint64, c_f32p,
                                      ctypes.c_int, c_stream]),
    "rdgan_critic_grad_after": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_uint64,
                                               c_f32p, ctypes.c_int, ctypes.c_void_p, c_stream]),
    "rdgan_gen_grad_after": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_uint64, c_f32p,
                                            ctypes.c_int, ctypes.c_void_p, c_stream]),
    "rdgan_adam": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_float, c_stream]),
    "rdgan_set_weight_versions": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]),
    "rdgan_form_builds": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    "rdgan_gen_param_layout": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    "rdgan_critic_param_layout": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    "rdgan_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    "rdgan_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint]),
    "rdgan_flop_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "rdgan_profile_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_long)]),
    "rdgan_profile_launches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rdgan_launch_table": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(LaunchStat), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "rdgan_data_gather": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, ctypes.c_void_p, c_stream]),
    "rdgan_data_valid_tiles": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_float, ctypes.c_int, ctypes.c_void_p, c_stream]),
    "rdgan_crps_ensemble": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_long, c_stream]),
    "rdgan_op_conv3d": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 14 + [c_stream]),
    "rdgan_op_conv3d_bf16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 14 + [c_stream]),
    "rdgan_op_conv3d_dgrad": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 13 + [c_stream]),
    "rdgan_op_conv3d_wgrad_bf16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 13 + [c_stream]),
    "rdgan_op_conv3d_dgrad_bf16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 13 + [c_stream]),
    "rdgan_op_conv3d_wgrad": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 14 + [c_stream]),
    "rdgan_op_fastd_wgrad": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 8 + [c_stream]),
    "rdgan_op_g9_wgrad": (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 4 + [c_stream]),
    "rdgan_op_upconv_slab16": (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_int, c_stream]),
    "rdgan_op_upconv_slab_t16": (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_int] * 3 + [c_stream]),
    "rdgan_op_upconv2_slab16": (ctypes.c_int, [c_f32p] * 5 + [ctypes.c_int, c_stream]),
    "rdgan_op_upconv_wgrad_slab16": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int, c_stream]),
    "rdgan_op_d2_fwd_slab16": (ctypes.c_int, [c_f32p] * 4 + [ctypes.c_int, ctypes.c_uint64, c_stream]),
    "rdgan_op_d3_wgrad_slab16": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int, c_stream]),
    "rdgan_op_d2_wgrad_slab16": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int, c_stream]),
    "rdgan_op_d2_wgrad_slab_t16": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int] * 3 + [c_stream]),
    "rdgan_op_d2_dgrad_slab16": (ctypes.c_int, [c_f32p] * 4 + [ctypes.c_int, ctypes.c_int, c_stream]),
    "rdgan_op_d2_dgrad_slab_t16": (ctypes.c_int, [c_f32p] * 4 + [ctypes.c_int] * 4 + [c_stream]),
    "rdgan_op_pixelnorm_lrelu": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_long, ctypes.c_int, c_stream]),
    "rdgan_op_pixelnorm_lrelu_bwd": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_long, ctypes.c_int, c_stream]),
    "rdgan_debug_activation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_f32p, ctypes.c_long, c_stream]),
    "rdgan_op_rng": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint32, c_f32p, c_f32p, ctypes.c_long, c_stream]),
}

TAG_GCONV_FWD, TAG_GCONV_DGRAD, TAG_GCONV_WGRAD, TAG_CRITIC_GEMM, TAG_ELEMENTWISE, TAG_GCONV3_FWD = range(6)

_lib = None


class RdganError(RuntimeError):
    pass


class NumericsError(RdganError, ArithmeticError):
    """tf.debugging.check_numerics behind the generator's softmax (reference T:349-350) fired: NaN/Inf in the output."""


def load():
    """Load the HIP library and bind every declared symbol.  Raises (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdganError(
            f"{LIB_PATH} is missing: build it with `python -m pr_disagg_radar_gan_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, handle=None, what=""):
    if rc == 0:
        return
    msg = ""
    if handle is not None and _lib is not None:
        m = _lib.rdgan_last_error(handle)
        msg = m.decode() if m else ""
    raise RdganError(f"{what} failed with code {rc}" + (f": {msg}" if msg else ""))
