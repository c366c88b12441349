"""ctypes binding of libds2hip.so (the C ABI declared in include/ds2hip.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing or a kernel launch fails,
this module raises -- it never routes to torch ops or to the CPU oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libds2hip.so")

F32, BF16 = 0, 1
CELL_GRU, CELL_LSTM, CELL_RNN_TANH = 0, 1, 2

_vp, _i, _l, _f, _ll = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_longlong

# name -> (restype, [argtypes]); mirrors include/ds2hip.h one to one (tests check every symbol is exported)
SIGNATURES = {
    "ds2_version": (_i, []),
    "ds2_error_string": (C.c_char_p, [_i]),
    "ds2_gemm_nt": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _l, _l, _l, _l, _i, _vp]),
    "ds2_gemm_nt_coresident": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _l, _l, _l, _l, _i, _vp]),
    "ds2_gemm_nt_rows2": (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _i, _vp]),
    "ds2_gemm8_nt": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _vp]),
    "ds2_gemm8_tn_grouped": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ds2_gemm8_nt_rows": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _vp, _i, _vp]),
    "ds2_gemm8_tn_grouped_rows": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ds2_gemm8_wgrad_dx_rows": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _vp, _i, _vp]),
    "ds2_zero_pad_rows": (_i, [_i, _vp, _l, _i, _vp, _i, _i, _vp]),
    "ds2_gemm8_wgrad_dx": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _vp]),
    "ds2_norm_partials": (_i, [_l]),
    "ds2_bn_fwd": (_i, [_i, _i, _i, _vp, _vp, _l, _i, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp,
                        _vp, _vp, _vp, _vp]),
    "ds2_bn_bwd": (_i, [_i, _i, _vp, _vp, _vp, _l, _i, _l, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ds2_colsum": (_i, [_i, _vp, _l, _i, _l, _vp, _f, _vp, _vp]),
    "ds2_conv_rows": (_i, [_i, _vp, _vp]),
    "ds2_conv1_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ds2_conv1_wgrad_ws_floats": (_l, [_i, _i, _i]),
    "ds2_conv1_wgrad": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ds2_conv2_fwd_ws_bytes": (_l, [_i, _i, _i, _i]),
    "ds2_conv2_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ds2_conv2_dgrad": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ds2_conv2_wgrad_ws_floats": (_l, [_i, _i]),
    "ds2_conv2_wgrad": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ds2_rnn_gates": (_i, [_i]),
    "ds2_rnn_saved_planes": (_i, [_i]),
    "ds2_rnn_state_bytes": (_l, [_i, _i, _i]),
    "ds2_rnn_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp]),
    "ds2_rnn_bwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ds2_rnn_persist_supported": (_i, [_i, _i, _i, _i, _i, C.c_uint]),
    "ds2_rnn_persist_shape_covered": (_i, [_i, _i, _i, _i, _i]),
    "ds2_rnn_persist_kind": (_i, [_i, _i, _i, _i, _i, C.c_uint]),
    "ds2_rnn_persist_ws_bytes": (_l, [_i, _i, _i, _i, _i, C.c_uint]),
    "ds2_rnn_persist_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ds2_rnn_persist_bwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "ds2_add2": (_i, [_i, _vp, _vp, _vp, _l, _vp]),
    "ds2_sum_slices": (_i, [_vp, _vp, _l, _i, _vp]),
    "ds2_transpose": (_i, [_i, _vp, _vp, _l, _i, _l, _l, _vp]),
    "ds2_split3_bf16": (_i, [_vp, _l, _l, _i, _i, _i, _vp, _l, _vp]),
    "ds2_cast_transpose_bf16": (_i, [_vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp]),
    "ds2_small_weight_layouts": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ds2_scale_by": (_i, [_vp, _vp, _l, _vp]),
    "ds2_copy_words": (_i, [_vp, _vp, _l, _vp]),
    "ds2_lookahead_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ds2_lookahead_ws_floats": (_l, [_i, _i, _i, _i]),
    "ds2_lookahead_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ds2_rnn_bias_grads": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ds2_softmax_rows": (_i, [_vp, _vp, _l, _i, _l, _l, _vp]),
    "ds2_opt_max_tensors": (_i, []),
    "ds2_clip_ws_floats": (_l, [_i, _vp]),
    "ds2_clip_coef": (_i, [_i, _vp, _vp, _f, _vp, _vp, _vp]),
    "ds2_opt_multi": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ds2_opt_matrix": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _i, _vp, _vp]),
    "ds2_opt_matrices": (_i, [_i, _i] + [_vp] * 13 + [_vp, _i, _vp, _vp]),
    "ds2_spect_frames": (_i, [_i]),
    "ds2_spect_ws_bytes": (_l, [_i, _i]),
    "ds2_spectrogram": (_i, [_vp, _l, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ds2_greedy_decode": (_i, [_vp, _l, _l, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ds2_ctc_ws_floats": (_l, [_i, _i, _i, _i]),
    "ds2_ctc_loss_grad": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _l, _vp, _i, _vp]),
}



class PersistOpts(C.Structure):
    """ds2_persist_opts (include/ds2hip.h): per-launch options of the persistent sweeps."""
    _fields_ = [("variant", C.c_uint), ("spin_limit", C.c_uint), ("startup_ms", C.c_uint)]


_lib = None


class Ds2HipError(RuntimeError):
    pass


def load():
    """Loads the shared object once; raises Ds2HipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Ds2HipError(
            "libds2hip.so not found at %s -- build it with `python -m deepspeech.pytorch_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI and this table ever diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def error_string(code):
    return load().ds2_error_string(int(code)).decode()


def call(name, *args):
    """Calls a status-returning entry; raises on a non-zero code."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise Ds2HipError("%s failed with code %d: %s" % (name, rc, error_string(rc)))


def query(name, *args):
    """Calls a value-returning (size query) entry."""
    return getattr(load(), name)(*args)
