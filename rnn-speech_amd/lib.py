"""ctypes binding of libamdspeech.so (the C ABI in include/amdspeech.h).

There is NO CPU fallback: if the shared library is missing or a call fails the
import / call raises.  The library is built in-tree by build.py (hipcc, gfx950).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMDSPEECH_LIB") or os.path.join(HERE, "libamdspeech.so")   # env: dev override only


class AmdSpeechError(RuntimeError):
    pass


class DataflowTimeout(AmdSpeechError):
    """amdspeech_lstm_status returned AMDSPEECH_ETIMEOUT: a whole-sequence launch gave up waiting.  The mini-batch's results are
    invalid; it can be repeated on the launch-per-diagonal kernels.  Any OTHER error of that call is a device fault."""


ETIMEOUT = -4                     # include/amdspeech.h


class LstmDesc(C.Structure):
    _fields_ = [("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("L", C.c_int),
                ("keep_in", C.c_float), ("keep_out", C.c_float), ("seed", C.c_uint64),
                ("precision", C.c_int), ("flags", C.c_int)]


class CtcHead(C.Structure):          # amdspeech_ctc_head (include/amdspeech.h): the CTC head fused into the whole-sequence LSTM kernels
    _fields_ = [("w_out", C.c_void_p), ("b_out", C.c_void_p), ("logits", C.c_void_p), ("dense_labels", C.c_void_p),
                ("loss", C.c_void_p), ("dlogits", C.c_void_p), ("ctc_ws", C.c_void_p), ("C", C.c_int), ("U", C.c_int)]


LSTM_ARMED, LSTM_ARM_NEXT, LSTM_SAME_WS, LSTM_PER_DIAGONAL, LSTM_INJECT_TIMEOUT = 1, 2, 4, 8, 16      # amdspeech_lstm_desc.flags (include/amdspeech.h)


COMM_ID_BYTES = 128
WS_Z0, WS_ZTOP, WS_DZTOP, WS_DZ0, WS_HFINAL, WS_CFINAL = range(6)

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_long
_SZ = C.c_size_t

# name -> (restype, argtypes); must list EVERY symbol include/amdspeech.h declares.
PROTOTYPES = {
    "amdspeech_version": (_I, []),
    "amdspeech_last_error": (C.c_char_p, []),
    "amdspeech_device_cu_count": (_I, []),
    "amdspeech_linear_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "amdspeech_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "amdspeech_gemm_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I]),
    "amdspeech_gemm_bf16x3": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I]),
    "amdspeech_gemm_bf16": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I]),
    "amdspeech_gemm_bf16_packed_scratch_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "amdspeech_gemm_bf16_packed": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _SZ]),
    "amdspeech_batchnorm_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F]),
    "amdspeech_batchnorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "amdspeech_batchnorm_sum": (_I, [_P, _P, _P, _I, _P, _I, _I, _I]),
    "amdspeech_batchnorm_apply": (_I, [_P, _P, _P, _P, _I, _F, _P, _P, _P, _I, _I, _I]),
    "amdspeech_batchnorm_bwd_sums": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "amdspeech_batchnorm_bwd_apply": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I]),
    "amdspeech_lstm_workspace_bytes": (_SZ, [C.POINTER(LstmDesc)]),
    "amdspeech_lstm_ws_ptr": (_P, [C.POINTER(LstmDesc), _P, _I]),
    "amdspeech_lstm_fwd": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _L, _P, _L, _P, _P, _P]),
    "amdspeech_lstm_status": (_I, [C.POINTER(LstmDesc), _P]),
    "amdspeech_lstm_workspace_release": (_I, [_P, _P]),
    "amdspeech_lstm_beside_forward": (_I, [_P, _P]),
    "amdspeech_lstm_beside_tail": (_I, [_P, _P]),
    "amdspeech_lstm_bwd": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _L, _P, _P, _L, _P]),
    "amdspeech_lstm_ctc_fusable": (_I, [C.POINTER(LstmDesc), _I, _I]),
    "amdspeech_lstm_pair_fusable": (_I, [C.POINTER(LstmDesc)]),
    "amdspeech_lstm_fwd_pair": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _P, C.POINTER(LstmDesc), _P, _P, _P, _L, _L, _P, _P, _P]),
    "amdspeech_lstm_bwd_pair": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _P, _P, C.POINTER(LstmDesc), _P, _P, _P, _P, _L, _L, _P]),
    "amdspeech_lstm_fwd_ctc": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _L, _P, _L, _P, _P, _P, C.POINTER(CtcHead)]),
    "amdspeech_lstm_bwd_ctc": (_I, [_P, C.POINTER(LstmDesc), _P, _P, _L, _P, _P, _L, _P, C.POINTER(CtcHead)]),
    "amdspeech_lstm_dropout_multipliers": (_I, [_P, C.POINTER(LstmDesc), _I, _I, _P]),
    "amdspeech_ctc_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "amdspeech_ctc_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "amdspeech_ctc_loss_fwd_bwd_staged": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I]),
    "amdspeech_ctc_greedy_decode": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "amdspeech_merge_repeated": (_I, [_P, _P, _P, _I, _I, _I]),
    "amdspeech_edit_distance": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "amdspeech_ctc_beam_search_host": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "amdspeech_ctc_beam_search_host_mt": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I]),
    "amdspeech_edit_distance_host": (_I, [_P, _P, _I, _P, _P, _I, _I, _P]),
    "amdspeech_crc32c": (C.c_uint32, [_P, _SZ, C.c_uint32]),
    "amdspeech_resample_workspace_bytes": (_SZ, [C.c_int]),
    "amdspeech_resample_num_samples": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "amdspeech_resample": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "amdspeech_audio_probe": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long)]),
    "amdspeech_audio_decode": (C.c_int, [C.c_char_p, _P, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_int]),
    "amdspeech_optim_workspace_bytes": (_SZ, [_L]),
    "amdspeech_clip_adam": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P]),
    "amdspeech_frontend_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "amdspeech_frontend_num_frames": (_I, [_I, _I, _I]),
    "amdspeech_frontend_mfcc": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "amdspeech_frontend_fbank": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "amdspeech_profile_enable": (_I, [_I]),
    "amdspeech_profile_get": (_I, [_I, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "amdspeech_profile_get_flops": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "amdspeech_comm_available": (_I, []),
    "amdspeech_comm_unique_id": (_I, [_P]),
    "amdspeech_comm_init": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "amdspeech_comm_destroy": (_I, [_P]),
    "amdspeech_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    "amdspeech_allreduce_sum_f32": (_I, [_P, _P, _P, _L]),
    "amdspeech_broadcast_f32": (_I, [_P, _P, _P, _L, _I]),
    "amdspeech_reverse_sequences": (_I, [_P, _P, _P, _P, _I, _I, _I, _I]),
    "amdspeech_axpy": (_I, [_P, _F, _P, _P, _L]),
    "amdspeech_fill": (_I, [_P, _P, _F, _L]),
}

_lib = None


def load():
    """Load the library (once).  Raises AmdSpeechError when it is absent --
    the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmdSpeechError(
            "libamdspeech.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().amdspeech_last_error().decode("utf-8", "replace")
        raise (DataflowTimeout if rc == ETIMEOUT else AmdSpeechError)("%s failed (%d): %s" % (what or "amdspeech call", rc, msg))
