"""ctypes binding of include/currennt_hip.h (one declaration per exported symbol)."""
import ctypes as C
import importlib.util
import os
import sys
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
COMM_ID_BYTES = 128

# cn_layer_kind, keyed by the type strings of LayerFactory.cu:52-87
LAYER_KINDS = {
    "input": 0, "lstm": 1, "blstm": 2,
    "feedforward_tanh": 3, "feedforward_logistic": 4, "feedforward_identity": 5,
    "softmax": 6, "sse": 7, "multiclass_classification": 8,
    "weightedsse": 9, "wf": 10, "ce": 11, "rmse": 12, "binary_classification": 13,
}

# cn_buffer
BUF = {
    "outputs": 0, "outputErrors": 1, "weights": 2, "weightUpdates": 3, "weightDeltas": 4,
    "cellStates": 5, "niActs": 6, "igActs": 7, "fgActs": 8, "ogActs": 9,
    "niDeltas": 10, "igDeltas": 11, "fgDeltas": 12, "ogDeltas": 13, "tmpOutputs": 14,
}

# every symbol include/currennt_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "cn_ctx_create", "cn_ctx_destroy", "cn_ctx_synchronize", "cn_ctx_set_option", "cn_ctx_get_option", "cn_ctx_join", "cn_layer_join", "cn_layer_join_stream", "cn_ctx_stream", "cn_last_error", "cn_device_arch", "cn_device_count", "cn_device_name",
    "cn_version", "cn_layer_create", "cn_layer_destroy", "cn_layer_size", "cn_layer_kind_of",
    "cn_layer_weight_count", "cn_fraction_load", "cn_fraction_load_resident", "cn_fraction_prefetch_resident", "cn_fraction_prefetch", "cn_layer_forward",
    "cn_layer_backward", "cn_loss_eval", "cn_loss_accumulate", "cn_loss_read", "cn_layer_set_weights", "cn_layer_read", "cn_layer_write_output_errors", "cn_layer_upload",
    "cn_layer_device_ptr", "cn_ctx_param_arena", "cn_ctx_weights_touched", "cn_sgd_update",
    "cn_sgd_update_all", "cn_ctx_arm_update", "cn_ctx_accumulate_updates", "cn_ctx_take_accumulated", "cn_layer_set_learning_rate", "cn_ctx_timing_enable", "cn_ctx_timing_read", "cn_ctx_timing_reset",
    "cn_layer_recurrent_kernel",
    "cn_comm_unique_id", "cn_comm_init", "cn_comm_destroy", "cn_comm_info", "cn_comm_backend", "cn_allreduce_grads", "cn_loss_read_global",
    # include/currennt_hip_debug.h
    "cn_dbg_gemm_nt", "cn_dbg_gemm_tn", "cn_dbg_row_map_counts", "cn_dbg_prefetch_hits",
]


class CurrenntHipError(RuntimeError):
    """A non-zero cn_status; mirrors the reference's std::runtime_error (main.cpp:492-495)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class Fraction(C.Structure):
    _fields_ = [("max_seq_length", C.c_int), ("min_seq_length", C.c_int), ("num_sequences", C.c_int),
                ("input_pattern_size", C.c_int), ("output_pattern_size", C.c_int),
                ("pat_types", C.c_void_p), ("inputs", C.c_void_p),
                ("target_classes", C.c_void_p), ("targets", C.c_void_p)]


def lib_path():
    # CURRENNT_HIP_LIB: alternative build of the same library (A/B timing of two builds on one device)
    return os.environ.get("CURRENNT_HIP_LIB") or os.path.join(_HERE, "libcurrennt_hip.so")


def build_library(verbose=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def load_library():
    """Load libcurrennt_hip.so and declare the prototypes.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise CurrenntHipError(-5, "libcurrennt_hip.so is not built (run __graft_entry__.build()); "
                                   "there is no CPU fallback for the HIP path")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, the library is linked against the system's.
    # Whichever is loaded first serves both (same SONAME); when the system's comes first, torch later reports
    # "No HIP GPUs are available".  So a process that can import torch loads torch's runtime before the library.
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None and not os.environ.get("CURRENNT_HIP_NO_TORCH_PRELOAD"):
        import torch  # noqa: F401
    L = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.cn_ctx_create.argtypes = [ci, ci, vp, C.POINTER(vp)]
    L.cn_ctx_destroy.argtypes = [vp]
    L.cn_ctx_synchronize.argtypes = [vp]
    L.cn_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.cn_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.cn_ctx_join.argtypes = [vp]
    L.cn_layer_join.argtypes = [vp]
    L.cn_layer_join_stream.argtypes = [vp, vp]
    L.cn_ctx_stream.argtypes = [vp]
    L.cn_ctx_stream.restype = vp
    L.cn_last_error.argtypes = [vp]
    L.cn_last_error.restype = C.c_char_p
    L.cn_device_arch.argtypes = [vp]
    L.cn_device_arch.restype = C.c_char_p
    L.cn_device_count.argtypes = []
    L.cn_device_name.argtypes = [ci, C.c_char_p, ci]
    L.cn_version.restype = C.c_char_p
    L.cn_layer_create.argtypes = [vp, ci, vp, ci, cf, ci, ci, C.POINTER(vp)]
    L.cn_layer_destroy.argtypes = [vp]
    L.cn_layer_size.argtypes = [vp]
    L.cn_layer_kind_of.argtypes = [vp]
    L.cn_layer_weight_count.argtypes = [vp]
    L.cn_fraction_load.argtypes = [vp, vp, vp, C.POINTER(Fraction)]
    L.cn_fraction_load_resident.argtypes = [vp, vp, vp, C.POINTER(Fraction)]
    L.cn_fraction_prefetch_resident.argtypes = [vp, vp, vp, C.POINTER(Fraction)]
    L.cn_fraction_prefetch.argtypes = [vp, vp, vp, C.POINTER(Fraction)]
    L.cn_loss_accumulate.argtypes = [vp]
    L.cn_loss_read.argtypes = [vp, C.POINTER(cf), C.POINTER(C.c_int64), ci]
    L.cn_layer_forward.argtypes = [vp]
    L.cn_layer_backward.argtypes = [vp]
    L.cn_loss_eval.argtypes = [vp, C.POINTER(cf), C.POINTER(ci)]
    L.cn_layer_set_weights.argtypes = [vp, vp, ci]
    L.cn_layer_read.argtypes = [vp, ci, ci, vp, C.c_size_t]
    L.cn_layer_write_output_errors.argtypes = [vp, vp, C.c_size_t]
    L.cn_layer_upload.argtypes = [vp, ci, vp, C.c_size_t]
    L.cn_layer_device_ptr.argtypes = [vp, ci]
    L.cn_layer_device_ptr.restype = vp
    L.cn_ctx_param_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cn_ctx_weights_touched.argtypes = [vp]
    L.cn_sgd_update.argtypes = [vp, cf, cf]
    L.cn_sgd_update_all.argtypes = [vp, cf, cf]
    L.cn_ctx_arm_update.argtypes = [vp, cf, cf]
    L.cn_ctx_accumulate_updates.argtypes = [vp, ci]
    L.cn_ctx_take_accumulated.argtypes = [vp]
    L.cn_ctx_timing_enable.argtypes = [vp, ci]
    L.cn_ctx_timing_read.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.cn_ctx_timing_reset.argtypes = [vp]
    L.cn_layer_set_learning_rate.argtypes = [vp, cf]
    L.cn_layer_recurrent_kernel.argtypes = [vp, ci]
    L.cn_layer_recurrent_kernel.restype = C.c_char_p
    L.cn_comm_unique_id.argtypes = [C.c_char_p]
    L.cn_comm_init.argtypes = [vp, C.c_char_p, ci, ci]
    L.cn_comm_destroy.argtypes = [vp]
    L.cn_comm_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.cn_comm_backend.argtypes = [vp, C.POINTER(C.c_int64)]
    L.cn_comm_backend.restype = C.c_char_p
    L.cn_allreduce_grads.argtypes = [vp, C.POINTER(vp), ci]
    L.cn_loss_read_global.argtypes = [vp, C.POINTER(cf), C.POINTER(C.c_int64), ci]
    L.cn_dbg_gemm_nt.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, ci]
    L.cn_dbg_gemm_tn.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.cn_dbg_row_map_counts.argtypes = [vp, vp]
    L.cn_dbg_prefetch_hits.argtypes = [vp, C.POINTER(C.c_int)]
    _LIB = L
    return L


def check(rc, ctx=None):
    if rc != 0:
        msg = load_library().cn_last_error(ctx)
        raise CurrenntHipError(rc, msg.decode() if msg else "cn_status %d" % rc)
