"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module (see the header of currennt_oracle.c).  `OracleNetwork` drives the C
restatement layer by layer in the call order of the reference:

  NeuralNetwork::loadSequences / computeForwardPass / calculateError /
  computeBackwardPass            (currennt_lib/src/NeuralNetwork.cpp:161-190)
  Optimizer::_processDataSet     (currennt_lib/src/optimizers/Optimizer.cu:37-104)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcurrennt_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")

ACT = {"feedforward_tanh": 0, "feedforward_logistic": 1, "feedforward_identity": 2}
POST = {"sse": 0, "weightedsse": 1, "wf": 2, "sse_mask": 2, "ce": 3, "rmse": 4, "binary_classification": 5}


def build():
    """Compile oracle/libcurrennt_oracle.so with gcc (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcurrennt_oracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(_HERE, "currennt_oracle.c")
    if (not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        build()
    L = C.CDLL(_LIB_PATH)
    ci, cf, vp = C.c_int, C.c_float, C.c_void_p
    L.orc_matmul.argtypes = [ci, f32p, f32p, ci, ci, f32p, ci, ci, ci]
    L.orc_lstm_weight_count.argtypes = [ci, ci, ci]
    L.orc_lstm_weight_count.restype = ci
    L.orc_lstm_internal_count.restype = ci
    L.orc_lstm_forward.argtypes = [ci, ci, ci, cf, ci, ci, ci, ci, i8p, f32p, f32p, f32p, f32p]
    L.orc_lstm_backward.argtypes = [ci, ci, ci, cf, ci, ci, ci, ci, i8p, f32p, f32p,
                                    f32p, vp, f32p, f32p]
    L.orc_ff_weight_count.argtypes = [ci, ci]
    L.orc_ff_weight_count.restype = ci
    L.orc_ff_forward.argtypes = [ci, ci, ci, cf, ci, f32p, f32p, f32p]
    L.orc_ff_backward.argtypes = [ci, ci, ci, cf, ci, f32p, f32p, f32p, f32p, vp, f32p]
    L.orc_softmax_forward.argtypes = [ci, ci, cf, ci, i8p, f32p, f32p, f32p, f32p]
    L.orc_softmax_backward.argtypes = [ci, ci, cf, ci, i8p, f32p, f32p, f32p, f32p, vp, f32p, f32p]
    L.orc_mcc_error.argtypes = [ci, ci, i32p, f32p]
    L.orc_mcc_error.restype = cf
    L.orc_mcc_correct.argtypes = [ci, ci, i32p, f32p]
    L.orc_mcc_correct.restype = ci
    L.orc_mcc_backward.argtypes = [ci, ci, i32p, f32p, f32p]
    L.orc_sse_error.argtypes = [ci, ci, i8p, f32p, f32p]
    L.orc_sse_error.restype = cf
    L.orc_sse_backward.argtypes = [ci, ci, i8p, f32p, f32p, f32p]
    L.orc_sgd_update.argtypes = [ci, cf, cf, f32p, f32p, f32p]
    L.orc_post_error.argtypes = [ci, ci, ci, i8p, f32p, f32p]
    L.orc_post_error.restype = cf
    L.orc_binary_correct.argtypes = [ci, i8p, f32p, f32p]
    L.orc_binary_correct.restype = ci
    L.orc_post_backward.argtypes = [ci, ci, ci, i8p, f32p, f32p, f32p]
    L.orc_set_threads.argtypes = [ci]
    L.orc_set_operand_rounding.argtypes = [ci]
    L.orc_get_operand_rounding.restype = ci
    L.orc_set_preact_rounding.argtypes = [ci]
    L.orc_get_preact_rounding.restype = ci
    L.orc_get_threads.restype = ci
    L.orc_set_threads(int(os.environ.get("ORACLE_THREADS", "1")))
    _lib = L
    return L


def set_threads(n):
    """Worker threads of the oracle (default 1 = the reference's single-threaded Thrust-host build).  Outputs are
    split over threads, every sum stays serial in the reference's order: results are bit-identical for any n.
    The test suite raises it so that config-size cases finish in seconds; bench.py's cpu_baseline keeps 1."""
    lib().orc_set_threads(int(n))


def get_threads():
    return int(lib().orc_get_threads())


def set_operand_rounding(mode):
    """None / "f32": the reference's fp32 arithmetic (default).  "bf16": a MODEL of CN_PREC_BF16 -- every matrix product
    rounds both operands to bf16 (x, y[t-1], W_in / W_rec, the four deltas, the output layer's deltas) and LSTM layers store
    their outputs rounded; accumulation, states, activations (libm), bias / peephole terms and their gradient sums stay
    fp32 in the reference's order (currennt_oracle.c, "operand rounding").  The HIP bf16 path is then held to the oracle at
    summation-order + v_exp_f32 / v_rcp_f32 distance instead of 3e-2.  Only the C restatement has the mode
    (OracleNetwork(backend="oracle")); oracle/_ref is the reference's object code and has no such switch."""
    if mode not in (None, "f32", "bf16"):
        raise ValueError("operand rounding mode must be None, 'f32' or 'bf16'")
    lib().orc_set_operand_rounding(1 if mode == "bf16" else 0)


def get_operand_rounding():
    return "bf16" if lib().orc_get_operand_rounding() else None


class operand_rounding:
    """with oracle.operand_rounding("bf16"): ...  (restores the previous mode)"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_operand_rounding()
        set_operand_rounding(self.mode)
        return self

    def __exit__(self, *exc):
        set_operand_rounding(self.prev)
        return False


_REF_PATH = os.path.join(_HERE, "_ref", "libcurrennt_ref.so")
_ref = None
# the functions oracle/_ref implements with the reference's own functors (oracle/ref/*.cpp): same signatures as orc_*
_REF_FUNCS = ["matmul", "lstm_forward", "lstm_backward", "ff_forward", "ff_backward", "softmax_forward", "softmax_backward",
              "mcc_error", "mcc_correct", "mcc_backward",
              "sse_error", "sse_backward", "post_error", "post_backward", "binary_correct"]


class _RefLib:
    """oracle/_ref/libcurrennt_ref.so behind the oracle's function names: orc_X -> ref_X for the functions the reference's
    own object code covers, the oracle's C restatement for the rest (weight counts, SGD update)."""

    def __init__(self, ref, orc):
        self._ref, self._orc = ref, orc

    def __getattr__(self, name):
        if name.startswith("orc_") and name[4:] in _REF_FUNCS:
            return getattr(self._ref, "ref_" + name[4:])
        return getattr(self._orc, name)


def ref_available():
    """True when oracle/_ref/libcurrennt_ref.so exists (built here from /root/reference by `make -C oracle _ref`, or
    shipped prebuilt to the GPU box)."""
    return os.path.exists(_REF_PATH)


def ref_lib():
    global _ref
    if _ref is not None:
        return _ref
    if not ref_available():
        raise RuntimeError("oracle/_ref is not built (needs /root/reference: make -C oracle _ref)")
    R, L = C.CDLL(_REF_PATH), lib()
    for f in _REF_FUNCS:
        o, r = getattr(L, "orc_" + f), getattr(R, "ref_" + f)
        r.argtypes, r.restype = o.argtypes, o.restype
    _ref = _RefLib(R, L)
    return _ref


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


INTERNAL_NAMES = ["tmpOutputs", "tmpOutputErrors", "cellStates", "cellStateErrors",
                  "niActs", "igActs", "fgActs", "ogActs",
                  "niDeltas", "igDeltas", "fgDeltas", "ogDeltas"]


class _Layer:
    def __init__(self, desc, prev, PS, maxT, weights):
        self.desc = desc
        self.name = desc["name"]
        self.type = desc["type"]
        self.size = int(desc["size"])
        self.prev = prev
        self.PS, self.maxT = PS, maxT
        self.bias = float(desc.get("bias", 0.0))
        self.trainable = self.type in ("lstm", "blstm", "softmax") or self.type in ACT
        self.post = self.type == "multiclass_classification" or self.type in POST
        n = PS * maxT * self.size
        # PostOutputLayer writes into the preceding layer's outputErrors (PostOutputLayer.cpp:43-47)
        self.outputs = np.zeros(n, np.float32) if not self.post else None
        self.outputErrors = np.zeros(n, np.float32) if not self.post else None
        self.weights = self.weightUpdates = None
        if self.trainable:
            P = prev.size
            if self.type in ("lstm", "blstm"):
                self.bidir = self.type == "blstm"
                if self.bidir and self.size % 2:
                    raise RuntimeError("Cannot create a bidirectional layer with an odd layer size")
                nw = lib().orc_lstm_weight_count(P, self.size, int(self.bidir))
                dirs = 2 if self.bidir else 1
                self.H = self.size // dirs
                self.bufs = np.zeros(dirs * 12 * PS * maxT * self.H, np.float32)
            else:
                nw = lib().orc_ff_weight_count(P, self.size)
                self.patTmp = np.zeros(PS * maxT, np.float32)
            w = weights.get(self.name) if weights else None
            if w is None:
                raise RuntimeError("oracle networks need explicit weights (SURVEY Q13)")
            flat = np.concatenate([np.asarray(w["input"], np.float32),
                                   np.asarray(w["bias"], np.float32),
                                   np.asarray(w["internal"], np.float32)])
            if flat.size != nw:
                raise RuntimeError("Invalid number of weights for layer '%s'" % self.name)
            self.weights = np.ascontiguousarray(flat)
            self.weightUpdates = np.zeros(nw, np.float32)

    def internal(self, which, d=0):
        """LSTM per-direction internal vector by reference name (LstmLayer.hpp:88-100)."""
        per = self.PS * self.maxT * self.H
        b = INTERNAL_NAMES.index(which)
        return self.bufs[(d * 12 + b) * per:(d * 12 + b + 1) * per]


class OracleNetwork:
    """Layer stack driven like NeuralNetwork.cpp:37-130,161-190."""

    def __init__(self, layers, weights, parallel_sequences, max_seq_length, backend="oracle"):
        """backend "oracle": the C restatement; "ref": the same call sequence through oracle/_ref, i.e. the reference's own
        compiled functors and Cpu GEMM (tests/test_oracle_ref.py holds the two bit-equal)."""
        self._L = lib() if backend == "oracle" else ref_lib()
        self.PS, self.maxT = parallel_sequences, max_seq_length
        self.layers = []
        prev = None
        for desc in layers:
            lay = _Layer(desc, prev, self.PS, self.maxT, weights)
            self.layers.append(lay)
            prev = lay
        if self.layers[0].type != "input":
            raise RuntimeError("The first layer is not an input layer")
        if not self.layers[-1].post:
            raise RuntimeError("The last layer is not a post output layer")

    # -- NeuralNetwork::loadSequences (NeuralNetwork.cpp:161-166)
    def load_sequences(self, frac):
        if frac["inputs"].shape[-1] != self.layers[0].size:
            raise RuntimeError("Input layer size of %d != data input pattern size of %d"
                               % (self.layers[0].size, frac["inputs"].shape[-1]))
        self.T, self.Tmin = int(frac["T"]), int(frac["Tmin"])
        self.N = self.T * self.PS
        self.patTypes = np.ascontiguousarray(frac["patTypes"], np.int8)
        x = np.ascontiguousarray(frac["inputs"], np.float32).reshape(-1)
        self.layers[0].outputs[:x.size] = x
        post = self.layers[-1]
        if post.type == "multiclass_classification":
            self.targetClasses = np.ascontiguousarray(frac["targetClasses"], np.int32)
        elif post.type == "binary_classification":      # BinaryClassificationLayer.cu:157-164
            self.targets = np.ascontiguousarray(frac["targetClasses"], np.float32).reshape(-1)
        else:
            self.targets = np.ascontiguousarray(frac["targets"], np.float32).reshape(-1)

    # -- NeuralNetwork::computeForwardPass (NeuralNetwork.cpp:168-173)
    def compute_forward_pass(self):
        L = self._L
        for lay in self.layers[1:-1]:
            P, x = lay.prev.size, lay.prev.outputs
            if lay.type in ("lstm", "blstm"):
                # model switch of the operand-rounding mode (C restatement only): this layer's input-projection pre-activations
                # are stored in bf16 (layer attribute round_preacts, set by the tests from the kernel the HIP path reports)
                pre = bool(getattr(lay, "round_preacts", False))
                if pre:
                    lib().orc_set_preact_rounding(1)
                try:
                    L.orc_lstm_forward(P, lay.size, int(lay.bidir), lay.bias, self.PS, self.maxT,
                                       self.T, self.Tmin, self.patTypes, lay.weights, x,
                                       lay.outputs, lay.bufs)
                finally:
                    if pre:
                        lib().orc_set_preact_rounding(0)
            elif lay.type == "softmax":
                L.orc_softmax_forward(P, lay.size, lay.bias, self.N, self.patTypes,
                                      lay.weights, x, lay.outputs, lay.patTmp)
            else:
                L.orc_ff_forward(ACT[lay.type], P, lay.size, lay.bias, self.N,
                                 lay.weights, x, lay.outputs)

    # -- PostOutputLayer::calculateError
    def calculate_error(self):
        L = self._L
        post, out = self.layers[-1], self.layers[-2]
        if post.type == "multiclass_classification":
            return float(L.orc_mcc_error(post.size, self.N, self.targetClasses, out.outputs))
        return float(L.orc_post_error(POST[post.type], out.size, self.N, self.patTypes, self.targets, out.outputs))

    def count_correct_classifications(self):
        post, out = self.layers[-1], self.layers[-2]
        if post.type == "binary_classification":
            return int(self._L.orc_binary_correct(self.N, self.patTypes, self.targets, out.outputs))
        return int(self._L.orc_mcc_correct(post.size, self.N, self.targetClasses, out.outputs))

    # -- NeuralNetwork::computeBackwardPass (NeuralNetwork.cpp:175-184), reverse order
    def compute_backward_pass(self):
        L = self._L
        post, out = self.layers[-1], self.layers[-2]
        if post.type == "multiclass_classification":
            L.orc_mcc_backward(post.size, self.N, self.targetClasses, out.outputs, out.outputErrors)
        else:
            L.orc_post_backward(POST[post.type], out.size, self.N, self.patTypes, self.targets,
                                out.outputs, out.outputErrors)
        for lay in reversed(self.layers[1:-1]):
            P, x = lay.prev.size, lay.prev.outputs
            # only a trainable preceding layer receives errors (LstmLayer.cu:991-992)
            prev_err = lay.prev.outputErrors if lay.prev.trainable else None
            if lay.type in ("lstm", "blstm"):
                L.orc_lstm_backward(P, lay.size, int(lay.bidir), lay.bias, self.PS, self.maxT,
                                    self.T, self.Tmin, self.patTypes, lay.weights, x,
                                    lay.outputErrors, _ptr(prev_err), lay.weightUpdates, lay.bufs)
            elif lay.type == "softmax":
                L.orc_softmax_backward(P, lay.size, lay.bias, self.N, self.patTypes, lay.weights,
                                       x, lay.outputs, lay.outputErrors, _ptr(prev_err),
                                       lay.weightUpdates, lay.patTmp)
            else:
                L.orc_ff_backward(ACT[lay.type], P, lay.size, lay.bias, self.N, lay.weights,
                                  x, lay.outputs, lay.outputErrors, _ptr(prev_err),
                                  lay.weightUpdates)

    # -- SteepestDescentOptimizer::_updateWeights (SteepestDescentOptimizer.cu:67-94)
    def update_weights(self, learning_rate, momentum, deltas=None):
        if deltas is None:
            deltas = getattr(self, "_deltas", None)
            if deltas is None:
                deltas = self._deltas = {l.name: np.zeros_like(l.weights)
                                         for l in self.layers if l.trainable}
        for lay in self.layers:
            if not lay.trainable:
                continue
            lr = learning_rate
            if float(lay.desc.get("learningRate", -1.0)) >= 0.0:
                lr = float(lay.desc["learningRate"])
            lib().orc_sgd_update(lay.weights.size, lr, momentum, lay.weights,
                                 lay.weightUpdates, deltas[lay.name])

    def trainable_layers(self):
        return [l for l in self.layers if l.trainable]

    def layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)

    def outputs(self):
        """Output layer activations [T][PS][C] of the current fraction."""
        out = self.layers[-2]
        return out.outputs[:self.N * out.size].reshape(self.T, self.PS, out.size)
