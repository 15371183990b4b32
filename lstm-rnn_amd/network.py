"""Host-side mirror of NeuralNetwork<TDevice> / Layer<TDevice> over the C ABI.

Method names and call order follow currennt_lib/src/NeuralNetwork.cpp:37-130 (construction from
the "layers"/"weights" JSON sections), :161-190 (loadSequences / computeForwardPass /
computeBackwardPass / calculateError), :237-262 (getOutputs) and
optimizers/SteepestDescentOptimizer.cu:67-94 (_updateWeights).  All arithmetic happens in
libcurrennt_hip.so; nothing here touches the CPU oracle.
"""
import ctypes as C

import numpy as np

from . import binding as B

_TRAINABLE = ("lstm", "blstm", "softmax", "feedforward_tanh", "feedforward_logistic",
              "feedforward_identity")
_POST = ("sse", "multiclass_classification", "weightedsse", "wf", "ce", "rmse", "binary_classification")


class Layer:
    """One layer handle (layers/Layer.hpp:40-179, TrainableLayer.hpp:84-155)."""

    def __init__(self, net, desc, prev):
        self.net, self.desc, self.prev = net, desc, prev
        if "name" not in desc:
            raise RuntimeError("Missing value 'name' in layer description")           # Layer.cpp:52-53
        if "size" not in desc:
            raise RuntimeError("Missing value 'size' in layer '%s'" % desc["name"])   # Layer.cpp:56-57
        self.name, self.type, self.size = desc["name"], desc["type"], int(desc["size"])
        if self.type not in B.LAYER_KINDS:
            raise RuntimeError("Unknown layer type '%s'" % self.type)                  # LayerFactory.cu:86
        self.trainable = self.type in _TRAINABLE
        self.post = self.type in _POST
        if self.trainable and "bias" not in desc:
            raise RuntimeError("Missing value 'bias' in layer '%s'" % self.name)       # TrainableLayer.cu:61-62
        self.bias = float(desc.get("bias", 0.0))
        self.learning_rate = float(desc.get("learningRate", -1.0))                     # TrainableLayer.cu:58
        h = C.c_void_p()
        L = net.lib
        B.check(L.cn_layer_create(net.ctx, B.LAYER_KINDS[self.type], prev.handle if prev else None,
                                  self.size, self.bias,
                                  net.parallel_sequences if prev is None else 0,
                                  net.max_seq_length if prev is None else 0, C.byref(h)), net.ctx)
        self.handle = h
        self.weight_count = L.cn_layer_weight_count(h) if self.trainable else 0
        if self.trainable and self.learning_rate >= 0.0:       # cn_sgd_update_all honours it too (SteepestDescentOptimizer.cu:78-80)
            B.check(L.cn_layer_set_learning_rate(h, self.learning_rate), net.ctx)
        if self.type in ("lstm", "blstm"):
            self.dirs = 2 if self.type == "blstm" else 1
            self.H = self.size // self.dirs

    # -- buffers in reference layout -----------------------------------------------------------
    def _read(self, which, count, direction=0):
        out = np.empty(count, np.float32)
        B.check(self.net.lib.cn_layer_read(self.handle, B.BUF[which], direction,
                                           out.ctypes.data_as(C.c_void_p), count), self.net.ctx)
        return out

    def outputs(self):
        return self._read("outputs", self.net.N * self.size).reshape(self.net.T, self.net.PS, self.size)

    def output_errors(self):
        return self._read("outputErrors", self.net.N * self.size).reshape(self.net.T, self.net.PS, self.size)

    def weights(self):
        return self._read("weights", self.weight_count)

    def weight_updates(self):
        return self._read("weightUpdates", self.weight_count)

    def weight_updates_tensor(self, torch):
        """The layer's weightUpdates in HBM as a torch tensor aliasing the library's memory (no copy)."""
        if getattr(self, "_wu_tensor", None) is None:
            from .parallel import DeviceArray
            ptr = self.net.lib.cn_layer_device_ptr(self.handle, B.BUF["weightUpdates"])
            if not ptr:
                raise RuntimeError("layer '%s' has no weightUpdates in device memory" % self.name)
            self._wu_tensor = torch.as_tensor(DeviceArray(ptr, self.weight_count), device="cuda:%d" % self.net.device)
        return self._wu_tensor

    def internal(self, which, direction=0):
        """LSTM per-direction internal vector by its reference name (LstmLayer.hpp:88-100)."""
        return self._read(which, self.net.N * self.H, direction).reshape(self.net.T, self.net.PS, self.H)

    def set_weights(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        B.check(self.net.lib.cn_layer_set_weights(self.handle, flat.ctypes.data_as(C.c_void_p), flat.size),
                self.net.ctx)

    def write_output_errors(self, err):
        err = np.ascontiguousarray(err, np.float32).reshape(-1)
        B.check(self.net.lib.cn_layer_write_output_errors(self.handle, err.ctypes.data_as(C.c_void_p), err.size),
                self.net.ctx)


class NeuralNetwork:
    def __init__(self, layers, weights=None, parallel_sequences=1, max_seq_length=1,
                 precision=B.PREC_F32, device=0, stream=None, seed=None, deterministic=None):
        self.lib = B.load_library()
        self.parallel_sequences, self.max_seq_length = int(parallel_sequences), int(max_seq_length)
        self.PS = self.parallel_sequences
        self.precision = precision
        self.device = int(device)
        ctx = C.c_void_p()
        B.check(self.lib.cn_ctx_create(device, precision, stream, C.byref(ctx)))
        self.ctx = ctx
        if deterministic is not None:                 # None: the library's default (on in the parity modes, off for bf16)
            self.set_option("deterministic", 1 if deterministic else 0)
        self.layers = []
        self.T = self.Tmin = self.N = 0
        try:
            names = set()
            prev = None
            for desc in layers:                                                        # NeuralNetwork.cpp:60-95
                lay = Layer(self, desc, prev)
                if lay.name in names:
                    raise RuntimeError("Different layers have the same name '%s'" % lay.name)   # :85-86
                names.add(lay.name)
                self.layers.append(lay)
                prev = lay
            if len(self.layers) < 3:
                raise RuntimeError("Not enough layers defined")                        # :98-99
            if self.layers[0].type != "input":
                raise RuntimeError("The first layer is not an input layer")            # :102-105 (paraphrased)
            if not self.layers[-1].post:
                raise RuntimeError("The last layer is not a post output layer")        # :116-117
            rng = np.random.RandomState(seed) if seed is not None else None
            for lay in self.layers:
                if not lay.trainable:
                    continue
                if weights is not None and lay.name in weights:                        # TrainableLayer.cu:68-101
                    w = weights[lay.name]
                    for key in ("input", "bias", "internal"):
                        if key not in w:
                            raise RuntimeError("Missing array 'weights/%s/%s'" % (lay.name, key))
                    flat = np.concatenate([np.asarray(w["input"], np.float32).reshape(-1),
                                           np.asarray(w["bias"], np.float32).reshape(-1),
                                           np.asarray(w["internal"], np.float32).reshape(-1)])
                    if flat.size != lay.weight_count:
                        raise RuntimeError("Invalid number of weights for layer '%s'" % lay.name)
                elif rng is not None:
                    # uniform [-0.1, 0.1] like Configuration.cpp:186-187; the reference draws from
                    # boost::mt19937, which is not reproducible here
                    flat = rng.uniform(-0.1, 0.1, lay.weight_count).astype(np.float32)
                else:
                    raise RuntimeError("no weights given for layer '%s' and no seed" % lay.name)
                lay.set_weights(flat)
        except Exception:
            self.close()
            raise

    # -- life cycle ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.cn_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- NeuralNetwork API ---------------------------------------------------------------------
    def input_layer(self):
        return self.layers[0]

    def post_output_layer(self):
        return self.layers[-1]

    def output_layer(self):
        return self.layers[-2]

    def trainable_layers(self):
        return [l for l in self.layers if l.trainable]

    def layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)

    def _host_descriptor(self, frac):
        x = np.ascontiguousarray(frac["inputs"], np.float32)
        pat = np.ascontiguousarray(frac["patTypes"], np.int8)
        f = B.Fraction()
        f.max_seq_length, f.min_seq_length = int(frac["T"]), int(frac["Tmin"])
        f.num_sequences = int(frac.get("numSeqs", self.PS))
        f.input_pattern_size = int(x.shape[-1])
        keep = [x, pat]
        f.pat_types = pat.ctypes.data_as(C.c_void_p)
        f.inputs = x.ctypes.data_as(C.c_void_p)
        post = self.layers[-1]
        if "targetClasses" in frac and frac["targetClasses"] is not None:
            tc = np.ascontiguousarray(frac["targetClasses"], np.int32)
            keep.append(tc)
            f.target_classes = tc.ctypes.data_as(C.c_void_p)
            f.output_pattern_size = int(frac.get("outputPatternSize", post.size))
        if "targets" in frac and frac["targets"] is not None:
            tg = np.ascontiguousarray(frac["targets"], np.float32)
            keep.append(tg)
            f.targets = tg.ctypes.data_as(C.c_void_p)
            f.output_pattern_size = int(tg.shape[-1])
        return f, keep

    def load_sequences(self, frac):                                                    # NeuralNetwork.cpp:161-166
        f, keep = self._host_descriptor(frac)
        # (the library copies the host arrays into pinned staging memory before it returns: nothing to wait for)
        B.check(self.lib.cn_fraction_load(self.ctx, self.layers[0].handle, self.layers[-1].handle, C.byref(f)), self.ctx)
        self.T, self.Tmin = f.max_seq_length, f.min_seq_length
        self.N = self.T * self.PS

    def prefetch_sequences(self, frac):
        """cn_fraction_prefetch: `frac` (host arrays, as for load_sequences) is what the next load_sequences will load -- the
        reference's loader thread one fraction ahead (DataSet.cpp:202-240), here across PCIe and through the re-layout as well.
        The arrays must be the same objects (already contiguous float32 / int8 / int32) at that load: it matches by address."""
        f, keep = self._host_descriptor(frac)
        B.check(self.lib.cn_fraction_prefetch(self.ctx, self.layers[0].handle, self.layers[-1].handle, C.byref(f)), self.ctx)

    def _resident_descriptor(self, dfrac):
        f = B.Fraction()
        f.max_seq_length, f.min_seq_length = int(dfrac["T"]), int(dfrac["Tmin"])
        f.num_sequences = int(dfrac.get("numSeqs", self.PS))
        f.input_pattern_size = int(dfrac["inputPatternSize"])
        f.output_pattern_size = int(dfrac.get("outputPatternSize", self.layers[-1].size))
        f.pat_types, f.inputs = dfrac["patTypes"], dfrac["inputs"]
        f.target_classes, f.targets = dfrac.get("targetClasses"), dfrac.get("targets")
        return f

    def prefetch_sequences_resident(self, dfrac):
        """cn_fraction_prefetch_resident: `dfrac` is what the next load_sequences_resident will load; it is re-laid out
        beside the coming backward pass."""
        f = self._resident_descriptor(dfrac)
        B.check(self.lib.cn_fraction_prefetch_resident(self.ctx, self.layers[0].handle, self.layers[-1].handle,
                                                       C.byref(f)), self.ctx)

    def load_sequences_resident(self, dfrac):
        """Like load_sequences, but `dfrac` holds DEVICE pointers (ints) for inputs / patTypes /
        targetClasses / targets: the fraction is already resident in HBM."""
        f = self._resident_descriptor(dfrac)
        B.check(self.lib.cn_fraction_load_resident(self.ctx, self.layers[0].handle, self.layers[-1].handle,
                                                   C.byref(f)), self.ctx)
        self.T, self.Tmin = f.max_seq_length, f.min_seq_length
        self.N = self.T * self.PS

    def loss_accumulate(self):
        """Add this fraction's error / #correct to the device-side epoch sums (no host sync)."""
        B.check(self.lib.cn_loss_accumulate(self.layers[-1].handle), self.ctx)

    def loss_read(self, reset=True):
        err, cor = C.c_float(), C.c_int64()
        B.check(self.lib.cn_loss_read(self.ctx, C.byref(err), C.byref(cor), 1 if reset else 0), self.ctx)
        return float(err.value), int(cor.value)

    def compute_forward_pass(self):                                                    # NeuralNetwork.cpp:168-173
        for lay in self.layers:
            B.check(self.lib.cn_layer_forward(lay.handle), self.ctx)

    def compute_backward_pass(self):                                                   # NeuralNetwork.cpp:175-184
        for lay in reversed(self.layers):
            B.check(self.lib.cn_layer_backward(lay.handle), self.ctx)

    def compute_backward_pass_allreduce(self, dist, torch):
        """Backward pass with the data-parallel gradient exchange folded in (SURVEY.md 8e "Overlap", bucket = layer):
        as soon as the backward pass of a layer is enqueued, its weightUpdates (whose gradient GEMMs run on the
        library's side stream) are all-reduced asynchronously from a communication stream that waits for that
        layer's gradient work only (cn_layer_join_stream), i.e. beside the recurrent kernels of the layers below
        it; only the first layer's exchange is exposed.  Returns after every reduction has been ordered before the
        context's stream; follow with update_weights_fused()."""
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(device=self.device)
        works = []
        main = self.torch_stream(torch)
        for lay in reversed(self.layers):
            B.check(self.lib.cn_layer_backward(lay.handle), self.ctx)
            if lay.trainable:
                B.check(self.lib.cn_layer_join_stream(lay.handle, C.c_void_p(self._comm_stream.cuda_stream)), self.ctx)
                with torch.cuda.stream(self._comm_stream):
                    works.append(dist.all_reduce(lay.weight_updates_tensor(torch), op=dist.ReduceOp.SUM, async_op=True))
        with torch.cuda.stream(main):
            for w in works:
                w.wait()            # orders the context's stream behind the reduction

    # -- data-parallel training through the library's own RCCL communicator -----------------------
    def set_option(self, name, value):
        """cn_ctx_set_option: named integer options of the context ("deterministic": fixed-order gradient sums)."""
        B.check(self.lib.cn_ctx_set_option(self.ctx, name.encode(), int(value)), self.ctx)

    def get_option(self, name):
        v = C.c_int()
        B.check(self.lib.cn_ctx_get_option(self.ctx, name.encode(), C.byref(v)), self.ctx)
        return v.value

    def comm_unique_id(self):
        """Rank 0: 128 rendezvous bytes (ncclGetUniqueId) to hand to every rank out of band."""
        buf = C.create_string_buffer(B.COMM_ID_BYTES)
        B.check(self.lib.cn_comm_unique_id(buf), self.ctx)
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Collective: bind an RCCL communicator for this context's GPU (cn_comm_init)."""
        if len(unique_id) != B.COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % B.COMM_ID_BYTES)
        B.check(self.lib.cn_comm_init(self.ctx, unique_id, int(rank), int(world)), self.ctx)

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        B.check(self.lib.cn_comm_info(self.ctx, C.byref(r), C.byref(w)), self.ctx)
        return r.value, w.value

    def comm_backend(self):
        """(name, exchanges): which exchange the bound communicator runs ("rccl", "p2p", "ipc"; "" without one) and how many
        all-reduces cn_allreduce_grads has enqueued on it (cn_comm_backend)."""
        k = C.c_int64()
        name = self.lib.cn_comm_backend(self.ctx, C.byref(k))
        return (name or b"").decode(), k.value

    def allreduce_grads(self, layers=None):
        """SUM all-reduce of the weightUpdates of `layers` (None: the whole arena in one exchange) on the library's
        communication stream; the next update waits for it on the device."""
        if not layers:
            B.check(self.lib.cn_allreduce_grads(self.ctx, None, 0), self.ctx)
            return
        arr = (C.c_void_p * len(layers))(*[l.handle for l in layers])
        B.check(self.lib.cn_allreduce_grads(self.ctx, arr, len(layers)), self.ctx)

    def compute_backward_pass_dp(self):
        """Backward pass with the gradient exchange folded in, bucket = layer (SURVEY.md 8e "Overlap"): the reduction of
        layer k is enqueued as soon as its backward pass is, and runs beside the recurrent kernels of the layers below.
        All RCCL calls are made by the library (cn_allreduce_grads); follow with update_weights_fused()."""
        for lay in reversed(self.layers):
            B.check(self.lib.cn_layer_backward(lay.handle), self.ctx)
            if lay.trainable:
                self.allreduce_grads([lay])

    def loss_read_global(self, reset=True):
        err, cor = C.c_float(), C.c_int64()
        B.check(self.lib.cn_loss_read_global(self.ctx, C.byref(err), C.byref(cor), 1 if reset else 0), self.ctx)
        return float(err.value), int(cor.value)

    def recurrent_kernel(self, backward):
        """Name of the recurrent kernel the first LSTM layer launches (for bench.py's roofline record)."""
        for lay in self.layers:
            if lay.type in ("lstm", "blstm"):
                return self.lib.cn_layer_recurrent_kernel(lay.handle, 1 if backward else 0).decode()
        return ""

    def prefetch_hits(self):
        """Loads that found their fraction announced and re-laid out ahead (cn_dbg_prefetch_hits)."""
        n = C.c_int(0)
        B.check(self.lib.cn_dbg_prefetch_hits(self.ctx, C.byref(n)), self.ctx)
        return n.value

    def row_map_counts(self):
        """(computed frames, dummy frames, T x padded parallel sequences) of the loaded fraction's row map (cn_dbg_row_map_counts)."""
        out = (C.c_int * 3)()
        B.check(self.lib.cn_dbg_row_map_counts(self.ctx, out), self.ctx)
        return tuple(out)

    def bf16_preactivation_layers(self):
        """Names of the LSTM layers whose input projection hands its pre-activations to the recurrent kernel as bf16
        (CN_PREC_BF16 with option pre16, the two-sequence forward kernels; LstmRec::pre16) -- as of the LAST forward pass, read off the kernel the
        library reports per layer.  (The parity tests model the rounding of exactly those layers.)"""
        out = []
        if self.precision != B.PREC_BF16:
            return out
        for lay in self.layers:
            if lay.type in ("lstm", "blstm"):
                k = self.lib.cn_layer_recurrent_kernel(lay.handle, 0).decode()
                if "_s2_" in k and self.get_option("pre16") == 1:
                    out.append(lay.name)
        return out

    def _loss(self):
        err, cor = C.c_float(), C.c_int()
        B.check(self.lib.cn_loss_eval(self.layers[-1].handle, C.byref(err), C.byref(cor)), self.ctx)
        return float(err.value), int(cor.value)

    def calculate_error(self):                                                         # NeuralNetwork.cpp:186-190
        return self._loss()[0]

    def count_correct_classifications(self):
        return self._loss()[1]

    def error_and_correct(self):
        return self._loss()

    def update_weights(self, learning_rate, momentum):                                 # SteepestDescentOptimizer.cu:67-94
        for lay in self.trainable_layers():
            lr = lay.learning_rate if lay.learning_rate >= 0.0 else learning_rate
            B.check(self.lib.cn_sgd_update(lay.handle, lr, momentum), self.ctx)

    def arm_update(self, learning_rate, momentum):
        """cn_ctx_arm_update: the coming backward pass applies each layer's momentum-SGD step as soon as that layer's gradient is
        complete; follow the backward pass with update_weights_fused(same values) (or update_weights), which completes the step."""
        B.check(self.lib.cn_ctx_arm_update(self.ctx, learning_rate, momentum), self.ctx)

    def update_weights_fused(self, learning_rate, momentum):
        """One launch for all layers; layers with a JSON learningRate of their own keep it (cn_layer_set_learning_rate)."""
        B.check(self.lib.cn_sgd_update_all(self.ctx, learning_rate, momentum), self.ctx)

    def accumulate_updates(self, first):
        """Batch learning (Optimizer.cu:72-85): add this fraction's weightUpdates of all layers to the epoch sum on the device
        (`first`: copy instead of add)."""
        B.check(self.lib.cn_ctx_accumulate_updates(self.ctx, 1 if first else 0), self.ctx)

    def take_accumulated(self):
        """The epoch sum becomes every layer's weightUpdates again (in front of the one update of the epoch, :95-97)."""
        B.check(self.lib.cn_ctx_take_accumulated(self.ctx), self.ctx)

    def outputs(self):
        """Output layer activations [T][PS][C] (NeuralNetwork.cpp:237-262 de-interleaves per sequence)."""
        return self.output_layer().outputs()

    def torch_stream(self, torch):
        """The context's HIP stream as a torch stream.  Collectives and tensor ops that must be ordered against the
        library's work run under `with torch.cuda.stream(net.torch_stream(torch))` -- torch's current stream is NOT the
        context's stream unless the caller made it so (a default-stream handle of 0 given to the constructor means
        "library-owned stream")."""
        if getattr(self, "_torch_stream", None) is None:
            self._torch_stream = torch.cuda.ExternalStream(int(self.lib.cn_ctx_stream(self.ctx)), device=torch.device("cuda", self.device))
        return self._torch_stream

    def join(self):
        """Order the ctx stream behind the internal side stream (before an external all-reduce)."""
        B.check(self.lib.cn_ctx_join(self.ctx), self.ctx)

    def synchronize(self):
        B.check(self.lib.cn_ctx_synchronize(self.ctx), self.ctx)

    def param_arena(self):
        """(weights_ptr, weight_updates_ptr, weight_deltas_ptr, count) raw device pointers."""
        w, g, d, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        B.check(self.lib.cn_ctx_param_arena(self.ctx, C.byref(w), C.byref(g), C.byref(d), C.byref(n)), self.ctx)
        return w.value, g.value, d.value, n.value

    def export_weights(self):
        """The "weights" JSON section (TrainableLayer.cu:211-248: input / bias / internal)."""
        out = {}
        for lay in self.trainable_layers():
            w = lay.weights()
            P, L = lay.prev.size, lay.size
            per_in = 4 if lay.type in ("lstm", "blstm") else 1
            n_in, n_b = L * per_in * P, L * per_in
            out[lay.name] = {"input": w[:n_in].tolist(), "bias": w[n_in:n_in + n_b].tolist(),
                             "internal": w[n_in + n_b:].tolist()}
        return out

    # -- timing ---------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        B.check(self.lib.cn_ctx_timing_enable(self.ctx, 1 if on else 0), self.ctx)

    def timing_reset(self):
        B.check(self.lib.cn_ctx_timing_reset(self.ctx), self.ctx)

    def timing_read(self):
        names = ["rec_fwd", "rec_bwd", "gemm_wide", "gemm_grad", "other", "exchange"]
        out = {}
        for k, nm in enumerate(names):
            ms, n = C.c_double(), C.c_int64()
            B.check(self.lib.cn_ctx_timing_read(self.ctx, k, C.byref(ms), C.byref(n)), self.ctx)
            out[nm] = (ms.value, n.value)
        return out
