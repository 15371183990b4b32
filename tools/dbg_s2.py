"""bring-up script: one forward / backward / update of a blstm stack in a given mode and size (python tools/dbg_s2.py T PS layers mode)"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
from helpers import net_desc, random_sequences, random_weights
pkg = g.load_package()
T, PS, NL = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = {"bf16": pkg.PREC_BF16, "x3": pkg.PREC_BF16X3}[sys.argv[4]]
maxT = int(sys.argv[5]) if len(sys.argv) > 5 else T
rng = np.random.RandomState(1)
P, C = 39, 183
layers = net_desc(P, [("blstm", 250)] * NL, C)
weights = random_weights(layers, rng, 0.08)
xs, ts = random_sequences(rng, [T] * PS, P, C=C)
frac = pkg.make_fraction(xs, ts, PS)
with pkg.NeuralNetwork(layers, weights, PS, maxT, precision=mode) as net:
    import os
    for it in range(3):
        net.load_sequences(frac); net.compute_forward_pass(); net.synchronize(); print(it, "fwd ok", flush=True)
        e, c = net.error_and_correct(); print(it, e, flush=True)
        if os.environ.get("DBG_LAYERWISE"):
            for lay in reversed(net.layers):
                pkg.binding.check(net.lib.cn_layer_backward(lay.handle), net.ctx); net.synchronize(); print(it, "bwd layer", lay.name, "ok", flush=True)
        else:
            net.compute_backward_pass(); net.synchronize()
        print(it, "bwd ok", net.recurrent_kernel(False), net.recurrent_kernel(True), flush=True)
        net.update_weights_fused(1e-4, 0.9); net.synchronize(); print(it, "update ok", flush=True)
