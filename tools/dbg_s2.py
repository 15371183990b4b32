import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
from helpers import net_desc, random_sequences, random_weights
pkg = g.load_package()
T = int(sys.argv[1])
rng = np.random.RandomState(1)
P, C, PS = 9, 7, 12
layers = net_desc(P, [("blstm", 250)], C)
weights = random_weights(layers, rng, 0.08)
xs, ts = random_sequences(rng, [T] * PS, P, C=C)
frac = pkg.make_fraction(xs, ts, PS)
with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16) as net:
    net.load_sequences(frac); net.compute_forward_pass(); print("fwd ok", flush=True)
    e, c = net.error_and_correct(); print(e, flush=True)
    net.compute_backward_pass(); net.synchronize(); print("bwd ok", net.recurrent_kernel(True), flush=True)
