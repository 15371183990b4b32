import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from bench import make_weights, net_desc, synth_fraction
pkg = ge.load_package()
P, C, PS, T = 40, 8000, 64, 120
layers = net_desc(P, [("blstm", 512), ("blstm", 512)], C)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, P, C, T - 20, T)
t0 = time.time()
for i in range(5):
    net.load_sequences(frac); net.compute_forward_pass(); e, c = net.error_and_correct(); net.compute_backward_pass(); net.update_weights_fused(1e-4, 0.9)
    print(i, e, c)
net.synchronize(); print("ok", time.time() - t0)
net.close()
