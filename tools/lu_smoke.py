"""Long-utterance stress shape (BASELINE.json configs[4]): 39 -> n x blstm1024 (H = 512) -> softmax183, PS = 16."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from bench import make_weights, net_desc, synth_fraction
pkg = ge.load_package()
P, C, PS = 39, 183, 16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
layers = net_desc(P, [("blstm", 1024)] * nl, C)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, P, C, T, T)
frames = pkg.fraction.real_frames(frac)
def step():
    net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate(); net.compute_backward_pass(); net.update_weights_fused(1e-5, 0.9)
for _ in range(2): step()
net.synchronize(); t0 = time.time()
n = 5
for _ in range(n): step()
net.synchronize(); dt = (time.time() - t0) / n
print("T=%d layers=%d: %.2f ms per fraction, %.3f M frames/s, %.2f us per time step and layer pass pair" % (T, nl, dt * 1e3, frames / dt / 1e6, dt / T / nl * 1e6))
print("error", net.loss_read())
net.close()
