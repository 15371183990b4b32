"""The literal CHiME example topology (examples/speech_recognition_chime/no_subsampling/network.jsn):
39 -> blstm156 -> blstm300 -> blstm102 -> softmax51 (H = 78 / 150 / 51 per direction), PS = 50, bf16."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from bench import make_weights, net_desc, synth_fraction
pkg = ge.load_package()
P, C, PS, T = 39, 51, 50, 150
layers = net_desc(P, [("blstm", 156), ("blstm", 300), ("blstm", 102)], C)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, P, C, 113, T)
frames = pkg.fraction.real_frames(frac)
def step():
    net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate(); net.compute_backward_pass(); net.update_weights_fused(1e-5, 0.9)
for _ in range(3): step()
net.synchronize(); t0 = time.time()
n = 20
for _ in range(n): step()
net.synchronize(); dt = (time.time() - t0) / n
print("chime topology: %.3f ms per fraction, %.2f M frames/s" % (dt * 1e3, frames / dt / 1e6), "error", net.loss_read())
net.close()
