"""Synthetic LVCSR shape (BASELINE.json configs[3]): 40 -> 4 x blstm1024?  No: 4 x 512 BLSTM (H = 256) -> 8000 states,
PS = 64, T in [300, 800] -- here with a fixed T to keep the run short."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from bench import make_weights, net_desc, synth_fraction
pkg = ge.load_package()
P, C, PS = 40, 8000, 64
T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
layers = net_desc(P, [("blstm", 512)] * 4, C)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, P, C, T - 100, T)
frames = pkg.fraction.real_frames(frac)
def step():
    net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate(); net.compute_backward_pass(); net.update_weights_fused(1e-5, 0.9)
for _ in range(2): step()
net.synchronize()
net.timing_enable(True); net.timing_reset()
t0 = time.time(); n = 5
for _ in range(n): step()
net.synchronize(); dt = (time.time() - t0) / n
tm = net.timing_read()
print("LV shape T=%d: %.2f ms per fraction, %.3f M frames/s" % (T, dt * 1e3, frames / dt / 1e6))
print("  per-class ms per fraction:", {k: round(v[0] / n, 2) for k, v in tm.items()})
net.close()
