"""Stage-by-stage diff of the HIP path against the oracle on a tiny network (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
from helpers import net_desc, random_sequences, random_weights, real_mask

pkg, orc = ge.load_package(), ge.load_oracle()
kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.RandomState(11)
P, C, PS = 7, 5, 3
layers = net_desc(P, [(kind, size)], C)
weights = random_weights(layers, rng, 0.4)
xs, ts = random_sequences(rng, [20, 17, 9], P, C=C)
frac = pkg.make_fraction(xs, ts, PS)
ref = orc.OracleNetwork(layers, weights, PS, frac["T"])
ref.load_sequences(frac); ref.compute_forward_pass(); print("ref err", ref.calculate_error()); ref.compute_backward_pass()
net = pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=prec)
net.load_sequences(frac); net.compute_forward_pass(); print("hip err", net.error_and_correct()); net.compute_backward_pass()
real = real_mask(frac)
lay, rl = net.layers[1], ref.layers[1]
T = frac["T"]
for d in range(lay.dirs):
    for name in ("niActs", "igActs", "fgActs", "ogActs", "cellStates", "tmpOutputs", "niDeltas", "igDeltas", "fgDeltas", "ogDeltas"):
        a = lay.internal(name, d).reshape(T, PS, lay.H)
        b = rl.internal(name, d)[:net.N * lay.H].reshape(T, PS, lay.H)
        m = real.reshape(T, PS)
        diff = np.abs(a - b) * m[:, :, None]
        per_t = diff.reshape(T, -1).max(1)
        print(d, name, "max", diff.max(), "first bad t", int(np.argmax(per_t > 1e-4)) if (per_t > 1e-4).any() else -1,
              "t0", per_t[0], "tlast", per_t[-1])
y = net.outputs().reshape(-1, C)[real]; yr = ref.outputs().reshape(-1, C)[real]
print("posterior max diff", np.abs(y - yr).max())
for l in net.trainable_layers():
    g, gr = l.weight_updates(), ref.layer(l.name).weightUpdates
    print(l.name, "grad rel", np.abs(g - gr).max() / np.abs(gr).max())
    if l.type in ("lstm", "blstm"):
        Pp, L, H = l.prev.size, l.size, l.H
        n_in, n_b, n_rec = 4 * L * Pp, 4 * L, 4 * L * H
        for nm, a, b in (("input", 0, n_in), ("bias", n_in, n_in + n_b), ("rec", n_in + n_b, n_in + n_b + n_rec), ("peep", n_in + n_b + n_rec, g.size)):
            print("   ", nm, np.abs(g[a:b] - gr[a:b]).max() / max(1e-12, np.abs(gr[a:b]).max()))
