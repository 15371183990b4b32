"""Repeat ONE backward pass many times on the multi-CU recurrent kernels and demand bit-identical gradients (option
"deterministic": every sum in a fixed order, so any difference is a wrong value that crossed a CU boundary -- a stale or
mis-tagged exchange granule).    python tools/xch_soak.py <size> <PS> <T> <layers> <repeats>"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from bench import make_weights, net_desc, synth_fraction  # noqa: E402

size, PS, T, nl, reps = (int(a) for a in sys.argv[1:6])
pkg = ge.load_package()
layers = net_desc(39, [("blstm", size)] * nl, 183)
bad = 0
with pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16, deterministic=True) as net:
    fracs = [synth_fraction(pkg, np.random.RandomState(s), PS, 39, 183, T - 7, T) for s in range(2)]
    ref = {}
    for k in range(reps):
        f = k % 2
        net.load_sequences(fracs[f]); net.compute_forward_pass(); net.compute_backward_pass()
        h = hashlib.md5(b"".join(l.weight_updates().tobytes() for l in net.trainable_layers())).hexdigest()
        if f not in ref:
            ref[f] = h
            print("kernels:", net.recurrent_kernel(False), net.recurrent_kernel(True))
        elif h != ref[f]:
            bad += 1
            print("repeat %d (fraction %d): gradients differ from the first pass" % (k, f))
print("%d of %d repeats differ" % (bad, reps))
sys.exit(1 if bad else 0)
