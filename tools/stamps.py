"""Segment timing inside the recurrent kernels (s_memtime deltas summed per step segment by every wave of
workgroup 0).  Needs `make -C lstm-rnn_amd/csrc stamp`; run on the GPU box:
    CURRENNT_HIP_LIB=lstm-rnn_amd/libcurrennt_hip_stamp.so python tools/stamps.py [H] [PS] [T]
Segments: 0 = step top -> prefetch issued, 1 = first LDS operand landed, 2 = MFMAs done, 3 = cell math done,
4 = stores issued, 5 = barrier passed."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from bench import make_weights, net_desc, synth_fraction  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 250
PS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pkg = ge.load_package()
layers = net_desc(39, [("blstm", H)], 183)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, 39, 183, T, T)
for _ in range(3):
    net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.synchronize()
lib = net.lib
buf = (C.c_ulonglong * (2 * 16 * 8))()
lib.cn_dbg_read_stamps.argtypes = [C.c_void_p]
assert lib.cn_dbg_read_stamps(buf) == 0
a = np.array(buf, np.float64).reshape(2, 16, 8) / T
for k, name in enumerate(("forward", "backward")):
    print(name, "cycles per step by segment (rows = waves of workgroup 0), total last:")
    for w in range(16):
        if a[k, w, :6].sum() > 0:
            print("  wave %2d  " % w + "  ".join("%7.1f" % v for v in a[k, w, :6]) + "   | %8.1f" % a[k, w, :6].sum())
net.close()
