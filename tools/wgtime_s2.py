"""Loop time of every workgroup of the LAST backward launch of the hand-written s2 loop (diag build:
make -C lstm-rnn_amd/csrc variants2 NAME=wgtime DEFS=-DCN_S2_WGTIME): which workgroups are the slow ones, and where they sit.
    CURRENNT_HIP_LIB=lstm-rnn_amd/libcurrennt_hip_wgtime.so python tools/wgtime_s2.py [layers]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from bench import make_weights, net_desc, synth_fraction  # noqa: E402
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pkg = ge.load_package()
layers = net_desc(39, [("blstm", 250)] * NL, 183)
PS, T = 50, 300
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, 39, 183, T, T)
for _ in range(4):
    net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.synchronize()
buf = (C.c_uint * 512)()
net.lib.cn_dbg_read_wgtime_s2.argtypes = [C.c_void_p]
assert net.lib.cn_dbg_read_wgtime_s2(buf) == 0
a = np.array(buf, np.int64).reshape(256, 2)[:52]
t = a[:, 0] / 100.0
print("loop time per workgroup [us]: min %.1f median %.1f max %.1f" % (t.min(), np.median(t), t.max()))
order = np.argsort(t)
for i in list(order[:4]) + list(order[-8:]):
    hw = a[i, 1]
    print("  wg %2d  %.1f us  xcc %d se %d sh %d cu %d" % (i, t[i], (hw >> 16) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15))
net.close()
