"""Segment timing inside the delta-exchange backward cluster kernel (s_memtime deltas summed per step segment by the waves of
workgroup 0 = member 0 of cluster 0).  Needs `make -C lstm-rnn_amd/csrc variantc NAME=clstamp DEFS=-DCN_CL_STAMP`; on the GPU box:
    CURRENNT_HIP_LIB=lstm-rnn_amd/libcurrennt_hip_clstamp.so python tools/stamps_cl.py [size] [PS] [T]
Segments: 0 stage copies (step top -> first product issued), 1 own part of the product done, 2 partners' deltas polled,
3 their LDS write + barrier, 4 second part of the product done (incl. the prefetch issue), 5 block errors + publish + stores
issued, 6 barrier."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from bench import make_weights, net_desc, synth_fraction  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 500
PS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pkg = ge.load_package()
layers = net_desc(39, [("blstm", size)] * 2, 183)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, 39, 183, T, T)
import time
for _ in range(3):
    net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.synchronize()
print("kernels:", net.recurrent_kernel(False), net.recurrent_kernel(True))
lib = net.lib
buf = (C.c_ulonglong * 64)()
lib.cn_dbg_read_stamps_cl.argtypes = [C.c_void_p]
assert lib.cn_dbg_read_stamps_cl(buf) == 0
a = np.array(buf, np.float64).reshape(8, 8) / T
print("ticks of s_memtime per step by segment (rows = waves of workgroup 0), total last:")
for w in range(8):
    if a[w].sum() > 0:
        print("  wave %d  " % w + "  ".join("%7.1f" % v for v in a[w, :7]) + "   | %8.1f" % a[w, :7].sum())
if hasattr(lib, "cn_dbg_read_stamps_cl_fwd"):
    lib.cn_dbg_read_stamps_cl_fwd.argtypes = [C.c_void_p]
    assert lib.cn_dbg_read_stamps_cl_fwd(buf) == 0
    a = np.array(buf, np.float64).reshape(8, 8) / T
    print("forward cluster kernel (segments: 0 top, 1 own part done, 2 polled, 3 LDS write + barrier, 4 partners' parts done, 5 cell update + publish + stores, 6 barrier):")
    for w in range(8):
        if a[w].sum() > 0:
            print("  wave %d  " % w + "  ".join("%7.1f" % v for v in a[w, :7]) + "   | %8.1f" % a[w, :7].sum())
net.close()
