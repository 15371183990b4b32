"""Segment timing inside the hand-written two-CU backward loop (cn_lstm_cluster.hip: lstm_bwd_s2c_asm_kernel; s_memtime deltas summed
per step segment by every wave of workgroup 0).  Needs `make -C lstm-rnn_amd/csrc variantc NAME=s2cstamp DEFS=-DCN_S2C_STAMP`; on
the GPU box:   CURRENNT_HIP_LIB=lstm-rnn_amd/libcurrennt_hip_s2cstamp.so python tools/stamps_s2c.py [H] [PS] [T] [layers]
Segments (cycles per step; each ends with a stamp that waits for lgkmcnt(0), ~45 cycles by itself, and perturbs the schedule):
0 barrier + loop control   1 poll / prefetch issue + own K half (16 reads, 16 MFMAs)   2 waiting for the poll
3 partner rows -> LDS + barrier   4 partner K half   5 sums, block errors, publish, stores   6 wait for the next stage + its block"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from bench import make_weights, net_desc, synth_fraction  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 500
PS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # layers: the stamps are those of the LAST backward launch = the lowest layer
pkg = ge.load_package()
layers = net_desc(39, [("blstm", H)] * NL, 183)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, 39, 183, T, T)
for _ in range(3):
    net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.synchronize()
print(net.recurrent_kernel(True))
lib = net.lib
buf = (C.c_uint * 32)()
lib.cn_dbg_read_stamps_s2c.argtypes = [C.c_void_p]
assert lib.cn_dbg_read_stamps_s2c(buf) == 0
raw = np.array(buf, np.float64).reshape(4, 8)
a = raw / T
for w in range(4):
    print("  wave %d  " % w + "  ".join("%7.1f" % v for v in a[w, :7]) + "   | %8.1f" % a[w, :7].sum()
          + "   in-kernel clock %.0f MHz" % (raw[w, :7].sum() / max(raw[w, 7], 1) * 100))
net.close()
