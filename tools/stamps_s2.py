"""Segment timing inside the hand-written backward loop of cn_lstm_s2.hip (s_memtime deltas summed per step segment by every
wave of workgroup 0).  Needs `make -C lstm-rnn_amd/csrc variants2 NAME=s2stamp DEFS=-DCN_S2_STAMP`; run on the GPU box:
    CURRENNT_HIP_LIB=lstm-rnn_amd/libcurrennt_hip_s2stamp.so python tools/stamps_s2.py [H] [PS] [T]
Segments (cycles per step, each ending with a stamp that waits for lgkmcnt(0), which perturbs the schedule):
0 barrier -> stage landed (vmcnt)   1 LDS reads issued + landed   2 16 MFMAs (+ fillers) issued   3 prefetch issued, MFMA
results summed   4 error arithmetic   5 LDS write, store, sums, write landed   6 barrier"""
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
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # layers: the stamps are those of the LAST backward launch = the lowest layer
pkg = ge.load_package()
layers = net_desc(39, [("blstm", H)] * NL, 183)
net = pkg.NeuralNetwork(layers, make_weights(layers, 1), PS, T, precision=pkg.PREC_BF16)
frac = synth_fraction(pkg, np.random.RandomState(0), PS, 39, 183, T, T)
for _ in range(3):
    net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.synchronize()
FWD = os.environ.get("STAMP_FWD") == "1"        # -DCN_S2_STAMP_F build: the forward loop (segments: barrier -> stage landed / LDS
print(net.recurrent_kernel(not FWD))              # operands + first gate's MFMAs / 12 MFMAs + fillers / tail / LDS write landed / barrier)
lib = net.lib
buf = (C.c_uint * 32)()
fn = lib.cn_dbg_read_stamps_s2f if FWD else lib.cn_dbg_read_stamps_s2
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
raw = np.array(buf, np.float64).reshape(4, 8)
a = raw / T
for w in range(4):
    print("  wave %d  " % w + "  ".join("%7.1f" % v for v in a[w, :7]) + "   | %8.1f" % a[w, :7].sum()
          + "   in-kernel clock %.0f MHz" % (raw[w, :7].sum() / max(raw[w, 7], 1) * 100))
net.close()
