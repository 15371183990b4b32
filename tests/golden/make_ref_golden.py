"""Generates tests/golden/ref_golden.npz from oracle/_ref, i.e. from the REFERENCE's own compiled functors and Cpu GEMM
(oracle/ref/ref_common.h; needs /root/reference, so it runs in the build container only).  The fixture holds inputs and the
outputs the reference's code produced for them -- data, no source.  tests/test_oracle_golden.py checks the oracle (bit-exact)
and, on the GPU, the HIP path (fp32 tolerance) against it; it runs anywhere, also where neither /root/reference nor
oracle/_ref exist.

    python tests/golden/make_ref_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402
from helpers import net_desc, random_sequences, random_weights  # noqa: E402

CASES = {
    # name: (P, hidden, C, lengths, PS, weight scale, seed)
    "blstm_stack": (13, [("blstm", 24), ("blstm", 16)], 9, [17, 15, 15, 8, 3], 6, 0.3, 101),
    "lstm_tanh": (7, [("feedforward_tanh", 10), ("lstm", 20)], 5, [12, 11, 4], 3, 0.4, 102),
}


def main():
    import subprocess
    pkg, orc = ge.load_package(), ge.load_oracle()
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"), "_ref"])
    out = {}
    for name, (P, hidden, C, lengths, PS, scale, seed) in CASES.items():
        rng = np.random.RandomState(seed)
        layers = net_desc(P, hidden, C)
        weights = random_weights(layers, rng, scale)
        xs, ts = random_sequences(rng, lengths, P, C=C)
        frac = pkg.make_fraction(xs, ts, PS)
        net = orc.OracleNetwork(layers, weights, PS, frac["T"], backend="ref")
        net.load_sequences(frac); net.compute_forward_pass()
        err, cor = net.calculate_error(), net.count_correct_classifications()
        net.compute_backward_pass()
        out[name + "/layers_json"] = np.array(json.dumps(layers))
        out[name + "/PS"] = np.int32(PS)
        out[name + "/seqLengths"] = np.array(lengths, np.int32)
        out[name + "/inputs"] = np.concatenate(xs).astype(np.float32)
        out[name + "/targetClasses"] = np.concatenate(ts).astype(np.int32)
        for lname, w in weights.items():
            for k, v in w.items():
                out["%s/w/%s/%s" % (name, lname, k)] = np.asarray(v, np.float32)
        out[name + "/error"] = np.float32(err)
        out[name + "/correct"] = np.int32(cor)
        N = net.N
        for lay in net.layers[1:-1]:
            out["%s/outputs/%s" % (name, lay.name)] = lay.outputs[:N * lay.size].copy()
            out["%s/outputErrors/%s" % (name, lay.name)] = lay.outputErrors[:N * lay.size].copy()
            out["%s/weightUpdates/%s" % (name, lay.name)] = lay.weightUpdates.copy()
            if lay.type in ("lstm", "blstm"):
                for d in range(2 if lay.type == "blstm" else 1):
                    for b in ("cellStates", "igActs", "ogDeltas", "cellStateErrors"):
                        out["%s/internal/%s/%d/%s" % (name, lay.name, d, b)] = lay.internal(b, d)[:N * lay.H].copy()
        # ten momentum-SGD steps on the same fraction: trained weights (update = the oracle's 4-line UpdateWeightFn
        # restatement; gradients from the reference functors)
        for _ in range(10):
            net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.update_weights(5e-3, 0.9)
        for lay in net.trainable_layers():
            out["%s/trained/%s" % (name, lay.name)] = lay.weights.copy()
    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
