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
    # name: P, hidden, C, lengths, PS, weight scale, seed, post output layer, SGD steps whose result is kept (0: none)
    "blstm_stack": dict(P=13, hidden=[("blstm", 24), ("blstm", 16)], C=9, lengths=[17, 15, 15, 8, 3], PS=6, scale=0.3, seed=101, train=10),
    "lstm_tanh": dict(P=7, hidden=[("feedforward_tanh", 10), ("lstm", 20)], C=5, lengths=[12, 11, 4], PS=3, scale=0.4, seed=102, train=10),
    # the shapes the HIP kernels specialise on: Hp = 128 register-resident recurrent kernels (the headline layer width) ...
    # (two stacked: K = 256 input projection and the error to a preceding LSTM layer; weights on a 2^-10 grid so that the
    # fixture compresses -- any float32 is a valid input) ...
    "blstm250_x2": dict(P=39, hidden=[("blstm", 250), ("blstm", 250)], C=183, lengths=[12, 12, 11, 9, 6, 2], PS=6, scale=0.08, seed=104, grid=1024, train=0),
    # ... Hp = 256: multi-CU cluster kernels (bf16: 2 x 128 units, split-bf16: 4 x 64), streamed W_rec in exact-fp32 mode
    "blstm500_cluster": dict(P=39, hidden=[("blstm", 500)], C=61, lengths=[10, 8, 5], PS=4, scale=0.06, seed=105, grid=1024, train=0),
    # SsePostOutputLayer behind a linear output layer (regression targets)
    "sse_regression": dict(P=9, hidden=[("lstm", 32), ("feedforward_logistic", 12)], C=4, lengths=[14, 9, 9, 2], PS=5, scale=0.4, seed=106,
                           post="sse", train=10),
    # Q3: softmax rows whose logits are all negative (centred on min/2), large logits into both safeExp clamps
    "softmax_q3": dict(P=4, hidden=[("lstm", 5)], C=9, lengths=[8, 5], PS=3, scale=0.5, seed=107, tweak="q3", train=0),
    # T = 1 (the only step is first and last of both directions)
    "t1": dict(P=6, hidden=[("blstm", 14)], C=5, lengths=[1, 1, 1], PS=4, scale=0.4, seed=108, train=5),
    # a partial fraction: three sequences in eight parallel slots
    "partial_fraction": dict(P=8, hidden=[("blstm", 64), ("lstm", 32)], C=7, lengths=[9, 6, 6], PS=8, scale=0.3, seed=109, train=5),
}


def build_case(name):
    """(layers, weights, xs, ts, PS, post) of one case -- everything drawn from the case's seed."""
    c = CASES[name]
    rng = np.random.RandomState(c["seed"])
    post = c.get("post", "multiclass_classification")
    layers = net_desc(c["P"], c["hidden"], c["C"], post=post)
    weights = random_weights(layers, rng, c["scale"])
    if c.get("grid"):
        for w in weights.values():
            for k in w:
                w[k] = (np.round(np.asarray(w[k], np.float32) * c["grid"]) / c["grid"]).astype(np.float32)
    if c.get("tweak") == "q3":
        H = c["hidden"][-1][1]
        weights["output"]["input"] = rng.uniform(-60, 60, c["C"] * H).astype(np.float32)
        weights["output"]["bias"] = (-np.abs(rng.uniform(50, 200, c["C"]))).astype(np.float32)
    if post == "sse":
        xs, ts = random_sequences(rng, c["lengths"], c["P"], L=c["C"])
    else:
        xs, ts = random_sequences(rng, c["lengths"], c["P"], C=c["C"])
    return layers, weights, xs, ts, c["PS"], post


def main():
    import subprocess
    pkg, orc = ge.load_package(), ge.load_oracle()
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"), "_ref"])
    out = {}
    for name, c in CASES.items():
        layers, weights, xs, ts, PS, post = build_case(name)
        classification = post != "sse"
        frac = pkg.make_fraction(xs, ts, PS, classification=classification)
        net = orc.OracleNetwork(layers, weights, PS, frac["T"], backend="ref")
        net.load_sequences(frac); net.compute_forward_pass()
        err = net.calculate_error()
        cor = net.count_correct_classifications() if classification else -1
        net.compute_backward_pass()
        out[name + "/layers_json"] = np.array(json.dumps(layers))
        out[name + "/PS"] = np.int32(PS)
        out[name + "/seqLengths"] = np.array(c["lengths"], np.int32)
        out[name + "/inputs"] = np.concatenate(xs).astype(np.float32)
        if classification:
            out[name + "/targetClasses"] = np.concatenate(ts).astype(np.int32)
        else:
            out[name + "/targets"] = np.concatenate(ts).astype(np.float32)
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
        # momentum-SGD steps on the same fraction: trained weights (update = the oracle's 4-line UpdateWeightFn restatement --
        # SteepestDescentOptimizer.cu pulls in Boost and cannot be compiled here; gradients from the reference functors)
        for _ in range(c["train"]):
            net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass(); net.update_weights(5e-3, 0.9)
        if c["train"]:
            out[name + "/trainSteps"] = np.int32(c["train"])
            for lay in net.trainable_layers():
                out["%s/trained/%s" % (name, lay.name)] = lay.weights.copy()
    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
