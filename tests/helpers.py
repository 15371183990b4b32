"""Shared helpers of the test suite: fixtures on disk, synthetic networks and fractions."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kat0():
    """KAT-0 inputs: test1 network + first 10 sequences of val_1_speaker.nc."""
    z = np.load(os.path.join(GOLDEN, "kat0_test1.npz"))
    layers = json.loads(str(z["layers_json"]))
    weights = {}
    for k in z.files:
        if k.startswith("w/"):
            _, name, key = k.split("/")
            weights.setdefault(name, {})[key] = z[k]
    lens = z["seqLengths"]
    off = np.concatenate([[0], np.cumsum(lens)])
    xs = [z["inputs"][off[i]:off[i + 1]] for i in range(len(lens))]
    ts = [z["targetClasses"][off[i]:off[i + 1]] for i in range(len(lens))]
    return layers, weights, xs, ts


def lstm_weight_count(P, L, bidir):
    return L * (4 * (P + 1) + (2 if bidir else 4) * L + 3)


def random_weights(layers, rng, scale=0.1):
    """Explicit weights for every trainable layer, split like the JSON sections."""
    out = {}
    prev = None
    for d in layers:
        t, L = d["type"], int(d["size"])
        if t in ("lstm", "blstm"):
            P = int(prev["size"])
            H = L // (2 if t == "blstm" else 1)
            out[d["name"]] = {"input": rng.uniform(-scale, scale, 4 * L * P).astype(np.float32),
                              "bias": rng.uniform(-scale, scale, 4 * L).astype(np.float32),
                              "internal": rng.uniform(-scale, scale, 4 * L * H + 3 * L).astype(np.float32)}
        elif t == "softmax" or t.startswith("feedforward"):
            P = int(prev["size"])
            out[d["name"]] = {"input": rng.uniform(-scale, scale, L * P).astype(np.float32),
                              "bias": rng.uniform(-scale, scale, L).astype(np.float32),
                              "internal": np.zeros(0, np.float32)}
        prev = d
    return out


def net_desc(P, hidden, C, post="multiclass_classification", bias=1.0):
    """hidden: list of (type, size) tuples."""
    layers = [{"name": "input", "type": "input", "size": P}]
    for i, (t, s) in enumerate(hidden):
        layers.append({"name": "%s_%d" % (t, i), "type": t, "size": s, "bias": bias})
    layers.append({"name": "output", "type": "softmax" if post == "multiclass_classification" else "feedforward_identity",
                   "size": C, "bias": bias})
    layers.append({"name": "postoutput", "type": post, "size": C})
    return layers


def random_sequences(rng, lengths, P, C=None, L=None):
    xs = [rng.randn(n, P).astype(np.float32) for n in lengths]
    if C is not None:
        ts = [rng.randint(0, C, n).astype(np.int32) for n in lengths]
    else:
        ts = [rng.randn(n, L).astype(np.float32) for n in lengths]
    return xs, ts


def real_mask(frac):
    return np.asarray(frac["patTypes"]).reshape(-1) != 0


def load_ref_golden(name):
    """One case of tests/golden/ref_golden.npz (outputs of the reference's own compiled functors, made by
    tests/golden/make_ref_golden.py): (layers, weights, xs, ts, PS, expected dict)."""
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    pre = name + "/"
    layers = json.loads(str(z[pre + "layers_json"]))
    weights, exp = {}, {}
    for k in z.files:
        if not k.startswith(pre):
            continue
        parts = k[len(pre):].split("/")
        if parts[0] == "w":
            weights.setdefault(parts[1], {})[parts[2]] = z[k]
        elif parts[0] not in ("layers_json", "PS", "seqLengths", "inputs", "targetClasses", "targets"):
            exp["/".join(parts)] = z[k]
    lens = z[pre + "seqLengths"]
    off = np.concatenate([[0], np.cumsum(lens)])
    xs = [z[pre + "inputs"][off[i]:off[i + 1]] for i in range(len(lens))]
    tkey = pre + ("targetClasses" if pre + "targetClasses" in z.files else "targets")      # real-valued targets: sse case
    ts = [z[tkey][off[i]:off[i + 1]] for i in range(len(lens))]
    return layers, weights, xs, ts, int(z[pre + "PS"]), exp
