"""Real-data convergence of the three arithmetic modes through the C++ driver: the literal CHiME example network
(examples/speech_recognition_chime/no_subsampling/network.jsn: 39 -> blstm156 -> blstm300 -> blstm102 -> softmax51, BASELINE
configs[2]) on the reference's one real data file (tests/golden/val_1_speaker.nc, 102 sequences / 13 878 frames), split 90 / 12
into a training and a validation file, identical initial weights (normal, sigma 0.1, like the example's config.cfg), same options,
`--precision f32 | bf16x3 | bf16`.  Prints one JSON object with the three curves (training / validation error and class error per
epoch); tests/test_gpu_configs.py asserts the bf16 figures against the f32 ones, DESIGN.md section 3 quotes them.

    python tools/chime_convergence.py [--epochs 20] [--ps 10] [--lr 1e-4] [--out gpurun_out/chime_convergence.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
from scipy.io import netcdf_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NC = os.path.join(ROOT, "tests", "golden", "val_1_speaker.nc")
BIN = os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip")

LAYERS = [{"size": 39, "name": "input", "type": "input"},
          {"size": 156, "name": "blstm_level_0", "bias": 1.0, "type": "blstm"},
          {"size": 300, "name": "blstm_level_1", "bias": 1.0, "type": "blstm"},
          {"size": 102, "name": "blstm_level_2", "bias": 1.0, "type": "blstm"},
          {"size": 51, "name": "output", "bias": 1.0, "type": "softmax"},
          {"size": 51, "name": "postoutput", "type": "multiclass_classification"}]


def read_sequences(path=NC):
    f = netcdf_file(path, "r", mmap=False)
    lens = np.array(f.variables["seqLengths"][:], np.int64)
    x = np.array(f.variables["inputs"][:], np.float32)
    t = np.array(f.variables["targetClasses"][:], np.int32)
    tags = ["".join(c.decode() for c in row).rstrip("\0") for row in f.variables["seqTags"][:]]
    num_labels = f.dimensions["numLabels"]
    f.close()
    off = np.concatenate([[0], np.cumsum(lens)])
    return [x[off[i]:off[i + 1]] for i in range(len(lens))], [t[off[i]:off[i + 1]] for i in range(len(lens))], tags, num_labels


def write_nc(path, xs, ts, tags, num_labels):
    """CURRENNT classification file (reference README:600-646)."""
    f = netcdf_file(path, "w")
    width = max(len(t) for t in tags) + 1
    f.createDimension("numSeqs", len(xs)); f.createDimension("numTimesteps", sum(len(x) for x in xs))
    f.createDimension("inputPattSize", xs[0].shape[1]); f.createDimension("numLabels", num_labels)
    f.createDimension("maxSeqTagLength", width)
    v = f.createVariable("seqTags", "c", ("numSeqs", "maxSeqTagLength"))
    for i, t in enumerate(tags):
        v[i] = np.array(list(t.ljust(width, "\0")), "c")
    v = f.createVariable("seqLengths", "i", ("numSeqs",)); v[:] = np.array([len(x) for x in xs], np.int32)
    v = f.createVariable("targetClasses", "i", ("numTimesteps",)); v[:] = np.concatenate(ts).astype(np.int32)
    v = f.createVariable("inputs", "f", ("numTimesteps", "inputPattSize")); v[:] = np.concatenate(xs).astype(np.float32)
    f.close()


def initial_weights(seed=1, sigma=0.1):
    rng = np.random.RandomState(seed)
    out, prev = {}, None
    for d in LAYERS:
        L = d["size"]
        if d["type"] == "blstm":
            P, H = prev["size"], L // 2
            out[d["name"]] = {"input": rng.normal(0, sigma, 4 * L * P), "bias": rng.normal(0, sigma, 4 * L), "internal": rng.normal(0, sigma, 4 * L * H + 3 * L)}
        elif d["type"] == "softmax":
            out[d["name"]] = {"input": rng.normal(0, sigma, L * prev["size"]), "bias": rng.normal(0, sigma, L), "internal": np.zeros(0)}
        prev = d
    return {k: {a: np.asarray(b, np.float32).tolist() for a, b in w.items()} for k, w in out.items()}


def parse_table(text):
    rows = []
    for l in text.splitlines():
        if l.strip()[:1].isdigit() and l.count("|") >= 5:
            c = [x.strip() for x in l.split("|")]
            tr = [float(v.rstrip("%")) for v in c[2].split()]
            va = [float(v.rstrip("%")) for v in c[3].split()]
            rows.append({"epoch": int(c[0]), "train_class_err": tr[0], "train_err": tr[1], "val_class_err": va[0], "val_err": va[1], "best": c[5]})
    return rows


def run(workdir, epochs=20, ps=10, lr=1e-4, momentum=0.9, n_train=90, modes=("f32", "bf16x3", "bf16"), extra=("--deterministic", "true")):
    """`extra`: further driver options; by default every mode sums its gradients in a fixed order (bf16 opts in), so a run is
    reproducible bit for bit and a difference between two modes is the arithmetic's, not the atomics'."""
    xs, ts, tags, num_labels = read_sequences()
    train, val = os.path.join(workdir, "train.nc"), os.path.join(workdir, "val.nc")
    write_nc(train, xs[:n_train], ts[:n_train], tags[:n_train], num_labels)
    write_nc(val, xs[n_train:], ts[n_train:], tags[n_train:], num_labels)
    net = os.path.join(workdir, "network.jsn")
    json.dump({"layers": LAYERS, "weights": initial_weights()}, open(net, "w"))
    result = {"network": "39-blstm156-blstm300-blstm102-softmax51", "train_sequences": n_train, "val_sequences": len(xs) - n_train,
              "train_frames": int(sum(len(x) for x in xs[:n_train])), "val_frames": int(sum(len(x) for x in xs[n_train:])),
              "parallel_sequences": ps, "learning_rate": lr, "momentum": momentum, "epochs": epochs, "modes": {}}
    for mode in modes:
        out = subprocess.run([BIN, "--train", "true", "--stochastic", "true", "--train_file", train, "--val_file", val, "--network", net,
                              "--parallel_sequences", str(ps), "--max_epochs", str(epochs), "--max_epochs_no_best", str(epochs + 1),
                              "--learning_rate", str(lr), "--momentum", str(momentum), "--precision", mode, "--random_seed", "1",
                              "--save_network", os.path.join(workdir, "trained_%s.jsn" % mode)] + list(extra),
                             capture_output=True, text=True, timeout=1200)
        if out.returncode != 0:
            raise RuntimeError("driver failed in mode %s:\n%s\n%s" % (mode, out.stdout[-2000:], out.stderr[-2000:]))
        result["modes"][mode] = parse_table(out.stdout)
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20); ap.add_argument("--ps", type=int, default=10)
    ap.add_argument("--lr", type=float, default=1e-4); ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        res = run(d, a.epochs, a.ps, a.lr, a.momentum)
    for mode, rows in res["modes"].items():
        print(mode, " ".join("%.2f/%.1f" % (r["val_class_err"], r["val_err"]) for r in rows), file=sys.stderr)
    text = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text + "\n")
    print(text)
