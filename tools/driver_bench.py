"""Throughput of the C++ driver (lstm-rnn_amd/currennt_hip) on a synthetic TIMIT-shaped NetCDF file:
39 -> 3 x blstm250 -> softmax183, parallel_sequences 50, stochastic momentum SGD, bf16, 3 epochs."""
import json, os, re, subprocess, sys, tempfile, time
import numpy as np
from scipy.io import netcdf_file
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import net_desc, WORKLOADS
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 400
wlname = sys.argv[2] if len(sys.argv) > 2 else "timit_3x250_blstm_H125"
rng = np.random.RandomState(0)
wl = WORKLOADS[wlname]; P_, C_, PS_ = wl["P"], wl["C"], wl.get("PS", 50)
lens = rng.randint(250, 351, nseq)
n = int(lens.sum())
d = tempfile.mkdtemp()
nc = os.path.join(d, "train.nc")
f = netcdf_file(nc, "w")
f.createDimension("numSeqs", nseq); f.createDimension("numTimesteps", n); f.createDimension("inputPattSize", P_)
f.createDimension("numLabels", C_); f.createDimension("maxSeqTagLength", 16)
tags = f.createVariable("seqTags", "c", ("numSeqs", "maxSeqTagLength"))
for i in range(nseq): tags[i] = np.array(list(("s%05d" % i).ljust(16, "\0")), "c")
f.createVariable("seqLengths", "i", ("numSeqs",))[:] = lens.astype(np.int32)
f.createVariable("targetClasses", "i", ("numTimesteps",))[:] = rng.randint(0, C_, n).astype(np.int32)
f.createVariable("inputs", "f", ("numTimesteps", "inputPattSize"))[:] = rng.randn(n, P_).astype(np.float32)
f.close()
wl = WORKLOADS[wlname]
net = os.path.join(d, "network.jsn")
json.dump({"layers": net_desc(wl["P"], wl["hidden"], wl["C"])}, open(net, "w"))
cmd = [os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip"), "--train", "true", "--stochastic", "true", "--train_file", nc, "--network", net,
       "--parallel_sequences", str(PS_), "--max_epochs", "3", "--learning_rate", "1e-5", "--momentum", "0.9", "--precision", "bf16",
       "--save_network", os.path.join(d, "trained.jsn"), "--random_seed", "1"]
t0 = time.time()
out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
print(out.stdout[-1200:])
rows = re.findall(r"^\s*(\d+) \|\s*([\d.]+) \|", out.stdout, re.M)
for ep, dur in rows:
    print("epoch %s: %.1f s -> %.2f M frames/s" % (ep, float(dur), n / max(float(dur), 1e-3) / 1e6))
print("frames per epoch", n, "wall", time.time() - t0)
