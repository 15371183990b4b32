"""Generates tests/golden/kat0_test1.npz -- runs in the build container only.

Inputs are DATA FILES of the reference's own test and example (no reference source code):
  /root/reference/tests/test1/network.jsn                        layers + explicit weights
  /root/reference/examples/speech_recognition_chime/val_1_speaker.nc   first 10 sequences, file order
This is KAT-0 of SURVEY.md Appendix A; the expected values recorded there (from the reference's
own Cpu build) are asserted in tests/test_oracle_kat0.py.
"""
import json
import os

import numpy as np
from scipy.io import netcdf_file

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    net = json.load(open(os.path.join(REF, "tests/test1/network.jsn")))
    f = netcdf_file(os.path.join(REF, "examples/speech_recognition_chime/val_1_speaker.nc"), "r", mmap=False)
    lens = np.array(f.variables["seqLengths"].data[:10], np.int32)
    n = int(lens.sum())
    x = np.array(f.variables["inputs"].data[:n], np.float32)
    tc = np.array(f.variables["targetClasses"].data[:n], np.int32)
    out = {"layers_json": np.array(json.dumps(net["layers"])), "seqLengths": lens, "inputs": x, "targetClasses": tc}
    for name, w in net["weights"].items():
        for key in ("input", "bias", "internal"):
            out["w/%s/%s" % (name, key)] = np.asarray(w[key], np.float32)
    np.savez_compressed(os.path.join(HERE, "kat0_test1.npz"), **out)
    print("frames", n, "sum inputs", x.sum())


if __name__ == "__main__":
    main()
