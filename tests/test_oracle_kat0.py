"""Pins the CPU oracle against KAT-0: values recorded in SURVEY.md Appendix A from the reference's
own Cpu instantiation (tests/test1/network.jsn on the first 10 sequences of val_1_speaker.nc,
one forward + backward, no update)."""
import numpy as np

from helpers import load_kat0

# SURVEY.md Appendix A, table "KAT-0"
KAT0_ERROR = 5293.397461
KAT0_CORRECT = 126
KAT0_LAYERS = {   # name: (#weights, sum outputs, sum dW, ||dW||_2)
    "blstm_level_0":     (1830, 1.988155e+01, 2.815368e-03, 3.940761e-03),
    "subsample_level_0": (55, -1.816304e+01, 1.069653e-03, 3.377306e-03),
    "blstm_level_1":     (470, 3.017492e+01, 5.380240e-02, 3.435415e-01),
    "subsample_level_1": (55, 7.334888e-03, 8.634804e-02, 2.900339e-01),
    "blstm_level_2":     (470, -2.027932e+02, 4.609958e+01, 3.138710e+01),
    "output":            (561, 1.175607e+03, -4.359866e-03, 2.528478e+02),
}


def test_kat0_inputs(pkg):
    layers, weights, xs, ts = load_kat0()
    assert [len(x) for x in xs] == [130, 140, 141, 142, 130, 123, 130, 130, 127, 152]
    frac = pkg.make_fraction(xs, ts, 10)
    assert frac["T"] == 152 and frac["Tmin"] == 123
    assert int((frac["patTypes"] != 0).sum()) == 1345 and int((frac["patTypes"] == 0).sum()) == 175
    assert abs(float(frac["inputs"].sum()) - 515.2587) < 1e-3


def test_kat0_oracle(pkg, orc):
    layers, weights, xs, ts = load_kat0()
    frac = pkg.make_fraction(xs, ts, 10)
    net = orc.OracleNetwork(layers, weights, 10, frac["T"])
    net.load_sequences(frac)
    net.compute_forward_pass()
    err = net.calculate_error()
    cor = net.count_correct_classifications()
    net.compute_backward_pass()
    assert abs(err - KAT0_ERROR) < 6e-4, err          # one fp32 ulp at 5293 (4.9e-4): the recorded value is matched to the last bit
    assert cor == KAT0_CORRECT
    N = frac["T"] * 10
    for name, (nw, s_out, s_dw, n_dw) in KAT0_LAYERS.items():
        lay = net.layer(name)
        assert lay.weights.size == nw
        out = lay.outputs[:N * lay.size].astype(np.float64)
        dw = lay.weightUpdates.astype(np.float64)
        # the survey printed 7 significant digits: 1e-6 relative is what that printing allows (measured: <= 1.4e-7 on every
        # ||dW||); sums that cancel (subsample_level_1's outputs add up to 7e-3 from terms of order 1) get an absolute term
        # scaled by the sum of magnitudes
        assert abs(out.sum() - s_out) <= 1e-6 * abs(s_out) + 5e-7 * np.abs(out).sum(), (name, out.sum(), s_out)
        assert abs(dw.sum() - s_dw) <= 1e-6 * abs(s_dw) + 1e-6 * np.abs(dw).sum(), (name, dw.sum(), s_dw)
        assert abs(np.sqrt((dw * dw).sum()) - n_dw) <= 1e-6 * n_dw, (name, np.sqrt((dw * dw).sum()), n_dw)
