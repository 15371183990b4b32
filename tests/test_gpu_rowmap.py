"""-m gpu: the fraction's row map (GemmNT::rowmap, cn_internal.h; option "no_nt_rowmap" switches it off; by default the panel
kernel takes it -- the error products of the headline step --, with "nt_rowmap_tiled" gemm_nt_kernel does too: both run here).

The reference multiplies ALL T x PS frames of a fraction in its N-wide products (LstmLayer.cu:771-786,990-1009,
FeedForwardLayer.cu:143-160,188-198), dummy ones included; every operand row of a dummy frame is zero there (y = 0 and
deltas = 0 by the checkPatType rule), so the product's rows for them are bias[n] (or 0).  The one-panel-per-CU kernel
(cn_gemm_nt_panel.hip) multiplies the REAL rows only and writes bias / 0 into the dummy rows.  That must leave every bit of
every visible result where it was: posteriors, errors, gradients and weights of a short training run, with sequences of very
different lengths (many dummy frames), a fraction with empty slots, and a fraction where nothing is dummy."""
import numpy as np
import pytest

from helpers import net_desc, random_weights

pytestmark = pytest.mark.gpu


def _fractions(pkg, rng, P, C, PS, lens_list):
    fracs = []
    for lens in lens_list:
        xs = [rng.randn(n, P).astype(np.float32) for n in lens]
        ts = [rng.randint(0, C, n).astype(np.int32) for n in lens]
        fracs.append(pkg.make_fraction(xs, ts, PS))
    return fracs


def _run(pkg, layers, weights, fracs, PS, T, off, mode):
    prec = {"bf16": pkg.PREC_BF16, "bf16x3": pkg.PREC_BF16X3}[mode]
    with pkg.NeuralNetwork(layers, weights, PS, T, precision=prec, deterministic=True) as net:
        net.set_option("no_nt_rowmap", off)
        net.set_option("nt_rowmap_tiled", 1)      # (by default only the panel kernel takes the map; the tiled kernel's path is tested here too)
        out = {"err": [], "post": [], "oerr": []}
        for k, f in enumerate(fracs * 2):
            net.load_sequences(f); net.compute_forward_pass(); out["err"].append(net.calculate_error())
            out["post"].append(net.outputs().copy())
            net.compute_backward_pass()
            out["oerr"].append([l.output_errors().copy() for l in net.trainable_layers()])
            if k == len(fracs) * 2 - 1:
                out["g"] = [l.weight_updates().copy() for l in net.trainable_layers()]
            net.update_weights(1e-3, 0.9)
        out["w"] = [l.weights().copy() for l in net.trainable_layers()]
    return out


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_row_map_changes_no_visible_bit(pkg, mode):
    rng = np.random.RandomState(17)
    P, C, PS, T = 39, 183, 50, 300
    layers = net_desc(P, [("blstm", 250)] * 3, C)
    weights = random_weights(layers, rng, 0.1)
    lens_list = [np.sort(rng.randint(250, 301, PS)),                 # the headline's fractions: ~10 % dummy frames
                 np.sort(rng.randint(20, 301, PS)),                  # half of the frames are dummy
                 np.sort(rng.randint(100, 281, 37)),                 # 13 empty slots: dummy only from the shortest sequence's end on
                 np.full(PS, 264)]                                   # nothing is dummy
    fracs = _fractions(pkg, rng, P, C, PS, lens_list)
    on = _run(pkg, layers, weights, fracs, PS, T, 0, mode)
    off = _run(pkg, layers, weights, fracs, PS, T, 1, mode)
    assert on["err"] == off["err"], [a - b for a, b in zip(on["err"], off["err"])]
    for k, (a, b) in enumerate(zip(on["post"], off["post"])):
        assert np.array_equal(a, b), (k, np.abs(a - b).max())
    for k, (la, lb) in enumerate(zip(on["oerr"], off["oerr"])):
        for i, (a, b) in enumerate(zip(la, lb)):
            assert np.array_equal(a, b), ("outputErrors", k, i, np.abs(a - b).max())
    for name in ("g", "w"):
        for i, (a, b) in enumerate(zip(on[name], off[name])):
            assert np.array_equal(a, b), (name, i, np.abs(a - b).max())


def test_row_map_lists_the_frames_nobody_checks_as_real(pkg):
    """The map itself (cn_dbg_row_map_counts): real = every frame in front of the shortest sequence's end (nobody checks patTypes
    there: empty and pad slots are computed like any other, by the reference too) + the frames whose patType is not NONE behind
    it; dummy = the rest; together all T x padded-PS rows."""
    rng = np.random.RandomState(3)
    P, C, PS, T = 39, 183, 50, 120
    layers = net_desc(P, [("blstm", 250)] * 2, C)
    weights = random_weights(layers, rng, 0.1)
    lens = np.sort(rng.randint(40, 121, 44))
    frac = _fractions(pkg, rng, P, C, PS, [lens])[0]
    with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_BF16) as net:
        net.load_sequences(frac)
        counts = net.row_map_counts()
    tmax, tmin = int(lens.max()), int(lens.min())
    real = int(lens.sum()) + (PS - len(lens)) * tmin        # empty slots count as real while nobody checks patTypes
    assert counts[0] >= real and counts[0] + counts[1] == counts[2]
    assert counts[0] - real == (counts[2] // tmax - PS) * tmin          # pad slots of the device layout, the same rule
