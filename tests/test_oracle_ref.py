"""The oracle (oracle/currennt_oracle.c, a restatement) against oracle/_ref: the REFERENCE's own object code for the
arithmetic of the path -- helpers/Matrix.cu compiled as it lies, and the functors of LstmLayer.cu, FeedForwardLayer.cu,
SoftmaxLayer.cu and MulticlassClassificationLayer.cu driven in the reference's call order (oracle/ref/ref_common.h says
exactly what is and is not the reference's code).  Every comparison is BIT-exact: same statements, same summation order.

oracle/_ref is built where /root/reference exists (this container) and shipped as a binary to the GPU box; without it
these tests skip and tests/test_oracle_golden.py (fixtures generated from it, committed) carries the pin."""
import os

import numpy as np
import pytest

from helpers import load_kat0, net_desc, random_sequences, random_weights


@pytest.fixture(scope="module")
def ref(orc):
    if os.path.isdir("/root/reference/currennt_lib/src"):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.abspath(orc.__file__)), "_ref"])
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here and no prebuilt library)")
    return orc.ref_lib()


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("add", [0, 1])
def test_matrix_products_bit_equal(orc, ref, kind, add):
    """helpers::Matrix<Cpu>::assignProduct / addProduct (Matrix.cu:218-349) for the three transposition cases."""
    rng = np.random.RandomState(3 + kind)
    m, k, n = 7, 13, 5
    if kind == 0:   a, b = rng.randn(k, m), rng.randn(n, k)          # column-major A(m x k), B(k x n)
    elif kind == 1: a, b = rng.randn(m, k), rng.randn(n, k)          # A(k x m)^T, B(k x n)
    else:           a, b = rng.randn(k, m), rng.randn(k, n)          # A(m x k), B(n x k)^T
    a, b = a.astype(np.float32).reshape(-1), b.astype(np.float32).reshape(-1)
    rowsA, colsA = (m, k) if kind != 1 else (k, m)
    rowsB, colsB = (k, n) if kind != 2 else (n, k)
    c0 = rng.randn(m * n).astype(np.float32)
    c1, c2 = c0.copy(), c0.copy()
    orc.lib().orc_matmul(kind, c1, a, rowsA, colsA, b, rowsB, colsB, add)
    ref.orc_matmul(kind, c2, a, rowsA, colsA, b, rowsB, colsB, add)
    assert np.array_equal(c1, c2) and not np.array_equal(c1, c0)


def run(orc, backend, layers, weights, frac, PS, steps=1, lr=1e-2):
    net = orc.OracleNetwork(layers, weights, PS, frac["T"], backend=backend)
    out = []
    for _ in range(steps):
        net.load_sequences(frac); net.compute_forward_pass()
        e = net.calculate_error()
        c = net.count_correct_classifications() if layers[-1]["type"] == "multiclass_classification" else -1
        net.compute_backward_pass()
        out.append((e, c))
        if steps > 1:
            net.update_weights(lr, 0.9)
    return net, out


def assert_networks_bit_equal(a, b, N):
    for la, lb in zip(a.layers, b.layers):
        if la.outputs is not None:
            assert np.array_equal(la.outputs[:N * la.size], lb.outputs[:N * la.size]), (la.name, "outputs")
            assert np.array_equal(la.outputErrors[:N * la.size], lb.outputErrors[:N * la.size]), (la.name, "outputErrors")
        if la.trainable:
            assert np.array_equal(la.weightUpdates, lb.weightUpdates), (la.name, "weightUpdates")
            assert np.array_equal(la.weights, lb.weights), (la.name, "weights")
        if la.type in ("lstm", "blstm"):
            assert np.array_equal(la.bufs, lb.bufs), (la.name, "internals")


@pytest.mark.parametrize("kind,size,lengths,PS", [("lstm", 12, [20, 17, 9], 3), ("blstm", 10, [20, 17, 9], 3), ("blstm", 16, [6, 6, 1], 5),
                                                 ("lstm", 7, [1], 1), ("blstm", 250, [9, 8], 2)])
def test_lstm_layer_bit_equal(orc, ref, pkg, kind, size, lengths, PS):
    """All twelve per-direction LSTM internals (LstmLayer.hpp:88-100), outputs, propagated errors and every weight gradient:
    ragged lengths, unused parallel slots, T = 1, peepholes, both directions -- bit for bit."""
    rng = np.random.RandomState(11)
    P, C = 7, 5
    layers = net_desc(P, [("feedforward_tanh", 6), (kind, size)], C)        # a trainable layer below: the error to the preceding layer is computed
    weights = random_weights(layers, rng, 0.4)
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    a, ra = run(orc, "oracle", layers, weights, frac, PS)
    b, rb = run(orc, "ref", layers, weights, frac, PS)
    assert ra == rb
    assert_networks_bit_equal(a, b, a.N)


def test_softmax_with_negative_logits_and_dummies_bit_equal(orc, ref, pkg):
    """Q3: the centring offset starts its max at FLT_MIN, so all-negative rows are centred on min/2; dummy rows keep the raw
    pre-activation.  Large-magnitude weights drive safeExp into both clamps."""
    rng = np.random.RandomState(12)
    P, C, PS = 4, 9, 3
    layers = net_desc(P, [("lstm", 5)], C)
    weights = random_weights(layers, rng, 0.5)
    weights["output"]["input"] = (rng.uniform(-60, 60, C * 5)).astype(np.float32)
    weights["output"]["bias"] = (-np.abs(rng.uniform(50, 200, C))).astype(np.float32)
    xs, ts = random_sequences(rng, [8, 5], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    a, ra = run(orc, "oracle", layers, weights, frac, PS)
    b, rb = run(orc, "ref", layers, weights, frac, PS)
    assert ra == rb
    assert_networks_bit_equal(a, b, a.N)


def test_kat0_network_bit_equal_and_trains_alike(orc, ref, pkg):
    """The reference's tests/test1 network on real CHiME frames (KAT-0), five momentum-SGD steps: oracle and reference
    functors stay bit-equal through training (any divergence in one gradient would compound)."""
    layers, weights, xs, ts = load_kat0()
    frac = pkg.make_fraction(xs, ts, 10)
    a, ra = run(orc, "oracle", layers, weights, frac, 10, steps=5)
    b, rb = run(orc, "ref", layers, weights, frac, 10, steps=5)
    assert ra == rb and ra[-1][0] < ra[0][0]
    assert abs(ra[0][0] - 5293.397461) < 0.05 and ra[0][1] == 126
    assert_networks_bit_equal(a, b, a.N)


def test_three_layer_blstm_stack_bit_equal(orc, ref, pkg):
    rng = np.random.RandomState(13)
    P, C, PS = 39, 183, 4
    layers = net_desc(P, [("blstm", 50)] * 3, C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, [15, 14, 9], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    a, ra = run(orc, "oracle", layers, weights, frac, PS, steps=2)
    b, rb = run(orc, "ref", layers, weights, frac, PS, steps=2)
    assert ra == rb
    assert_networks_bit_equal(a, b, a.N)


@pytest.mark.parametrize("act", ["feedforward_tanh", "feedforward_logistic", "feedforward_identity"])
def test_feedforward_layers_bit_equal(orc, ref, pkg, act):
    rng = np.random.RandomState(14)
    P, L, PS = 6, 4, 4
    layers = net_desc(P, [("lstm", 8), (act, 6)], L, post="sse")
    weights = random_weights(layers, rng, 0.5)
    xs, ts = random_sequences(rng, [9, 5], P, L=L)
    frac = pkg.make_fraction(xs, ts, PS, classification=False)
    a, ra = run(orc, "oracle", layers, weights, frac, PS)
    b, rb = run(orc, "ref", layers, weights, frac, PS)
    assert ra == rb
    assert_networks_bit_equal(a, b, a.N)


def post_case(pkg, post, out_type, seed=31):
    """A small net ending in the given post output layer, with ragged sequences and an unused parallel slot."""
    rng = np.random.RandomState(seed)
    P, PS = 5, 4
    L = 1 if post == "binary_classification" else 6
    W = 2 * L if post in ("weightedsse", "wf") else L
    layers = [{"name": "input", "type": "input", "size": P},
              {"name": "lstm_0", "type": "blstm", "size": 12, "bias": 1.0},
              {"name": "output", "type": out_type, "size": L, "bias": 1.0},
              {"name": "postoutput", "type": post, "size": W}]
    weights = random_weights(layers, rng, 0.5)
    lengths = [11, 7, 9]
    xs = [rng.randn(n, P).astype(np.float32) for n in lengths]
    if post == "binary_classification":
        ts = [rng.randint(0, 2, n).astype(np.int32) for n in lengths]
        return layers, weights, pkg.make_fraction(xs, ts, PS, classification=True), PS
    if post == "ce":            # a target distribution with exact zeros (max(FLT_MIN, t)) ...
        ts = []
        for n in lengths:
            t = rng.rand(n, L).astype(np.float32); t[:, 0] = 0; ts.append(t / t.sum(1, keepdims=True))
        weights["output"]["bias"] = np.array([0, -100, 0, 0, 0, 5], np.float32)    # ... and posteriors that underflow to 0 (the +-100 clip)
    else:
        ts = [rng.randn(n, W).astype(np.float32) for n in lengths]
    return layers, weights, pkg.make_fraction(xs, ts, PS, classification=False), PS


@pytest.mark.parametrize("post,out_type", [("sse", "feedforward_identity"), ("weightedsse", "feedforward_identity"),
                                           ("wf", "feedforward_tanh"), ("ce", "softmax"), ("rmse", "feedforward_identity"),
                                           ("binary_classification", "feedforward_logistic")])
def test_post_output_layers_bit_equal(orc, ref, pkg, post, out_type):
    """calculateError / countCorrectClassifications / computeBackwardPass of SsePostOutputLayer.cu:39-155, WeightedSse...:40-167,
    SseMask...:40-167, CePostOutputLayer.cu:43-170, RmsePostOutputLayer.cu:40-174, BinaryClassificationLayer.cu:44-207:
    the oracle's restatement against the reference's own functors and thrust calls (oracle/ref/ref_post.cpp), and everything
    the injected errors flow into below (output layer and LSTM gradients)."""
    layers, weights, frac, PS = post_case(pkg, post, out_type)
    a, ra = run(orc, "oracle", layers, weights, frac, PS, steps=2, lr=1e-5)
    b, rb = run(orc, "ref", layers, weights, frac, PS, steps=2, lr=1e-5)
    assert ra == rb and np.all(np.isfinite([e for e, _ in ra]))
    if post == "binary_classification":
        assert a.count_correct_classifications() == b.count_correct_classifications() > 0
    assert_networks_bit_equal(a, b, a.N)
    assert np.any(a.layers[-2].outputErrors != 0)
