"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, fp32 mode.  Tolerances are stated per assertion; BASELINE.json asks for frame-posterior
max-abs error < 1e-4 against the CPU reference."""
import numpy as np
import pytest

from helpers import load_kat0, net_desc, random_sequences, random_weights, real_mask

pytestmark = pytest.mark.gpu

POSTERIOR_TOL = 1e-4      # BASELINE.json north_star


def run_both(pkg, orc, layers, weights, frac, PS, precision=0, lr=None, backend="oracle"):
    ref = orc.OracleNetwork(layers, weights, PS, frac["T"], backend=backend)
    ref.load_sequences(frac)
    ref.compute_forward_pass()
    e_ref = ref.calculate_error()
    c_ref = ref.count_correct_classifications() if layers[-1]["type"] == "multiclass_classification" else -1
    ref.compute_backward_pass()
    net = pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=precision)
    net.load_sequences(frac)
    net.compute_forward_pass()
    e, c = net.error_and_correct()
    net.compute_backward_pass()
    return ref, net, (e_ref, c_ref), (e, c)


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def check_network(pkg, orc, layers, weights, frac, PS, grad_tol=2e-4, precision=0, backend="oracle"):
    ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS, precision=precision, backend=backend)
    try:
        real = real_mask(frac)
        C = layers[-1]["size"]
        y = net.outputs().reshape(-1, C)[real]
        yr = ref.outputs().reshape(-1, C)[real]
        assert np.abs(y - yr).max() < POSTERIOR_TOL
        assert abs(e - e_ref) <= 1e-4 * max(1.0, abs(e_ref))
        assert c == c_ref
        for lay in net.trainable_layers():
            rl = ref.layer(lay.name)
            g, gr = lay.weight_updates(), rl.weightUpdates
            assert rel_err(g, gr) < grad_tol, (lay.name, rel_err(g, gr))
            if lay.prev.trainable:
                pe = lay.prev.output_errors().reshape(-1, lay.prev.size)[real]
                per = rl.prev.outputErrors[:net.N * lay.prev.size].reshape(-1, lay.prev.size)[real]
                assert rel_err(pe, per) < grad_tol, (lay.name, "prev errors")
        return ref, net
    except Exception:
        net.close()
        raise


@pytest.mark.parametrize("kind,size", [("lstm", 12), ("blstm", 10), ("blstm", 40), ("lstm", 128)])
def test_single_lstm_layer_internals(pkg, orc, kind, size):
    """Every LSTM internal vector of LstmLayer.hpp:88-100 on real slots, ragged lengths {20,17,9}."""
    rng = np.random.RandomState(11)
    P, C, PS = 7, 5, 3
    layers = net_desc(P, [(kind, size)], C)
    weights = random_weights(layers, rng, 0.4)
    xs, ts = random_sequences(rng, [20, 17, 9], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    with net:
        lay, rl = net.layers[1], ref.layers[1]
        real = real_mask(frac)
        for d in range(lay.dirs):
            for name in ("cellStates", "niActs", "igActs", "fgActs", "ogActs", "tmpOutputs",
                         "niDeltas", "igDeltas", "fgDeltas", "ogDeltas"):
                a = lay.internal(name, d).reshape(-1, lay.H)[real]
                b = rl.internal(name, d)[:net.N * lay.H].reshape(-1, lay.H)[real]
                assert np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(b).max()), (name, d, np.abs(a - b).max())
        # bw direction zeroes the cell state of dummy slots (LstmLayer.cu:81-82); outputs of dummies are 0
        out = lay.outputs().reshape(-1, lay.size)[~real]
        assert np.all(out == 0)


def test_kat0_network(pkg, orc):
    """The reference's tests/test1 network (blstm/feedforward_tanh stack) on real CHiME frames."""
    layers, weights, xs, ts = load_kat0()
    frac = pkg.make_fraction(xs, ts, 10)
    ref, net = check_network(pkg, orc, layers, weights, frac, 10)
    with net:
        e, c = net.error_and_correct()
        assert abs(e - 5293.397461) < 0.05 and c == 126


def test_timit_like_stack_3xblstm(pkg, orc):
    """39 -> 3 x blstm -> softmax 183 (reading A uses size 250; a 3 x 50 stand-in keeps the oracle fast)."""
    rng = np.random.RandomState(5)
    P, C, PS = 39, 183, 6
    layers = net_desc(P, [("blstm", 50), ("blstm", 50), ("blstm", 50)], C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, [31, 30, 28, 25, 25, 12], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    net.close()


def test_full_width_blstm250(pkg, orc):
    """One blstm of size 250 (H = 125 -> padded to 128: the register-resident kernel) + softmax 183."""
    rng = np.random.RandomState(6)
    P, C, PS = 39, 183, 20
    layers = net_desc(P, [("blstm", 250)], C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, list(rng.randint(20, 41, PS)), P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    net.close()


def test_streaming_kernel_blstm320(pkg, orc):
    """H = 160 per direction does not fit the register-resident kernel: W_rec is streamed."""
    rng = np.random.RandomState(8)
    P, C, PS = 13, 9, 5
    layers = net_desc(P, [("blstm", 320)], C)
    weights = random_weights(layers, rng, 0.08)
    xs, ts = random_sequences(rng, [12, 12, 10, 7, 3], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    net.close()


def test_partial_fraction_and_sse(pkg, orc):
    """Fewer sequences than parallel_sequences (missing columns stay PATTYPE_NONE, DataSet.cpp:339-341)
    and the SSE post output layer on a feedforward_identity output."""
    rng = np.random.RandomState(9)
    P, L, PS = 6, 4, 4
    layers = net_desc(P, [("lstm", 8), ("feedforward_logistic", 6)], L, post="sse")
    weights = random_weights(layers, rng, 0.5)
    xs, ts = random_sequences(rng, [9, 5], P, L=L)
    frac = pkg.make_fraction(xs, ts, PS, classification=False)
    ref, net, (e_ref, _), (e, _) = run_both(pkg, orc, layers, weights, frac, PS)
    with net:
        real = real_mask(frac)
        assert abs(e - e_ref) <= 1e-5 * max(1.0, abs(e_ref))
        y = net.outputs().reshape(-1, L)[real]
        yr = ref.outputs().reshape(-1, L)[real]
        assert np.abs(y - yr).max() < 1e-5
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 2e-4, lay.name


@pytest.mark.parametrize("post,out_type", [("weightedsse", "feedforward_identity"), ("wf", "feedforward_tanh"),
                                           ("ce", "softmax"), ("rmse", "feedforward_identity"),
                                           ("binary_classification", "feedforward_logistic")])
def test_remaining_post_output_layers(pkg, orc, post, out_type):
    """the post output layers of LayerFactory.cu:52-87 besides sse / multiclass.
    Error, class count, injected output errors and every weight gradient against the oracle, with
    ragged sequences and an unused parallel slot."""
    rng = np.random.RandomState(31)
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
        frac = pkg.make_fraction(xs, ts, PS, classification=True)
    else:
        if post == "ce":          # a target distribution with exact zeros (exercises max(FLT_MIN, t))
            ts = []
            for n in lengths:
                t = rng.rand(n, L).astype(np.float32); t[:, 0] = 0; ts.append(t / t.sum(1, keepdims=True))
        else:
            ts = [rng.randn(n, W).astype(np.float32) for n in lengths]
        frac = pkg.make_fraction(xs, ts, PS, classification=False)
    ref = orc.OracleNetwork(layers, weights, PS, frac["T"])
    ref.load_sequences(frac); ref.compute_forward_pass()
    e_ref = ref.calculate_error(); ref.compute_backward_pass()
    with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=0) as net:
        net.load_sequences(frac); net.compute_forward_pass()
        e, c = net.error_and_correct()
        net.compute_backward_pass()
        real = real_mask(frac)
        assert abs(e - e_ref) <= 1e-5 * max(1.0, abs(e_ref)), (e, e_ref)
        if post == "binary_classification":
            assert c == ref.count_correct_classifications()
        else:
            assert c == -1
        out = net.layer("output")
        oe = out.output_errors().reshape(-1, L)
        oer = ref.layer("output").outputErrors[:net.N * L].reshape(-1, L)
        if out_type != "softmax":            # softmax rewrites its outputErrors in place during backward
            assert rel_err(oe[real], oer[real]) < 1e-5
            assert np.all(oe[~real] == 0)
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 2e-4, lay.name


def test_sgd_momentum_steps(pkg, orc):
    """Three fractions of stochastic training: post-update weights track the oracle (Q10)."""
    rng = np.random.RandomState(10)
    P, C, PS = 5, 4, 3
    layers = net_desc(P, [("blstm", 12)], C)
    weights = random_weights(layers, rng, 0.3)
    ref = orc.OracleNetwork(layers, weights, PS, 12)
    with pkg.NeuralNetwork(layers, weights, PS, 12) as net:
        for step in range(3):
            xs, ts = random_sequences(rng, [12, 8, 5], P, C=C)
            frac = pkg.make_fraction(xs, ts, PS)
            for n in (ref, net):
                n.load_sequences(frac); n.compute_forward_pass(); n.compute_backward_pass()
            ref.update_weights(1e-2, 0.9); net.update_weights(1e-2, 0.9)
            for lay in net.trainable_layers():
                assert np.abs(lay.weights() - ref.layer(lay.name).weights).max() < 5e-6, (step, lay.name)


def test_weight_noise_perturbs_the_backward_pass_only(pkg, orc):
    """Q9 (Optimizer.cu:47-50,59-70,83-84): forward pass and error on the CLEAN weights, `injectWeightNoise`, backward pass on
    the NOISY weights (every product with a weight in it: the BPTT products, the error to the preceding layer; the gradients
    themselves are activations x deltas), clean weights restored BEFORE the update, update applied to the clean weights.
    Through the C ABI (cn_layer_set_weights between the passes) against the oracle doing the same with the same noise vectors.
    Three stochastic steps; the counter-example (noise injected BEFORE the forward pass) must differ visibly."""
    rng = np.random.RandomState(77)
    P, C, PS, sigma = 5, 4, 3, 0.1
    layers = net_desc(P, [("blstm", 12), ("lstm", 7)], C)
    weights = random_weights(layers, rng, 0.3)
    ref = orc.OracleNetwork(layers, weights, PS, 12)
    wrong = orc.OracleNetwork(layers, weights, PS, 12)
    with pkg.NeuralNetwork(layers, weights, PS, 12) as net:
        for step in range(3):
            xs, ts = random_sequences(rng, [12, 8, 5], P, C=C)
            frac = pkg.make_fraction(xs, ts, PS)
            noise = {l.name: rng.normal(0.0, sigma, l.weights.size).astype(np.float32) for l in ref.trainable_layers()}
            # reference protocol on the oracle
            ref.load_sequences(frac); ref.compute_forward_pass(); e_ref = ref.calculate_error()
            clean = {l.name: l.weights.copy() for l in ref.trainable_layers()}
            for l in ref.trainable_layers():
                l.weights += noise[l.name]                                   # TrainableLayer.cu:188-209
            ref.compute_backward_pass()
            for l in ref.trainable_layers():
                l.weights[:] = clean[l.name]
            ref.update_weights(1e-2, 0.9)
            # the same calls through the C ABI
            net.load_sequences(frac); net.compute_forward_pass(); e, _ = net.error_and_correct()
            assert abs(e - e_ref) < 1e-4 * max(1.0, abs(e_ref))                # the forward pass saw the clean weights
            for lay in net.trainable_layers():
                assert np.array_equal(lay.weights(), clean[lay.name]) or np.abs(lay.weights() - clean[lay.name]).max() < 5e-6
                lay.set_weights(lay.weights() + noise[lay.name])
            net.compute_backward_pass()
            for lay in net.trainable_layers():
                assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 2e-4, (step, lay.name)
            net.join()                                                         # the gradient products still read the noisy operands
            for lay in net.trainable_layers():
                lay.set_weights(clean[lay.name])
            net.update_weights(1e-2, 0.9)
            for lay in net.trainable_layers():
                assert np.abs(lay.weights() - ref.layer(lay.name).weights).max() < 5e-6, (step, lay.name)
            # counter-example: noise in front of the forward pass
            wrong.load_sequences(frac)
            for l in wrong.trainable_layers():
                l.weights[:] = clean[l.name] + noise[l.name]
            wrong.compute_forward_pass(); e_wrong = wrong.calculate_error(); wrong.compute_backward_pass()
            assert abs(e_wrong - e_ref) > 1e-3 * abs(e_ref)
            assert max(rel_err(l.weightUpdates, ref.layer(l.name).weightUpdates) for l in wrong.trainable_layers()) > 1e-2
            for l in wrong.trainable_layers():
                l.weights[:] = ref.layer(l.name).weights


def test_batch_accumulation_on_the_device(pkg, orc):
    """Batch learning (Optimizer.cu:72-85,95-97): cn_ctx_accumulate_updates sums the fractions' weightUpdates on the device
    (first fraction: copy), cn_ctx_take_accumulated hands the sum to the one update of the epoch.  Two epochs of three
    fractions against the oracle summing in the same order; call-order errors are reported, not ignored."""
    rng = np.random.RandomState(78)
    P, C, PS = 5, 4, 3
    layers = net_desc(P, [("blstm", 12), ("feedforward_tanh", 6)], C)
    weights = random_weights(layers, rng, 0.3)
    fracs = [pkg.make_fraction(*random_sequences(rng, lens, P, C=C), PS) for lens in ([12, 8, 5], [9, 9], [4, 11, 7])]
    ref = orc.OracleNetwork(layers, weights, PS, 12)
    with pkg.NeuralNetwork(layers, weights, PS, 12) as net:
        with pytest.raises(pkg.CurrenntHipError):
            net.take_accumulated()                                             # nothing accumulated
        with pytest.raises(pkg.CurrenntHipError):
            net.accumulate_updates(False)                                      # an epoch starts with first = True
        for epoch in range(2):
            acc = None
            for k, f in enumerate(fracs):
                for n in (ref, net):
                    n.load_sequences(f); n.compute_forward_pass(); n.compute_backward_pass()
                g = [l.weightUpdates.copy() for l in ref.trainable_layers()]
                acc = g if acc is None else [a + b for a, b in zip(acc, g)]
                net.accumulate_updates(k == 0)
            for l, a in zip(ref.trainable_layers(), acc):
                l.weightUpdates[:] = a
            net.take_accumulated()
            for lay in net.trainable_layers():
                assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 2e-4, (epoch, lay.name)
            ref.update_weights(1e-2, 0.9); net.update_weights_fused(1e-2, 0.9)
            for lay in net.trainable_layers():
                assert np.abs(lay.weights() - ref.layer(lay.name).weights).max() < 1e-5, (epoch, lay.name)
        with pytest.raises(pkg.CurrenntHipError):
            net.take_accumulated()                                             # the sum was taken


@pytest.mark.parametrize("hidden", [[64, 64], [300], [384, 320], [440], [600]])
def test_bf16_mode_close(pkg, orc, hidden):
    """Throughput mode (bf16 MFMA operands, fp32 accumulate/state): not a parity mode; posteriors
    stay within 3e-2 of the fp32 oracle on a 2-layer stack and the loss within 1 %.  Sizes 300 / 320 / 384
    (H = 150 / 160 / 192 -> Hp = 160 / 160 / 192) run the 10- and 12-wave register-resident kernels; 440 and 600
    (H = 220 / 300) are padded up to the 2-CU / 8-CU cluster shapes (Hp = 256 / 512)."""
    rng = np.random.RandomState(12)
    P, C, PS = 39, 20, 8
    layers = net_desc(P, [("blstm", h) for h in hidden], C)
    weights = random_weights(layers, rng, 0.1 if max(hidden) <= 64 else 0.05)
    xs, ts = random_sequences(rng, [30] * PS, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net, (e_ref, _), (e, _) = run_both(pkg, orc, layers, weights, frac, PS, precision=1)
    with net:
        assert np.abs(net.outputs() - ref.outputs()).max() < 3e-2
        assert abs(e - e_ref) < 1e-2 * e_ref
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 5e-2, lay.name


X3_CASES = {
    # name: (P, hidden, C, lengths, PS, weight scale)
    "uni_lstm128": (7, [("lstm", 128)], 5, [20, 17, 9], 3, 0.4),                         # register-resident, Hp = 128
    "blstm_stack": (39, [("blstm", 50)] * 3, 183, [31, 30, 28, 25, 25, 12], 6, 0.1),      # Hp = 32 x 3 layers
    "blstm250": (39, [("blstm", 250)], 183, [40, 38, 33, 33, 30, 29, 21, 20], 8, 0.1),    # the headline layer shape
    "blstm320_streamed": (13, [("blstm", 320)], 9, [12, 12, 10, 7, 3], 5, 0.08),          # Hp = 160: W_rec streamed and split per step
    "tanh_lstm_softmax700": (6, [("feedforward_tanh", 40), ("lstm", 24)], 700, [9, 7, 7, 4], 5, 0.3),
    "two_sequences_per_lane": (4, [("blstm", 20)], 3, [6, 5] * 260, 520, 0.5),            # rpl = 2
    "blstm500_cluster": (20, [("blstm", 500)], 12, [25, 25, 24, 22, 22, 20, 17, 15, 9, 4], 10, 0.06),   # Hp = 256: 4 CUs x 64 units
    "blstm440_padded_to_cluster": (20, [("blstm", 440), ("blstm", 500)], 12, [14, 13, 13, 9, 2], 5, 0.06),
}


@pytest.mark.parametrize("case", sorted(X3_CASES) + ["blstm500_cluster/delta_exchange"])
def test_bf16x3_parity_mode(pkg, orc, case, monkeypatch):
    """CN_PREC_BF16X3 (operands split into bf16 hi + lo inside the kernels, three bf16 MFMAs per product, fp32
    accumulation): the SAME tolerances as the exact-fp32 mode -- posteriors max-abs < 1e-4 (BASELINE.json), error 1e-4
    relative, #correct exact, gradients and propagated errors within 2e-4 of the layer's max."""
    if case.endswith("/delta_exchange"):          # the backward cluster kernel that exchanges deltas (three MFMAs per chunk), kept for A/B
        monkeypatch.setenv("CN_NO_BWD_PSUM", "1")
    P, hidden, C, lengths, PS, scale = X3_CASES[case.split("/")[0]]
    rng = np.random.RandomState(71)
    layers = net_desc(P, hidden, C)
    weights = random_weights(layers, rng, scale)
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS, precision=pkg.PREC_BF16X3)
    with net:
        if "cluster" in case:
            assert net.recurrent_kernel(False) == "lstm_fwd_cluster_kernel<2,256,64,1>"
            # backward: partial sums exchanged, lo halves in the spare rows of the row quads (two MFMAs per chunk)
            assert net.recurrent_kernel(True) == ("lstm_bwd_cluster_kernel<2,256,64,1>" if case.endswith("/delta_exchange") else "lstm_bwd_cluster_psum_kernel<2,256,64>")


def test_bf16x3_mode_stays_on_the_oracle_through_training(pkg, orc):
    """Sixty momentum-SGD steps on a learnable task in CN_PREC_BF16X3 and in the oracle, from the same weights: after
    training (error down by more than a third, posteriors far from uniform) the posteriors still agree to 1e-4 and the
    trained weights to 2e-4 of their range -- a split-product error that compounded through recurrence or training
    would show here, not at initial weights."""
    rng = np.random.RandomState(21)
    P, C, PS, T = 8, 4, 12, 24
    layers = net_desc(P, [("blstm", 32)], C)
    weights = random_weights(layers, rng, 0.1)
    proj = rng.randn(2 * P, C)
    fracs = []
    for _ in range(4):
        xs = [rng.randn(T - (i % 3), P).astype(np.float32) for i in range(PS)]
        ts = []
        for x in xs:
            prev = np.vstack([np.zeros((1, P), np.float32), x[:-1]])
            ts.append(np.argmax(np.hstack([x, prev]) @ proj, axis=1).astype(np.int32))
        fracs.append(pkg.make_fraction(xs, ts, PS))
    ref = orc.OracleNetwork(layers, weights, PS, T)
    with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_BF16X3) as net:
        errs = []
        for step in range(60):
            for n in (ref, net):
                n.load_sequences(fracs[step % 4]); n.compute_forward_pass()
            errs.append((ref.calculate_error(), net.calculate_error()))
            for n in (ref, net):
                n.compute_backward_pass(); n.update_weights(2e-3, 0.9)
        for n in (ref, net):
            n.load_sequences(fracs[0]); n.compute_forward_pass()
        first, last = np.mean([e[0] for e in errs[:4]]), np.mean([e[0] for e in errs[-4:]])
        assert last < 0.67 * first
        real = real_mask(fracs[0])
        y, yr = net.outputs().reshape(-1, C)[real], ref.outputs().reshape(-1, C)[real]
        assert yr.max() > 0.9 and np.abs(y - yr).max() < POSTERIOR_TOL, (yr.max(), np.abs(y - yr).max())
        assert abs(errs[-1][1] - errs[-1][0]) < 1e-4 * errs[-1][0]
        for lay in net.trainable_layers():
            w, wr = lay.weights(), ref.layer(lay.name).weights
            assert np.abs(w - wr).max() < 2e-4 * np.abs(wr).max(), (lay.name, np.abs(w - wr).max())


def test_error_texts(pkg):
    """Shape errors carry the reference's messages (InputLayer.cpp:52-55, LstmLayer.cu:528-529)."""
    rng = np.random.RandomState(1)
    layers = net_desc(5, [("blstm", 8)], 3)
    weights = random_weights(layers, rng)
    with pkg.NeuralNetwork(layers, weights, 2, 4) as net:
        xs, ts = random_sequences(rng, [4, 3], 6, C=3)
        with pytest.raises(pkg.CurrenntHipError, match="Input layer size of 5 != data input pattern size of 6"):
            net.load_sequences(pkg.make_fraction(xs, ts, 2))
    bad = net_desc(5, [("blstm", 7)], 3)
    with pytest.raises(pkg.CurrenntHipError, match="Cannot create a bidirectional layer with an odd layer size"):
        pkg.NeuralNetwork(bad, None, 2, 4, seed=1)


@pytest.mark.parametrize("size,scale", [(500, 0.06), (1024, 0.04)])
def test_cluster_kernel_matches_streaming_kernel(pkg, orc, size, scale):
    """H = 250 per direction (reading B: CURRENNT size 500 -> Hp = 256): in bf16 mode W_rec is split over a
    2-CU cluster with a per-step hand-off (cn_lstm_cluster.hip); H = 512 (the long-utterance config, size 1024):
    8 CUs x 64 units.  Same arithmetic as the single-CU streaming kernel, so the two agree to bf16 noise; both
    track the fp32 oracle loosely."""
    import os
    rng = np.random.RandomState(14)
    P, C, PS = 20, 12, 10
    layers = net_desc(P, [("blstm", size)], C)
    weights = random_weights(layers, rng, scale)
    xs, ts = random_sequences(rng, [25, 25, 24, 22, 22, 20, 17, 15, 9, 4], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    res = {}
    for mode in ("cluster", "stream"):
        if mode == "stream":
            os.environ["CN_NO_CLUSTER"] = "1"
        try:
            with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=1) as net:
                net.load_sequences(frac); net.compute_forward_pass()
                e, c = net.error_and_correct()
                net.compute_backward_pass()
                res[mode] = (net.outputs(), e, [l.weight_updates() for l in net.trainable_layers()],
                             net.layers[1].internal("tmpOutputs", 1))
        finally:
            os.environ.pop("CN_NO_CLUSTER", None)
    (y1, e1, g1, h1), (y2, e2, g2, h2) = res["cluster"], res["stream"]
    assert np.abs(h1 - h2).max() < 1e-2 and np.abs(y1 - y2).max() < 2e-3
    assert abs(e1 - e2) < 1e-3 * abs(e2)
    for a, b in zip(g1, g2):
        assert rel_err(a, b) < 1e-2
    ref = orc.OracleNetwork(layers, weights, PS, frac["T"])
    ref.load_sequences(frac); ref.compute_forward_pass(); ref.compute_backward_pass()
    assert np.abs(y1 - ref.outputs()).max() < 3e-2
    for a, lay in zip(g1, ref.trainable_layers()):
        assert rel_err(a, lay.weightUpdates) < 6e-2, lay.name


@pytest.mark.parametrize("T", [1, 3, 8])
@pytest.mark.parametrize("shape", ["s2", "s2asm", "rpl1", "rpl2"])
@pytest.mark.parametrize("kind,size", [("blstm", 128), ("lstm", 125), ("blstm", 250)])
def test_row_pair_sparse_products(pkg, orc, monkeypatch, kind, size, shape, T):
    """The 2:4 row-pair MFMA path of the register-resident kernels (bf16 and split-bf16 modes, Hp = 64 / 128) in its three
    cuts: "s2" / "s2asm" = two sequences per workgroup, 32 units per wave (cn_lstm_s2.hip; compiled kernels / the hand-written
    loops the mode takes by default at Hp = 128), and the 4- and 8-sequence workgroups of cn_lstm.hip with one and two
    sequences per lane (CN_RPL).  Loop shapes T = 1, 3, 8, ragged lengths, a partly filled last sequence group, one- and two-directional.
    Checked in the split-bf16 mode at the fp32 tolerances, which a misplaced operand element cannot meet."""
    rpl = 2 if shape == "rpl2" else 1
    monkeypatch.setenv("CN_RPL", str(rpl))
    if shape == "s2":
        monkeypatch.setenv("CN_S2_X3", "1"); monkeypatch.setenv("CN_NO_S2_ASM", "1")      # the compiled kernels of the cut
    if shape == "rpl1":
        monkeypatch.setenv("CN_NO_S2", "1")
    rng = np.random.RandomState(300 + T + rpl + size)          # (same data for "s2" and "rpl1")
    P, C, PS = 6, 4, 13
    layers = net_desc(P, [(kind, size)], C)
    weights = random_weights(layers, rng, 0.15)
    lengths = [max(1, T - (i % 3)) for i in range(PS)]
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS, precision=pkg.PREC_BF16X3)
    with net:
        Hp = 64 if size == 128 else 128
        if shape == "s2":
            assert net.recurrent_kernel(False) == "lstm_fwd_s2_kernel<2,%d>" % Hp
            assert net.recurrent_kernel(True) == "lstm_bwd_s2_kernel<2,%d>" % Hp
        elif shape == "s2asm":          # what the mode takes by default: the hand-written loops where they exist (Hp = 128), else 4-sequence kernels
            want = "lstm_%s_s2_x3_asm_kernel" if Hp == 128 else "lstm_%s_kernel<2,64,1,1>"
            assert net.recurrent_kernel(False) == want % "fwd" and net.recurrent_kernel(True) == want % "bwd"
        else:
            assert net.recurrent_kernel(False) == "lstm_fwd_kernel<2,%d,1,%d>" % (Hp, rpl)
            assert net.recurrent_kernel(True) == "lstm_bwd_kernel<2,%d,1,%d>" % (Hp, rpl)


@pytest.mark.parametrize("size,rpl,kernel", [(384, 1, "lstm_fwd_kernel<0,192,1,1>"), (384, 2, "lstm_fwd_kernel<0,192,1,2>"),
                                             (500, 2, "lstm_fwd_cluster_kernel<0,256,128,2>"), (1024, 2, "lstm_fwd_cluster_kernel<0,512,64,2>")])
def test_row_pair_sparse_products_bf16_shapes(pkg, orc, monkeypatch, size, rpl, kernel):
    """Shapes that exist in bf16 mode only: Hp = 192 register resident (one and two sequences per lane) and the 2-CU / 8-CU
    clusters with two sequences per lane.  bf16 tolerances: these catch a misplaced operand (errors of order one), not
    rounding."""
    monkeypatch.setenv("CN_RPL", str(rpl))
    rng = np.random.RandomState(41 + size + rpl)
    P, C, T, PS = 10, 6, 9, 13
    layers = net_desc(P, [("blstm", size)], C)
    weights = random_weights(layers, rng, 0.05)
    xs, ts = random_sequences(rng, [max(1, T - (i % 4)) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net, (e_ref, _), (e, _) = run_both(pkg, orc, layers, weights, frac, PS, precision=pkg.PREC_BF16)
    with net:
        assert net.recurrent_kernel(False) == kernel
        assert np.abs(net.outputs() - ref.outputs()).max() < 3e-2
        assert abs(e - e_ref) < 1e-2 * e_ref
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 5e-2, lay.name


@pytest.mark.parametrize("C", [20, 700])
@pytest.mark.parametrize("order", ["accumulate_then_backward", "accumulate_twice", "no_backward"])
def test_loss_accumulate_matches_the_per_fraction_errors(pkg, orc, C, order):
    """cn_loss_accumulate + cn_loss_read (the epoch sums of Optimizer.cu:46-55) against the per-fraction values of cn_loss_eval
    and the oracle, over two fractions (narrow and wide softmax rows), with and without backward passes in between and with a
    fraction accumulated twice."""
    rng = np.random.RandomState(9 + C)
    P, PS = 7, 5
    layers = net_desc(P, [("blstm", 16)], C)
    weights = random_weights(layers, rng, 0.3)
    fracs = []
    for lens in ([9, 9, 8, 5, 2], [7, 6, 6, 6, 1]):
        xs, ts = random_sequences(rng, lens, P, C=C)
        fracs.append(pkg.make_fraction(xs, ts, PS))
    ref = orc.OracleNetwork(layers, weights, PS, max(f["T"] for f in fracs))
    want_e, want_c = 0.0, 0
    with pkg.NeuralNetwork(layers, weights, PS, max(f["T"] for f in fracs)) as net:
        per_fraction = []
        for f in fracs:
            ref.load_sequences(f); ref.compute_forward_pass()
            want_e += ref.calculate_error(); want_c += ref.count_correct_classifications()
            net.load_sequences(f); net.compute_forward_pass()
            per_fraction.append(net.error_and_correct())
            net.loss_accumulate()
            if order == "accumulate_twice":
                net.loss_accumulate()
            if order != "no_backward":
                net.compute_backward_pass()
        e, c = net.loss_read()
        k = 2 if order == "accumulate_twice" else 1
        assert abs(e - k * want_e) <= 1e-5 * k * want_e and c == k * want_c
        assert abs(e - k * sum(x[0] for x in per_fraction)) <= 1e-6 * e
        assert net.loss_read() == (0.0, 0)                              # (the read reset the sums)


@pytest.mark.parametrize("C", [20, 700])
def test_softmax_output_errors_read_back_in_bf16_mode(pkg, orc, C):
    """bf16 mode: the fused softmax / multiclass backward kernel (narrow and wide rows) writes only the bf16 operand copy of the
    layer's outputErrors; cn_layer_read hands that copy back.  It must be the oracle's outputErrors up to bf16 rounding and the
    bf16 noise of the forward pass (SoftmaxLayer.cu:317-349 after MulticlassClassificationLayer.cu:220-240)."""
    rng = np.random.RandomState(5 + C)
    P, PS = 9, 6
    layers = net_desc(P, [("blstm", 24)], C)
    weights = random_weights(layers, rng, 0.3)
    xs, ts = random_sequences(rng, [11, 10, 10, 7, 4, 2], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net, _, _ = run_both(pkg, orc, layers, weights, frac, PS, precision=pkg.PREC_BF16)
    with net:
        real = real_mask(frac)
        sm, rsm = net.layers[-2], ref.layers[-2]
        oe = sm.output_errors().reshape(-1, C)[real]
        roe = rsm.outputErrors[:net.N * C].reshape(-1, C)[real]
        assert np.abs(oe - roe).max() < 2e-2 * np.abs(roe).max()
        assert np.abs(oe).max() > 0.1                                   # (it is there at all)


@pytest.mark.parametrize("T", [1, 2, 3, 6])
@pytest.mark.parametrize("PS", [3, 520, 1100])
def test_short_sequences_and_sequences_per_lane(pkg, orc, T, PS):
    """Loop structure of the recurrent kernels (peeled first pair, branch-free pairs, odd last step) for
    T = 1, 2, 3, 6 and the three sequences-per-lane shapes: PS = 3 -> 1, 520 -> 2, 1100 -> 4 sequences per lane
    (chosen so that the workgroups still fit the chip).  Every second sequence is one frame shorter."""
    rng = np.random.RandomState(100 + T + PS)
    P, C = 4, 3
    layers = net_desc(P, [("blstm", 20)], C)
    weights = random_weights(layers, rng, 0.5)
    lengths = [max(1, T - (i % 2)) for i in range(PS)]
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS)
    with net:
        real = real_mask(frac)
        y = net.outputs().reshape(-1, C)[real]
        yr = ref.outputs().reshape(-1, C)[real]
        assert np.abs(y - yr).max() < POSTERIOR_TOL
        assert abs(e - e_ref) <= 1e-4 * max(1.0, abs(e_ref)) and c == c_ref
        for lay in net.trainable_layers():
            g, gr = lay.weight_updates(), ref.layer(lay.name).weightUpdates
            assert rel_err(g, gr) < 2e-4, (lay.name, rel_err(g, gr))


def test_bf16_mode_trains_like_fp32_mode(pkg):
    """Sixty momentum-SGD steps on a learnable task (the class is a function of the current and the previous
    input frame) in both precision modes: the error falls by more than a third in each and the two trajectories
    end within 5 % of each other."""
    rng = np.random.RandomState(21)
    P, C, PS, T = 8, 4, 12, 24
    layers = net_desc(P, [("blstm", 32)], C)
    weights = random_weights(layers, rng, 0.1)
    proj = rng.randn(2 * P, C)
    fracs = []
    for _ in range(4):
        xs = [rng.randn(T - (i % 3), P).astype(np.float32) for i in range(PS)]
        ts = []
        for x in xs:
            prev = np.vstack([np.zeros((1, P), np.float32), x[:-1]])
            ts.append(np.argmax(np.hstack([x, prev]) @ proj, axis=1).astype(np.int32))
        fracs.append(pkg.make_fraction(xs, ts, PS))
    curves = {}
    for prec in (pkg.PREC_F32, pkg.PREC_BF16):
        with pkg.NeuralNetwork(layers, weights, PS, T, precision=prec) as net:
            errs = []
            for step in range(60):
                net.load_sequences(fracs[step % 4]); net.compute_forward_pass()
                errs.append(net.error_and_correct()[0])
                net.compute_backward_pass(); net.update_weights(2e-3, 0.9)
            curves[prec] = errs
    for prec, errs in curves.items():
        first, last = np.mean(errs[:4]), np.mean(errs[-4:])
        assert np.all(np.isfinite(errs)) and last < 0.67 * first, (prec, first, last)
    a, b = np.mean(curves[pkg.PREC_F32][-4:]), np.mean(curves[pkg.PREC_BF16][-4:])
    assert abs(a - b) < 0.05 * a, (a, b)


def test_wide_softmax_layer(pkg, orc):
    """Softmax layers wider than 256 classes (tied-state outputs of the LVCSR shape) take the block-per-pattern
    kernels: posteriors, error, #correct and all gradients against the oracle."""
    rng = np.random.RandomState(33)
    P, C, PS = 6, 700, 5
    layers = net_desc(P, [("lstm", 24)], C)
    weights = random_weights(layers, rng, 0.3)
    xs, ts = random_sequences(rng, [9, 7, 7, 4], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    net.close()


@pytest.mark.parametrize("precision", [0, 1])
def test_lazy_wide_softmax_equals_the_eager_kernels(pkg, precision, monkeypatch):
    """Wide softmax rows in training: the forward pass leaves the logits in place (softmax_fwd_wide_kernel<false> writes only
    {offset, sum} and the loss statistics per row) and the fused backward kernel recomputes the posteriors (SoftmaxLayer.cu:61-152
    element by element, the same expressions).  Three orders of calls must agree BIT FOR BIT on posteriors and output errors:
    A forward, backward, then read the posteriors (normalised on demand after the lazy backward kernel);
    B forward, read the posteriors (normalised on demand), backward (the eager backward kernel on them);
    C as B, then a second forward pass -- a layer whose posteriors were read runs the eager forward kernel -- and backward.
    Error, #correct (row statistics of either forward kernel) and the gradients (atomic column sums: 1e-6) agree too.
    The library is lazy by itself only in the bf16 mode (v_exp_f32 and one reciprocal per row: recomputing is cheap there);
    CN_LAZY_SOFTMAX=1 puts the fp32 mode's exact kernels (expf, a division per element) through the same three orders."""
    if precision == 0:
        monkeypatch.setenv("CN_LAZY_SOFTMAX", "1")
    rng = np.random.RandomState(35 + precision)
    P, C, PS = 6, 700, 5
    layers = net_desc(P, [("lstm", 24)], C)
    weights = random_weights(layers, rng, 0.3)
    xs, ts = random_sequences(rng, [9, 7, 7, 4], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    got = {}
    for mode in "ABC":
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=precision) as net:
            out = net.layer("output")
            net.load_sequences(frac); net.compute_forward_pass()
            if mode == "A":
                ec = net.error_and_correct()
                net.compute_backward_pass()
                oe = out.output_errors().copy(); y = net.outputs().copy()
            else:
                y = net.outputs().copy()
                ec = net.error_and_correct()
                net.compute_backward_pass()
                oe = out.output_errors().copy()
                if mode == "C":
                    net.compute_forward_pass()
                    y2 = net.outputs().copy(); ec2 = net.error_and_correct()
                    net.compute_backward_pass()
                    assert np.array_equal(y2, y) and ec2 == ec and np.array_equal(out.output_errors(), oe)
            got[mode] = (y, oe, ec, [l.weight_updates() for l in net.trainable_layers()])
    real = real_mask(frac)
    yA = got["A"][0].reshape(-1, C)
    assert np.allclose(yA[real].sum(1), 1.0, atol=1e-5) and np.abs(got["A"][1]).max() > 0
    for mode in "BC":
        assert np.array_equal(got[mode][0], got["A"][0]), mode
        assert np.array_equal(got[mode][1], got["A"][1]), mode
        assert got[mode][2] == got["A"][2], mode
        for a, b in zip(got[mode][3], got["A"][3]):
            assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(a).max()), mode


def test_two_contexts_interleaved(pkg):
    """Two networks alive in one process, trained alternately (they share the device's CU-masked gradient stream):
    each ends with the weights it reaches when trained alone."""
    rng = np.random.RandomState(41)
    P, C, PS, T = 12, 7, 6, 25
    specs = []
    for hidden in ([("blstm", 64), ("blstm", 64)], [("lstm", 48), ("blstm", 96)]):
        layers = net_desc(P, hidden, C)
        weights = random_weights(layers, rng, 0.1)
        xs, ts = random_sequences(rng, [T - i for i in range(PS)], P, C=C)
        specs.append((layers, weights, pkg.make_fraction(xs, ts, PS)))

    def train(nets, steps=4):
        for _ in range(steps):
            for net, (_, _, frac) in zip(nets, specs):
                net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass()
                net.update_weights_fused(1e-3, 0.9)
        return [np.concatenate([l.weights() for l in net.trainable_layers()]) for net in nets]

    solo = []
    for spec in specs:
        with pkg.NeuralNetwork(spec[0], spec[1], PS, T) as net:
            for _ in range(4):
                net.load_sequences(spec[2]); net.compute_forward_pass(); net.compute_backward_pass()
                net.update_weights_fused(1e-3, 0.9)
            solo.append(np.concatenate([l.weights() for l in net.trainable_layers()]))
    nets = [pkg.NeuralNetwork(s_[0], s_[1], PS, T) for s_ in specs]
    try:
        both = train(nets)
    finally:
        for n in nets:
            n.close()
    for a, b in zip(solo, both):
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())


def test_wide_layers_in_fp32_parity_mode(pkg, orc):
    """fp32 parity mode with 512 units per direction (the long-utterance topology's layer width): W_rec does not fit
    one CU, so the streaming kernels run with the compact operand tile (4 real rows + one zero row; a full 16-row
    fp32 delta tile of 4 x 512 units would need 263 KB of LDS).  Same tolerances as every other fp32 parity test."""
    rng = np.random.RandomState(33)
    P, C, PS = 10, 6, 3
    layers = net_desc(P, [("blstm", 1024)], C)
    weights = random_weights(layers, rng, 0.03)
    xs, ts = random_sequences(rng, [5, 4, 2], P, C=C)
    check_network(pkg, orc, layers, weights, pkg.make_fraction(xs, ts, PS), PS)


def test_layer_too_wide_for_fp32_mode_says_so(pkg):
    """1024 units per direction in fp32 mode: the backward kernel's two delta tiles exceed the LDS; the call fails
    with a message that names the limit instead of a bare launch error."""
    rng = np.random.RandomState(34)
    layers = net_desc(4, [("blstm", 2048)], 3)
    xs, ts = random_sequences(rng, [3, 2], 4, C=3)
    with pkg.NeuralNetwork(layers, None, 2, 3, seed=1) as net:
        net.load_sequences(pkg.make_fraction(xs, ts, 2))
        net.compute_forward_pass()
        with pytest.raises(pkg.CurrenntHipError, match="KB of LDS per workgroup"):
            net.compute_backward_pass()


@pytest.mark.parametrize("depth", [2, 4])
def test_destroy_deep_cluster_network(depth):
    """BLSTM layers of size 1024 (8-CU cluster kernels; BASELINE.json's long-utterance topology has five): one training
    step, then the context is destroyed and the process exits.  Run in a child process with a time limit: when every
    context owned (and destroyed) its CU-masked gradient stream, hipStreamDestroy never returned for some depths
    (cn_api.cpp: masked_stream)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "from helpers import net_desc, random_sequences, random_weights\n"
        "pkg = ge.load_package()\n"
        "rng = np.random.RandomState(5)\n"
        "layers = net_desc(39, [('blstm', 1024)] * %d, 20)\n"
        "weights = random_weights(layers, rng, 0.02)\n"
        "xs, ts = random_sequences(rng, [12] * 16, 39, C=20)\n"
        "frac = pkg.make_fraction(xs, ts, 16)\n"
        "for _ in range(2):\n"
        "    with pkg.NeuralNetwork(layers, weights, 16, 12, precision=pkg.PREC_BF16) as net:\n"
        "        net.load_sequences(frac); net.compute_forward_pass(); e = net.calculate_error()\n"
        "        net.compute_backward_pass(); net.update_weights_fused(1e-5, 0.9); net.synchronize()\n"
        "        assert np.isfinite(e)\n"
        "print('destroyed ok')\n" % (root, os.path.join(root, "tests"), depth))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "destroyed ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.parametrize("pre16", [0, 1])
@pytest.mark.parametrize("kind,size,T", [("blstm", 250, 1), ("blstm", 250, 2), ("blstm", 250, 3), ("blstm", 250, 4), ("blstm", 250, 5), ("blstm", 250, 6), ("lstm", 128, 9), ("blstm", 256, 37),
                                         ("blstm", 250, 7), ("blstm", 250, 64)])
def test_hand_written_time_loops_equal_the_compiled_kernels_bit_for_bit(pkg, monkeypatch, kind, size, T, pre16):
    """The hand-written (asm) time loops of cn_lstm_s2.hip against the C++ kernels of the same cut (CN_NO_S2_ASM): same
    operand order, same arithmetic, so EVERY value on a real slot must be bit-identical -- outputs, the four gate
    activations, cell states, all deltas, every weight gradient except the split-K atomics' last bits (compared at 1e-6).
    Loop shapes: T = 1 ... 7 (the backward loop leaves its four-step body after any step; the forward loop's pair count and
    tail length in every combination, T < 4: compiled forward kernel), longer odd / even passes, the pass as long as the
    buffers (prefetch into the guard steps), ragged lengths, unused slots, one- and two-directional.  A wrong wait count, a missed hazard or a misplaced operand cannot pass this.
    pre16 = 1: the second form of the forward loop's text (bf16 pre-activations out of LstmRec::pre16, widened by shift / mask)
    against the compiled kernel reading the same buffer."""
    rng = np.random.RandomState(500 + T)
    P, C, PS = 9, 7, 11                       # 11 slots -> padded to 12, the last group half filled
    layers = net_desc(P, [(kind, size), (kind, size)], C)
    weights = random_weights(layers, rng, 0.08)
    lengths = [max(1, T - (i % 4) * (T // 4)) for i in range(PS - 1)]      # one unused slot; maxT = T: the last steps prefetch guard rows
    lengths[0] = T
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    real = real_mask(frac)
    got = {}
    for mode in ("asm", "cpp"):
        if mode == "cpp":
            monkeypatch.setenv("CN_NO_S2_ASM", "1")
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16) as net:
            net.set_option("pre16", pre16)
            net.load_sequences(frac); net.compute_forward_pass()
            e, c = net.error_and_correct()
            assert len(net.bf16_preactivation_layers()) == (2 if pre16 else 0)
            net.compute_backward_pass()
            names = (net.recurrent_kernel(False), net.recurrent_kernel(True))
            vals = {"error": np.float32(e), "out": net.outputs().reshape(-1, C)[real]}
            for lay in net.layers[1:3]:
                for dd in range(lay.dirs):
                    for name in ("cellStates", "niActs", "igActs", "fgActs", "ogActs", "tmpOutputs", "niDeltas", "igDeltas", "fgDeltas", "ogDeltas"):
                        vals["%s/%d/%s" % (lay.name, dd, name)] = lay.internal(name, dd).reshape(-1, lay.H)[real]
            grads = {lay.name: lay.weight_updates() for lay in net.trainable_layers()}
            got[mode] = (names, vals, grads)
    assert got["asm"][0] == ("lstm_fwd_s2_asm_kernel" if T >= 4 else "lstm_fwd_s2_kernel<0,128>", "lstm_bwd_s2_asm_kernel")
    assert got["cpp"][0] == ("lstm_fwd_s2_kernel<0,128>", "lstm_bwd_s2_kernel<0,128>")
    for key, v in got["asm"][1].items():
        assert np.array_equal(v, got["cpp"][1][key]), (key, np.abs(v - got["cpp"][1][key]).max())
    for name, g in got["asm"][2].items():
        assert rel_err(g, got["cpp"][2][name]) < 1e-6, name


@pytest.mark.parametrize("kind,size,T", [("blstm", 500, 1), ("blstm", 500, 2), ("blstm", 500, 3), ("lstm", 250, 6), ("blstm", 512, 37), ("blstm", 500, 64)])
def test_hand_written_wide_forward_loop_equals_its_compiled_twin_bit_for_bit(pkg, orc, monkeypatch, kind, size, T):
    """Hp = 256 on one CU (cn_lstm_s2.hip: lstm_fwd_s2w_asm_kernel, generated step body with 64 MFMAs, 18 W fragments streamed
    from LDS per step) against lstm_fwd_s2w_kernel, the compiled kernel of the same cut (CN_NO_S2W_ASM): every forward value on a
    real slot bit-identical; and against the 2-CU cluster kernel it replaces (CN_NO_S2W) and the CPU oracle at the bf16
    tolerances.  T = 1, 2, 3: the loop is left after any step; T = 64 = the buffers' length: the prefetch runs into the guard steps."""
    rng = np.random.RandomState(700 + T)
    P, C, PS = 9, 7, 11
    layers = net_desc(P, [(kind, size), (kind, size)], C)
    weights = random_weights(layers, rng, 0.05)
    lengths = [max(1, T - (i % 4) * (T // 4)) for i in range(PS - 1)]
    lengths[0] = T
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    real = real_mask(frac)
    got = {}
    for mode, env in (("asm", None), ("cpp", "CN_NO_S2W_ASM"), ("cluster", "CN_NO_S2W")):
        if env:
            monkeypatch.setenv(env, "1")
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16) as net:
            net.load_sequences(frac); net.compute_forward_pass()
            e, c = net.error_and_correct()
            name = net.recurrent_kernel(False)
            vals = {"error": np.float32(e), "out": net.outputs().reshape(-1, C)[real]}
            for lay in net.layers[1:3]:
                for dd in range(lay.dirs):
                    for nm in ("cellStates", "niActs", "igActs", "fgActs", "ogActs", "tmpOutputs"):
                        vals["%s/%d/%s" % (lay.name, dd, nm)] = lay.internal(nm, dd).reshape(-1, lay.H)[real]
            got[mode] = (name, vals)
    assert got["asm"][0] == "lstm_fwd_s2w_asm_kernel" and got["cpp"][0] == "lstm_fwd_s2w_kernel"
    assert got["cluster"][0].startswith("lstm_fwd_cluster"), got["cluster"][0]
    for key, v in got["asm"][1].items():
        assert np.array_equal(v, got["cpp"][1][key]), (key, np.abs(v - got["cpp"][1][key]).max())
    ref = orc.OracleNetwork(layers, weights, PS, frac["T"])
    ref.load_sequences(frac); ref.compute_forward_pass()
    want = ref.outputs().reshape(-1, C)[real]
    assert np.abs(got["asm"][1]["out"] - want).max() < 3e-2
    assert np.abs(got["asm"][1]["out"] - got["cluster"][1]["out"]).max() < 3e-2


def test_first_layer_gradient_group_on_the_large_tiles_over_many_frames(pkg, monkeypatch):
    """The gradient group of the FIRST 256-wide layer (dW_in 2048 x 64 + two dW_rec 1024 x 256) has no member large enough for the
    256 x 256 kernel on its own; from 28 000 frames on (cn_gemm.hip: launch_gemm_tn_group) the dW_rec pair goes there anyway and
    dW_in follows on the small tiles -- it is the exposed tail of the LVCSR step.  Same products, another tiling and split-K: the
    layer's weightUpdates must agree with the all-small-tiles launch (CN_TNBIG_GROUP_MINK above the frame count) to the order of
    the fp32 atomics, here on 64 x 448 = 28 672 frames with ragged lengths."""
    rng = np.random.RandomState(77)
    P, C, PS, T = 40, 30, 64, 448
    layers = net_desc(P, [("blstm", 512)], C)
    weights = random_weights(layers, rng, 0.05)
    lengths = [T - (i % 7) * 40 for i in range(PS)]
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    got = {}
    for mode in ("large", "small"):
        if mode == "small":
            monkeypatch.setenv("CN_TNBIG_GROUP_MINK", "100000000")
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16) as net:
            net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass()
            got[mode] = {lay.name: lay.weight_updates() for lay in net.trainable_layers()}
    for name, g in got["large"].items():
        assert np.abs(g).max() > 0
        assert rel_err(g, got["small"][name]) < 2e-5, (name, rel_err(g, got["small"][name]))


@pytest.mark.parametrize("kind,size,T", [("blstm", 500, 1), ("blstm", 500, 2), ("blstm", 500, 3), ("blstm", 500, 5), ("lstm", 250, 6), ("blstm", 512, 37),
                                         ("blstm", 500, 64)])
def test_hand_written_two_cu_backward_loop_equals_its_compiled_twin_bit_for_bit(pkg, orc, monkeypatch, kind, size, T):
    """Hp = 256, backward, clusters of two CUs with two sequences each (cn_lstm_cluster.hip: lstm_bwd_s2c_asm_kernel, generated
    loop text, 32 MFMAs per wave and step around the exchange) against lstm_bwd_s2c_kernel, the compiled kernel of the same cut:
    all four deltas on every real slot bit-identical, gradients to the split-K atomics' last bits; and against the 8-wave cluster
    kernel it replaces at the bf16 tolerances (another summation order of the same product).  T = 1, 2, 3, 5: the loop is left
    after any step of its four-step body; T = 64 = the buffers' length: the prefetch runs into the guard steps."""
    rng = np.random.RandomState(800 + T)
    P, C, PS = 9, 7, 11
    layers = net_desc(P, [(kind, size), (kind, size)], C)
    weights = random_weights(layers, rng, 0.05)
    lengths = [max(1, T - (i % 4) * (T // 4)) for i in range(PS - 1)]
    lengths[0] = T
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    real = real_mask(frac)
    got = {}
    for mode, env in (("cluster", "CN_NO_S2C"), ("cpp", "CN_S2C"), ("asm", None)):
        if env:
            monkeypatch.setenv(env, "1")
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16) as net:
            net.load_sequences(frac); net.compute_forward_pass()
            net.compute_backward_pass()
            name = net.recurrent_kernel(True)
            vals = {}
            for lay in net.layers[1:3]:
                for dd in range(lay.dirs):
                    for nm in ("niDeltas", "igDeltas", "fgDeltas", "ogDeltas"):
                        vals["%s/%d/%s" % (lay.name, dd, nm)] = lay.internal(nm, dd).reshape(-1, lay.H)[real]
            grads = {lay.name: lay.weight_updates() for lay in net.trainable_layers()}
            got[mode] = (name, vals, grads)
        if env:
            monkeypatch.delenv(env)
    assert got["asm"][0] == "lstm_bwd_s2c_asm_kernel" and got["cpp"][0] == "lstm_bwd_s2c_kernel"
    assert got["cluster"][0].startswith("lstm_bwd_cluster_kernel"), got["cluster"][0]
    for key, v in got["asm"][1].items():
        assert np.array_equal(v, got["cpp"][1][key]), (key, np.abs(v - got["cpp"][1][key]).max())
        assert np.abs(v - got["cluster"][1][key]).max() <= 2.0 ** -7 * max(1e-3, np.abs(got["cluster"][1][key]).max()), key
    for name, g in got["asm"][2].items():
        assert rel_err(g, got["cpp"][2][name]) < 1e-6, name
        assert rel_err(g, got["cluster"][2][name]) < 2e-3, name


@pytest.mark.parametrize("kind,size,T", [("blstm", 250, 1), ("blstm", 250, 2), ("blstm", 250, 5), ("lstm", 128, 9), ("blstm", 256, 37), ("blstm", 250, 64)])
def test_hand_written_split_bf16_loops_equal_the_compiled_kernels_bit_for_bit(pkg, monkeypatch, kind, size, T):
    """The same for the split-bf16 (CN_PREC_BF16X3) hand-written loops of the s2 cut against the compiled kernels of that cut
    (CN_S2_X3 selects the cut for both runs, CN_NO_S2_ASM the compiled kernels): every value on a real slot bit-identical."""
    monkeypatch.setenv("CN_S2_X3", "1")
    rng = np.random.RandomState(600 + T)
    P, C, PS = 9, 7, 11
    layers = net_desc(P, [(kind, size), (kind, size)], C)
    weights = random_weights(layers, rng, 0.08)
    lengths = [max(1, T - (i % 4) * (T // 4)) for i in range(PS - 1)]
    lengths[0] = T
    xs, ts = random_sequences(rng, lengths, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    real = real_mask(frac)
    got = {}
    for mode in ("asm", "cpp"):
        if mode == "cpp":
            monkeypatch.setenv("CN_NO_S2_ASM", "1")
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_BF16X3) as net:
            net.load_sequences(frac); net.compute_forward_pass()
            e, c = net.error_and_correct()
            net.compute_backward_pass()
            names = (net.recurrent_kernel(False), net.recurrent_kernel(True))
            vals = {"error": np.float32(e), "out": net.outputs().reshape(-1, C)[real]}
            for lay in net.layers[1:3]:
                for dd in range(lay.dirs):
                    for name in ("cellStates", "niActs", "igActs", "fgActs", "ogActs", "tmpOutputs", "niDeltas", "igDeltas", "fgDeltas", "ogDeltas"):
                        vals["%s/%d/%s" % (lay.name, dd, name)] = lay.internal(name, dd).reshape(-1, lay.H)[real]
            grads = {lay.name: lay.weight_updates() for lay in net.trainable_layers()}
            got[mode] = (names, vals, grads)
    assert got["asm"][0] == ("lstm_fwd_s2_x3_asm_kernel", "lstm_bwd_s2_x3_asm_kernel")
    assert got["cpp"][0] == ("lstm_fwd_s2_kernel<2,128>", "lstm_bwd_s2_kernel<2,128>")
    for key, v in got["asm"][1].items():
        assert np.array_equal(v, got["cpp"][1][key]), (key, np.abs(v - got["cpp"][1][key]).max())
    for name, g in got["asm"][2].items():
        assert rel_err(g, got["cpp"][2][name]) < 1e-6, name


@pytest.mark.parametrize("precision", [0, 1])
def test_armed_update_equals_update_after_the_backward_pass(pkg, precision):
    """cn_ctx_arm_update: each layer's UpdateWeightFn step (SteepestDescentOptimizer.cu:39-59) rides on the launch that unpacks
    its gradient, behind its own gradient GEMMs, instead of one launch for all layers behind the last backward kernel.  A layer's
    weights are last read by its own backward pass, so four momentum steps must end in the same weights (gradient sums are
    split-K atomics: equal to ~1e-6), the flat weightUpdates stay readable, a JSON learningRate of a layer is honoured, and both
    ways of completing the step (cn_sgd_update_all, cn_sgd_update per layer) work."""
    rng = np.random.RandomState(90 + precision)
    P, C, PS = 13, 9, 8
    layers = net_desc(P, [("blstm", 64), ("feedforward_tanh", 24), ("lstm", 32)], C)
    layers[2]["learningRate"] = 3e-3
    weights = random_weights(layers, rng, 0.2)
    fracs = []
    for k in range(2):
        xs, ts = random_sequences(rng, [17, 16, 16, 12, 9, 9, 4], P, C=C)
        fracs.append(pkg.make_fraction(xs, ts, PS))
    lr, mom = 1e-2, 0.9
    res = {}
    for mode in ("plain", "armed_all", "armed_per_layer"):
        with pkg.NeuralNetwork(layers, weights, PS, 17, precision=precision) as net:
            for step in range(4):
                net.load_sequences(fracs[step % 2]); net.compute_forward_pass(); net.calculate_error()
                if mode != "plain":
                    net.arm_update(lr, mom)
                net.compute_backward_pass()
                if mode == "armed_per_layer":
                    net.update_weights(lr, mom)
                else:
                    net.update_weights_fused(lr, mom)
            res[mode] = ({l.name: l.weights() for l in net.trainable_layers()}, {l.name: l.weight_updates() for l in net.trainable_layers()})
    for mode in ("armed_all", "armed_per_layer"):
        for name, w in res["plain"][0].items():
            assert np.abs(res[mode][0][name] - w).max() < 2e-6 * max(1.0, np.abs(w).max()), (mode, name)
            g = res["plain"][1][name]
            assert np.abs(res[mode][1][name] - g).max() < 1e-5 * max(1e-3, np.abs(g).max()), (mode, name, "weightUpdates")
    moved = max(np.abs(res["plain"][0][l["name"]] - np.concatenate([np.ravel(weights[l["name"]][k]) for k in ("input", "bias", "internal")])).max()
                for l in layers if l["name"] in weights)
    assert moved > 1e-3


def test_armed_update_state_errors(pkg):
    """A second backward pass before the armed step was completed, and a completing call with other values, are refused."""
    rng = np.random.RandomState(91)
    layers = net_desc(5, [("lstm", 8)], 3)
    weights = random_weights(layers, rng, 0.2)
    xs, ts = random_sequences(rng, [6, 5], 5, C=3)
    frac = pkg.make_fraction(xs, ts, 2)
    with pkg.NeuralNetwork(layers, weights, 2, 6) as net:
        net.load_sequences(frac); net.compute_forward_pass(); net.calculate_error()
        net.arm_update(1e-2, 0.9); net.compute_backward_pass()
        with pytest.raises(pkg.CurrenntHipError, match="differ from what cn_ctx_arm_update"):
            net.update_weights_fused(2e-2, 0.9)
        net.arm_update(1e-2, 0.9) if False else None
        net.load_sequences(frac); net.compute_forward_pass(); net.calculate_error()
        with pytest.raises(pkg.CurrenntHipError, match="armed update"):
            net.compute_backward_pass()
