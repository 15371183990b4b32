"""-m gpu: the context option "deterministic" (include/currennt_hip.h, cn_ctx_set_option).

The reference's Cpu path is reproducible: every weight's gradient is ONE logical thread adding its patterns serially
(ComputeWeightUpdateFn, LstmLayer.cu:289-512; Matrix.cu:41-183; FeedForwardLayer.cu:82-102,200-207).  The HIP path cuts
those sums over workgroups (split-K gradient products, per-workgroup bias / peephole / column sums); with the option on
the partials are STORED and added in a fixed order by one thread per output, so two runs are bit-identical -- the default
in the parity modes (CN_PREC_F32, CN_PREC_BF16X3), opt-in for CN_PREC_BF16.  Tested here:
  * training the headline net twice in one process gives the same bits (all three arithmetic modes; with the option off the
    bf16 run is allowed to differ, which is what the option is for);
  * the fixed-order sums are the same sums: against the atomics path within fp32 summation noise, and against the oracle
    at the standard tolerances (the whole of test_gpu_parity.py runs with the option on, it is the f32 default);
  * every producer of partials: the small and the 256 x 256 gradient GEMM, the one-CU, two-sequence and cluster recurrent
    kernels, narrow and wide softmax rows, plain dense layers."""
import numpy as np
import pytest

from helpers import net_desc, random_sequences, random_weights

pytestmark = pytest.mark.gpu


def _learnable(pkg, rng, P, C, nseq, tlen, nfrac=2):
    proj = rng.randn(2 * P, C).astype(np.float32)
    fracs = []
    for _ in range(nfrac):
        xs = [rng.randn(tlen - (i % 7), P).astype(np.float32) for i in range(nseq)]
        ts = [np.argmax(np.hstack([x, np.vstack([np.zeros((1, P), np.float32), x[:-1]])]) @ proj, axis=1).astype(np.int32) for x in xs]
        fracs.append(pkg.make_fraction(xs, ts, nseq))
    return fracs


def _train(pkg, layers, weights, fracs, PS, T, prec, det, updates, lr=5e-4, armed=False):
    with pkg.NeuralNetwork(layers, weights, PS, T, precision=prec, deterministic=det) as net:
        errs = []
        for k in range(updates):
            net.load_sequences(fracs[k % len(fracs)]); net.compute_forward_pass(); errs.append(net.calculate_error())
            if armed:
                net.arm_update(lr, 0.9)        # every layer's update rides on the launch behind its gradient products (bench.py's step)
            net.compute_backward_pass(); net.update_weights_fused(lr, 0.9)
        g = [l.weight_updates().copy() for l in net.trainable_layers()]
        w = [l.weights().copy() for l in net.trainable_layers()]
    return errs, g, w


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_headline_net_trains_to_the_same_bits_twice(pkg, mode):
    """BASELINE configs[1]: 39 -> 3 x blstm250 -> softmax183 at PS = 50 (the benchmarked kernels), 20 momentum-SGD updates on a
    learnable task, twice in one process: errors of every fraction, final gradients and final weights are BIT-identical."""
    rng = np.random.RandomState(91)
    P, C, PS, T = 39, 183, 50, 64
    layers = net_desc(P, [("blstm", 250)] * 3, C)
    weights = random_weights(layers, rng, 0.1)
    fracs = _learnable(pkg, rng, P, C, PS, T)
    prec = {"f32": pkg.PREC_F32, "bf16x3": pkg.PREC_BF16X3, "bf16": pkg.PREC_BF16}[mode]
    det = True if mode == "bf16" else None                    # None: the library default, which must be ON in the parity modes
    runs = [_train(pkg, layers, weights, fracs, PS, T, prec, det, 20) for _ in range(2)]
    (e0, g0, w0), (e1, g1, w1) = runs
    assert e0[-1] < 0.97 * e0[0]                              # it trains: the weights have moved
    assert e0 == e1, (mode, [a - b for a, b in zip(e0, e1)])
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b), (mode, float(np.abs(a - b).max()))
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b), (mode, float(np.abs(a - b).max()))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_armed_update_adds_the_partials_itself_and_gets_the_same_bits(pkg, mode):
    """With cn_ctx_arm_update the launch that unpacks a layer's gradient, applies the momentum step and rebuilds the operand copies
    (pack_group_kernel, update = 2) ALSO adds the stored partial sums (PackFold) -- no fold launch at all on bench.py's step.  It
    forms the same sums in the same order as the fold launch of the unarmed path: 12 updates either way end in the same bits, and
    the armed run repeats itself."""
    rng = np.random.RandomState(92)
    P, C, PS, T = 39, 183, 50, 48
    layers = net_desc(P, [("blstm", 250)] * 2, C)
    weights = random_weights(layers, rng, 0.1)
    fracs = _learnable(pkg, rng, P, C, PS, T)
    prec = {"f32": pkg.PREC_F32, "bf16": pkg.PREC_BF16}[mode]
    plain = _train(pkg, layers, weights, fracs, PS, T, prec, True, 12)
    armed = [_train(pkg, layers, weights, fracs, PS, T, prec, True, 12, armed=True) for _ in range(2)]
    for run in armed:
        assert run[0] == plain[0], mode
        for a, b in zip(run[1] + run[2], plain[1] + plain[2]):
            assert np.array_equal(a, b), (mode, float(np.abs(a - b).max()))


def test_option_default_and_switch(pkg):
    layers = net_desc(5, [("blstm", 8)], 4)
    for prec, want in ((pkg.PREC_F32, 1), (pkg.PREC_BF16X3, 1), (pkg.PREC_BF16, 0)):
        with pkg.NeuralNetwork(layers, None, 3, 7, precision=prec, seed=3) as net:
            assert net.get_option("deterministic") == want
            net.set_option("deterministic", 1 - want)
            assert net.get_option("deterministic") == 1 - want
            with pytest.raises(pkg.CurrenntHipError, match="unknown option"):
                net.set_option("no_such_option", 1)


SHAPES = [
    # name, P, hidden, C, PS, T, modes
    ("one_cu_small", 20, [("blstm", 64), ("lstm", 48)], 11, 6, 24, ("f32", "bf16x3", "bf16")),
    ("headline_s2", 39, [("blstm", 250)] * 2, 183, 50, 40, ("f32", "bf16x3", "bf16")),
    ("two_cu_cluster_big_tn", 40, [("blstm", 512), ("blstm", 512)], 300, 16, 300, ("bf16x3", "bf16")),   # 4 800 frames: the 256 x 256 gradient GEMM, wide softmax rows
    ("eight_cu_cluster", 12, [("blstm", 1024)], 6, 8, 40, ("bf16",)),
    ("dense_hidden", 10, [("feedforward_tanh", 40), ("blstm", 32), ("feedforward_logistic", 24)], 7, 5, 30, ("f32", "bf16")),
]


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_fixed_order_sums_equal_the_atomic_sums_and_repeat(pkg, shape):
    """Every producer of partial sums, one backward pass: the deterministic gradient (i) repeats bit for bit over three passes
    on the same fraction, (ii) equals the atomics path's within fp32 summation noise (2e-5 of the layer's largest gradient in
    the fp32-operand modes; the bf16 mode's operands are the same in both paths, so the same bound holds)."""
    name, P, hidden, C, PS, T, modes = shape
    rng = np.random.RandomState(17)
    layers = net_desc(P, hidden, C)
    weights = random_weights(layers, rng, 0.08)
    xs, ts = random_sequences(rng, [T - (i % 5) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    for mode in modes:
        prec = {"f32": pkg.PREC_F32, "bf16x3": pkg.PREC_BF16X3, "bf16": pkg.PREC_BF16}[mode]
        grads = {}
        for det in (True, False):
            with pkg.NeuralNetwork(layers, weights, PS, T, precision=prec, deterministic=det) as net:
                passes = []
                for _ in range(3):
                    net.load_sequences(frac); net.compute_forward_pass(); e = net.calculate_error(); net.compute_backward_pass()
                    passes.append((e, [l.weight_updates().copy() for l in net.trainable_layers()]))
                grads[det] = passes
        e_first, g_first = grads[True][0]
        for e, g in grads[True][1:]:
            assert e == e_first, (name, mode)
            for a, b in zip(g, g_first):
                assert np.array_equal(a, b), (name, mode, float(np.abs(a - b).max()))
        for a, b in zip(g_first, grads[False][0][1]):
            scale = float(np.abs(b).max())
            assert scale > 0
            assert float(np.abs(a - b).max()) <= 2e-5 * scale, (name, mode, float(np.abs(a - b).max()) / scale)
