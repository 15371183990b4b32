"""CPU: the oracle's operand-rounding mode (oracle.set_operand_rounding("bf16"), currennt_oracle.c "operand rounding") against
an independent numpy statement of the same model.  Mode None must stay the reference's arithmetic (test_oracle_ref.py holds it
bit-equal to oracle/_ref; here: switching the mode on and off leaves it unchanged)."""
import numpy as np

from helpers import net_desc, random_sequences, random_weights


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def test_products_round_both_operands(orc):
    rng = np.random.RandomState(0)
    L = orc.lib()
    rowsA, colsA, colsB = 7, 13, 5
    a = rng.randn(rowsA * colsA).astype(np.float32); b = rng.randn(colsA * colsB).astype(np.float32)
    A, Bm = a.reshape(colsA, rowsA).T, b.reshape(colsB, colsA).T                 # column-major views
    c0 = np.zeros(rowsA * colsB, np.float32); c1 = np.zeros_like(c0); c2 = np.zeros_like(c0)
    L.orc_matmul(0, c0, a, rowsA, colsA, b, colsA, colsB, 0)
    with orc.operand_rounding("bf16"):
        assert orc.get_operand_rounding() == "bf16"
        L.orc_matmul(0, c1, a, rowsA, colsA, b, colsA, colsB, 0)
    assert orc.get_operand_rounding() is None
    L.orc_matmul(0, c2, a, rowsA, colsA, b, colsA, colsB, 0)
    assert np.array_equal(c0, c2)                                                # the mode leaves nothing behind
    want = (bf16_round(A).astype(np.float64) @ bf16_round(Bm).astype(np.float64)).T.reshape(-1)
    assert np.abs(c1 - want).max() < 1e-5 and np.abs(c0 - want).max() > 1e-4   # rounded operands, fp32 sums


def test_lstm_layer_in_rounding_mode_against_numpy(pkg, orc):
    """One unidirectional LSTM layer + softmax, forward and the input-weight gradient, restated in numpy float64 with the
    same rounding points: x, W_in, W_rec, y[t-1] rounded; deltas rounded inside products only; y stored rounded."""
    rng = np.random.RandomState(3)
    P, H, C, PS = 5, 6, 4, 2
    layers = net_desc(P, [("lstm", H)], C)
    weights = random_weights(layers, rng, 0.5)
    xs, ts = random_sequences(rng, [7, 7], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    with orc.operand_rounding("bf16"):
        net = orc.OracleNetwork(layers, weights, PS, 7)
        net.load_sequences(frac); net.compute_forward_pass(); net.calculate_error(); net.compute_backward_pass()
    lay = net.layers[1]
    T, N = 7, 14
    w = lay.weights.astype(np.float64)
    Win = bf16_round(lay.weights[:4 * H * P]).astype(np.float64).reshape(4, H, P)
    bias = w[4 * H * P:4 * H * P + 4 * H].reshape(4, H)
    Wrec = bf16_round(lay.weights[4 * H * (P + 1):4 * H * (P + 1) + 4 * H * H]).astype(np.float64).reshape(4, H, H)
    peep = w[4 * H * (P + 1) + 4 * H * H:].reshape(3, H)
    x = bf16_round(frac["inputs"].reshape(N, P)).astype(np.float64).reshape(T, PS, P)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    tanh = lambda v: 2.0 * sig(2.0 * v) - 1.0
    y = np.zeros((T, PS, H)); c = np.zeros((T, PS, H))
    for t in range(T):
        pre = np.einsum("ghp,sp->gsh", Win, x[t]) + bias[:, None, :]
        if t:
            pre += np.einsum("ghk,sk->gsh", Wrec, y[t - 1])
            pre[1] += c[t - 1] * peep[0]; pre[2] += c[t - 1] * peep[1]
        ni, ig, fg = tanh(pre[0]), sig(pre[1]), sig(pre[2])
        c[t] = ni * ig + (c[t - 1] * fg if t else 0.0)
        og = sig(pre[3] + c[t] * peep[2])
        y[t] = bf16_round((tanh(c[t]) * og).astype(np.float32))
    got = lay.outputs[:N * H].reshape(T, PS, H)
    assert np.array_equal(got, bf16_round(got))                                  # stored rounded
    d = np.abs(got - y)
    assert d.max() <= 2.0 ** -7 and (d > 0).mean() < 0.05                        # equal up to a rare rounding-boundary flip
    # input-weight gradient of the ni gate: sum_n bf16(x[n][p]) * bf16(delta_ni[n][h]); bias gradient from the UNROUNDED deltas
    dni = lay.internal("niDeltas")[:N * H].reshape(N, H)
    g = (x.reshape(N, P)[:, None, :] * bf16_round(dni).astype(np.float64)[:, :, None]).sum(0)            # [H][P]
    assert np.abs(lay.weightUpdates[:H * P].reshape(H, P) - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    gb = dni.astype(np.float64).sum(0)
    assert np.abs(lay.weightUpdates[4 * H * P:4 * H * P + H] - gb).max() < 1e-5 * max(1.0, np.abs(gb).max())
    assert np.abs(bf16_round(dni) - dni).max() > 0                               # (the two differ: the test can tell them apart)
