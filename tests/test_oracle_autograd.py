"""Second, independent check of the oracle: an fp64 autograd model of the equations of SURVEY.md
section 3.3a must reproduce the oracle's outputs and (absent delta clipping) its weightUpdates, which
the survey found equal to the exact gradient of the summed cross entropy."""
import numpy as np
import torch

from helpers import net_desc, random_sequences, random_weights, real_mask


def sig(x):
    return 1.0 / (1.0 + torch.exp(-x))


def tanh_(x):
    return 2.0 * sig(2.0 * x) - 1.0


def lstm_layer(x, lens, w, P, L, bidir, bias):
    """x: [T][PS][P] float64 tensor; returns [T][PS][L]."""
    T, PS, _ = x.shape
    dirs = 2 if bidir else 1
    H = L // dirs
    Win = w[:4 * L * P].reshape(4, dirs, H, P)
    Wb = w[4 * L * P:4 * L * (P + 1)].reshape(4, dirs, H)
    Wr = w[4 * L * (P + 1):4 * L * (P + 1) + 4 * L * H].reshape(4, dirs, H, H)
    Wp = w[4 * L * (P + 1) + 4 * L * H:].reshape(3, dirs, H)
    outs = []
    for d in range(dirs):
        ys = [None] * T
        y_prev = torch.zeros(PS, H, dtype=torch.float64)
        c_prev = torch.zeros(PS, H, dtype=torch.float64)
        order = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in order:
            a = [x[t] @ Win[g, d].T + y_prev @ Wr[g, d].T + bias * Wb[g, d] for g in range(4)]
            n = tanh_(a[0])
            i = sig(a[1] + Wp[0, d] * c_prev)
            f = sig(a[2] + Wp[1, d] * c_prev)
            c = n * i + f * c_prev
            o = sig(a[3] + Wp[2, d] * c)
            y = tanh_(c) * o
            m = torch.tensor([[1.0 if t < ln else 0.0] for ln in lens], dtype=torch.float64)
            y, c = y * m, c * m          # dummy slots: output 0, state 0 (Q1)
            ys[t] = y
            y_prev, c_prev = y, c
        outs.append(torch.stack(ys))
    return torch.cat(outs, dim=2)


def test_autograd_matches_oracle(pkg, orc):
    rng = np.random.RandomState(3)
    P, C, PS = 3, 3, 3
    layers = net_desc(P, [("blstm", 8), ("lstm", 4)], C)
    weights = random_weights(layers, rng, 0.5)
    lens = [6, 4, 2]
    xs, ts = random_sequences(rng, lens, P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    T = frac["T"]
    ref = orc.OracleNetwork(layers, weights, PS, T)
    ref.load_sequences(frac)
    ref.compute_forward_pass()
    err = ref.calculate_error()
    ref.compute_backward_pass()

    params = {}
    for lay in ref.trainable_layers():
        params[lay.name] = torch.tensor(lay.weights.astype(np.float64), requires_grad=True)
    x = torch.tensor(frac["inputs"].reshape(T, PS, P).astype(np.float64))
    h = lstm_layer(x, lens, params["blstm_0"], P, 8, True, 1.0)
    h = lstm_layer(h, lens, params["lstm_1"], 8, 4, False, 1.0)
    wo = params["output"]
    z = h @ wo[:C * 4].reshape(C, 4).T + 1.0 * wo[C * 4:]
    logp = torch.log_softmax(z, dim=2)
    tc = torch.tensor(frac["targetClasses"].reshape(T, PS).astype(np.int64))
    mask = tc >= 0
    loss = -(logp.gather(2, tc.clamp(min=0).unsqueeze(2)).squeeze(2) * mask).sum()
    loss.backward()

    assert abs(float(loss) - err) < 1e-4 * max(1.0, err)
    real = real_mask(frac)
    post = torch.exp(logp).detach().numpy().reshape(-1, C)[real]
    assert np.abs(post - ref.outputs().reshape(-1, C)[real]).max() < 1e-5
    for lay in ref.trainable_layers():
        g = params[lay.name].grad.numpy()
        gr = lay.weightUpdates
        assert np.abs(np.concatenate([lay.internal(n, d) for n in ("niDeltas", "igDeltas", "fgDeltas", "ogDeltas")
                                      for d in range(2 if lay.type == "blstm" else 1)])).max() < 1.0 \
            if lay.type in ("lstm", "blstm") else True            # no delta reached the +-1 clip
        assert np.abs(g - gr).max() < 2e-4 * max(1.0, np.abs(g).max()), (lay.name, np.abs(g - gr).max())
