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


def _post_loss64(kind, y, t, L):
    """fp64 statement of calculateError() of the post output layers (formulas of the layer headers)."""
    if kind == "sse":
        return 0.5 * ((t - y) ** 2).sum()
    if kind == "weightedsse":
        tt = t.reshape(-1, L, 2)
        return 0.5 * (((y - tt[..., 0]) * tt[..., 1]) ** 2).sum()
    if kind == "wf":
        tt = t.reshape(-1, L, 2)
        return 0.5 * ((y * tt[..., 1] - tt[..., 0]) ** 2).sum()
    if kind == "ce":
        tiny = torch.tensor(np.finfo(np.float32).tiny, dtype=torch.float64)
        return (t * torch.log(torch.maximum(t, tiny) / torch.maximum(y, tiny))).sum()
    if kind == "rmse":
        return torch.sqrt(((y - t) ** 2).sum(1) / L).sum()
    p = torch.where(t > 0, y, 1 - y)
    return (-torch.log(p)).sum()


def test_post_output_layers_against_fp64_autograd(orc):
    """orc_post_error / orc_post_backward against an fp64 autograd statement of the same losses.  Two
    layers are documented exceptions where the reference does not inject the derivative of its error:
    rmse injects rmse * (y - t) (RmsePostOutputLayer.cu:73-97) and weightedsse injects (y - t) * w, one
    factor w short of the derivative (WeightedSsePostOutputLayer.cu:66-93)."""
    import ctypes  # noqa: F401
    rng = np.random.RandomState(4)
    N, L = 23, 5
    pat = np.ones(N, np.int8); pat[[3, 17]] = 0
    lib = orc.lib()
    for kind in ("sse", "weightedsse", "wf", "ce", "rmse", "binary_classification"):
        Lk = 1 if kind == "binary_classification" else L
        W = 2 * Lk if kind in ("weightedsse", "wf") else Lk
        if kind in ("ce", "binary_classification"):
            y = rng.uniform(0.05, 0.95, (N, Lk)).astype(np.float32)
        else:
            y = rng.randn(N, Lk).astype(np.float32)
        if kind == "ce":
            t = rng.rand(N, W).astype(np.float32); t[:, 1] = 0
        elif kind == "binary_classification":
            t = rng.randint(0, 2, (N, W)).astype(np.float32)
        else:
            t = rng.randn(N, W).astype(np.float32)
        k = orc.POST[kind]
        e = lib.orc_post_error(k, Lk, N, pat, t.reshape(-1), y.reshape(-1))
        err = np.full(N * Lk, 7.0, np.float32)
        lib.orc_post_backward(k, Lk, N, pat, t.reshape(-1), y.reshape(-1), err)
        err = err.reshape(N, Lk)
        real = pat != 0
        y64 = torch.tensor(y[real].astype(np.float64), requires_grad=True)
        t64 = torch.tensor(t[real].astype(np.float64))
        loss = _post_loss64(kind, y64, t64, Lk)
        assert abs(e - loss.item()) <= 2e-6 * max(1.0, abs(loss.item())), kind
        assert np.all(err[~real] == 0), kind
        if kind == "rmse":
            rm = np.sqrt(((y[real] - t[real]).astype(np.float64) ** 2).sum(1, keepdims=True) / Lk)
            want = rm * (y[real] - t[real])
        elif kind == "weightedsse":
            tt = t[real].astype(np.float64).reshape(-1, Lk, 2)
            want = (y[real] - tt[..., 0]) * tt[..., 1]
        else:
            loss.backward()
            want = y64.grad.numpy()
        assert np.abs(err[real] - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), kind
    # ce clips the injected error to +-100 (CePostOutputLayer.cu:95)
    y = np.full((1, 2), 1e-6, np.float32); t = np.array([[1.0, 0.0]], np.float32)
    err = np.zeros(2, np.float32)
    lib.orc_post_backward(orc.POST["ce"], 2, 1, np.ones(1, np.int8), t.reshape(-1), y.reshape(-1), err)
    assert err[0] == -100.0 and err[1] == 0.0
