"""-m gpu parity tests at the sizes BASELINE.json's `configs` name ("configs not exercised"):

  configs[0]  39 -> lstm128 -> softmax183                       (examples/phoneme_recognition_timit topology)
  configs[2]  39 -> blstm156 -> blstm300 -> blstm102 -> softmax51   (examples/speech_recognition_chime/
              no_subsampling/network.jsn, on KAT-0's real CHiME frames)
  configs[3]  40 -> 4 x blstm512 -> softmax8000                  (synthetic LVCSR; first test of the C = 8000 kernels)
  configs[4]  39 -> blstm1024 -> softmax183 at T = 2000, and 5 x blstm1024 at T = 2000 (cluster kernels,
              LDS dummy table T*4*RPL, 32-bit xch_epoch accumulation)

All through the C ABI against the CPU oracle on the same seeded inputs.  fp32 parity mode uses the standard
tolerances (posterior max-abs < 1e-4 = BASELINE.json north star, gradients < 2e-4 of the layer's max); bf16
throughput mode uses the bf16 tolerances of test_bf16_mode_close.  Reference call sequences being matched:
LstmLayer.cu:763-1051, SoftmaxLayer.cu:250-353, MulticlassClassificationLayer.cu:159-240."""
import os

import numpy as np
import pytest

from helpers import load_kat0, net_desc, random_sequences, random_weights, real_mask
from test_gpu_parity import POSTERIOR_TOL, check_network, rel_err, run_both

pytestmark = pytest.mark.gpu


def test_config0_literal_lstm128_softmax183(pkg, orc):
    """BASELINE configs[0]: 39 -> lstm 128 -> softmax 183, ragged lengths, a partial fraction (one unused slot)."""
    rng = np.random.RandomState(50)
    P, C, PS = 39, 183, 8
    layers = net_desc(P, [("lstm", 128)], C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, [60, 57, 57, 49, 41, 33, 20], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    with net:
        # three momentum steps on the same fraction: weights track the oracle (Q10)
        for step in range(3):
            for n in (ref, net):
                if step:
                    n.load_sequences(frac); n.compute_forward_pass(); n.compute_backward_pass()
            ref.update_weights(1e-3, 0.9); net.update_weights(1e-3, 0.9)
            for lay in net.trainable_layers():
                assert np.abs(lay.weights() - ref.layer(lay.name).weights).max() < 5e-6, (step, lay.name)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16x3_s4"])
def test_config1_headline_net_against_the_live_reference_library(pkg, orc, monkeypatch, mode):
    """BASELINE configs[1] as benchmarked -- 39 -> 3 x blstm250 -> softmax183, PS = 50 (one sequence per lane, 26 workgroups
    of the Hp = 128 register-resident kernels) -- against oracle/_ref, the REFERENCE's own compiled functors and Cpu GEMM
    (oracle/ref/ref_common.h), run live on this box: no restatement between the HIP path and the reference's arithmetic.
    oracle/_ref is built where /root/reference exists and travels to the GPU box as a binary; skipped where it is absent
    (tests/test_oracle_golden.py then carries the pin through committed vectors)."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libcurrennt_ref.so not shipped")
    rng = np.random.RandomState(60)
    P, C, PS = 39, 183, 50
    layers = net_desc(P, [("blstm", 250)] * 3, C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, sorted(rng.randint(14, 25, PS - 2).tolist(), reverse=True), P, C=C)     # two unused slots
    frac = pkg.make_fraction(xs, ts, PS)
    if mode == "bf16x3_s4":
        monkeypatch.setenv("CN_NO_S2", "1")
    ref, net = check_network(pkg, orc, layers, weights, frac, PS, backend="ref",
                             precision=pkg.PREC_F32 if mode == "f32" else pkg.PREC_BF16X3)
    with net:
        # f32: 4 sequences per workgroup, exact-fp32 MFMAs; bf16x3: the hand-written row-quad loops of the s2 cut (cn_lstm_s2.hip);
        # bf16x3_s4: the 4-sequence kernels in that mode (three MFMAs per product)
        want = {"f32": "lstm_%s_kernel<1,128,1,1>", "bf16x3": "lstm_%s_s2_x3_asm_kernel", "bf16x3_s4": "lstm_%s_kernel<2,128,1,1>"}[mode]
        assert net.recurrent_kernel(False) == want % "fwd" and net.recurrent_kernel(True) == want % "bwd"


CHIME_LAYERS = [("blstm", 156), ("blstm", 300), ("blstm", 102)]


def chime_fraction(pkg):
    _, _, xs, ts = load_kat0()              # first 10 sequences of val_1_speaker.nc, 39-d, 51 classes
    return xs, ts, pkg.make_fraction(xs, ts, 10)


def test_config2_literal_chime_network_fp32(pkg, orc):
    """BASELINE configs[2]: the literal CHiME example topology (no_subsampling/network.jsn) in fp32 parity mode on
    KAT-0's real CHiME frames (T = 152, 1 345 real frames, 175 dummy slots)."""
    rng = np.random.RandomState(51)
    layers = net_desc(39, CHIME_LAYERS, 51)
    weights = random_weights(layers, rng, 0.1)
    xs, ts, frac = chime_fraction(pkg)
    ref, net = check_network(pkg, orc, layers, weights, frac, 10)
    net.close()


def test_config2_literal_chime_network_bf16x3(pkg, orc):
    """The same net and frames in CN_PREC_BF16X3, at the fp32 tolerances (H = 150 -> Hp = 160: streamed W_rec, split per step)."""
    rng = np.random.RandomState(51)
    layers = net_desc(39, CHIME_LAYERS, 51)
    weights = random_weights(layers, rng, 0.1)
    xs, ts, frac = chime_fraction(pkg)
    ref, net = check_network(pkg, orc, layers, weights, frac, 10, precision=pkg.PREC_BF16X3)
    net.close()


def test_config2_literal_chime_network_bf16(pkg, orc):
    """The same net in the bf16 throughput mode (H = 78 / 150 / 51 -> the 6-, 10- and 4-wave register-resident kernels)."""
    rng = np.random.RandomState(51)
    layers = net_desc(39, CHIME_LAYERS, 51)
    weights = random_weights(layers, rng, 0.1)
    xs, ts, frac = chime_fraction(pkg)
    ref, net, (e_ref, _), (e, _) = run_both(pkg, orc, layers, weights, frac, 10, precision=1)
    with net:
        real = real_mask(frac)
        assert np.abs(net.outputs().reshape(-1, 51)[real] - ref.outputs().reshape(-1, 51)[real]).max() < 3e-2
        assert abs(e - e_ref) < 1e-2 * e_ref
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 5e-2, lay.name


def lvcsr_case(pkg, depth):
    rng = np.random.RandomState(52)
    P, C, PS = 40, 8000, 4
    layers = net_desc(P, [("blstm", 512)] * depth, C)
    weights = random_weights(layers, rng, 0.05)
    xs, ts = random_sequences(rng, [22, 20, 17, 9], P, C=C)
    return layers, weights, pkg.make_fraction(xs, ts, PS), PS, C


@pytest.mark.parametrize("depth", [2, 4])
def test_config3_lvcsr_softmax8000_fp32(pkg, orc, depth):
    """BASELINE configs[3]: 40 -> {2,4} x blstm512 -> softmax 8000 in fp32 parity mode: the block-per-pattern
    C = 8000 softmax / multiclass kernels (SMW_VPT = 32 values per thread) and the 256-unit-per-direction layers."""
    layers, weights, frac, PS, C = lvcsr_case(pkg, depth)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS)
    with net:
        # every posterior row of a real frame sums to 1, dummy rows are untouched by the softmax
        real = real_mask(frac)
        y = net.outputs().reshape(-1, C)
        assert np.abs(y[real].sum(1) - 1.0).max() < 1e-5


def test_config3_lvcsr_softmax8000_bf16x3(pkg, orc):
    """2 x blstm512 -> softmax 8000 in CN_PREC_BF16X3 at the fp32 tolerances."""
    layers, weights, frac, PS, C = lvcsr_case(pkg, 2)
    ref, net = check_network(pkg, orc, layers, weights, frac, PS, precision=pkg.PREC_BF16X3)
    net.close()


def test_config3_lvcsr_softmax8000_bf16(pkg, orc):
    """The same 4-layer stack in bf16 mode (2-CU cluster kernels for Hp = 256, 256x256 LDS-DMA GEMM when it applies)."""
    layers, weights, frac, PS, C = lvcsr_case(pkg, 4)
    ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS, precision=1)
    with net:
        real = real_mask(frac)
        y, yr = net.outputs().reshape(-1, C)[real], ref.outputs().reshape(-1, C)[real]
        assert np.abs(y - yr).max() < 3e-2 and np.abs(y.sum(1) - 1.0).max() < 1e-4
        assert abs(e - e_ref) < 1e-2 * e_ref
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 6e-2, lay.name


@pytest.mark.parametrize("precision,tol,gtol", [(0, 1e-4, 2e-4), (2, 1e-4, 2e-4), (1, 3e-2, 6e-2)])
def test_config3_lvcsr_stack_at_sixteen_sequences_of_up_to_96_frames(pkg, orc, precision, tol, gtol):
    """The LVCSR stack (40 -> 2 x blstm512 -> softmax 8000) at a size where the kernels run the shapes they run in the bench:
    PS = 16 (eight workgroups per direction in the one-CU-per-pair forward loop of the bf16 mode, four 2-CU clusters per direction
    in its backward pass and in the split-bf16 mode), T = 96 with ragged lengths, ~1 300 frames x 8000 classes against the
    multi-threaded oracle; f32 and bf16x3 at the fp32 tolerances, bf16 at its own."""
    orc.set_threads(8)
    try:
        rng = np.random.RandomState(53)
        P, C, PS = 40, 8000, 16
        layers = net_desc(P, [("blstm", 512)] * 2, C)
        weights = random_weights(layers, rng, 0.05)
        lengths = [96, 96, 91, 90, 88, 85, 85, 80, 77, 77, 70, 64, 50, 33, 17, 5]
        xs, ts = random_sequences(rng, lengths, P, C=C)
        frac = pkg.make_fraction(xs, ts, PS)
        ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS, precision=precision)
        with net:
            if precision == 1:
                assert net.recurrent_kernel(False) == "lstm_fwd_s2w_asm_kernel" and net.recurrent_kernel(True) == "lstm_bwd_s2c_asm_kernel"
            real = real_mask(frac)
            y, yr = net.outputs().reshape(-1, C)[real], ref.outputs().reshape(-1, C)[real]
            assert np.abs(y - yr).max() < tol and np.abs(y.sum(1) - 1.0).max() < 1e-4
            assert abs(e - e_ref) < (1e-2 if precision == 1 else 1e-4) * e_ref
            for lay in net.trainable_layers():
                assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < gtol, lay.name
    finally:
        orc.set_threads(1)


@pytest.mark.parametrize("precision,tol,gtol", [(0, 1e-4, 2e-4), (2, 1e-4, 2e-4), (1, 3e-2, 6e-2)])
def test_config1_reading_b_three_layers_of_250_units_per_direction(pkg, orc, precision, tol, gtol):
    """BASELINE configs[1] in its second reading (39 -> 3 x blstm500 -> softmax 183: 250 units per direction, Hp = 256) at PS = 16,
    T = 96, ragged, one unused slot: the one-CU-per-pair forward loop + 2-CU backward clusters (bf16), the 4-CU clusters
    (bf16x3), the streaming kernels (f32), against the multi-threaded oracle."""
    orc.set_threads(8)
    try:
        rng = np.random.RandomState(54)
        P, C, PS = 39, 183, 16
        layers = net_desc(P, [("blstm", 500)] * 3, C)
        weights = random_weights(layers, rng, 0.06)
        lengths = [96, 95, 93, 90, 90, 84, 80, 71, 66, 60, 52, 41, 30, 12, 3]
        xs, ts = random_sequences(rng, lengths, P, C=C)
        frac = pkg.make_fraction(xs, ts, PS)
        ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS, precision=precision)
        with net:
            if precision == 1:
                assert net.recurrent_kernel(False) == "lstm_fwd_s2w_asm_kernel"
            real = real_mask(frac)
            y, yr = net.outputs().reshape(-1, C)[real], ref.outputs().reshape(-1, C)[real]
            assert np.abs(y - yr).max() < tol
            assert abs(e - e_ref) < (1e-2 if precision == 1 else 1e-4) * e_ref
            if precision != 1:
                assert c == c_ref
            for lay in net.trainable_layers():
                assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < gtol, lay.name
    finally:
        orc.set_threads(1)


def test_config4_long_utterance_T2000_cluster_vs_oracle(pkg, orc):
    """BASELINE configs[4] layer shape: 39 -> blstm1024 -> softmax183, T = 2000, bf16 8-CU cluster kernels against the
    fp32 oracle (about 90 GFLOP of oracle work).  Ragged: one sequence ends at 1 501, the fraction has an unused slot.
    The oracle runs multi-threaded here (bit-identical to its single-threaded order, oracle.set_threads)."""
    rng = np.random.RandomState(53)
    P, C, PS = 39, 183, 3
    layers = net_desc(P, [("blstm", 1024)], C)
    weights = random_weights(layers, rng, 0.05)
    xs, ts = random_sequences(rng, [2000, 1999, 1501], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref, net, (e_ref, c_ref), (e, c) = run_both(pkg, orc, layers, weights, frac, PS, precision=1)
    with net:
        real = real_mask(frac)
        y, yr = net.outputs().reshape(-1, C)[real], ref.outputs().reshape(-1, C)[real]
        assert np.all(np.isfinite(y)) and np.abs(y - yr).max() < 3e-2
        assert abs(e - e_ref) < 1e-2 * e_ref
        # the last and the first frames of the longest sequence (both directions have run the full 2000 steps there)
        h, hr = net.layers[1].outputs()[:, 0, :], ref.layers[1].outputs[:net.N * 1024].reshape(2000, PS, 1024)[:, 0, :]
        assert np.abs(h[[0, 1999]] - hr[[0, 1999]]).max() < 2e-2
        for lay in net.trainable_layers():
            assert rel_err(lay.weight_updates(), ref.layer(lay.name).weightUpdates) < 6e-2, lay.name


def _run_longutt(pkg, layers, weights, frac, PS, T, steps, no_cluster):
    if no_cluster:
        os.environ["CN_NO_CLUSTER"] = "1"
    try:
        with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_BF16) as net:
            errs = []
            for _ in range(steps):
                net.load_sequences(frac); net.compute_forward_pass()
                errs.append(net.error_and_correct()[0])
                net.compute_backward_pass(); net.update_weights_fused(1e-6, 0.9)
            net.synchronize()
            return errs, net.outputs(), [l.weights() for l in net.trainable_layers()]
    finally:
        os.environ.pop("CN_NO_CLUSTER", None)


def test_config4_long_utterance_5x1024_T2000_property(pkg):
    """BASELINE configs[4] as written: 39 -> 5 x blstm1024 -> softmax183, PS = 16, T = 2000, bf16.  No oracle at this size
    (2.3 TFLOP per pass); size-independent properties instead: everything finite, posterior rows sum to 1, two passes
    over the same fraction through the cluster kernels (granule tags count on across the 20 cluster launches: 40 020
    tags) give the same error as the single-CU streaming kernels (CN_NO_CLUSTER=1) on the same weights, and training
    two steps moves both paths to the same weights."""
    rng = np.random.RandomState(54)
    P, C, PS, T = 39, 183, 16, 2000
    layers = net_desc(P, [("blstm", 1024)] * 5, C)
    weights = random_weights(layers, rng, 0.02)
    xs, ts = random_sequences(rng, [T] * 14 + [T - 7, T - 450], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    e1, y1, w1 = _run_longutt(pkg, layers, weights, frac, PS, T, 2, no_cluster=False)
    e2, y2, w2 = _run_longutt(pkg, layers, weights, frac, PS, T, 2, no_cluster=True)
    real = real_mask(frac)
    y1r = y1.reshape(-1, C)[real]
    assert np.all(np.isfinite(e1)) and np.all(np.isfinite(y1r))
    assert np.abs(y1r.sum(1) - 1.0).max() < 1e-4
    assert abs(e1[0] - e2[0]) < 2e-3 * abs(e2[0]) and abs(e1[1] - e2[1]) < 2e-3 * abs(e2[1]), (e1, e2)
    assert np.abs(y1 - y2).max() < 2e-2
    for a, b, w0 in zip(w1, w2, [np.concatenate([weights[l["name"]][k] for k in ("input", "bias", "internal")])
                                  for l in layers if l["name"] in weights]):
        moved = np.abs(a - w0).max()
        assert moved > 0 and np.abs(a - b).max() < 0.1 * moved + 1e-7, (moved, np.abs(a - b).max())


def test_cluster_launches_back_to_back_with_different_T(pkg):
    """The cluster kernels' granule tags continue across launches (LstmRec::xch_epoch is advanced by the launcher): forward /
    backward launches back to back over fractions of different T must not match a previous launch's granules.  The same
    three fractions through the streaming kernels are the reference."""
    rng = np.random.RandomState(55)
    P, C, PS = 12, 6, 8
    layers = net_desc(P, [("blstm", 512), ("blstm", 1024)], C)
    weights = random_weights(layers, rng, 0.04)
    fracs = []
    for T in (9, 31, 4, 17):
        xs, ts = random_sequences(rng, [T - (i % 3 if T > 3 else 0) for i in range(PS)], P, C=C)
        fracs.append(pkg.make_fraction(xs, ts, PS))
    res = {}
    for mode in ("cluster", "stream"):
        if mode == "stream":
            os.environ["CN_NO_CLUSTER"] = "1"
        try:
            with pkg.NeuralNetwork(layers, weights, PS, 31, precision=pkg.PREC_BF16) as net:
                out = []
                for f in fracs:
                    net.load_sequences(f); net.compute_forward_pass()
                    e = net.error_and_correct()[0]
                    net.compute_backward_pass()
                    out.append((e, [l.weight_updates() for l in net.trainable_layers()]))
                res[mode] = out
        finally:
            os.environ.pop("CN_NO_CLUSTER", None)
    for (e1, g1), (e2, g2) in zip(res["cluster"], res["stream"]):
        assert abs(e1 - e2) < 1e-3 * abs(e2)
        for a, b in zip(g1, g2):
            assert rel_err(a, b) < 1e-2


def test_all_dummy_fraction_contributes_nothing(pkg):
    """A data-parallel rank whose share of the last global fraction is empty loads an all-dummy fraction (T = 1, Tmin = 0,
    every slot PATTYPE_NONE; DataSet::setShard): zero error, zero #correct, exactly zero gradients, and the following
    update leaves the weights where momentum alone puts them -- so the rank can take part in the all-reduce."""
    rng = np.random.RandomState(56)
    P, C, PS = 9, 5, 6
    layers = net_desc(P, [("blstm", 40), ("lstm", 24)], C)
    weights = random_weights(layers, rng, 0.2)
    xs, ts = random_sequences(rng, [12, 11, 11, 8, 5, 3], P, C=C)
    real_frac = pkg.make_fraction(xs, ts, PS)
    dummy = {"T": 1, "Tmin": 0, "PS": PS, "numSeqs": 0, "inputs": np.zeros((PS, P), np.float32),
             "patTypes": np.zeros(PS, np.int8), "targetClasses": np.full(PS, -1, np.int32)}
    for prec in (pkg.PREC_F32, pkg.PREC_BF16):
        with pkg.NeuralNetwork(layers, weights, PS, 12, precision=prec) as net:
            net.load_sequences(real_frac); net.compute_forward_pass(); net.compute_backward_pass()     # leave non-zero state behind
            net.load_sequences(dummy); net.compute_forward_pass()
            e, c = net.error_and_correct()
            net.compute_backward_pass()
            assert e == 0.0 and c == 0
            for lay in net.trainable_layers():
                assert np.all(lay.weight_updates() == 0.0), lay.name


@pytest.mark.parametrize("mode,post_bound,w_bound", [("f32", 1e-4, 1e-4), ("bf16x3", 8e-3, 8e-3)])
def test_config1_drift_through_training_stays_inside_the_measured_bound(pkg, orc, mode, post_bound, w_bound):
    """The real 39 -> 3 x blstm250 -> softmax183 net trained for 20 momentum-SGD updates (lr 1e-2, momentum 0.9) on a learnable
    task, HIP path against the oracle doing the same.  Single-pass parity is < 1e-4 in both modes (test_gpu_parity.py).  Through
    training the f32 mode now holds the NORTH-STAR bound itself, 1e-4 on the posteriors and on the weights: its gradient sums
    run in a fixed order (option "deterministic", on by default in this mode; round 6), so what the updates amplify is fp32
    summation ORDER against the oracle's serial sums only, the same every run -- with split-K atomics it was 0.7-2.1e-4 and
    different every run (BENCH_r05: 1.3e-4, the profiled run 2.1e-4; this test allowed 2.5e-4).  bf16x3: 3.3-4.3e-3
    (2^-16 per product term), pinned with a margin of ~2x."""
    rng = np.random.RandomState(77)
    P, C, nseq, tlen = 39, 183, 6, 40
    layers = net_desc(P, [("blstm", 250)] * 3, C)
    weights = random_weights(layers, rng, 0.1)
    proj = rng.randn(2 * P, C).astype(np.float32)
    fracs = []
    for _ in range(2):
        xs = [rng.randn(tlen - (i % 3), P).astype(np.float32) for i in range(nseq)]
        ts = [np.argmax(np.hstack([x, np.vstack([np.zeros((1, P), np.float32), x[:-1]])]) @ proj, axis=1).astype(np.int32) for x in xs]
        fracs.append(pkg.make_fraction(xs, ts, nseq))

    def train(net):
        errs = []
        for k in range(20):
            net.load_sequences(fracs[k % 2]); net.compute_forward_pass(); errs.append(net.calculate_error())
            net.compute_backward_pass(); net.update_weights(1e-2, 0.9)
        net.load_sequences(fracs[0]); net.compute_forward_pass()
        return errs
    ref = orc.OracleNetwork(layers, weights, nseq, tlen)
    eref = train(ref)
    assert eref[-1] < 0.95 * eref[0]                      # the task is learnable: the posteriors have moved
    real = real_mask(fracs[0])
    yr = ref.outputs().reshape(-1, C)[real]
    with pkg.NeuralNetwork(layers, weights, nseq, tlen, precision=pkg.PREC_F32 if mode == "f32" else pkg.PREC_BF16X3) as net:
        e = train(net)
        y = net.outputs().reshape(-1, C)[real]
        post = float(np.abs(y - yr).max())
        wmax = max(float(np.abs(l.weights() - ref.layer(l.name).weights).max()) for l in net.trainable_layers())
        assert abs(e[-1] - eref[-1]) < 2e-3 * eref[-1], (e[-1], eref[-1])
        assert post < post_bound and wmax < w_bound, (mode, post, wmax)


def test_config2_real_data_convergence_of_the_three_arithmetic_modes(tmp_path):
    """BASELINE configs[2] as a TRAINING run on real data: the literal CHiME example network
    (examples/speech_recognition_chime/no_subsampling/network.jsn: 39 -> blstm156 -> blstm300 -> blstm102 -> softmax51) on the
    reference's one real data file (tests/golden/val_1_speaker.nc: 102 sequences, 13 878 frames), first 90 sequences to train on,
    last 12 to validate on, identical initial weights (normal, sigma 0.1 as in the example's config.cfg), stochastic momentum
    SGD through the C++ driver in f32, bf16x3 and bf16 (tools/chime_convergence.py).  What the benchmarked bf16 arithmetic
    does to TRAINING (the single-pass distance is pinned elsewhere):
      (i)  lr 1e-5 (the example's), 20 epochs -- the smooth part of training, validation class error 96 % -> ~75 %: bf16 follows
           f32 epoch by epoch (measured: <= 0.07 % absolute on the class error, 3e-4 relative on the error; bound here: 0.5 %
           absolute, 0.5 % relative.  From epoch ~25 on the runs begin to decorrelate -- 0.9 % / 1 % seen at epoch 27 for bf16x3,
           the fp32-tolerance mode -- which is what (ii) looks at);
      (ii) lr 3e-5, 60 epochs -- training to ~40 %: the three runs decorrelate like any three SGD runs (from epoch ~12 on the
           class error of ONE mode moves by +-3 % from epoch to epoch, and one mode differs from itself by 2-3 % between runs),
           so what is looked at is the best epoch (what early stopping keeps) and the mean of the last ten.  Round 6: every
           mode runs with its gradient sums in a fixed order (`--deterministic true`; tools/chime_convergence.py), so a run is
           REPRODUCIBLE BIT FOR BIT -- asserted: the bf16 run twice gives the same table -- and the difference between two modes
           is a number, not noise: best / last ten = 38.83 / 43.04 (f32), 37.91 / 42.57 (bf16x3), 38.40 / 41.04 (bf16), i.e.
           the bf16 "penalty" on this task is -0.4 % / -2.0 % absolute (bf16 ends BETTER; three trajectories of a chaotic
           regime, each exactly repeatable).  Asserted one-sided: bf16 and bf16x3 end no more than 1.5 % absolute above f32
           on either figure."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import chime_convergence as cc
    smooth = cc.run(str(tmp_path), epochs=20, ps=10, lr=1e-5)
    f32, b16, x3 = smooth["modes"]["f32"], smooth["modes"]["bf16"], smooth["modes"]["bf16x3"]
    assert len(f32) == len(b16) == len(x3) == 20
    assert f32[-1]["val_class_err"] < f32[0]["val_class_err"] - 15.0                 # it trains
    for a, b, c in zip(f32, b16, x3):
        assert abs(a["val_class_err"] - b["val_class_err"]) <= 0.5, (a, b)
        assert abs(a["val_err"] - b["val_err"]) <= 5e-3 * a["val_err"], (a, b)
        assert abs(a["train_err"] - b["train_err"]) <= 5e-3 * a["train_err"], (a, b)
        assert abs(a["val_class_err"] - c["val_class_err"]) <= 0.5 and abs(a["val_err"] - c["val_err"]) <= 5e-3 * a["val_err"], (a, c)
    far = cc.run(str(tmp_path), epochs=60, ps=10, lr=3e-5)
    again = cc.run(str(tmp_path), epochs=60, ps=10, lr=3e-5, modes=("bf16",))
    assert again["modes"]["bf16"] == far["modes"]["bf16"]        # run-to-run spread = 0: the same table, every digit the driver prints
    stat = {}
    for mode, rows in far["modes"].items():
        ce = [r["val_class_err"] for r in rows]
        stat[mode] = (min(ce), float(np.mean(ce[-10:])))
    # With atomics (rounds 4-5) ONE mode differed from itself by 2-3 % between runs: best / last-ten of f32 38.0-41.3 / 41.1-43.8 %,
    # of bf16 39.1-42.7 / 40.9-45.6 over eight runs, and the mode-to-mode bound had to go.  With fixed-order sums the figures are
    # exact (docstring); a mode's penalty against f32 is asserted one-sided, and every mode must get from 91 % into the region.
    for mode in ("f32", "bf16", "bf16x3"):
        assert stat[mode][0] < 46.0 and stat[mode][1] < 49.0, (mode, stat)
    for mode in ("bf16", "bf16x3"):
        assert stat[mode][0] <= stat["f32"][0] + 1.5 and stat[mode][1] <= stat["f32"][1] + 1.5, (mode, stat)
