"""What "posterior max-abs after 40 updates" can be asked of ANY fp32 implementation of this path (CPU, no GPU needed).

The reference sums a weight's gradient over the patterns serially in slot order (ComputeWeightUpdateFn, LstmLayer.cu:289-512;
one logical thread per weight, :502-510).  Put the same sequences into the slots of each fraction in reverse order: the
gradient is mathematically the same, its fp32 summation order is another.  Training the headline net (39 -> 3 x blstm250 ->
softmax183) with the oracle -- bit-equal to the reference's compiled functors, tests/test_oracle_ref.py -- on bench.py's
`parity_vs_cpu` task, the two orders stay within 1e-6 of each other through 20 updates and are 2.9e-4 apart after 40: the
reference differs from ITSELF by more than the north-star 1e-4 there.  The HIP f32 mode's distance to the oracle after the
same 40 updates (1.9e-4, BENCH `parity_vs_cpu.f32`; reproducible since round 6: fixed-order sums) is at that floor; the 1e-4
bound is asserted where it can hold: single pass (tests/test_gpu_parity.py) and through 20 updates (tests/test_gpu_configs.py).
This test pins the floor so that DESIGN.md's statement is checked, not claimed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_reference_arithmetic_differs_from_itself_after_forty_updates(pkg, orc):
    import bench
    wl = bench.WORKLOADS["timit_3x250_blstm_H125"]
    P, C, nseq = wl["P"], wl["C"], 6
    layers = bench.net_desc(P, wl["hidden"], C)
    weights = bench.make_weights(layers, 1234)
    rng = np.random.RandomState(77)
    tlen = int(max(6, min(60, 90e9 / 41 / bench.flops_per_frame(P, wl["hidden"], C) / nseq)))
    proj = rng.randn(2 * P, C).astype(np.float32)
    fwd, rev = [], []
    for _ in range(2):
        xs = [rng.randn(tlen - (i % 3), P).astype(np.float32) for i in range(nseq)]
        ts = [np.argmax(np.hstack([x, np.vstack([np.zeros((1, P), np.float32), x[:-1]])]) @ proj, axis=1).astype(np.int32) for x in xs]
        fwd.append(pkg.make_fraction(xs, ts, nseq)); rev.append(pkg.make_fraction(xs[::-1], ts[::-1], nseq))
    orc.set_threads(min(8, len(os.sched_getaffinity(0))))
    try:
        nets = [orc.OracleNetwork(layers, weights, nseq, tlen) for _ in range(2)]
        T0 = int(fwd[0]["T"])
        real = np.asarray(fwd[0]["patTypes"]).reshape(T0, nseq) != 0

        def distance():
            for net, fr in zip(nets, (fwd, rev)):
                net.load_sequences(fr[0]); net.compute_forward_pass()
            ya = nets[0].outputs().reshape(T0, nseq, C)
            yb = nets[1].outputs().reshape(T0, nseq, C)[:, ::-1, :]
            return float(np.abs(ya - yb)[real].max())
        seen = {}
        for k in range(40):
            for net, fr in zip(nets, (fwd, rev)):
                net.load_sequences(fr[k % 2]); net.compute_forward_pass(); net.calculate_error()
                net.compute_backward_pass(); net.update_weights(1e-2, 0.9)
            if k + 1 in (20, 40):
                seen[k + 1] = distance()
    finally:
        orc.set_threads(1)
    assert seen[20] < 1e-5, seen              # measured 1.5e-7: through 20 updates summation order does not matter
    assert seen[40] > 1e-4, seen              # measured 2.9e-4: after 40 it exceeds the north-star bound, reference vs reference
