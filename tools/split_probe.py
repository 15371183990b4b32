#!/usr/bin/env python3
"""Probe (round 4): what would splitting a fraction into k sub-fractions by SEQUENCES buy?  k independent contexts of PS/k
sequences each (own weights: an upper bound on the update cost), enqueued layer by layer in turn on k streams, the later ones
delayed at the step start so that one sub-fraction's N-wide products run beside the other's recurrent kernels.  Compared with
one context of PS sequences.  Never a measurement of the product: it tells whether the split is worth building into the library."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench as Bn

def main():
    import torch
    pkg = ge.load_package()
    from lstm_rnn_amd import binding as B
    wl = Bn.WORKLOADS[os.environ.get("WL", "timit_3x250_blstm_H125")]
    PS = int(os.environ.get("PS", wl.get("PS", 50))); tmin, tmax = wl.get("tmin", 250), wl.get("tmax", 350)
    P, C, hidden = wl["P"], wl["C"], wl["hidden"]
    layers = Bn.net_desc(P, hidden, C); weights = Bn.make_weights(layers, 1234)
    dev = torch.device("cuda", 0)
    steps, warm = int(os.environ.get("STEPS", 20)), 3
    lib = pkg.load_library()
    for k in [int(x) for x in os.environ.get("KS", "1,2,3,4").split(",")]:
        for delay_us in [float(x) for x in os.environ.get("DELAYS", "0,10,20,40").split(",")]:
            if k == 1 and delay_us: continue
            rng = np.random.RandomState(1234)
            nfrac = 4
            nets = [pkg.NeuralNetwork(layers, weights, PS // k, tmax, precision=pkg.PREC_BF16) for _ in range(k)]
            keep, dfr = [], [[] for _ in range(k)]
            frames = 0
            for i in range(nfrac):
                xs, ts = Bn.synth_sequences(rng, PS, P, C, tmin, tmax)        # sorted ascending; sub-fraction j takes a contiguous slice
                for j in range(k):
                    sl = slice(j * (PS // k), (j + 1) * (PS // k))
                    f = pkg.make_fraction(xs[sl], ts[sl], PS // k)
                    x = torch.from_numpy(f["inputs"]).to(dev); pt = torch.from_numpy(f["patTypes"]).to(dev); tc = torch.from_numpy(f["targetClasses"]).to(dev)
                    keep += [x, pt, tc]
                    dfr[j].append({"T": f["T"], "Tmin": f["Tmin"], "numSeqs": f["numSeqs"], "inputPatternSize": P, "outputPatternSize": C,
                                   "inputs": x.data_ptr(), "patTypes": pt.data_ptr(), "targetClasses": tc.data_ptr(), "frames": pkg.fraction.real_frames(f)})
            streams = [n.torch_stream(torch) for n in nets]
            cyc = int(delay_us * 100)          # torch.cuda._sleep counts ~ 100 MHz realtime ticks? calibrated below
            def step(i):
                fr = 0
                for j, n in enumerate(nets):
                    if j and delay_us:
                        with torch.cuda.stream(streams[j]):
                            torch.cuda._sleep(int(j * delay_us * SLEEP_PER_US))
                    n.load_sequences_resident(dfr[j][i % nfrac]); fr += dfr[j][i % nfrac]["frames"]
                for li in range(len(nets[0].layers)):
                    for n in nets:
                        B.check(lib.cn_layer_forward(n.layers[li].handle), n.ctx)
                for n in nets: n.loss_accumulate()
                for li in reversed(range(len(nets[0].layers))):
                    for n in nets:
                        B.check(lib.cn_layer_backward(n.layers[li].handle), n.ctx)
                for n in nets: n.update_weights_fused(1e-4, 0.9)
                return fr
            for i in range(warm): step(i)
            best = []
            for rep in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fr = 0
                for i in range(steps): fr += step(warm + i)
                torch.cuda.synchronize(); best.append((time.perf_counter() - t0, fr))
            dt, fr = sorted(best)[len(best) // 2]
            print("k=%d delay=%4.0f us: %.3f ms/step  %.2f M frames/s" % (k, delay_us, 1e3 * dt / steps, fr / dt / 1e6), flush=True)
            for n in nets: n.close()
            del keep

if __name__ == "__main__":
    import torch
    # calibrate torch.cuda._sleep
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    a.record(); torch.cuda._sleep(1000000); b.record(); torch.cuda.synchronize()
    SLEEP_PER_US = 1000000 / (a.elapsed_time(b) * 1e3)
    print("sleep cycles per us: %.1f" % SLEEP_PER_US)
    main()
