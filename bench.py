#!/usr/bin/env python3
"""bench.py -- train frames/sec of the CURRENNT LSTM hot path on MI355X.

Metric (BASELINE.json): train frames/sec (node), 3x250 BLSTM 39->183, at 1/2/4/8 MI355X.
A step = one fraction (parallel_sequences sequences) through
    load (resident in HBM) -> forward -> loss -> backward (all weight gradients) -> [all-reduce] -> SGD update
i.e. Optimizer::_processDataSet's loop body with --stochastic semantics (Optimizer.cu:46-97).

One process per GPU (torch.distributed / RCCL when launched with --nproc-per-node N); the PS
sequences of a fraction are independent, so ranks take disjoint sequences (weak scaling: PS per GPU
is fixed) and meet only in one all-reduce(SUM) of the flat weightUpdates arena.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

# SURVEY.md section 8(d) workloads.  "size" is the CURRENNT JSON size (units of both directions).
WORKLOADS = {
    # reading A: the repo's own TIMIT recipe uses "size": 250 = 125 units per direction
    "timit_3x250_blstm_H125": dict(P=39, hidden=[("blstm", 250)] * 3, C=183),
    # reading B: Graves ASRU'13 literal, 250 units per direction = CURRENNT "size": 500
    "timit_3x500_blstm_H250": dict(P=39, hidden=[("blstm", 500)] * 3, C=183),
    # BASELINE.json configs[0] topology
    "timit_1x128_lstm": dict(P=39, hidden=[("lstm", 128)], C=183),
    # BASELINE.json configs[3]: synthetic LVCSR, 40-d fbank -> 4 x 512 BLSTM (256 per direction) -> 8000 tied states
    "lvcsr_4x512_blstm_8000": dict(P=40, hidden=[("blstm", 512)] * 4, C=8000, PS=64, tmin=300, tmax=800),
    # BASELINE.json configs[4]: long-utterance stress, 5 x 1024 BLSTM (512 per direction), T = 2000
    "longutt_5x1024_blstm": dict(P=39, hidden=[("blstm", 1024)] * 5, C=183, PS=16, tmin=2000, tmax=2000),
}
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak


def pmc_bytes_per_launch(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc.json, made by
    tools/make_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench's default
    command; FETCH_SIZE doubled per MI355X_MICROARCH.md).  None when no summary matches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for k, v in data.items():
        if kernel in k:
            return v["bytes_per_launch"], os.path.basename(files[-1])
    return None, None
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def mfma_busy_records(workload):
    """MFMA-busy per kernel from the newest committed SQ-counter summary (profiles/*_sq.md, made by tools/pmc_sq6.sh from separate
    rocprofv3 --pmc passes of this bench's command): {kernel: {"chip": share of all SIMD cycles of the chip with an MFMA in the
    pipe while the kernel runs, "active_cus": the same over the CUs that hold a wave of the kernel}}.  None without a summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq.md")))
    if not files:
        return None, None
    out, on = {}, False
    for line in open(files[-1]):
        if line.startswith("## "):
            on = line[3:].strip() == workload
        elif on and line.startswith("| `"):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            try:
                out[c[0].strip("`")] = {"chip": float(c[-3]), "active_cus": float(c[-1])}
            except ValueError:
                pass
    return (out or None), os.path.basename(files[-1])


def net_desc(P, hidden, C):
    layers = [{"name": "input", "type": "input", "size": P}]
    for i, (t, s) in enumerate(hidden):
        layers.append({"name": "%s_%d" % (t, i + 1), "type": t, "size": s, "bias": 1.0})
    layers.append({"name": "output", "type": "softmax", "size": C, "bias": 1.0})
    layers.append({"name": "postoutput", "type": "multiclass_classification", "size": C})
    return layers


def make_weights(layers, seed):
    """uniform [-0.1, 0.1] (Configuration.cpp:186-187) from a seeded generator, flat layout split like the JSON."""
    rng = np.random.RandomState(seed)
    out, prev = {}, None
    for d in layers:
        t, L = d["type"], d["size"]
        if t in ("lstm", "blstm"):
            P, H = prev["size"], L // (2 if t == "blstm" else 1)
            out[d["name"]] = {"input": rng.uniform(-.1, .1, 4 * L * P).astype(np.float32),
                              "bias": rng.uniform(-.1, .1, 4 * L).astype(np.float32),
                              "internal": rng.uniform(-.1, .1, 4 * L * H + 3 * L).astype(np.float32)}
        elif t == "softmax":
            out[d["name"]] = {"input": rng.uniform(-.1, .1, L * prev["size"]).astype(np.float32),
                              "bias": rng.uniform(-.1, .1, L).astype(np.float32), "internal": np.zeros(0, np.float32)}
        prev = d
    return out


def synth_sequences(rng, PS, P, C, tmin, tmax):
    """i.i.d. N(0,1) features, uniform targets, lengths U[tmin,tmax] sorted ascending (DataSet.cpp:603-605)."""
    lens = np.sort(rng.randint(tmin, tmax + 1, PS))
    xs = [rng.randn(n, P).astype(np.float32) for n in lens]
    ts = [rng.randint(0, C, n).astype(np.int32) for n in lens]
    return xs, ts


def synth_fraction(pkg, rng, PS, P, C, tmin, tmax):
    xs, ts = synth_sequences(rng, PS, P, C, tmin, tmax)
    return pkg.make_fraction(xs, ts, PS)


def flops_per_frame(P, hidden, C):
    """F_lstm = dirs*8H*[(P+H)*2 + H + (first?0:P)], softmax 6*C*P."""
    total, first, prev = 0.0, True, P
    for t, size in hidden:
        dirs = 2 if t == "blstm" else 1
        H = size // dirs
        total += dirs * 8 * H * ((prev + H) * 2 + H + (0 if first else prev))
        first, prev = False, size
    return total + 6 * C * prev


def rec_algorithmic(hidden):
    """per real frame: HBM bytes of the fused cell kernels (44 B fwd / 64 B bwd per unit-frame,
    SURVEY 8(a) rows a3/a6) and recurrent MFMA flops (2*4*H*H per direction, fwd and bwd each)."""
    units = sum(s for _, s in hidden)
    fl = sum((2 if t == "blstm" else 1) * 8 * (s // (2 if t == "blstm" else 1)) ** 2 for t, s in hidden)
    return 44.0 * units, 64.0 * units, fl


def gemm_products(wl, precision):
    """The N-wide products of one training step, per class, as (name, flop per frame, algorithmic HBM bytes per frame): what a
    product must move when every operand is read once and every output written once (weights, read once per launch, are noise
    beside N = 15 000 frames and left out).  `e` = bytes of an operand element, outputs that the NEXT kernel consumes as fp32
    (gate pre-activations, outputErrors, logits) are 4 bytes.  Call sites: LstmLayer.cu:774-785 (K1), :996-1006 (K8),
    :1038-1043 (K9), FeedForwardLayer.cu:148-206."""
    e = 2.0 if precision == "bf16" else 4.0
    wide, grad = [], []
    prev, first = wl["P"], True
    for i, (t, size) in enumerate(wl["hidden"]):
        dirs = 2 if t == "blstm" else 1
        H = size // dirs
        R = dirs * 4 * H
        wide.append(("L%d.K1 x->preacts" % (i + 1), 2.0 * R * prev, e * prev + 4.0 * R))
        if not first:
            wide.append(("L%d.K8 delta->err" % (i + 1), 2.0 * R * prev, e * R + 4.0 * prev))
        grad.append(("L%d.K9 dWin" % (i + 1), 2.0 * R * prev, e * (R + prev)))
        grad.append(("L%d.K9 dWrec" % (i + 1), 2.0 * R * H, e * (R + size)))
        prev, first = size, False
    C = wl["C"]
    wide.append(("out.fwd", 2.0 * C * prev, e * prev + 4.0 * C))
    wide.append(("out.K8", 2.0 * C * prev, e * C + 4.0 * prev))
    grad.append(("out.dW", 2.0 * C * prev, e * (C + prev)))
    return {"gemm_wide": wide, "gemm_grad": grad}


def gemm_roofline(wl, precision, tm, frames, steps):
    """`roofline_gemm`: per GEMM class the device time of the event-timed pass against the class's ATTAINABLE time, the sum over
    its products of max(flop / MFMA peak, algorithmic bytes / HBM peak) -- the gate GEMMs of these nets are bound by their fp32
    outputs, not by the matrix cores, so flop / 2.5 PF alone reads 7 % for ever."""
    peak_fl = PEAK_MFMA_TFLOPS.get(precision, 2500.0 / 3) * 1e12
    out = {}
    for cls, prods in gemm_products(wl, precision).items():
        ms = tm[cls][0]
        if not ms:
            continue
        roof_s, fl, by, hbm_bound = 0.0, 0.0, 0.0, 0
        worst = None
        for name, f, b in prods:
            t_f, t_b = f * frames / peak_fl, b * frames / (PEAK_HBM_GBS * 1e9)
            roof_s += max(t_f, t_b); fl += f * frames; by += b * frames
            hbm_bound += t_b >= t_f
        out[cls] = {"achieved_ms_per_step": ms / steps, "roof_ms_per_step": 1e3 * roof_s / steps, "frac": 1e3 * roof_s / ms,
                    "tflops": fl / (ms * 1e-3) / 1e12, "algorithmic_GBps": by / (ms * 1e-3) / 1e9,
                    "products": len(prods), "products_hbm_bound": int(hbm_bound), "launches": tm[cls][1]}
    out["note"] = ("roof = sum over the class's products of max(flop / %.0f TFLOP/s, algorithmic bytes / %.0f GB/s); frac = roof / achieved device "
                   "time (hipEvents on the launching stream); gemm_grad runs on side streams beside the recurrent kernels" % (peak_fl / 1e12, PEAK_HBM_GBS))
    # why a class sits below 0.6 of its roof, in one line each (measured: NOTEBOOK A.7)
    out["below_roof_because"] = {
        "gemm_wide": "every product runs with cold operands behind a recurrent kernel (the same kernels back to back: 0.7 x the time); the "
                     "input projections write 61 MB of fp32 pre-activations each (their stores alone take 14 of 27 us, bf16 results cost the "
                     "bf16 mode's pin a factor of two: option pre16); a fraction's rows are T_max x PS, 15-30 % of them dummy frames that the roof "
                     "does not count -- only the panel kernel (long-K, narrow-N products) skips them",
        "gemm_grad": "split-K over the frames in 64 x 64 tiles on the 80 CUs of a masked stream beside the recurrent kernels (hidden there: "
                     "only the first layer's group, ~30 us on the whole chip, is on the critical path); deeper prefetch measured slower",
    }
    return out


def spawn_ranks(n, argv):
    """One process per GPU through torch.distributed.run on 127.0.0.1; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # This process has touched no GPU and may therefore wait, time out and kill: the ranks run in a process group of their
    # own; if the job is not done within CN_BENCH_TIMEOUT seconds (a rank that never reaches the rendezvous leaves the others
    # waiting in ncclCommInitRank / a barrier for ever) the whole group is terminated and the exit code says so.
    import signal
    limit = float(os.environ.get("CN_BENCH_TIMEOUT", "1200"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: the %d-rank job did not finish within %.0f s; terminating its process group %d\n" % (n, limit, proc.pid))
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def roofline_records(res, wl, workload, PS, precision, value):
    """`roofline` (the recurrent kernel with the larger share of device time), `roofline_other` (the other one), `roofline_pair`
    (both together) and `roofline_mfma` from the event-timed pass of run_workload."""
    tm, fr = res["timing"], res["timing_frames"]
    b_fwd, b_bwd, fl_rec = rec_algorithmic(wl["hidden"])
    nl_f, nl_b = max(1, tm["rec_fwd"][1]), max(1, tm["rec_bwd"][1])
    nlayers = len(wl["hidden"])
    total_ms = sum(v[0] for k, v in tm.items() if k != "exchange")          # (the exchange runs on a stream of its own, beside the rest)

    def roof(bwd):
        dom = res["kernels"][1] if bwd else res["kernels"][0]          # the kernel the launcher instantiated (cn_layer_recurrent_kernel)
        ms, nl, bpf = (tm["rec_bwd"][0], nl_b, b_bwd) if bwd else (tm["rec_fwd"][0], nl_f, b_fwd)
        frames_per_launch = fr / (nl / float(nlayers))      # one launch = one layer pass over one fraction
        bytes_per_launch = bpf / nlayers * frames_per_launch
        avg_s = ms / nl * 1e-3
        ach = bytes_per_launch / avg_s / 1e9
        traffic, src = (None, None)
        if workload == "timit_3x250_blstm_H125" and PS == 50 and precision == "bf16":
            tb, src = pmc_bytes_per_launch(dom.split("<")[0])
            traffic = tb / avg_s / 1e9 if tb else None
        return {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": ms / nl, "launches": nl,
                "share_of_device_time": ms / total_ms}

    out = {}
    # the larger share of device time, whichever it is; two shares within 1.5 % of each other are a tie inside the run-to-run spread
    # of the event times (headline: 116.5 against 116.2 us per launch) -- then the backward kernel, which moves the more bytes
    # (64 against 44 B per unit-frame) and is the one every earlier round reported, so the series stays comparable and the record
    # does not flip between 0.25 and 0.18 on a coin toss; `roofline_other` carries the other kernel either way
    tie = abs(tm["rec_bwd"][0] - tm["rec_fwd"][0]) <= 0.015 * max(tm["rec_bwd"][0], tm["rec_fwd"][0])
    bwd_dom = tie or tm["rec_bwd"][0] >= tm["rec_fwd"][0]
    out["roofline"] = roof(bwd_dom)
    out["roofline"]["note"] = ("latency-bound persistent kernel (T sequential steps); " +
                               ("forward and backward kernel tie in device time (within 1.5 %): the backward kernel is reported, the forward one "
                                "is `roofline_other`; " if tie else "") +
                               "per-class device time [ms] over the event-timed pass: " + ", ".join("%s=%.2f" % (k, v[0]) for k, v in tm.items()))
    out["roofline_other"] = roof(not bwd_dom)
    pair_ms = tm["rec_fwd"][0] + tm["rec_bwd"][0]
    pair_bytes = (b_fwd + b_bwd) * fr
    out["roofline_pair"] = {"kernels": list(res["kernels"]), "bound": "hbm", "achieved": pair_bytes / (pair_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": pair_bytes / (pair_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "share_of_device_time": pair_ms / total_ms}
    fpf = flops_per_frame(wl["P"], wl["hidden"], wl["C"])
    gemm_ms = tm["gemm_wide"][0] + tm["gemm_grad"][0]
    gemm_fl = (fpf - fl_rec * 2) * fr
    out["roofline_mfma"] = {"gate_gemms_tflops": gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
                            "recurrent_tflops": fl_rec * 2 * fr / (pair_ms * 1e-3) / 1e12 if pair_ms else None,
                            "whole_step_tflops": fpf * value / 1e12, "peak": PEAK_MFMA_TFLOPS.get(precision, 2500.0 / 3),
                            "flop_per_frame": fpf, "event_pass_total_ms": total_ms}
    if precision == "bf16":
        busy, src = mfma_busy_records(workload)
        if busy:
            out["roofline_mfma"]["mfma_busy"] = busy
            out["roofline_mfma"]["mfma_busy_source"] = src + " (SQ_VALU_MFMA_BUSY_CYCLES / (4 x 32 x GRBM_GUI_ACTIVE) and / (4 x SQ_BUSY_CU_CYCLES))"
    out["roofline_gemm"] = gemm_roofline(wl, precision, tm, fr, res["timing_steps"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="timit_3x250_blstm_H125", choices=sorted(WORKLOADS))
    ap.add_argument("--parallel-sequences", type=int, default=None, help="per GPU (default: the workload's, 50 as in examples/*/config.cfg)")
    ap.add_argument("--tmin", type=int, default=None, help="sequence lengths are U[tmin, tmax] (default: the workload's, 250..350)")
    ap.add_argument("--tmax", type=int, default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "bf16x3"],
                    help="bf16: throughput mode; f32: exact-fp32 MFMA parity mode; bf16x3: split-bf16 parity mode (fp32 tolerance at a third of the bf16 MFMA rate)")
    ap.add_argument("--no-also", action="store_true", help="skip the informational extra workloads of the default run")
    ap.add_argument("--no-driver-leg", action="store_true", help="skip the end-to-end run of the C++ driver on a synthetic NetCDF file")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--also", default="", help="comma list of extra workloads (name or name:precision) measured and reported under 'also'")
    args = ap.parse_args()
    launched = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if not launched and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes BEFORE torch is
        # imported or any GPU call is made here (a process that has initialised the GPU must not exec), relay
        # rank 0's JSON line (the children inherit stdout) and hand the launcher's exit code on.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if launched and os.environ.get("CN_BENCH_TEST_HANG") in (os.environ.get("RANK"), "all"):
        time.sleep(3600)        # test hook (tests/test_parallel_gloo.py): a rank that never reaches the rendezvous
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    wl0 = WORKLOADS[args.workload]
    if args.parallel_sequences is None: args.parallel_sequences = wl0.get("PS", 50)
    if args.tmin is None: args.tmin = wl0.get("tmin", 250)
    if args.tmax is None: args.tmax = max(wl0.get("tmax", 350), args.tmin)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / cross-process device memory need dmabuf IPC on this host driver
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # CN_BENCH_BACKEND=gloo: test mode for boxes with fewer GPUs than ranks (ranks share devices; the reductions go
    # through the host).  It exercises the world > 1 control flow and stream ordering with real sums; never a measurement.
    # CN_BENCH_BACKEND=ipc: the same test situation, but the exchange is the LIBRARY's (cn_comm_init / cn_allreduce_grads /
    # cn_loss_read_global on its CN_COMM_BACKEND=ipc test backend); torch.distributed (gloo) carries the control plane only.
    # CN_BENCH_BACKEND=p2p: the library's NATIVE exchange (CN_COMM_BACKEND=p2p: one stream-ordered kernel per bucket over
    # peer-mapped memory, cn_comm_p2p.hip) in place of RCCL; one rank per GPU when the node has enough of them (a measurement),
    # ranks sharing devices otherwise (a test).  gloo carries the control plane.
    backend = os.environ.get("CN_BENCH_BACKEND", "nccl")
    if backend in ("ipc", "p2p"):
        os.environ["CN_COMM_BACKEND"] = backend
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d needs %d GPUs, this node has %d (one rank per GPU over RCCL; no device sharing)"
                         % (world, world, torch.cuda.device_count()))
    shared_devices = backend != "nccl" and (backend != "p2p" or world > torch.cuda.device_count())
    device_index = local_rank % torch.cuda.device_count() if shared_devices else local_rank
    torch.cuda.set_device(device_index)
    rendezvous_s = float(os.environ.get("CN_BENCH_RENDEZVOUS_TIMEOUT", "180"))
    if world > 1 or os.environ.get("CN_BENCH_FORCE_ALLREDUCE") == "1":
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(seconds=rendezvous_s))
        else:
            dist.init_process_group("gloo" if backend in ("ipc", "p2p") else backend, timeout=datetime.timedelta(seconds=rendezvous_s))

    def bounded(what, seconds, fn, *a):
        """Run a blocking collective set-up call under a watchdog: a rank whose call does not return in time says which rank and
        which call on stderr and ends the process with code 3 (exiting is always allowed; the launcher then ends the others)."""
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(seconds):
                sys.stderr.write("bench.py: rank %d of %d: %s did not complete within %.0f s (a rank missing from the rendezvous?)\n" % (rank, world, what, seconds))
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            return fn(*a)
        finally:
            done.set()

    pkg = ge.load_package()
    dev = torch.device("cuda", device_index)
    PRECISIONS = {"bf16": pkg.PREC_BF16, "f32": pkg.PREC_F32}
    if hasattr(pkg, "PREC_BF16X3"):
        PRECISIONS["bf16x3"] = pkg.PREC_BF16X3
    if args.precision not in PRECISIONS:
        raise SystemExit("bench.py: precision %s is not built into this library" % args.precision)
    use_comm = world > 1 or os.environ.get("CN_BENCH_FORCE_ALLREDUCE") == "1"
    flat_exchange = os.environ.get("CN_BENCH_FLAT_ALLREDUCE") == "1"
    prefetch = os.environ.get("CN_BENCH_NO_PREFETCH") != "1"   # A/B switch: the next fraction's re-layout on the main stream
    armed = os.environ.get("CN_BENCH_NO_ARM") != "1"       # A/B switch: the update of all layers behind the last backward kernel
    # The gradient exchange is the library's own RCCL communicator (cn_comm_init / cn_allreduce_grads); torch.distributed
    # carries the control plane only (rendezvous id, barrier, max-over-ranks timing).  CN_BENCH_BACKEND=gloo swaps in the
    # torch path (compute_backward_pass_allreduce) as the test double for boxes with fewer GPUs than ranks.
    native_comm = use_comm and backend in ("nccl", "ipc", "p2p")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def allmax(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def allsum(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t]

    def run_workload(name, steps, warmup, precision, roofline_pass=False, host_pass=False, min_seconds=0.5, deterministic=None):
        """-> (result dict, workload): `steps` timed steps per repetition (median of enough repetitions for min_seconds)"""
        wl = WORKLOADS[name]
        PS = args.parallel_sequences if name == args.workload else wl.get("PS", 50)
        tmin = args.tmin if name == args.workload else wl.get("tmin", 250)
        tmax = args.tmax if name == args.workload else max(wl.get("tmax", 350), tmin)
        P, C, hidden = wl["P"], wl["C"], wl["hidden"]
        layers = net_desc(P, hidden, C)
        weights = make_weights(layers, 1234)                       # identical replicas on every rank
        rng = np.random.RandomState(1234 + rank)                   # seed = 1234 + rank
        nfrac = 4
        fracs = [synth_fraction(pkg, rng, PS, P, C, tmin, tmax) for _ in range(nfrac)]
        # (the library runs on a stream of its own; net.torch_stream() is that stream for torch ordering)
        net = pkg.NeuralNetwork(layers, weights, PS, tmax, precision=PRECISIONS[precision], device=device_index, deterministic=deterministic)
        det_on = bool(net.get_option("deterministic"))
        if native_comm:
            uid = [net.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            bounded("cn_comm_init (ncclCommInitRank)", rendezvous_s, net.comm_init, uid[0], rank, world)
            comm_world = net.comm_info()[1]                       # what RCCL itself reports (ncclCommCount), not WORLD_SIZE
        # fractions resident in HBM (torch owns the device memory)
        dfr, keep = [], []
        for f in fracs:
            x = torch.from_numpy(f["inputs"]).to(dev)
            pt = torch.from_numpy(f["patTypes"]).to(dev)
            tc = torch.from_numpy(f["targetClasses"]).to(dev)
            keep += [x, pt, tc]
            dfr.append({"T": f["T"], "Tmin": f["Tmin"], "numSeqs": f["numSeqs"], "inputPatternSize": P,
                        "outputPatternSize": C, "inputs": x.data_ptr(), "patTypes": pt.data_ptr(),
                        "targetClasses": tc.data_ptr(), "frames": pkg.fraction.real_frames(f)})
        wptr, gptr, dptr, count = net.param_arena()
        wts = torch.as_tensor(pkg.parallel.DeviceArray(wptr, count), device=dev)
        net.synchronize()
        w0 = wts.clone()                                           # initial weights, for check.update_l2 / update_sum
        grads = torch.as_tensor(pkg.parallel.DeviceArray(gptr, count), device=dev) if (use_comm and not native_comm) else None

        def step(i, from_host=False):
            f = dfr[i % nfrac]
            if from_host:
                net.load_sequences(fracs[i % nfrac])        # host buffers through cn_fraction_load (pinned staging + PCIe)
            else:
                net.load_sequences_resident(f)
            net.compute_forward_pass()
            net.loss_accumulate()
            if prefetch and not from_host:
                net.prefetch_sequences_resident(dfr[(i + 1) % nfrac])   # the next step's re-layout, beside this backward pass
            elif prefetch:
                net.prefetch_sequences(fracs[(i + 1) % nfrac])          # host buffers: packed and uploaded now, re-laid out beside this backward pass
            # every layer's momentum-SGD step behind its own gradient (cn_ctx_arm_update; with the library's communicator: behind
            # its all-reduce); the torch test double reduces outside the library, so it keeps the update behind the whole pass
            if armed and (not use_comm or (native_comm and not flat_exchange)):
                net.arm_update(args.lr, args.momentum)
            if not use_comm:
                net.compute_backward_pass()
            elif native_comm and not flat_exchange:
                net.compute_backward_pass_dp()              # per-layer RCCL all-reduce beside the backward pass of the layers below
            elif native_comm:
                net.compute_backward_pass(); net.allreduce_grads(None)
            elif not flat_exchange:
                net.compute_backward_pass_allreduce(dist, torch)
            else:
                net.compute_backward_pass()
                net.join()                                  # gradient GEMMs run on the library's side stream
                with torch.cuda.stream(net.torch_stream(torch)):
                    dist.all_reduce(grads, op=dist.ReduceOp.SUM)
            net.update_weights_fused(args.lr, args.momentum)
            return f["frames"]

        def timed(from_host=False):
            """EXACTLY `steps` steps between barrier + synchronize on both sides; (seconds, frames) of this rank."""
            barrier()
            t0 = time.perf_counter()
            fr = 0
            for i in range(steps):
                fr += step(warmup + i, from_host)
            barrier()
            return time.perf_counter() - t0, fr

        for i in range(warmup):
            step(i)
        dt0, frames = timed()
        # the headline region is short (20 steps x 1.4 ms): repeat the same K steps until >= min_seconds have been timed
        # and report the MEDIAN repetition (each one bracketed like the first); the count is agreed across ranks
        reps = int(min(60, max(1, np.ceil(min_seconds / max(allmax([dt0])[0], 1e-6))))) if min_seconds > 0 else 1
        dts = [dt0] + [timed()[0] for _ in range(reps - 1)]
        dts = allmax(dts)                                          # per repetition: the slowest rank
        res = {"frames": allsum([float(frames)])[0], "seconds": float(np.median(dts)), "repeats": reps,
               "seconds_min": float(min(dts)), "seconds_max": float(max(dts)), "timed_total_s": float(sum(dts)),
               "weights": int(count), "PS": PS, "tmin": tmin, "tmax": tmax,
               "kernels": (net.recurrent_kernel(False), net.recurrent_kernel(True)), "deterministic": det_on}
        # epoch sums: over all ranks through the library's communicator when it is bound (cn_loss_read_global), so that the
        # record compares with a single-process run over the union of the ranks' fractions
        err_sum, correct = net.loss_read_global() if native_comm else net.loss_read()
        res["error_sum"] = err_sum
        if native_comm:
            lo_hi = allmax([float(comm_world), -float(comm_world)])
            res["rccl_ranks"] = (int(-lo_hi[1]), int(lo_hi[0]))      # (min, max) over ranks
            res["comm_backend"] = net.comm_backend()                 # (name, exchanges enqueued) as the LIBRARY reports them
        # what warm-up + the FIRST repetition's steps did to the weights is not separable from later repetitions; the
        # data-parallel equivalence test runs with min_seconds = 0 (CN_BENCH_MIN_SECONDS=0), i.e. one repetition
        upd = (wts.double() - w0.double())
        res["update_l2"], res["update_sum"] = float(upd.norm()), float(upd.sum())
        if world > 1:
            # data-parallel replicas must stay bit-identical: same reduced gradients, same update on every rank
            with torch.cuda.stream(net.torch_stream(torch)):
                sig = torch.stack([wts.double().sum(), wts.double().abs().sum()])
                lo, hi = sig.clone(), sig.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize(dev)
            res["replicas_identical"] = bool(torch.equal(lo, hi))
        if host_pass:
            # the same steps with the fractions handed over as host buffers (cn_fraction_load: packed into pinned memory,
            # one upload on a copy stream under the previous fraction's compute, then the re-layout kernel)
            for i in range(warmup):
                step(i, from_host=True)
            hits0 = net.prefetch_hits()
            hd = [timed(from_host=True) for _ in range(max(1, min(reps, 20)))]
            res["host_frames_per_s"] = frames / float(np.median(allmax([h[0] for h in hd])))
            res["host_prefetch_hits"] = (net.prefetch_hits() - hits0, len(hd) * steps)
        if roofline_pass:
            # one more pass of the same steps with hipEvents around every kernel class, on the stream they are launched on
            net.timing_enable(True); net.timing_reset()
            fr2 = 0
            for i in range(steps):
                fr2 += step(warmup + i)
            net.synchronize()
            res["timing"] = net.timing_read(); res["timing_frames"] = fr2; res["timing_steps"] = steps
            net.timing_enable(False)
        net.close()
        del keep
        return res, wl

    min_seconds = float(os.environ.get("CN_BENCH_MIN_SECONDS", "0.5"))
    res, wl = run_workload(args.workload, args.steps, args.warmup, args.precision, roofline_pass=not args.no_roofline_pass,
                           host_pass=(world == 1), min_seconds=min_seconds)
    seconds, frames = res["seconds"], res["frames"]
    value = frames / seconds

    # informational extra lines: the Graves-literal reading of "3x250" and the parity (fp32-tolerance) arithmetic mode
    also_spec = [a for a in args.also.split(",") if a]
    if not args.also and world == 1 and args.workload == "timit_3x250_blstm_H125" and not args.no_also:
        also_spec = (["timit_3x500_blstm_H250"] + (["timit_3x250_blstm_H125:bf16x3", "timit_3x500_blstm_H250:bf16x3"] if "bf16x3" in PRECISIONS else [])
                     + ["timit_3x250_blstm_H125:f32"])
        # BASELINE.json configs[3] and configs[4] as written (one GPU's share of the 8-GPU configs: PS per GPU as in WORKLOADS)
        also_spec += ["lvcsr_4x512_blstm_8000", "longutt_5x1024_blstm"]
        # ... and at fp32 tolerance (every config carries an at-tolerance figure beside its bf16 one)
        if "bf16x3" in PRECISIONS:
            also_spec += ["lvcsr_4x512_blstm_8000:bf16x3", "longutt_5x1024_blstm:bf16x3"]
        # the headline with the gradient sums in a fixed order (option "deterministic", off by default in bf16): what reproducible costs
        also_spec += ["timit_3x250_blstm_H125:bf16:det"]
    also = {}
    for spec in also_spec:
        name, _, pr = spec.partition(":")
        pr, _, opt = pr.partition(":")
        pr = pr or args.precision
        big = name in ("lvcsr_4x512_blstm_8000", "longutt_5x1024_blstm")
        st2, wu2 = (min(args.steps, 5), min(args.warmup, 2)) if big else (args.steps, args.warmup)
        roof2 = (big and pr == args.precision) or spec == "timit_3x500_blstm_H250"      # (reading B: the other reading of the headline config gets its roofline too)
        r2, wl2 = run_workload(name, st2, wu2, pr, roofline_pass=roof2, min_seconds=0 if big else min(min_seconds, 0.25),
                               deterministic=True if opt == "det" else None)
        v2 = r2["frames"] / r2["seconds"]
        also[spec] = {"value": v2, "unit": "frames/s", "dtype": pr, "ms_per_step": 1e3 * r2["seconds"] / st2, "steps": st2, "repeats": r2["repeats"],
                      "parallel_sequences": r2["PS"], "seq_len": "U[%d,%d]" % (r2["tmin"], r2["tmax"]), "deterministic": r2["deterministic"]}
        if roof2 and "timing" in r2:
            rr = roofline_records(r2, wl2, name, r2["PS"], pr, v2)
            also[spec].update({"roofline": rr["roofline"], "roofline_other": rr["roofline_other"], "roofline_pair": rr["roofline_pair"],
                               "roofline_mfma": rr["roofline_mfma"], "roofline_gemm": rr["roofline_gemm"]})

    if rank == 0:
        exch = "none"
        if use_comm:
            how = {"ipc": " (library communicator on its ipc TEST backend: not a measurement)",
                   "p2p": " (library communicator, p2p backend: one kernel per bucket over peer-mapped memory" +
                          ("; ranks SHARE devices: not a measurement)" if shared_devices else ")"),
                   "nccl": " (library RCCL communicator)"}
            exch = ("flat" if flat_exchange else "per-layer, overlapped") + (how[backend] if native_comm else " (torch.distributed test double)")
        out = {
            "metric": "train frames/sec (node), 3x250 BLSTM 39->183" if args.workload.startswith("timit_3x") else "train frames/sec (node), " + args.workload,
            "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * seconds / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": args.workload, "topology": "%d -> " % wl["P"] + " -> ".join("%s%d" % h for h in wl["hidden"]) + " -> softmax%d" % wl["C"],
                       "parallel_sequences_per_gpu": args.parallel_sequences, "seq_len": "U[%d,%d]" % (args.tmin, args.tmax),
                       "weights": res["weights"], "update": "stochastic momentum SGD every fraction",
                       "parallelism": "dp%d over sequences" % world, "inputs": "fractions resident in HBM (cn_fraction_load_resident)"},
            "timing": {"repeats": res["repeats"], "reported": "median repetition of `steps` steps, each bracketed by barrier + synchronize, max over ranks",
                       "ms_per_step_min": 1e3 * res["seconds_min"] / args.steps, "ms_per_step_max": 1e3 * res["seconds_max"] / args.steps,
                       "timed_total_s": res["timed_total_s"]},
        }
        out["config"]["deterministic_sums"] = res["deterministic"]
        if "timit_3x250_blstm_H125:bf16:det" in also and args.precision == "bf16":
            dv = also["timit_3x250_blstm_H125:bf16:det"]["value"]
            out["deterministic"] = {"headline_default": res["deterministic"], "value_with_fixed_order_sums": dv, "cost": 1.0 - dv / value,
                                    "note": "option \"deterministic\" (cn_ctx_set_option): gradient sums in a fixed order, bit-identical runs; default on in "
                                            "f32 / bf16x3, opt-in for bf16 (this line: the same bf16 steps with it on)"}
        out["check"] = {"error_sum": res["error_sum"], "update_l2": res["update_l2"], "update_sum": res["update_sum"],
                        **({"replicas_identical": res["replicas_identical"]} if "replicas_identical" in res else {}), "allreduce": exch}
        if use_comm:
            # backend: what the bound communicator says it runs (cn_comm_backend), not what the environment asked for
            name, count = res.get("comm_backend", ("torch.distributed " + backend + " (test double)", None))
            out["exchange"] = {"backend": name, "allreduces_enqueued": count, "granularity": "flat arena" if flat_exchange else "per layer",
                               "one_rank_per_gpu": not shared_devices}
        if "host_frames_per_s" in res:
            out["load_path"] = {"value": res["host_frames_per_s"], "unit": "frames/s", "vs_resident": res["host_frames_per_s"] / value,
                                "prefetched_loads": "%d of %d" % tuple(res.get("host_prefetch_hits", (0, 0))),
                                "note": "same steps with every fraction handed over as HOST buffers: cn_fraction_prefetch of fraction k+1 behind the forward pass of "
                                        "fraction k (packed into pinned staging and uploaded at once, re-laid out beside the backward pass), cn_fraction_load "
                                        "then exchanges buffers; PCIe-inclusive, informational, never `value`"}
        if "timing" in res:
            out.update(roofline_records(res, wl, args.workload, args.parallel_sequences, args.precision, value))
            if native_comm:
                ex = res["timing"].get("exchange", (0.0, 0))
                out["exchange"].update({"ranks_min": res["rccl_ranks"][0], "ranks_max": res["rccl_ranks"][1],
                                   "allreduce_ms_per_step": ex[0] / max(1, res["timing_steps"]), "allreduces_per_step": ex[1] / max(1, res["timing_steps"]),
                                   "note": "ranks as the library's communicator reports them (cn_comm_info); device time of the per-layer all-reduces from "
                                           "hipEvents on the library's communication stream during the event-timed pass"})
        if also:
            out["also"] = also
            # the other readings of BASELINE configs[1] and the at-tolerance arithmetic modes, inside `config` so that a consumer
            # that keeps only the contract's keys still sees them: B = Graves-literal 250 units per direction; bf16x3 / f32 = the
            # modes that hold the north-star 1e-4 posterior tolerance single-pass (bf16 holds 3e-2 against the fp32 reference and
            # 2e-4 against the bf16-operand oracle, tests/test_gpu_bf16_pinned.py)
            names = {"timit_3x500_blstm_H250": "B_H250_bf16", "timit_3x250_blstm_H125:bf16x3": "A_H125_bf16x3",
                     "timit_3x500_blstm_H250:bf16x3": "B_H250_bf16x3", "timit_3x250_blstm_H125:f32": "A_H125_f32"}
            other = {names[k]: {"value": v["value"], "unit": "frames/s", "ms_per_step": v["ms_per_step"], "dtype": v["dtype"]}
                     for k, v in also.items() if k in names}
            if other:
                out["config"]["other_readings"] = other
                # ... and once more as flat scalars (round 4's driver record kept only the scalar entries of `config`)
                for k, v in other.items():
                    out["config"]["frames_per_s_" + k] = round(v["value"])
        if world == 1 and not args.no_driver_leg:
            out["driver_leg"] = driver_leg(wl, args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, wl, args, PRECISIONS)
        print(json.dumps(out))
    if use_comm and dist.is_initialized():
        dist.destroy_process_group()


def time_oracle_step(pkg, orc, wl, args, PS, tlo, thi, backend="oracle"):
    """One training step (load, forward, error, backward, update) of the oracle on one synthetic fraction; 1 thread.
    backend "ref": the same call sequence through oracle/_ref, the reference's own compiled functors and Cpu GEMM."""
    layers = net_desc(wl["P"], wl["hidden"], wl["C"])
    weights = make_weights(layers, 1234)
    rng = np.random.RandomState(99)
    frac = synth_fraction(pkg, rng, PS, wl["P"], wl["C"], tlo, thi)
    net = orc.OracleNetwork(layers, weights, PS, frac["T"], backend=backend)
    t0 = time.perf_counter()
    net.load_sequences(frac); net.compute_forward_pass(); net.calculate_error(); net.count_correct_classifications()
    net.compute_backward_pass(); net.update_weights(args.lr, args.momentum)
    dt = time.perf_counter() - t0
    frames = pkg.fraction.real_frames(frac)
    return frames / dt, "1 fraction, %d sequences U[%d,%d] (%d frames), same topology, fp32, %.1f s" % (PS, tlo, thi, frames, dt)


def driver_leg(wl, args):
    """The same workload end to end through the C++ driver (lstm-rnn_amd/currennt_hip, host/main.cpp): a synthetic
    NetCDF-3 file on disk -> reader -> length sort -> fraction packer (worker thread) -> cn_fraction_load (host buffers)
    -> forward / backward / momentum SGD every fraction, error sums read once per epoch.  Fractions are cut from the
    length-SORTED sequence list as the reference does (DataSet.cpp:603-605), so they pad less than the bench's
    independently drawn ones; informational, never `value`."""
    import re
    import subprocess
    import tempfile
    from scipy.io import netcdf_file
    binary = os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip")
    if not os.path.exists(binary):
        return {"error": "driver not built"}
    rng = np.random.RandomState(4321)
    nseq, P, C = 3000, wl["P"], wl["C"]
    lens = rng.randint(args.tmin, args.tmax + 1, nseq)
    n = int(lens.sum())
    with tempfile.TemporaryDirectory() as d:
        nc = os.path.join(d, "train.nc")
        f = netcdf_file(nc, "w")
        f.createDimension("numSeqs", nseq); f.createDimension("numTimesteps", n); f.createDimension("inputPattSize", P)
        f.createDimension("numLabels", C); f.createDimension("maxSeqTagLength", 16)
        tags = f.createVariable("seqTags", "c", ("numSeqs", "maxSeqTagLength"))
        tags[:] = np.array([list(("s%05d" % i).ljust(16, "\0")) for i in range(nseq)], "c")
        f.createVariable("seqLengths", "i", ("numSeqs",))[:] = lens.astype(np.int32)
        f.createVariable("targetClasses", "i", ("numTimesteps",))[:] = rng.randint(0, C, n).astype(np.int32)
        f.createVariable("inputs", "f", ("numTimesteps", "inputPattSize"))[:] = rng.randn(n, P).astype(np.float32)
        f.close()
        net = os.path.join(d, "network.jsn")
        json.dump({"layers": net_desc(P, wl["hidden"], C)}, open(net, "w"))
        epochs = 4
        cmd = [binary, "--train", "true", "--stochastic", "true", "--train_file", nc, "--network", net,
               "--parallel_sequences", str(args.parallel_sequences), "--max_epochs", str(epochs), "--learning_rate", str(args.lr),
               "--momentum", str(args.momentum), "--precision", args.precision, "--save_network", os.path.join(d, "trained.jsn"), "--random_seed", "1"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, CN_DRIVER_TIMING="1"))
    if out.returncode != 0:
        return {"error": out.stdout[-300:]}
    secs = [float(m.group(1)) for m in re.finditer(r"TIMING epoch \d+ ([\d.]+) s", out.stderr)]
    if len(secs) < 2:
        return {"error": "no epoch timings"}
    med = float(np.median(secs[1:]))                       # the first epoch carries allocation and first-touch costs
    return {"value": n / med, "unit": "frames/s", "epoch_seconds": secs, "frames_per_epoch": n, "sequences": nseq,
            "note": "C++ driver end to end from a NetCDF file (length-sorted fractions, cn_fraction_load from host buffers, error read once per epoch); informational"}


def cpu_baseline(pkg, wl, args, precisions):
    """The oracle (scalar fp32 restatement of the reference's Cpu path, 1 thread like its Thrust-host build)
    timed on a bounded sample of the same workload, plus BASELINE.json configs[0] (the config the reference itself runs on
    a CPU: 39 -> lstm128 -> softmax183), plus a parity record of the HIP path against it AFTER 40 weight updates."""
    orc = ge.load_oracle()
    orc.set_threads(1)
    # keep the sample near 10 s of CPU work (the oracle runs at ~3 GFLOP/s): 16 sequences of the workload's lengths
    # for reading A (~5k frames), fewer and shorter sequences for the larger topologies
    PS, tlo, thi = 16, args.tmin, args.tmax
    budget = 30e9 / flops_per_frame(wl["P"], wl["hidden"], wl["C"])          # frames
    if PS * (tlo + thi) / 2 > budget:
        PS = 8 if budget >= 8 * 20 else 4
        thi = max(4, int(budget / PS * 1.15)); tlo = max(2, int(thi * 0.75))
    v, sample = time_oracle_step(pkg, orc, wl, args, PS, tlo, thi)
    out = {"value": v, "unit": "frames/s", "cores": 1, "kind": "port", "sample": sample}
    if orc.ref_available():
        # oracle/_ref travelled to this box: the reference's OWN object code (its functors and helpers::Matrix<Cpu>, compiled from
        # /root/reference in the build container) on the same sample is the baseline; the restatement's figure stays beside it
        vr, sample_r = time_oracle_step(pkg, orc, wl, args, PS, tlo, thi, backend="ref")
        out = {"value": vr, "unit": "frames/s", "cores": 1, "kind": "reference", "sample": sample_r + " (oracle/_ref: reference functors + Cpu GEMM)",
               "port": {"value": v, "unit": "frames/s", "cores": 1, "kind": "port", "sample": sample}}
    w0 = WORKLOADS["timit_1x128_lstm"]
    v0, sample0 = time_oracle_step(pkg, orc, w0, args, 16, 250, 350)
    out["configs0_timit_1x128_lstm"] = {"value": v0, "unit": "frames/s", "cores": 1, "kind": "port", "sample": sample0}

    # Parity of the HIP path against the oracle on the same topology and weights after FORTY momentum-SGD updates on a
    # learnable task (the class is a fixed random projection of the current and previous frame), i.e. with posteriors
    # that have moved away from 1/C -- at initial weights every posterior is ~1/183 and any arithmetic passes.
    # fp32 parity mode carries the north-star tolerance (posterior max-abs < 1e-4); the measured precision is
    # reported beside it.
    layers = net_desc(wl["P"], wl["hidden"], wl["C"])
    weights = make_weights(layers, 1234)
    rng = np.random.RandomState(77)
    P, C = wl["P"], wl["C"]
    fpf = flops_per_frame(P, wl["hidden"], C)
    nseq = 6
    tlen = int(max(6, min(60, 90e9 / 41 / fpf / nseq)))                        # <= ~90 GFLOP of oracle work for the 41 passes
    orc.set_threads(min(8, len(os.sched_getaffinity(0))))                      # checker, not the timed baseline: bit-identical for any thread count
    proj = rng.randn(2 * P, C).astype(np.float32)
    fracs, fracs_rev = [], []
    for _ in range(2):
        xs = [rng.randn(tlen - (i % 3), P).astype(np.float32) for i in range(nseq)]
        ts = [np.argmax(np.hstack([x, np.vstack([np.zeros((1, P), np.float32), x[:-1]])]) @ proj, axis=1).astype(np.int32) for x in xs]
        fracs.append(pkg.make_fraction(xs, ts, nseq))
        fracs_rev.append(pkg.make_fraction(xs[::-1], ts[::-1], nseq))          # the same sequences, slots in reverse order
    lr, mom, nupd = 1e-2, 0.9, 40

    def train(net, fr=None):
        fr = fr or fracs
        errs = []
        for k in range(nupd):
            net.load_sequences(fr[k % 2]); net.compute_forward_pass(); errs.append(net.calculate_error())
            net.compute_backward_pass(); net.update_weights(lr, mom)
        net.load_sequences(fr[0]); net.compute_forward_pass()
        return errs

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    ref = orc.OracleNetwork(layers, weights, nseq, tlen)
    eref = train(ref)
    yr = ref.outputs()
    real = np.asarray(fracs[0]["patTypes"]).reshape(-1) != 0
    parity = {"task": "%d updates (lr %g, momentum %g) on 2 fractions of %d sequences x %d frames, learnable targets; oracle error %.1f -> %.1f, "
                      "largest posterior %.3f" % (nupd, lr, mom, nseq, tlen, eref[0], eref[-1], float(yr.reshape(-1, C)[real].max()))}
    # What "after 40 updates" can mean in fp32 at all: the REFERENCE arithmetic itself (the oracle, bit-equal to oracle/_ref) trained
    # on the same sequences with the slots of each fraction in reverse order -- the same gradient mathematically, its sum over the
    # patterns taken in another order (LstmLayer.cu:502-510 walks the patterns in slot order) -- against its own first run.  A HIP
    # figure at or below this one is at the noise floor of fp32 summation order under this many updates, not an arithmetic defect;
    # the north-star 1e-4 is a single-pass bound (tests/test_gpu_parity.py) and holds through 20 updates (tests/test_gpu_configs.py).
    ref2 = orc.OracleNetwork(layers, weights, nseq, tlen)
    train(ref2, fracs_rev)
    T0 = int(fracs[0]["T"])
    y_a = yr.reshape(T0, nseq, C); y_b = ref2.outputs().reshape(T0, nseq, C)[:, ::-1, :]
    real3 = real.reshape(T0, nseq)
    parity["f32_reference_slot_order_noise"] = {
        "posterior_max_abs": float(np.abs(y_a - y_b)[real3].max()),
        "weights_max_abs": max(float(np.abs(ref2.layer(l["name"]).weights - ref.layer(l["name"]).weights).max()) for l in layers if l["type"] in ("lstm", "blstm", "softmax")),
        "note": "oracle vs oracle: the same sequences with the slots of every fraction reversed (another fp32 summation order of the same gradient), same 40 updates"}
    for name in dict.fromkeys(["f32", "bf16x3", args.precision]):
        if name not in precisions:
            continue
        with pkg.NeuralNetwork(layers, weights, nseq, tlen, precision=precisions[name]) as hip:
            e = train(hip)
            y = hip.outputs()
            post = float(np.abs(y.reshape(-1, C)[real] - yr.reshape(-1, C)[real]).max())
            wrel = max(rel(l.weights(), ref.layer(l.name).weights) for l in hip.trainable_layers())
            wmax = max(float(np.abs(l.weights() - ref.layer(l.name).weights).max()) for l in hip.trainable_layers())
            parity[name] = {"posterior_max_abs": post, "weights_rel_l2": wrel, "weights_max_abs": wmax,
                            "error_first": float(e[0]), "error_last": float(e[-1]), "error_last_oracle": float(eref[-1])}
    if "bf16" in precisions and hasattr(orc, "operand_rounding"):
        # The benchmarked arithmetic against the oracle's bf16-operand model (oracle.set_operand_rounding: both operands of every
        # matrix product rounded to bf16, fp32 accumulation / state / libm activations): (i) ONE forward + backward pass at the
        # weights the model reached after the 40 updates, i.e. with peaked posteriors; (ii) the same 40 updates in both.
        with orc.operand_rounding("bf16"):
            refq = orc.OracleNetwork(layers, weights, nseq, tlen)
            with pkg.NeuralNetwork(layers, weights, nseq, tlen, precision=precisions["bf16"]) as hip:
                hip.load_sequences(fracs[0]); hip.compute_forward_pass()
                for name in hip.bf16_preactivation_layers():       # (the model rounds the pre-activations the HIP path keeps in bf16)
                    refq.layer(name).round_preacts = True
            eq = train(refq)
            yq = refq.outputs().copy()
            refq.calculate_error(); refq.compute_backward_pass()
            with pkg.NeuralNetwork(layers, weights, nseq, tlen, precision=precisions["bf16"]) as hip:
                for l in hip.trainable_layers():
                    l.set_weights(refq.layer(l.name).weights)
                hip.load_sequences(fracs[0]); hip.compute_forward_pass(); hip.calculate_error(); hip.compute_backward_pass()
                y1 = hip.outputs()
                g1 = max(float(np.abs(l.weight_updates() - refq.layer(l.name).weightUpdates).max() / max(1e-30, np.abs(refq.layer(l.name).weightUpdates).max()))
                         for l in hip.trainable_layers())
            with pkg.NeuralNetwork(layers, weights, nseq, tlen, precision=precisions["bf16"]) as hip:
                e = train(hip)
                y40 = hip.outputs()
                w40 = max(rel(l.weights(), refq.layer(l.name).weights) for l in hip.trainable_layers())
        parity["bf16_vs_bf16_operand_oracle"] = {
            "single_pass_at_trained_weights": {"posterior_max_abs": float(np.abs(y1.reshape(-1, C)[real] - yq.reshape(-1, C)[real]).max()),
                                               "gradient_max_rel_to_layer_max": g1, "largest_posterior": float(yq.reshape(-1, C)[real].max())},
            "after_40_updates": {"posterior_max_abs": float(np.abs(y40.reshape(-1, C)[real] - yq.reshape(-1, C)[real]).max()), "weights_rel_l2": w40,
                                 "error_last": float(e[-1]), "error_last_oracle": float(eq[-1])},
            "note": "oracle with matrix-product operands rounded to bf16 (test infrastructure): what remains is summation order and v_exp_f32 / v_rcp_f32"}
    out["parity_vs_cpu"] = parity
    orc.set_threads(1)
    return out


if __name__ == "__main__":
    main()
