"""CPU-side checks: the C-ABI library loads and exports every symbol include/*.h declares, the
fraction packer follows DataSet::_makeFractionTask, and the oracle's primitive products follow
helpers/Matrix.cu.  No compute call is made without a GPU."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(pkg, hiplib):
    from lstm_rnn_amd import binding
    decl = declared_symbols()
    assert decl, "no declarations found in include/*.h"
    missing = [n for n in sorted(decl) if not hasattr(hiplib, n)]
    assert not missing, missing
    assert set(binding.EXPORTS) == decl, sorted(set(binding.EXPORTS) ^ decl)
    assert hiplib.cn_version().decode().startswith("currennt_hip")


def test_no_cpu_fallback(pkg, hiplib):
    """Without a GPU the product path must fail loudly (CN_ERR_NO_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = hiplib.cn_ctx_create(0, 0, None, C.byref(ctx))
    assert rc == -5 and not ctx.value
    assert b"no HIP device" in hiplib.cn_last_error(None)
    layers = [{"name": "i", "type": "input", "size": 2}, {"name": "o", "type": "softmax", "size": 2, "bias": 1.0},
              {"name": "p", "type": "multiclass_classification", "size": 2}]
    with pytest.raises(pkg.CurrenntHipError):
        pkg.NeuralNetwork(layers, None, 1, 2, seed=0)


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (test infrastructure)."""
    for path in glob.glob(os.path.join(ROOT, "lstm-rnn_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
            assert "oracle" not in open(path, errors="ignore").read().lower().replace("nothing here touches the cpu oracle", ""), path


def test_fraction_packer_layout(pkg):
    xs = [np.arange(3 * 2, dtype=np.float32).reshape(3, 2) + 10, np.arange(1 * 2, dtype=np.float32).reshape(1, 2) + 50]
    ts = [np.array([1, 2, 0], np.int32), np.array([2], np.int32)]
    f = pkg.make_fraction(xs, ts, 3)
    assert (f["T"], f["Tmin"], f["PS"], f["numSeqs"]) == (3, 1, 3, 2)
    x = f["inputs"].reshape(3, 3, 2)
    assert np.all(x[:, 0] == xs[0]) and np.all(x[0, 1] == xs[1][0]) and np.all(x[1:, 1] == 0) and np.all(x[:, 2] == 0)
    pat = f["patTypes"].reshape(3, 3)
    assert pat[:, 0].tolist() == [1, 2, 3]          # FIRST, NORMAL, LAST (DataSet.cpp:400-407)
    assert pat[:, 1].tolist() == [1, 0, 0]          # a length-1 sequence is FIRST (timestep == 0 wins)
    assert pat[:, 2].tolist() == [0, 0, 0]          # missing column stays NONE (:331)
    tc = f["targetClasses"].reshape(3, 3)
    assert tc[:, 0].tolist() == [1, 2, 0] and tc[:, 1].tolist() == [2, -1, -1] and tc[:, 2].tolist() == [-1, -1, -1]
    fr = pkg.make_fractions(xs * 3, ts * 3, 4, sort_by_length=True)
    assert [q["numSeqs"] for q in fr] == [4, 2] and fr[0]["seqLengths"] == [1, 1, 1, 3]


def test_oracle_matmul_kinds(orc):
    """orc_matmul restates the three column-major products of helpers/Matrix.cu:41-183."""
    rng = np.random.RandomState(0)
    L = orc.lib()
    A = rng.randn(5, 7).astype(np.float32); B = rng.randn(7, 3).astype(np.float32)
    c = np.zeros(5 * 3, np.float32)
    L.orc_matmul(0, c, np.ascontiguousarray(A.T).reshape(-1), 5, 7, np.ascontiguousarray(B.T).reshape(-1), 7, 3, 0)
    assert np.allclose(c.reshape(3, 5).T, A @ B, atol=1e-5)
    A2 = rng.randn(7, 5).astype(np.float32)     # C = A2^T B
    c = np.ones(5 * 3, np.float32)
    L.orc_matmul(1, c, np.ascontiguousarray(A2.T).reshape(-1), 7, 5, np.ascontiguousarray(B.T).reshape(-1), 7, 3, 1)
    assert np.allclose(c.reshape(3, 5).T, 1 + A2.T @ B, atol=1e-5)
    B2 = rng.randn(3, 7).astype(np.float32)     # C = A B2^T
    c = np.zeros(5 * 3, np.float32)
    L.orc_matmul(2, c, np.ascontiguousarray(A.T).reshape(-1), 5, 7, np.ascontiguousarray(B2.T).reshape(-1), 3, 7, 0)
    assert np.allclose(c.reshape(3, 5).T, A @ B2.T, atol=1e-5)


def test_oracle_quirks(orc):
    """Q3 (softmax offset with max initialised to FLT_MIN), Q1 (dummy slots), Q10 (momentum SGD)."""
    L = orc.lib()
    P, C_, N = 2, 3, 2
    w = np.array([0, 0, 0, 0, 0, 0, -5.0, -6.0, -7.0], np.float32)   # zero input weights, negative biases
    x = np.zeros((N, P), np.float32); y = np.zeros(N * C_, np.float32); tmp = np.zeros(N, np.float32)
    pat = np.array([2, 0], np.int8)
    L.orc_softmax_forward(P, C_, 1.0, N, pat, w, x.reshape(-1), y, tmp)
    e = np.exp(np.array([-5.0, -6.0, -7.0]) - 0.5 * (-7.0 + 0.0))     # max stays ~0: offset = -3.5
    assert np.allclose(y[:3], e / e.sum(), rtol=1e-6)
    assert np.allclose(y[3:], [-5, -6, -7])                           # dummy slot keeps the pre-activation
    wts = np.array([1.0, 2.0], np.float32); g = np.array([0.5, -1.0], np.float32); dl = np.array([0.1, 0.0], np.float32)
    L.orc_sgd_update(2, 0.1, 0.9, wts, g, dl)
    assert np.allclose(dl, [0.9 * 0.1 - 0.05, 0.1]) and np.allclose(wts, [1.04, 2.1])


def test_generated_step_body_of_the_wide_forward_loop_is_in_sync_with_its_generator():
    """lstm-rnn_amd/csrc/cn_lstm_s2w_loop.inc is the output of tools/gen_s2w_loop.py (schedule table -> asm string with derived
    lgkmcnt counts): an edit of either without the other would ship a loop nobody generated."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_s2w_loop.py")], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(root, "lstm-rnn_amd", "csrc", "cn_lstm_s2w_loop.inc")) as f:
        assert f.read() == out
    body = out.split("#else")[1]
    assert body.count("v_smfmac_f32_16x16x64_bf16") == 2 * 64        # two step bodies of 64 MFMAs
    assert body.count("ds_read_b128") == 2 * (8 + 36)               # operand reads + streamed fragments


def test_persistent_gemm_never_touches_its_bias_registers_before_a_counted_wait_retires_the_load(tmp_path):
    """gemm_nt_big8_kernel fetches the next tile's bias with an asm load the compiler knows nothing about (a plain load would
    make hipcc drain vmcnt at every tile seam); that is only sound while the compiled code leaves the destination registers alone
    until one of the kernel's own counted waits has retired the load -- the first `s_waitcnt vmcnt(6)` behind it (third k-tile
    of the tile, phase 1) or the `vmcnt(0)` in front of the last tile's stores.  Checked on the ISA hipcc produces here: in the
    text between the load and that wait the register is not named, and the control flow stays inside that text -- every branch in
    it goes FORWARD to a label in front of the wait (the skipped `if (ok)` stores of the seam), so no path leaves the region, or
    re-enters it from elsewhere, without passing the wait."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "big.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(root, "lstm-rnn_amd", "csrc", "cn_gemm_big.hip")], check=True, capture_output=True)
    lines = out.read_text().splitlines()
    loads = [i for i, l in enumerate(lines) if re.match(r"\s*global_load_dword v\d+,", l)]
    assert len(loads) == 4                                   # two per instantiation (C or C2 / both)
    for i in loads:
        reg = re.match(r"\s*global_load_dword (v\d+),", lines[i]).group(1)
        region = None
        for k, l in enumerate(lines[i + 1:], i + 1):
            if re.search(r"s_waitcnt vmcnt\((6|0)\)", l):
                region = (i + 1, k)
                break
            assert not re.search(r"\b%s\b" % reg, l), (reg, l)
        assert region, "no wait behind the bias load"
        labels = {m.group(1): k for k in range(*region) for m in [re.match(r"^(\.?\w+):", lines[k])] if m}
        for k in range(*region):
            m = re.match(r"\s*s_c?branch\w*\s+(\.?\w+)", lines[k])
            if m:
                assert m.group(1) in labels and labels[m.group(1)] > k, ("a branch leaves the region between the bias load and its wait", lines[k])
        # nothing outside jumps into the region either
        for name in labels:
            for k, l in enumerate(lines):
                if not (region[0] <= k < region[1]) and re.match(r"\s*s_c?branch\w*\s+%s\b" % re.escape(name), l):
                    raise AssertionError(("a branch enters the region between the bias load and its wait", l))


def test_last_arriver_hand_offs_drain_their_memory_operations_before_they_are_counted(tmp_path):
    """softmax_mcc_bwd_kernel hands its column-sum replicas (float atomics) and, through rowstat_reduce_wave, its loss partials
    (stores) to the last workgroup to arrive at a counter.  That is only sound when the adds / stores have been acknowledged before
    the counter is bumped: in the ISA an `s_waitcnt vmcnt(0)` must sit between the last replica add (partial store) and the counter
    atomic -- for the replicas in EVERY wave, in front of the barrier that precedes the count -- and the reader must invalidate
    (`buffer_inv sc1`) before its loads.  (A workgroup-scope release fence compiled to nothing here in round 4.)"""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "ew.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(root, "lstm-rnn_amd", "csrc", "cn_elementwise.hip")], check=True, capture_output=True)
    lines = out.read_text().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*softmax_mcc_bwd_kernel\w*:", l)]
    assert len(starts) == 2                                  # F32 = false / true
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [l.strip() for l in lines[st:end]]
        counters = [k for k, l in enumerate(body) if re.match(r"global_atomic_add v\d+, v\d+, v\d+, s\[", l)]   # returning u32 adds
        assert len(counters) == 2, counters                  # the loss partials' counter, the replicas' counter
        for k in counters:
            # no L2 write-back in front of the counter (what is handed over are atomics and sc1 stores; buffer_wbl2 there cost 1.6 us)
            assert not any(l.startswith("buffer_wbl2") for l in body[max(0, k - 6):k])
            # the acquire of the last arriver behind it
            tail = body[k:k + 40]
            assert any(l.startswith("buffer_inv sc1") for l in tail), tail
        # loss partials: two dword stores, then a drain, then the counter -- in straight-line code
        k0 = counters[0]
        stores = [k for k in range(max(0, k0 - 24), k0) if body[k].startswith("global_store_dword")]
        assert len(stores) == 2
        assert any("s_waitcnt vmcnt(0)" in body[k] for k in range(stores[-1], k0))
        # replicas: float adds, then every wave's own drain IN FRONT of the barrier that precedes the counter
        k1 = counters[1]
        adds = [k for k in range(k1) if body[k].startswith("global_atomic_add_f32")]
        assert adds
        barrier = max(k for k in range(adds[-1], k1) if body[k] == "s_barrier")
        assert any("s_waitcnt vmcnt(0)" in body[k] for k in range(adds[-1], barrier)), body[adds[-1]:barrier + 1]


def test_no_launch_path_reads_the_environment():
    """One options block per context (cn_internal.h: CN_OPTION_LIST, filled from the environment once in cn_ctx_create, changed by
    cn_ctx_set_option): no kernel source file calls getenv any more, and cn_api.cpp / cn_comm_ipc.cpp only where a context or a
    communicator is CREATED (46 switches were read on launch paths until round 5)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "lstm-rnn_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".inc", ".h")):
            assert "getenv(" not in open(os.path.join(csrc, name)).read(), name
    api = open(os.path.join(csrc, "cn_api.cpp")).read()
    # options_from_env (two macro lines), the RCCL library name, and the four context-level defaults inside cn_ctx_create
    lines = [l for l in api.splitlines() if "getenv(" in l]
    assert len(lines) == 8, lines
    create = api[api.index("int cn_ctx_create("):api.index("int cn_ctx_destroy(")]
    assert sum("getenv(" in l for l in create.splitlines()) == 5


def test_p2p_exchange_drains_every_wave_before_each_flag_store(tmp_path):
    """p2p_allreduce_kernel (cn_comm_p2p.hip) hands staged gradients to its peers through flag words in THEIR regions: READY
    (my staging half is written), REDUCED (my slice of it holds the sums), DONE (my reads of the peers' halves have returned).
    Each hand-off is only sound when every wave of the workgroup has had its own stores acknowledged (loads returned) before
    the barrier in front of the flag store.  A workgroup-scope release fence compiles to no vmcnt wait on gfx950 (round 5
    shipped that), so the wait is written in asm and the ISA is read here: exactly three asm `s_waitcnt vmcnt(0)`, each one in
    straight-line code in front of an `s_barrier`, the flag store (`global_store_dwordx2 ... sc0 sc1`) the first store behind
    that barrier, and the staging stores of the copy loop / the reduced slice in front of the drain."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "p2p.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(root, "lstm-rnn_amd", "csrc", "cn_comm_p2p.hip")], check=True, capture_output=True)
    lines = out.read_text().splitlines()
    st = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*p2p_allreduce_kernel\w*:", l))
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in lines[st:end]]
    drains = [k + 1 for k, l in enumerate(body[:-2]) if l.startswith(";;#ASMSTART") and body[k + 1] == "s_waitcnt vmcnt(0)" and body[k + 2].startswith(";;#ASMEND")]
    assert len(drains) == 3, drains                          # READY, REDUCED, DONE
    is_mem = lambda l: l.startswith(("global_", "buffer_", "flat_", "scratch_"))
    is_jump = lambda l: l.startswith(("s_branch", "s_cbranch", "s_endpgm")) or re.match(r"^\.?\w+:", l)
    for d in drains:
        bar = next(k for k in range(d, len(body)) if body[k] == "s_barrier")
        between = body[d + 1:bar]
        assert len(between) <= 6 and not any(is_mem(l) or is_jump(l) for l in between), between
        # behind the barrier: the flag store is the first store (the flag pointer a.flags[t] is a load from the argument block)
        first_store = next(l for l in body[bar + 1:] if l.startswith("global_store") or l.startswith("global_atomic"))
        assert re.match(r"global_store_dwordx2 .* sc0 sc1$", first_store), first_store
    # READY: the copy loop's staging stores sit in front of the first drain; REDUCED: the reduced slice's between drains 0 and 1
    staged = [k for k, l in enumerate(body) if re.match(r"global_store_dwordx2 .* sc0 sc1$", l)]
    assert any(k < drains[0] for k in staged)
    assert any(drains[0] < k < drains[1] for k in staged)
    # no barrier in the kernel is followed by a flag store without a drain directly in front of it: every s_barrier whose next
    # memory instruction (straight line) is an sc0 sc1 store must be one of the three above
    for k, l in enumerate(body):
        if l != "s_barrier":
            continue
        nxt = next((m for m in body[k + 1:] if is_mem(m) or is_jump(m)), "")
        if re.match(r"global_store_dwordx2 .* sc0 sc1$", nxt):
            assert any(d < k <= d + 7 for d in drains), body[max(0, k - 8):k + 4]


def test_generated_loop_of_the_two_cu_backward_kernel_is_in_sync_with_its_generator():
    """lstm-rnn_amd/csrc/cn_lstm_s2c_loop.inc is the output of tools/gen_s2c_loop.py (the step of lstm_bwd_s2c_asm_kernel as a
    schedule: poll, own K half, exchange, partner K half, block errors; the timing build's text beside it)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("S2C_POLL", "S2C_RMW")}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_s2c_loop.py")], capture_output=True, text=True, check=True, env=env).stdout
    with open(os.path.join(root, "lstm-rnn_amd", "csrc", "cn_lstm_s2c_loop.inc")) as f:
        assert f.read() == out
    plain = out.split("#define S2C_ASM_TEXT_STAMP")[0]
    assert plain.count("v_smfmac_f32_16x16x64_bf16") == 4 * 32          # four step bodies of 32 MFMAs
    assert plain.count("ds_read_b128") == 4 * 16                        # ONE view: eight reads per K half
    assert plain.count("s_barrier") == 4 * 2 and "s_memtime" not in plain
