#!/usr/bin/env python3
"""Writes lstm-rnn_amd/csrc/cn_lstm_s2w_loop.inc: the two step bodies (stage P / Q) of lstm_fwd_s2w_asm_kernel, the
hand-scheduled forward time loop for Hp = 256 on one CU (cn_lstm_s2.hip).  The schedule is a table -- 64 sparse MFMAs per wave
and step in a fixed order, every other instruction hung behind one of them -- and this script turns the table into the asm
string, counting the LDS operations in flight so that every `s_waitcnt lgkmcnt(n)` is derived instead of written by hand.

    python tools/gen_s2w_loop.py > lstm-rnn_amd/csrc/cn_lstm_s2w_loop.inc

Registers (fixed, clobbered by the asm statement):
    v[224:239]  accumulators n, i, f, o (registers 2, 3 of each stay zero: rows of zeros in both views of the tile)
    v[208:215]  stage P: pre-activations n, i, f, o of pair A, then of pair B;  v[216:223] stage Q
    v[176:199]  three buffers of one streamed W fragment each
    v[248:251]  n, i, f, o of pair A (one 16-byte store);  v[200:203] of pair B;  v[204:207] temporaries x0..x3
    v252 cell state before the dummy select, v253 tanh(cell state)
MFMA m = 32 S + 8 G + 2 K + V: pair S, gate G, K chunk K, view V (the order of lstm_fwd_s2w_kernel per accumulator).
"""
import sys

K1 = "0xbfb8aa3b"   # -log2(e)
K2 = "0xc038aa3b"   # -2 log2(e)
GATES = "nifo"
ACC = {"n": 224, "i": 228, "f": 232, "o": 236}
FBUF = [176, 184, 192]

# the streamed fragments in consumption order: (S, G, K, V); fragment j lives in buffer j % 3 and at LDS offset j * 2048
STREAM = [(0, "n", 2, 1), (0, "n", 3, 0), (0, "n", 3, 1), (0, "i", 3, 0), (0, "i", 3, 1), (0, "f", 3, 0), (0, "f", 3, 1),
          (0, "o", 3, 0), (0, "o", 3, 1), (1, "n", 2, 1), (1, "n", 3, 0), (1, "n", 3, 1), (1, "i", 3, 0), (1, "i", 3, 1),
          (1, "f", 3, 0), (1, "f", 3, 1), (1, "o", 3, 0), (1, "o", 3, 1)]
assert len(STREAM) % 3 == 0


def mfma_index(S, G, K, V):
    return 32 * S + 8 * GATES.index(G) + 2 * K + V


USE = {mfma_index(*f): j for j, f in enumerate(STREAM)}


class Step:
    def __init__(self, stage, rd, wr):
        self.stage = stage              # first register of the stage
        self.pt = "ptP" if stage == 208 else "ptQ"
        self.rd, self.wr = rd, wr       # LDS byte offsets of the tile read / written
        self.out = []
        self.lds = 0                    # LDS operations issued so far in this step
        self.ready = {}                 # name -> index of the LDS operation that delivers it
        self.done = 0                   # LDS operations known complete (by the waits emitted so far)
        self.nostream = False

    def emit(self, s):
        self.out.append(s)

    def lds_op(self, s, name=None):
        self.emit(s)
        self.lds += 1
        if name:
            self.ready[name] = self.lds

    def need(self, name):
        k = self.ready.get(name, 0)
        if k > self.done:
            n = self.lds - k
            assert n <= 15, (name, n)
            self.emit("s_waitcnt lgkmcnt(%d)" % n)
            self.done = k

    def load_fragment(self, j):
        if self.nostream:
            return
        b = FBUF[j % 3]
        self.lds_op("ds_read_b128 v[%d:%d], %%[wl] offset:%d" % (b, b + 3, j * 2048))
        self.lds_op("ds_read_b128 v[%d:%d], %%[wl] offset:%d" % (b + 4, b + 7, j * 2048 + 1024), "F%d" % j)

    def mfma(self, m):
        S, G, K, V = m >> 5, GATES[(m >> 3) & 3], (m >> 1) & 3, m & 1
        self.need("a%d%d" % (K, V))
        if m in USE:
            j = USE[m]
            self.need("F%d" % j)
            w = "v[%d:%d]" % (FBUF[j % 3], FBUF[j % 3] + 7)
        else:
            w = "%%[w%d%d%s%d]" % (S, V, G, K)
        a = ACC[G]
        self.emit("v_smfmac_f32_16x16x64_bf16 v[%d:%d], %%[a%d%d], %s, %%[spidx]" % (a, a + 3, K, V, w))


def chain(S, gate, x, out, cst, peep):
    """activation of one gate: list of instructions (strings); dependent transcendentals are kept apart by the caller's placement"""
    a = ACC[gate]
    if gate == "n":
        return ["v_add_f32 %s, v%d, v%d" % (x, a, a + 1), "v_mul_f32 %s, %s, %s" % (x, K2, x), "v_exp_f32 %s, %s" % (x, x),
                "v_add_f32 %s, 1.0, %s" % (x, x), "v_rcp_f32 %s, %s" % (x, x), "v_fma_f32 v%d, %s, 2.0, -1.0" % (out, x)]
    return ["v_add_f32 %s, v%d, v%d" % (x, a, a + 1), "v_fmac_f32 %s, %s, %s" % (x, peep, cst), "v_mul_f32 %s, %s, %s" % (x, K1, x),
            "v_exp_f32 %s, %s" % (x, x), "v_add_f32 %s, 1.0, %s" % (x, x), "v_rcp_f32 v%d, %s" % (out, x)]


STAMP = ["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %%[tq], s98, %%[tl]", "s_add_u32 %%[st%d], %%[st%d], %%[tq]", "s_mov_b32 %%[tl], s98"]


def build(stage, rd, wr, diag=""):
    st = Step(stage, rd, wr)
    F = {m: [] for m in range(-1, 65)}      # fillers behind MFMA m (-1: before the first, 64: behind the last one's guard distance)
    pa, pb = stage, stage + 4
    x0, x1, x2, x3, x4, x5 = "v204", "v205", "v206", "v207", "%[x4]", "%[x5]"

    def seed(gate, pre):
        a = ACC[gate]
        return ["v_mov_b32 v%d, v%d" % (a, pre), "v_mov_b32 v%d, 0" % (a + 1)]

    # ---- pair A: seeds, prefetch of pair A's stage two steps ahead, offsets
    F[-1] += seed("n", pa) + ["v_cmp_eq_u32 vcc, 0, %%[%s]" % st.pt]
    F[0] += seed("i", pa + 1)
    F[1] += seed("f", pa + 2)
    F[2] += seed("o", pa + 3)
    F[3] += ["global_load_ubyte %%[%s], %%[oP], %%[patpf]" % st.pt, "global_load_dwordx4 v[%d:%d], %%[oA], %%[actspf]" % (pa, pa + 3)]
    F[4] += ["v_add_u32 %[oA], %[oA], %[sA]"]
    F[5] += ["v_add_u32 %[oP], %[oP], %[sP]"]
    F[6] += ["v_add_u32 %[oC], %[oC], %[sC]"]
    F[7] += ["v_add_u32 %[oY], %[oY], %[sY]"]

    def place(instrs, first, spread=1):
        """one instruction per MFMA gap from `first` on (a transcendental's consumer then never follows it directly)"""
        m = first
        for ins in instrs:
            F[m].append(ins)
            m += spread
        return m

    # ---- pair A activations (a gate's accumulator is read four MFMAs after its last one at the earliest)
    place(chain(0, "n", x0, 248, None, None), 11)
    place(chain(0, "i", x1, 249, "%[cstA]", "%[piA]"), 19)
    place(chain(0, "f", x2, 250, "%[cstA]", "%[pfA]"), 27)
    F[33] += ["v_mul_f32 %s, %%[cstA], v250" % x3]
    F[34] += ["v_fma_f32 v252, v248, v249, %s" % x3]
    F[35] += ["v_add_f32 %s, v236, v237" % x4, "v_mul_f32 %s, %s, v252" % (x5, K2)]
    F[36] += ["v_fmac_f32 %s, %%[poA], v252" % x4, "v_exp_f32 %s, %s" % (x5, x5)]
    F[37] += ["v_mul_f32 %s, %s, %s" % (x4, K1, x4), "v_cndmask_b32_e64 %[cstA], v252, 0, vcc"]
    F[38] += ["v_exp_f32 %s, %s" % (x4, x4), "v_add_f32 %s, 1.0, %s" % (x5, x5)]
    F[39] += ["v_rcp_f32 %s, %s" % (x5, x5)]
    F[40] += ["v_add_f32 %s, 1.0, %s" % (x4, x4)]
    F[41] += ["v_rcp_f32 v251, %s" % x4, "v_fma_f32 v253, %s, 2.0, -1.0" % x5]
    F[42] += []
    F[43] += ["v_mul_f32 %s, v253, v251" % x4]
    F[44] += ["v_cvt_pk_bf16_f32 %s, %s, %s" % (x4, x4, x4)]
    F[45] += ["v_cndmask_b32_e64 %s, %s, 0, vcc" % (x4, x4)]
    F[46] += ["LDSW ds_write_b16 %%[oTA], %s offset:%d" % (x4, wr), "global_store_dwordx4 %[oA], v[248:251], %[acts1]"]
    F[47] += ["global_store_dword %[oC], %[cstA], %[cell1]"]
    F[48] += ["global_store_dword %[oC], v253, %[th1]"]
    F[49] += ["global_store_short %%[oY], %s, %%[yop1]" % x4]

    # ---- pair B: seeds behind the reads of pair A's sums, its stage's prefetch behind the last seed
    F[19] += ["s_waitcnt vmcnt(21)"]
    F[20] += seed("n", pb)
    F[26] += seed("i", pb + 1)
    F[31] += seed("f", pb + 2)
    F[40] += seed("o", pb + 3)
    F[42] += ["global_load_dwordx4 v[%d:%d], %%[oA], %%[actspf1] offset:512" % (pb, pb + 3)]
    # ---- pair B activations: n, i, f under its own MFMAs (x0..x2 are free again), the output gate behind the last MFMA
    place(chain(1, "n", x0, 200, None, None), 43)
    place(chain(1, "i", x1, 201, "%[cstB]", "%[piB]"), 51)
    place(chain(1, "f", x2, 202, "%[cstB]", "%[pfB]"), 59)   # (its last two instructions land behind the last MFMA)
    tail = ["s_nop 1",
            "v_mul_f32 %s, %%[cstB], v202" % x3,
            "v_fma_f32 v252, v200, v201, %s" % x3,
            "s_nop 7",                                    # (the last MFMA's result is not interlocked against VALU reads)
            "v_add_f32 %s, v236, v237" % x4,
            "v_mul_f32 %s, %s, v252" % (x5, K2),
            "v_fmac_f32 %s, %%[poB], v252" % x4,
            "v_exp_f32 %s, %s" % (x5, x5),
            "v_mul_f32 %s, %s, %s" % (x4, K1, x4),
            "v_cndmask_b32_e64 %[cstB], v252, 0, vcc",
            "v_exp_f32 %s, %s" % (x4, x4),
            "v_add_f32 %s, 1.0, %s" % (x5, x5),
            "v_rcp_f32 %s, %s" % (x5, x5),
            "v_add_f32 %s, 1.0, %s" % (x4, x4),
            "v_rcp_f32 v203, %s" % x4,
            "v_fma_f32 v253, %s, 2.0, -1.0" % x5,
            "s_nop 0",
            "v_mul_f32 %s, v253, v203" % x4,
            "v_cvt_pk_bf16_f32 %s, %s, %s" % (x4, x4, x4),
            "v_cndmask_b32_e64 %s, %s, 0, vcc" % (x4, x4),
            "LDSW ds_write_b16 %%[oTB], %s offset:%d" % (x4, wr),
            "global_store_dwordx4 %[oA], v[200:203], %[acts1] offset:512",
            "global_store_dword %[oC], %[cstB], %[cell1] offset:128",
            "global_store_dword %[oC], v253, %[th1] offset:128",
            "global_store_short %%[oY], %s, %%[yop1] offset:64" % x4]

    # ---- W stream: fragment j + 3 is fetched into fragment j's buffer right behind the MFMA that consumed j
    fetch = {}
    for m, j in USE.items():
        fetch.setdefault(m, []).append((j + 3) % len(STREAM))

    # ---- emit
    st.nostream = diag == "nostream"
    stamp = diag == "stamp"

    def mark(i):
        if stamp:
            for ins in STAMP:
                st.emit(ins % (i, i) if "%d" in ins else ins.replace("%%", "%"))
            st.done = st.lds            # (the stamp waits for every LDS operation)
    st.emit("s_waitcnt vmcnt(20)")
    mark(0)
    for k in range(4):
        for v in range(2):
            st.lds_op("ds_read_b128 %%[a%d%d], %%[av%d] offset:%d" % (k, v, v, rd + 64 * k), "a%d%d" % (k, v))
    # (fragments 0..2 of this step were fetched by the previous step; the barrier's lgkmcnt(0) retired them)
    for j in range(3):
        st.ready["F%d" % j] = 0

    def fillers(m):
        for ins in F[m]:
            if diag == "nofill" and ins.startswith(("v_add_f32", "v_mul_f32", "v_exp", "v_rcp", "v_fma", "v_fmac", "v_cvt", "v_cndmask")):
                continue
            if ins.startswith("LDSW "):
                st.lds_op(ins[5:])
            else:
                st.emit(ins)
        for j in fetch.get(m, []):
            st.load_fragment(j)

    fillers(-1)
    for m in range(64):
        st.mfma(m)
        fillers(m)
        if m == 7:
            mark(1)
        if m == 31:
            mark(2)
    fillers(64)
    mark(3)
    for ins in tail:
        if ins.startswith("LDSW "):
            st.lds_op(ins[5:])
        else:
            st.emit(ins)
    st.emit("s_waitcnt lgkmcnt(0)")
    mark(4)
    st.emit("s_barrier")
    mark(5)
    st.emit("s_sub_u32 %[cnt], %[cnt], 1")
    st.emit("s_cbranch_scc1 9f")
    return st.out


def main():
    print("// generated by tools/gen_s2w_loop.py -- do not edit; the schedule table and the register map are in the script")
    # diagnostic builds (never shipped): -DCN_S2W_DIAG_NOSTREAM leaves the streamed fragments' buffers alone (wrong results, timing
    # only), -DCN_S2W_STAMP sums s_memtime deltas per step segment (tools/stamps_s2.py)
    for cond, diag in (("#if defined(CN_S2W_DIAG_NOSTREAM)", "nostream"), ("#elif defined(CN_S2W_DIAG_NOFILL)", "nofill"), ("#elif defined(CN_S2W_STAMP)", "stamp"), ("#else", "")):
        print(cond)
        for name, stage, rd, wr in (("S2W_STEP_P", 208, 0, 1440), ("S2W_STEP_Q", 216, 1440, 0)):
            body = build(stage, rd, wr, diag)
            print("#define %s \\" % name)
            for i, ins in enumerate(body):
                print('    "%s\\n\\t"%s' % (ins, " \\" if i + 1 < len(body) else ""))
            print()
    print("#endif")
    # the W operands: chunks 0, 1 of every fragment in AGPRs, chunk 2 in VGPRs except the two streamed ones
    ag = ['[w%d%d%s%d] "a"(wa[%d][%d][%d][%d])' % (S, V, G, K, S, V, GATES.index(G), K) for S in range(2) for V in range(2) for G in GATES for K in range(2)]
    vg = ['[w%d%d%s2] "v"(wv[%d][%d][%d])' % (S, V, G, S, V, GATES.index(G)) for S in range(2) for V in range(2) for G in GATES if (S, G, 2, V) not in STREAM]
    print("#define S2W_W_OPERANDS \\")
    print("    " + ", \\\n    ".join(", ".join(ag[i:i + 4]) for i in range(0, len(ag), 4)) + ", \\")
    print("    " + ", \\\n    ".join(", ".join(vg[i:i + 4]) for i in range(0, len(vg), 4)))
    print()
    # where the kernel puts the streamed fragments: j -> pair, view, gate, chunk
    print("#define S2W_STREAM_COUNT %d" % len(STREAM))
    print("#define S2W_STREAM_TABLE { " + ", ".join("{%d, %d, %d, %d}" % (S, V, GATES.index(G), K) for (S, G, K, V) in STREAM) + " }")
    print("// instructions per step: %d" % len(build(208, 0, 1440)), file=sys.stderr)


if __name__ == "__main__":
    main()
