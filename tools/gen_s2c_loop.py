#!/usr/bin/env python3
"""Writes lstm-rnn_amd/csrc/cn_lstm_s2c_loop.inc: the asm text of lstm_bwd_s2c_asm_kernel's time loop -- the backward pass of a
256-units-per-direction layer on clusters of TWO CUs, two sequences per cluster, one wave per SIMD (cn_lstm_cluster.hip).

    python tools/gen_s2c_loop.py > lstm-rnn_amd/csrc/cn_lstm_s2c_loop.inc

The loop is lstm_bwd_s2_asm_kernel's (cn_lstm_s2.hip: four prefetch stages, the e-independent block of step t+1 formed at the end
of step t, every step the same code, left after any step) with the K = 1024 product cut in two halves and the exchange between
them.  A step:

    poll loads (the partner's two granules of the previous step): their trip through L2, ~1100 cycles, is the longest thing in a step
    8 tile reads + 16 sparse MFMAs: OWN K half -> accumulators A (one per unit group)     -- while the granules travel
    wait for the poll (counted: the prefetches issued behind it stay in flight), check the tags (retry path out of line)
    partner's deltas -> their rows of the tile, barrier
    8 tile reads + 16 MFMAs: PARTNER K half -> accumulators B
    e = A0 + A1 + B2 + B3 of the lane's unit group, block errors, deltas -> bf16: publish (two tagged 8-byte granules), own rows
    of the NEXT tile, delta_op; block of the next step; barrier
Where the poll is issued was measured (S2C_POLL=top | early | both; reading B / LVCSR ms per fraction, same box, interleaved):
top of the step 2.59 / 9.39 (later in the round, other box: 2.573 / 9.30); behind MFMA 2 / 4 / 6 / 9 of the own half, prefetch
behind it ("mid", S2C_POLL_AT; what a delayed first sample bought the 8-CU kernels): 2.80 / 9.93, 2.81 / 9.93, 2.82 / 10.0,
2.83 / 10.08 -- here the own half is long enough for the top sample to come back with the data; in front of the previous step's last barrier (its sample often arrives in L2 before the partner's
store: retries) 2.61 / 9.49; both samples 2.64 / 9.64 (the extra sample costs more than it finds, as in the 8-wave kernels).

ONE view of the tile (round 5; the s2 kernels read it through two zero-padded views, one per unit group): lanes c >= 8 read the
rows of lanes c - 8, so the A operand's rows 8 .. 15 repeat rows 0 .. 7 and ONE read serves both unit groups' MFMAs -- into
separate accumulators (the MFMA of group j leaves that group's sums for BOTH sequences in all four lane quarters; a lane keeps
the accumulator of its own group, a select per step).  Half the LDS reads (a phase was bound by them: 16 reads ~ 620 cycles).

Fixed registers (clobbered): v[128:159] the eight operand reads of a phase, v[160:163] the polled granules, stage k = 0..3 at
b = 168 + 20k: v[b : b+3] n,i,f,o, v[b+4 : b+7] / v[b+8 : b+11] accumulators A of unit group 0 / 1, v[b+12 : b+15] / v[b+16 : b+19]
accumulators B; v[248:249] the four bf16 deltas, v[250:253] the two granules {value, tag}.  W_rec^T fragments: "a" operands
w{j}k{kc} (256 AGPRs).  vmcnt per step, in order: poll x2, prefetch x4, outputErrors, publish x2, delta_op store = 10 operations.
"""
import os

POLL_AT = int(os.environ.get("S2C_POLL_AT", "4"))     # "mid": the own half's MFMA (0 .. 11) behind which the sample leaves
POLL = os.environ.get("S2C_POLL", "top")        # where a step's poll is issued: "top" of the step, "early" (in front of the previous step's
                                                # last barrier), "both" (two samples: the early one is looked at first)
STAMP = False        # the second text (S2C_ASM_TEXT_STAMP, -DCN_S2C_STAMP builds): s_memtime deltas per step segment summed into st0 .. st6
PLANE = 9 * 544          # tile plane in bytes (9 rows of lds_pitch(8 * 64) = 544)
ROW2 = 544 // 4          # second row of a ds_write2_b32, in dwords
G1 = 2048                # byte offset of a lane's second granule (NT * 8)
WIN = 12                 # tile reads in flight at most (lgkmcnt counts to 15)


def rd(kc):
    b = 128 + kc * 4
    return "v[%d:%d]" % (b, b + 3)


class Text:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)


def stamp(o, i):
    if STAMP:
        for t in ["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %[tz], s98, %[tl]", "s_add_u32 %%[st%d], %%[st%d], %%[tz]" % (i, i), "s_mov_b32 %[tl], s98"]:
            o(t)


def phase(o, acc0, acc1, wbase, plane_off, fillers):
    """8 reads (the chunks of one plane, one view) feeding 16 MFMAs: chunk kc x w0k{wbase + kc} -> acc0, x w1k{wbase + kc} -> acc1;
    counted waits; fillers[i] is issued behind MFMA i."""
    for kc in range(8):
        o("ds_read_b128 %s, %%[av] offset:%d" % (rd(kc), plane_off + 64 * kc))
    i = 0
    for kc in range(8):
        o("s_waitcnt lgkmcnt(%d)" % (7 - kc))
        for j, acc in ((0, acc0), (1, acc1)):
            o("v_smfmac_f32_16x16x64_bf16 %s, %s, %%[w%dk%d], %%[spidx]" % (acc, rd(kc), j, wbase + kc))
            for f in fillers.get(i, []):
                o(f)
            i += 1


def block(o, NI, IG, FG, OG, TH, CP, PT, CN):
    """S2B_BLOCK of cn_lstm_s2.hip: the e-independent terms of the step that this stage holds"""
    for s in ["v_cmp_eq_u32 vcc, 0, %%[%s]" % PT,
              "v_fma_f32 %%[x0], -%s, %s, %s" % (OG, OG, OG),
              "v_fma_f32 %%[x1], -%%[%s], %%[%s], 1.0" % (TH, TH),
              "v_cndmask_b32_e64 %%[%s], %%[%s], 0, %%[last]" % (CN, CP),
              "v_cndmask_b32_e64 %[m], 1.0, 0, vcc",
              "v_mul_f32 %%[t2m], %%[x0], %%[%s]" % TH,
              "v_mul_f32 %%[x1], %s, %%[x1]" % OG,
              "v_fma_f32 %%[x0], -%s, %s, 1.0" % (NI, NI),
              "v_mul_f32 %[car], %[fgn], %[ecn]",
              "v_fma_f32 %[wm], %[po], %[t2m], %[x1]",
              "v_mul_f32 %%[d2m], %s, %%[x0]" % IG,
              "v_fma_f32 %%[x0], -%s, %s, %s" % (FG, FG, FG),
              "v_fmac_f32 %[car], %[pi], %[dign]",
              "v_fma_f32 %%[x1], -%s, %s, %s" % (IG, IG, IG),
              "v_mul_f32 %%[d3m], %%[x0], %%[%s]" % CN,
              "v_fmac_f32 %[car], %[pf], %[dfgn]",
              "v_mul_f32 %%[d4m], %%[x1], %s" % NI,
              "v_mul_f32 %%[fgn], %s, %%[m]" % FG]:
        o(s)


def stage_regs(k):
    b = 168 + 20 * k
    return dict(N="v%d" % b, I="v%d" % (b + 1), F="v%d" % (b + 2), O="v%d" % (b + 3), AX="v[%d:%d]" % (b, b + 3),
                A=["v[%d:%d]" % (b + 4, b + 7), "v[%d:%d]" % (b + 8, b + 11)], A0=["v%d" % (b + 4), "v%d" % (b + 8)], A1=["v%d" % (b + 5), "v%d" % (b + 9)],
                B=["v[%d:%d]" % (b + 12, b + 15), "v[%d:%d]" % (b + 16, b + 19)], B2=["v%d" % (b + 14), "v%d" % (b + 18)], B3=["v%d" % (b + 15), "v%d" % (b + 19)],
                NEVER=[b + 6, b + 7, b + 10, b + 11, b + 12, b + 13, b + 16, b + 17],
                TH="th%d" % k, CP="cp%d" % k, PT="pt%d" % k)


RMW = os.environ.get("S2C_RMW", "1") == "1"       # poll with a returning atomic OR of zero (executes in the L2) instead of an sc1 load: reading B 2.600 -> 2.589, LVCSR 9.46 -> 9.38 ms (A.6: 13 % per hop in the probe, correct across XCDs too)


def poll_issue(o, XT, second=False):
    b = 164 if second else 160
    if RMW:
        o("global_atomic_or_x2 v[%d:%d], %%[%s], v[254:255], %%[xch] sc0" % (b, b + 1, XT))
        o("global_atomic_or_x2 v[%d:%d], %%[%s], v[254:255], %%[xch] offset:%d sc0" % (b + 2, b + 3, XT, G1))
        return
    o("global_load_dwordx2 v[%d:%d], %%[%s], %%[xch] sc1" % (b, b + 1, XT))
    o("global_load_dwordx2 v[%d:%d], %%[%s], %%[xch] offset:%d sc1" % (b + 2, b + 3, XT, G1))


def step(o, k):
    s, nx = stage_regs(k), stage_regs((k + 1) % 4)
    par = k & 1
    R = par * PLANE                                     # plane read
    WT = "oT1" if par == 0 else "oT"                    # own rows of the plane written (the next step's)
    TP = "oTp" if par == 0 else "oTp1"                  # partner rows of the plane read
    XM = "oXm%d" % par                                  # my granules of this step's slot set
    CS = "ccA" if par == 0 else "ccB"
    stamp(o, 0)                                          # segment 0: barrier / loop control of the previous step
    if POLL == "top":
        poll_issue(o, "oXt%d" % (1 - par))
    elif POLL == "both":
        poll_issue(o, "oXt%d" % (1 - par), second=True)
    o("v_mov_b32 %s, %s" % (s["A0"][1], s["A0"][0]))    # outputErrors = the C operand of both groups' first MFMA
    for r in (s["A1"][0], s["A1"][1], s["B2"][0], s["B3"][0], s["B2"][1], s["B3"][1]):
        o("v_mov_b32 %s, 0" % r)
    # ---- prefetch of step t+4 into this stage (its block was formed at the end of the previous step: the registers are dead)
    prefetch = ["global_load_dwordx4 %s, %%[oA], %%[actspf]" % s["AX"],
                "global_load_dword %%[%s], %%[oC], %%[thpf]" % s["TH"],
                "global_load_dword %%[%s], %%[oC], %%[cellpf]" % s["CP"],
                "global_load_ubyte %%[%s], %%[oP], %%[patpf]" % s["PT"]]
    if POLL != "mid":
        for t in prefetch:
            o(t)
    # ---- own K half while the granules travel; offsets and the factor-m products in its gaps (one VALU per gap)
    own_fill = {0: ["v_add_u32 %[oA], %[oA], %[sA]"], 1: ["v_add_u32 %[oC], %[oC], %[sC]"], 2: ["v_add_u32 %[oD], %[oD], %[sD]"],
                3: ["v_add_u32 %[oP], %[oP], %[sP]"], 4: ["v_mul_f32 %[t2m], %[t2m], %[m]"], 5: ["v_mul_f32 %[wm], %[wm], %[m]"],
                6: ["v_mul_f32 %[carm], %[car], %[m]"], 7: ["v_mul_f32 %[d2m], %[d2m], %[m]"], 8: ["v_mul_f32 %[d3m], %[d3m], %[m]"],
                9: ["v_mul_f32 %[d4m], %[d4m], %[m]"],
                # the previous step's gradient sums (its deltas are still in dni .. dog; c[prev] of that step = this step's cell state)
                10: ["v_add_f32 %[sb0], %[sb0], %[dni]"], 11: ["v_add_f32 %[sb1], %[sb1], %[dign]"],
                12: ["v_add_f32 %[sb2], %[sb2], %[dfgn]"], 13: ["v_add_f32 %[sb3], %[sb3], %[dog]"],
                14: ["v_fmac_f32 %%[spi], %%[%s], %%[dign]" % CS], 15: ["v_fmac_f32 %%[spf], %%[%s], %%[dfgn]" % CS]}
    if POLL == "mid":
        # the sample leaves behind MFMA number POLL_AT of the own half (a sample that leaves before the partner's granules are in L2
        # costs a whole extra round trip), the prefetch behind it (the order of the vector-memory queue stays poll, prefetch: the
        # wait for the sample leaves the prefetch in flight), the offsets' move to the next step behind the prefetch
        tmp = Text()
        poll_issue(tmp, "oXt%d" % (1 - par))
        moves = [own_fill[i][0] for i in range(4)]
        for i in range(4):
            own_fill[i] = []
        own_fill[POLL_AT] = own_fill.get(POLL_AT, []) + tmp.lines + prefetch
        for i in range(4):
            own_fill[POLL_AT + 1 + i] = own_fill.get(POLL_AT + 1 + i, []) + [moves[i]]
    phase(o, s["A"][0], s["A"][1], 0, R, own_fill)
    stamp(o, 1)                                          # 1: top + own half
    # ---- the partner's deltas: wait for the two poll loads (the four prefetch loads behind them stay in flight), check the tags
    if POLL == "both":
        # the early sample (two more loads, the second sample, sit between it and the prefetch); a lane it missed takes the
        # second sample, issued a barrier later
        o("s_waitcnt vmcnt(6)")
        o("v_cmp_ne_u32 vcc, %[tagc], v161")
        o("v_cmp_ne_u32 %[tq], %[tagc], v163")
        o("s_or_b64 vcc, vcc, %[tq]")
        o("s_cbranch_vccz %df" % (30 + k))
        o("s_waitcnt vmcnt(4)")
        o("v_mov_b32 v160, v164")
        o("v_mov_b32 v161, v165")
        o("v_mov_b32 v162, v166")
        o("v_mov_b32 v163, v167")
    else:
        o("s_waitcnt vmcnt(4)")
    o("v_cmp_ne_u32 vcc, %[tagc], v161")
    o("v_cmp_ne_u32 %[tq], %[tagc], v163")
    o("s_or_b64 vcc, vcc, %[tq]")
    o("s_cbranch_vccnz %df" % (20 + k))
    o("%d:" % (30 + k))
    stamp(o, 2)                                          # 2: waiting for the poll
    o("ds_write2_b32 %%[%s], v160, v162 offset0:0 offset1:%d" % (TP, ROW2))
    o("s_cmp_eq_u32 %[cnt], 1")
    o("s_cselect_b64 %[last], -1, 0")
    o("s_waitcnt lgkmcnt(0)")
    o("s_barrier")
    stamp(o, 3)                                          # 3: partner rows + barrier
    # ---- partner K half
    phase(o, s["B"][0], s["B"][1], 8, R, {})
    stamp(o, 4)                                          # 4: partner half
    o("s_nop 7")                                        # (the MFMAs' results: 11 wait states before the first read)
    o("s_nop 3")
    o("v_add_f32 %%[x0], %s, %s" % (s["A0"][0], s["A1"][0]))
    o("v_add_f32 %%[x1], %s, %s" % (s["B2"][0], s["B3"][0]))
    o("v_add_f32 %[x0], %[x0], %[x1]")
    o("v_add_f32 %%[car], %s, %s" % (s["A0"][1], s["A1"][1]))       # (car is free here: carm holds its product with m)
    o("v_add_f32 %%[x1], %s, %s" % (s["B2"][1], s["B3"][1]))
    o("v_add_f32 %[car], %[car], %[x1]")
    o("v_cndmask_b32_e64 %[x0], %[x0], %[car], %[ugm]")              # the lane's own unit group
    o("global_load_dword %s, %%[oC], %%[errpf]" % s["A0"][0])        # outputErrors of step t+4 = this stage's next C operand
    for t in ["v_mul_f32 %[dog], %[t2m], %[x0]",
              "v_fma_f32 %[ecn], %[x0], %[wm], %[carm]",
              "v_med3_f32 %[dog], %[dog], -1.0, 1.0",
              "v_mul_f32 %[dni], %[d2m], %[ecn]",
              "v_mul_f32 %[dfgn], %[d3m], %[ecn]",
              "v_mul_f32 %[dign], %[d4m], %[ecn]",
              "v_med3_f32 %[dni], %[dni], -1.0, 1.0",
              "v_med3_f32 %[dfgn], %[dfgn], -1.0, 1.0",
              "v_med3_f32 %[dign], %[dign], -1.0, 1.0",
              "v_cvt_pk_bf16_f32 v249, %[dfgn], %[dog]",
              "v_cvt_pk_bf16_f32 v248, %[dni], %[dign]",
              "v_add_u32 v251, 1, %[tagc]"]:
        o(t)
    # publish first (the partner waits for it), then the own rows of the next tile, then delta_op
    o("v_mov_b32 v250, v248")
    o("v_mov_b32 v252, v249")
    o("v_mov_b32 v253, v251")
    o("global_store_dwordx2 %%[%s], v[250:251], %%[xch] sc1" % XM)
    o("global_store_dwordx2 %%[%s], v[252:253], %%[xch] offset:%d sc1" % (XM, G1))
    o("ds_write2_b32 %%[%s], v248, v249 offset0:0 offset1:%d" % (WT, ROW2))
    o("global_store_dwordx2 %[oD], v[248:249], %[delta1]")
    o("v_fmac_f32 %%[spo], %%[%s], %%[dog]" % CS)
    o("v_mov_b32 %[tagc], v251")
    stamp(o, 5)                                          # 5: sums, chain, publish, stores
    # the next stage's loads (issued three steps ago) have landed: 5 + 20 + 8 operations have been issued behind them (two samples
    # per step: 5 + 24 + 10)
    o("s_waitcnt vmcnt(%d)" % (39 if POLL == "both" else 33))
    block(o, nx["N"], nx["I"], nx["F"], nx["O"], nx["TH"], nx["CP"], nx["PT"], CS)
    # the next step's poll: the partner published about as long ago as this member did (the block above), its store has had that
    # time to reach L2; the sample comes back behind the next step's own half
    if POLL in ("early", "both"):
        poll_issue(o, "oXt%d" % par)
    o("s_waitcnt lgkmcnt(0)")
    stamp(o, 6)                                          # 6: wait for the stage + block
    o("s_barrier")
    o("s_sub_u32 %[cnt], %[cnt], 1")
    o("s_cbranch_scc1 9f")


def retry(o, k):
    """out of line: a lane's granules had not arrived when the sample came back"""
    par = k & 1
    XT = "oXt%d" % (1 - par)
    o("%d:" % (20 + k))
    o("s_cmp_lg_u32 %[gave], 0")                        # a poll that timed out before: do not wait again
    o("s_cbranch_scc1 %db" % (30 + k))
    o("s_mov_b32 %[spin], 0")
    o("%d:" % (40 + k))
    o("s_sleep 1")
    poll_issue(o, XT)
    o("s_waitcnt vmcnt(0)")
    o("v_cmp_ne_u32 vcc, %[tagc], v161")
    o("v_cmp_ne_u32 %[tq], %[tagc], v163")
    o("s_or_b64 vcc, vcc, %[tq]")
    o("s_cbranch_vccz %db" % (30 + k))
    o("s_add_u32 %[spin], %[spin], 1")
    o("s_cmp_lt_u32 %[spin], 0x100000")
    o("s_cbranch_scc1 %db" % (40 + k))
    o("s_mov_b32 %[gave], 1")                           # ~1 s: the partner never arrived; report and go on
    o("v_mov_b32 %[x0], 1")
    o("v_mov_b32 %[x1], 0")
    o("global_store_dword %[x1], %[x0], %[fault]")
    o("s_branch %db" % (30 + k))


def first(o, k):
    s = stage_regs(k)
    o("global_load_dwordx4 %s, %%[x0], %%[acts]" % s["AX"])
    o("global_load_dword %%[%s], %%[x1], %%[th]" % s["TH"])
    o("global_load_dword %%[%s], %%[x1], %%[cell1]" % s["CP"])
    o("global_load_ubyte %%[%s], %%[m], %%[pat]" % s["PT"])
    o("global_load_dword %s, %%[x1], %%[err]" % s["A0"][0])
    o("v_add_u32 %[x0], %[x0], %[sA]")
    o("v_add_u32 %[x1], %[x1], %[sC]")
    o("v_add_u32 %[m], %[m], %[sP]")


def text():
    o = Text()
    # accumulator registers that are never read collect products of rows that belong elsewhere: cleared once
    for k in range(4):
        for r in stage_regs(k)["NEVER"]:
            o("v_mov_b32 v%d, 0" % r)
    if RMW:
        o("v_mov_b32 v254, 0")
        o("v_mov_b32 v255, 0")
    o("global_load_dword %[ccA], %[oC], %[cell]")
    o("v_mov_b32 %[x0], %[oA]")
    o("v_mov_b32 %[x1], %[oC]")
    o("v_mov_b32 %[m], %[oP]")
    for k in range(4):
        first(o, k)
    o("s_waitcnt vmcnt(0)")
    o("s_cmp_eq_u32 %[cnt], 0")
    o("s_cselect_b64 %[last], -1, 0")
    o("s_nop 1")
    s0 = stage_regs(0)
    block(o, s0["N"], s0["I"], s0["F"], s0["O"], s0["TH"], s0["CP"], s0["PT"], "ccB")
    if POLL in ("early", "both"):
        poll_issue(o, "oXt1")                           # step 0 polls the "step -1" granules (zeros) the members publish in front of the loop
    o("1:")
    for k in range(4):
        step(o, k)
    o("s_branch 1b")
    for k in range(4):
        retry(o, k)
    o("9:")
    o("s_waitcnt vmcnt(0)")
    # gradient sums of the last step (its c[prev] is 0: lastCall)
    o("v_add_f32 %[sb0], %[sb0], %[dni]")
    o("v_add_f32 %[sb1], %[sb1], %[dign]")
    o("v_add_f32 %[sb2], %[sb2], %[dfgn]")
    o("v_add_f32 %[sb3], %[sb3], %[dog]")
    return o.lines


def main():
    global STAMP
    print("// generated by tools/gen_s2c_loop.py -- do not edit (tests/test_abi_and_host.py keeps the two in sync)")
    for name, stamped in (("S2C_ASM_TEXT", False), ("S2C_ASM_TEXT_STAMP", True)):
        STAMP = stamped
        lines = text()
        print("#define %s \\" % name)
        for i, l in enumerate(lines):
            end = " \\" if i + 1 < len(lines) else ""
            print('    "%s\\n\\t"%s' % (l, end))
        print()


if __name__ == "__main__":
    main()
