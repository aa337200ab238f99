#!/usr/bin/env python3
"""Generator of the 64-rows-per-wave form of K5's pipelined block (rsa_attn_block64.h), head dim 128.

One wave owns 64 query rows (two 32-row halves h = 0, 1) and the WHOLE register file of its SIMD (one wave per SIMD, 512
registers): every K fragment and every V^T fragment read from LDS feeds TWO MFMAs (one per row half), so the LDS operand
reads per MFMA halve against the 32-row form (24 reads per 32 MFMAs instead of per 16).  One block = one 32-key sub-step:

    S_nxt[h]^T = K(sub-tile u+1) . Q[h]^T - m[h]     16 MFMAs   ks = 0..7, h = 0, 1  (A = K rows by ds_read_b128, B = Q[h]
                                                                from the ACCUMULATOR file, C of the first = the -m block)
    P[h]       = exp2(S_cur[h])                       in place, fp32; row sums; packed to the 2-byte type
    O[h]^T    += V(sub-tile u)^T . P[h]^T             16 MFMAs   (k2, dt) = 8 V^T fragments x h (A = V^T by ds_read_b64_tr_b16,
                                                                B = packed P[h], C = D = O[h][dt] in the ACCUMULATOR file)
    mx[h]      = row max of S_nxt[h]

Register map.  Accumulator file (owned by the asm statements, never seen by the compiler as values: a "+a" operand makes
hipcc keep the tile in arch VGPRs and copy 128 registers in and out around every statement):
    O[h][dt]  a[16*(4h+dt) : +15]   (a0..a127)         Q[h][ks]  a[128 + 4*(8h+ks) : +3]   (a128..a191)
Arch VGPRs, pinned through physical-register constraints:
    SA[h] v[16h : +15]   SB[h] v[32+16h : +15]   -m[h] v[64+16h : +15]   P[h] v[96+8h : +7]   K ring v[112:127]   V ring v[128:143]
    scratch v144..v149   K read addresses v[152:159]   V read addresses v[160:167]

Schedule: the vector work is dealt into the 32 MFMA gaps by issue cost (cdna_hip_programming.md: <= 5 single-issue
instructions per v_mfma_f32_32x32x16 gap, at most one of them a v_exp_f32, costs summing to <= 24 cycles hide) in deadline
order: exponentials + packing of the first 16 keys (needed by PV MFMA 16), of the second 16 keys (MFMA 24), row sums, then
the two row maxima interleaved with each other so that the v_permlane32_swap wait states are other useful instructions.

LDS-DMA inside the block: the wave's 8 one-KiB pieces of the tile being staged are issued in 8 gaps of the block (every
fourth), so that the pieces reach the CU's texture addresser spread over the sub-step instead of as a burst behind the
barrier (the 32-row kernel's staging point costs each wave ~450 cycles for 4 pieces: 8 waves x 4 pieces queue at one
addresser, profiles/r03_k5_block.md).  The source address walks in SGPRs (s90:s91 base, s92 LDS destination), the swizzled
per-lane source offset alternates between two VGPRs (pieces on even / odd 8-row groups).  When the sub-step stages nothing
from inside the block (boundary tiles, staged by the C++ side in front of it with clamped rows; the last tiles) operand
`dm` is 0 and every piece is branched over (s_cbranch_vccz): ONE statement per (slot parity, sub-step parity) -- alternative
statements that define the same pinned tiles make hipcc copy the tiles around every one of them.

usage: python3 gen_k5_block64.py > rsa_attn_block64.h
"""
import sys

AHEAD = 4
COST = dict(exp=8, cvt=5, add=4, max=4, mov=4, swap=4, lds=4, wait=1, dma=24)
D, KS, DT = 128, 8, 4
TILE = 64 * D * 2

# ---- register map ----
SA = [0, 16]
SB = [32, 48]
NM = [64, 80]
P = [96, 104]
KF, VF = 112, 128
PS = [144, 145]
T = [146, 147, 148, 149]
TMP0, TMP1 = 96, 150          # clobbered temporaries [TMP0, TMP1)
KA, VA = 152, 160
AO = lambda h, dt: 16 * (4 * h + dt)           # noqa: E731
AQ = lambda h, ks: 128 + 4 * (8 * h + ks)      # noqa: E731


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


DMA_GAPS = [4 * j + 1 for j in range(8)]      # the 8 gaps that carry the wave's LDS-DMA pieces: every fourth gap


def gen_block(dt, VS, SUB, budget=None):
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    cv = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    SC, SN = (SA, SB) if SUB == 0 else (SB, SA)
    kslot, ksub = (VS, 1) if SUB == 0 else (VS ^ 1, 0)
    koff = kslot * TILE + ksub * 32 * D * 2
    vbase = (2 + VS) * TILE
    lines, lds_seq = [], []

    def k_read(ks):
        lines.append(f"ds_read_b128 {vr(KF + 4 * (ks % AHEAD), 4)}, {vr(KA + ks)} offset:{koff}")
        lds_seq.append((("K", ks), 1))

    def v_read(p):
        k2, d = divmod(p, DT)
        off = vbase + (2 * SUB + k2) * 16 * D * 2
        b = VF + 4 * (p % AHEAD)
        lines.append(f"ds_read_b64_tr_b16 {vr(b, 2)}, {vr(VA + 2 * d)} offset:{off}")
        lines.append(f"ds_read_b64_tr_b16 {vr(b + 2, 2)}, {vr(VA + 2 * d + 1)} offset:{off}")
        lds_seq.append((("V", p), 2))

    def wait_for(tag):
        idx = [i for i, (t, _) in enumerate(lds_seq) if t == tag][-1]
        after = sum(c for _, c in lds_seq[idx + 1:])
        lines.append(f"s_waitcnt lgkmcnt({after})")

    # ---- vector work: an exponential stream and a stream of everything else, each item with what it waits for ----
    # exponentials in groups of four (both halves of two adjacent scores): after group g the packing of P word (g & 3) of
    # key half (g >> 2) and the row-sum adds of those scores are ready
    EXP = []
    for g in range(8):
        for h in (0, 1):
            EXP += [(h, 2 * g), (h, 2 * g + 1)]
    pos = {e: n for n, e in enumerate(EXP)}
    others = []        # (kind, text, exponentials that must have been issued, earliest gap, deadline MFMA or None)
    for k2 in (0, 1):
        for j in range(4):
            for h in (0, 1):
                need = max(pos[(h, 8 * k2 + 2 * j)], pos[(h, 8 * k2 + 2 * j + 1)]) + 1
                others.append(("cvt", f"{cv} {vr(P[h] + 4 * k2 + j)}, {vr(SC[h] + 8 * k2 + 2 * j)}, {vr(SC[h] + 8 * k2 + 2 * j + 1)}",
                               need, -1, 16 + 8 * k2))
    adds = []
    for h in (0, 1):
        adds.append([("add", f"v_add_f32 {vr(PS[h])}, {vr(SC[h])}, {vr(SC[h] + 1)}", pos[(h, 1)] + 1, -1, None)]
                    + [("add", f"v_add_f32 {vr(PS[h])}, {vr(PS[h])}, {vr(SC[h] + i)}", pos[(h, i)] + 1, -1, None) for i in range(2, 16)]
                    + [("add", f"v_add_f32 %[l{h}], %[l{h}], {vr(PS[h])}", 32, -1, None)])
    addq = [x for pair in zip(*adds) for x in pair]      # the two halves' chains interleaved
    E = 17             # S_nxt[1]'s last MFMA is MFMA 15: its readers sit two or more MFMAs behind it
    maxq = []
    for h in (0, 1):
        maxq += [("max", f"v_max_f32 {vr(T[2 * h])}, {vr(SN[h])}, {vr(SN[h] + 1)}", 0, E, None)]
    for h in (0, 1):
        maxq += [("max", f"v_max_f32 {vr(T[2 * h + 1])}, {vr(SN[h] + 2)}, {vr(SN[h] + 3)}", 0, E, None)]
    for i in range(2, 8):
        for h in (0, 1):
            t = T[2 * h + (i & 1)]
            maxq += [("max", f"v_max3_f32 {vr(t)}, {vr(t)}, {vr(SN[h] + 2 * i)}, {vr(SN[h] + 2 * i + 1)}", 0, E, None)]
    maxq += [("max", f"v_max_f32 {vr(T[2 * h])}, {vr(T[2 * h])}, {vr(T[2 * h + 1])}", 0, E, None) for h in (0, 1)]
    maxq += [("mov", f"v_mov_b32 {vr(T[2 * h + 1])}, {vr(T[2 * h])}", 0, E, None) for h in (0, 1)]
    # v_permlane32_swap: 2 wait states behind the VALU write of either operand and in front of a reader of its results: the
    # other half's mov / swap and one s_nop each way ("tail": emitted as one unit, never split by other vector work)
    tail = ["s_nop 0", f"v_permlane32_swap_b32 {vr(T[0])}, {vr(T[1])}", f"v_permlane32_swap_b32 {vr(T[2])}, {vr(T[3])}", "s_nop 0",
            f"v_max_f32 %[mx0], {vr(T[0])}, {vr(T[1])}", f"v_max_f32 %[mx1], {vr(T[2])}, {vr(T[3])}"]

    dgaps = DMA_GAPS
    pre = 64
    ei = 0             # exponentials issued
    cvq = list(others)
    last_exp_line = -10

    def emit_slot(cycles, gap, next_mfma, final=False):
        """Fill one slot: up to two exponentials while there are any, then whatever is ready, by issue cost."""
        nonlocal ei, last_exp_line
        used = 0
        nexp = 0
        progress = True
        while progress and (used < cycles or final):
            progress = False
            # 1. packing whose inputs are ready (deadline work)
            if cvq and cvq[0][2] <= ei and len(lines) - last_exp_line >= 1 + (1 if cvq[0][2] == ei else 0):
                k, t, need, e, dl = cvq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            # 2. an exponential (at most two per slot outside the prologue region)
            if ei < len(EXP) and (nexp < 2 or gap < 0 or final):
                h, i = EXP[ei]
                lines.append(f"v_exp_f32 {vr(SC[h] + i)}, {vr(SC[h] + i)}")
                last_exp_line = len(lines) - 1
                ei += 1; nexp += 1; used += COST["exp"]; progress = True
                continue
            # 3. row sums
            if addq and addq[0][2] <= ei and len(lines) - last_exp_line >= 2:
                k, t, need, e, dl = addq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            # 4. row maxima of S_nxt
            if maxq and maxq[0][3] <= gap:
                k, t, need, e, dl = maxq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            if not maxq and not addq and not cvq and ei == len(EXP) and tail and gap >= E:
                lines.extend(tail); used += 6 * 4; tail.clear(); progress = True
        return used

    dma_j = 0

    def dma_piece():
        nonlocal dma_j
        # skipped as a whole when this sub-step stages nothing from inside the block (vcc = 0, set at the head)
        lines.append(f"s_cbranch_vccz .Lk5w_%=_{dma_j}")
        # M0 <- LDS destination of this piece; the SALU add between the M0 write and the load is the required wait state
        lines.append("s_mov_b32 m0, s92")
        lines.append("s_add_u32 s92, s92, 2048")
        lines.append(f"global_load_lds_dwordx4 %[vo{dma_j & 1}], s[90:91]")
        lines.append("s_add_u32 s90, s90, %[st]")
        lines.append("s_addc_u32 s91, s91, 0")
        lines.append(f".Lk5w_%=_{dma_j}:")
        dma_j += 1

    lines.append("s_cmp_lg_u32 %[dm], 0")
    lines.append("s_cselect_b64 vcc, -1, 0")
    lines.append("s_mov_b32 s90, %[glo]")
    lines.append("s_mov_b32 s91, %[ghi]")
    lines.append("s_mov_b32 s92, %[ld]")
    for ks in range(AHEAD):
        k_read(ks)
    emit_slot(pre, -1, 0)
    usage = []
    for i in range(32):
        if i < 16:
            ks, h = divmod(i, 2)
            if h == 0:
                wait_for(("K", ks))
            c = vr(NM[h], 16) if ks == 0 else vr(SN[h], 16)
            lines.append(f"{mf} {vr(SN[h], 16)}, {vr(KF + 4 * (ks % AHEAD), 4)}, {ar(AQ(h, ks), 4)}, {c}")
            fixed = 0
            if h == 1 and ks + AHEAD < KS:
                k_read(ks + AHEAD); fixed += COST["lds"]
            if h == 0 and ks >= 4:           # V^T fragments 0..3 ride the last QK^T shadows (gaps 8, 10, 12, 14)
                v_read(ks - 4); fixed += 2 * COST["lds"]
        else:
            p, h = divmod(i - 16, 2)
            k2, d = divmod(p, DT)
            if h == 0:
                text = "\n".join(lines)
                for hh in (0, 1):
                    for jj in range(4):
                        assert f"{cv} {vr(P[hh] + 4 * k2 + jj)}," in text, (dt, VS, SUB, "P not packed before PV", p)
                wait_for(("V", p))
            lines.append(f"{mf} {ar(AO(h, d), 16)}, {vr(VF + 4 * (p % AHEAD), 4)}, {vr(P[h] + 4 * k2, 4)}, {ar(AO(h, d), 16)}")
            fixed = 0
            if h == 1 and p + AHEAD < 2 * DT:
                v_read(p + AHEAD); fixed += 2 * COST["lds"]
        if i in dgaps:
            dma_piece(); fixed += COST["dma"]
        usage.append(fixed + emit_slot(budget - fixed if budget else 24 - fixed, i, i + 1, final=(i == 31)))
    assert ei == len(EXP) and not cvq and not addq and not maxq and not tail, (dt, VS, SUB, "vector work left over")
    assert dma_j == len(dgaps)
    if STATS is not None:
        STATS.append((dt, VS, SUB, usage))
    return lines


STATS = None


def rowmax_lines(S):
    """mx[h] = row maximum of the 32 x 32 score tile S[h] (both halves), out of the pipelined block."""
    lines = []
    for h in (0, 1):
        lines.append(f"v_max_f32 {vr(T[2 * h])}, {vr(S[h])}, {vr(S[h] + 1)}")
    for i in range(1, 8):
        for h in (0, 1):
            lines.append(f"v_max3_f32 {vr(T[2 * h])}, {vr(T[2 * h])}, {vr(S[h] + 2 * i)}, {vr(S[h] + 2 * i + 1)}")
    for h in (0, 1):
        lines.append(f"v_mov_b32 {vr(T[2 * h + 1])}, {vr(T[2 * h])}")
    lines.append("s_nop 1")
    lines.append(f"v_permlane32_swap_b32 {vr(T[0])}, {vr(T[1])}")
    lines.append(f"v_permlane32_swap_b32 {vr(T[2])}, {vr(T[3])}")
    lines.append("s_nop 1")
    lines.append(f"v_max_f32 %[mx0], {vr(T[0])}, {vr(T[1])}")
    lines.append(f"v_max_f32 %[mx1], {vr(T[2])}, {vr(T[3])}")
    return lines


def gen_qk0(dt):
    """Prologue: S_A[h] = K(slot 0, sub-tile 0) . Q[h]^T - m[h], row maxima -- the block's first half without a softmax."""
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    SN = SA
    lines = []
    for ks in range(KS):
        lines.append(f"ds_read_b128 {vr(KF + 4 * (ks % AHEAD), 4)}, {vr(KA + ks)}" if ks < AHEAD else None)
    lines = [l for l in lines if l]
    for ks in range(KS):
        # reads outstanding behind fragment ks when it is needed: those issued after it so far
        lines.append(f"s_waitcnt lgkmcnt({min(AHEAD - 1, KS - 1 - ks)})")
        for h in (0, 1):
            c = vr(NM[h], 16) if ks == 0 else vr(SN[h], 16)
            lines.append(f"{mf} {vr(SN[h], 16)}, {vr(KF + 4 * (ks % AHEAD), 4)}, {ar(AQ(h, ks), 4)}, {c}")
        if ks + AHEAD < KS:
            lines.append(f"ds_read_b128 {vr(KF + 4 * (ks % AHEAD), 4)}, {vr(KA + ks + AHEAD)}")
    lines.append("s_nop 15")     # the last MFMA's passes (8 + margin) before the maxima read S
    lines.append("s_nop 3")
    lines += rowmax_lines(SN)
    return lines


def c_string(lines):
    return " \\\n".join(f'    "{l}\\n\\t"' for l in lines)


def main():
    out = ["// GENERATED by gen_k5_block64.py -- do not edit; edit the generator (its docstring says what this is).", "#pragma once", ""]
    for dt in ("bf16", "f16"):
        for VS in (0, 1):
            for SUB in (0, 1):
                out.append(f"#define RSA_K5W_BLOCK_{dt.upper()}_V{VS}_S{SUB} \\")
                out.append(c_string(gen_block(dt, VS, SUB)))
                out.append("")
        out.append(f"#define RSA_K5W_QK0_{dt.upper()} \\")
        out.append(c_string(gen_qk0(dt)))
        out.append("")
    # rare paths on the pinned arch registers, as asm as well (the compiler never computes on S / -m: it then keeps every
    # pinned tile in place between statements instead of shuffling 16-register tuples around the blocks)
    for nmx, S in (("A", SA), ("B", SB)):
        out.append(f"#define RSA_K5W_ROWMAX_{nmx} \\")
        out.append(c_string(rowmax_lines(S)))
        out.append("")
        # boundary mask: score i of lane half hh is key kfirst + 4 hh + (i & 3) + 8 (i >> 2); kept iff lo <= key < hi, tested
        # as (key - lo) <u (hi - lo): %[kb0/1] = kfirst + 4 hh - lo[h], %[sp0/1] = hi[h] - lo[h] (0 when the range is empty)
        lines = []
        for h in (0, 1):
            for i in range(16):
                off = (i & 3) + 8 * (i >> 2)
                lines.append(f"v_add_u32 {vr(T[0])}, {off}, %[kb{h}]")
                lines.append(f"v_cmp_gt_u32 vcc, %[sp{h}], {vr(T[0])}")
                lines.append(f"v_cndmask_b32 {vr(S[h] + i)}, %[ninf], {vr(S[h] + i)}, vcc")
        out.append(f"#define RSA_K5W_MASK_{nmx} \\")
        out.append(c_string(lines))
        out.append("")
        # deferred rescale of both halves: O[h] *= al[h], S[h] -= de[h], -m[h] = ng[h] (a half that does not move gets 1, 0
        # and its old -m: exact no-ops)
        lines = ["s_nop 11"]     # the last PV MFMA of the preceding block wrote O: 12 wait states before it is read
        for h in (0, 1):
            for g in range(0, 64, 8):
                base = AO(h, 0) + g
                lines += [f"v_accvgpr_read_b32 {vr(TMP0 + j)}, {ar(base + j)}" for j in range(8)]
                lines += [f"v_mul_f32 {vr(TMP0 + j)}, {vr(TMP0 + j)}, %[al{h}]" for j in range(8)]
                lines += [f"v_accvgpr_write_b32 {ar(base + j)}, {vr(TMP0 + j)}" for j in range(8)]
            lines += [f"v_sub_f32 {vr(S[h] + i)}, {vr(S[h] + i)}, %[de{h}]" for i in range(16)]
            lines += [f"v_mov_b32 {vr(NM[h] + i)}, %[ng{h}]" for i in range(16)]
        out.append(f"#define RSA_K5W_RESCALE_{nmx} \\")
        out.append(c_string(lines))
        out.append("")
    out.append("#define RSA_K5W_NMZERO \\")
    out.append(c_string([f"v_mov_b32 {vr(NM[0] + i)}, 0" for i in range(32)]))
    out.append("")
    # accumulator-file housekeeping: zero O, write one Q fragment, read one O tile
    out.append("#define RSA_K5W_OZERO \\")
    out.append(c_string([f"v_accvgpr_write_b32 {ar(i)}, 0" for i in range(128)]))
    out.append("")
    for h in (0, 1):
        for ks in range(KS):
            out.append(f"#define RSA_K5W_QWRITE_H{h}_K{ks} \\")
            out.append(c_string([f"v_accvgpr_write_b32 {ar(AQ(h, ks) + j)}, {vr(TMP0 + j)}" for j in range(4)]))
            out.append("")
    for h in (0, 1):
        for d in range(DT):
            out.append(f"#define RSA_K5W_OREAD_H{h}_D{d} \\")
            out.append(c_string([f"v_accvgpr_read_b32 {vr(TMP0 + j)}, {ar(AO(h, d) + j)}" for j in range(16)]))
            out.append("")
    # operand lists
    outs = [f'"+{{{vr(SA[h], 16)}}}"(SA[{h}])' for h in (0, 1)] + [f'"+{{{vr(SB[h], 16)}}}"(SB[{h}])' for h in (0, 1)]
    outs += ['[l0] "+v"(l[0])', '[l1] "+v"(l[1])', '[mx0] "=&v"(mx[0])', '[mx1] "=&v"(mx[1])']
    ins = [f'"{{{vr(NM[h], 16)}}}"(nm[{h}])' for h in (0, 1)] + [f'"{{{vr(KA, 8)}}}"(ka)', f'"{{{vr(VA, 8)}}}"(va)']
    dma = ['[dm] "s"(dm)', '[glo] "s"(glo)', '[ghi] "s"(ghi)', '[ld] "s"(ldst)', '[st] "s"(gstep)', '[vo0] "v"(vo0)', '[vo1] "v"(vo1)']
    out.append(f"#define RSA_K5W_OPS : {', '.join(outs)} : {', '.join(ins + dma)}")
    outs0 = [f'"+{{{vr(SA[h], 16)}}}"(SA[{h}])' for h in (0, 1)] + ['[mx0] "=&v"(mx[0])', '[mx1] "=&v"(mx[1])']
    out.append(f"#define RSA_K5W_OPS_QK0 : {', '.join(outs0)} : {', '.join(ins[:3])}")
    for nmx, S in (("A", SA), ("B", SB)):
        so = [f'"+{{{vr(S[h], 16)}}}"(S{nmx}[{h}])' for h in (0, 1)]
        out.append(f"#define RSA_K5W_OPS_ROWMAX_{nmx} : {', '.join(so)}, [mx0] \"=&v\"(mx[0]), [mx1] \"=&v\"(mx[1]) :")
        out.append(f"#define RSA_K5W_OPS_MASK_{nmx} : {', '.join(so)} : [kb0] \"v\"(kb0), [kb1] \"v\"(kb1), [sp0] \"v\"(sp0), "
                   f"[sp1] \"v\"(sp1), [ninf] \"v\"(ninf)")
        no = [f'"+{{{vr(NM[h], 16)}}}"(nm[{h}])' for h in (0, 1)]
        out.append(f"#define RSA_K5W_OPS_RESCALE_{nmx} : {', '.join(so + no)} : [al0] \"v\"(al0), [al1] \"v\"(al1), "
                   f"[de0] \"v\"(de0), [de1] \"v\"(de1), [ng0] \"v\"(ng0), [ng1] \"v\"(ng1)")
    out.append(f"#define RSA_K5W_OPS_NMZERO : \"={{{vr(NM[0], 16)}}}\"(nm[0]), \"={{{vr(NM[1], 16)}}}\"(nm[1])")
    tmp = ", ".join(f'"v{r}"' for r in range(TMP0, TMP1))
    acc_o = ", ".join(f'"a{r}"' for r in range(128))
    acc_q = ", ".join(f'"a{r}"' for r in range(128, 192))
    out.append(f"#define RSA_K5W_CLOBBER_TMP {tmp}")
    out.append(f"#define RSA_K5W_CLOBBER_O {acc_o}")
    out.append(f"#define RSA_K5W_CLOBBER_Q {acc_q}")
    out.append('#define RSA_K5W_CLOBBER_DMA "s90", "s91", "s92", "vcc"')
    out.append(f"// O a[0:127], Q a[128:191]; SA v[0:31], SB v[32:63], -m v[64:95], temporaries v[{TMP0}:{TMP1 - 1}] "
               f"(P v[96:111], K ring v[112:127], V ring v[128:143]), K addresses v[{KA}:{KA + 7}], V addresses v[{VA}:{VA + 7}]")
    print("\n".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stats":
        STATS = []
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            main()
        for dt, VS, SUB, usage in STATS:
            if dt == "bf16" and VS == 0:
                print(f"S{SUB}: per-gap issue cost {usage}  total {sum(usage)}")
    else:
        main()
