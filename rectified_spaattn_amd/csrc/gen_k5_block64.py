#!/usr/bin/env python3
# (round 6: head dim 64 beside 128 -- RSA_K5V_* beside RSA_K5W_*, configure(d) below; the docstring describes head dim 128)
"""Generator of the 64-rows-per-wave form of K5's main loop (rsa_attn_block64.h), head dim 128.

One wave owns 64 query rows (two 32-row halves h = 0, 1) and the WHOLE register file of its SIMD (one wave per SIMD, 512
registers): every K fragment and every V^T fragment read from LDS feeds TWO MFMAs (one per row half), so the LDS operand
reads per MFMA halve against the 32-row form (24 reads per 32 MFMAs instead of per 16).  One block = one 32-key sub-step u:

    S_nxt[h]^T = K(u+1) . Q[h]^T - m[h]               16 MFMAs   ks = 0..7, h = 0, 1  (A = K rows by ds_read_b128, B = Q[h]
                                                                from the ACCUMULATOR file, C of the first = the -m block)
    P[h]       = exp2(S_cur[h])                       in place, fp32; row sums; packed to the 2-byte type
    O[h]^T    += V(u)^T . P[h]^T                      16 MFMAs   (k2, dt) = 8 V^T fragments x h (A = V^T by ds_read_b64_tr_b16,
                                                                B = packed P[h], C = D = O[h][dt] in the ACCUMULATOR file)
    mx[h]      = row max of S_nxt[h]

Register map.  Accumulator file (owned by the asm statements, never seen by the compiler as values: a "+a" operand makes
hipcc keep the tile in arch VGPRs and copy 128 registers in and out around every statement):
    O[h][dt]  a[16*(4h+dt) : +15]   (a0..a127)         Q[h][ks]  a[128 + 4*(8h+ks) : +3]   (a128..a191)
Arch VGPRs, pinned through physical-register constraints:
    SA[h] v[16h : +15]   SB[h] v[32+16h : +15]   -m[h] v[64+16h : +15]   P[h] v[96+8h : +7]   K ring v[112:127]   V ring v[128:143]
    scratch v144..v149   K read addresses v[152:159]   V read addresses v[160:167]   DMA lane offsets v[168:171] (K), v[172:175] (V)

LDS: K and V each live in a ring of FOUR 32-key half-tiles (8 KiB, image and swizzle of rsa_attn_kernel.hip); half-tile x
sits in slot x & 3, so the slot is a compile-time constant of block U = u & 3.  Block u reads K(u+1) and V(u) and stages
V(u+3) into V(u-1)'s slot and K(u+4) into K(u)'s slot: every LDS-DMA piece (global_load_lds_dwordx4, 1 KiB) has two to three
sub-steps to land, and the wait in front of block u is the constant vmcnt(16) (the 8 + 8 pieces of blocks u-2, u-1 may fly).

Schedule inside a block: the vector work is dealt into the 32 MFMA gaps by issue cost (cdna_hip_programming.md: <= 5
single-issue instructions per v_mfma_f32_32x32x16 gap, costs summing to <= 24 cycles hide) in deadline order: exponentials +
packing of the first 16 keys (needed by PV MFMA 16), of the second 16 keys (MFMA 24), row sums, then the two row maxima
interleaved with each other.  The wave's 8 DMA pieces (4 of K, 4 of V) take every fourth gap: they reach the CU's texture
addresser spread over the sub-step instead of as a burst behind the barrier (the 32-row kernel's staging point costs each wave
~450 cycles for 4 pieces: 8 waves x 4 pieces queue at one addresser, profiles/r03_k5_block.md; four pieces back to back cost
this wave ~25 cycles each, one per four MFMAs ~2 when the lines are in the CU's own cache: tools/probes/dma_issue_probe2.hip --
what a piece costs beyond that is the memory system pushing back, not the instruction).  The source bases WALK in scalar
registers -- a half-tile is 32 consecutive keys, a kept block 128 -- one step per block, re-based once per kept block, all of it
inside MFMA gaps (a scalar instruction beside an MFMA is free, between two blocks it is not: `tight` below); the rows of piece
j come from lane-offset register j, the LDS destination from M0 + instruction offset.  Block alone (no staging, no boundary):
1 131 .. 1 163 cycles per 32 MFMAs (tools/probes/k5w_block_probe.hip), in the loop with its staging ~1 330.

Two products:
  * RSA_K5W_LOOP_*: the steady-state loop as ONE asm statement: per kept 128-key block four blocks (U = 0..3), each behind
    `s_waitcnt vmcnt(16)` + `s_barrier`; list entry of the block after next read from LDS in the shadow, scalar re-basing of
    the DMA walkers, the deferred-rescale test after every block with the rescale itself out of line.  With one wave per SIMD
    nothing overlaps the instructions BETWEEN blocks (the first form of this kernel spent ~300 cycles per sub-step in hipcc's
    glue: stamps in profiles/r04_k5_w64.md), so the steady state contains none.
  * RSA_K5W_BLOCK_*_U{0..3}: the same block without DMA as a statement of its own, for the sub-steps the loop does not take
    (boundary blocks: masked scores, clamped rows -- staged from C++), plus the rare-path helpers (mask, row maxima, rescale)
    and the accumulator-file housekeeping.  One statement per U: alternative statements that define the same pinned tiles make
    hipcc copy the tiles around every one of them.

usage: python3 gen_k5_block64.py > rsa_attn_block64.h        (python3 gen_k5_block64.py stats: per-gap issue costs)
"""
import os
import sys

AHEAD = 4
COST = dict(exp=8, cvt=5, add=4, max=4, mov=4, swap=4, lds=4, wait=1, dma=24)   # (dma: the piece keeps its gap to itself; pricing it at 4 changes nothing measurable: form 7)
D, KS, DT = 128, 8, 4
HALF = 32 * D * 2             # bytes of a 32-key half-tile
VRING = 4 * HALF              # LDS offset of the V ring
NQK, NPV, NMF = 2 * KS, 4 * DT, 2 * KS + 4 * DT      # MFMAs of a block: scores, P . V, all
PFX = "K5W"                   # macro prefix: K5W = head dim 128, K5V = head dim 64


def configure(d):
    """Head dim of the streams generated from here on (round 6: 64 beside 128).  Head dim 64: 8 + 8 MFMAs per 32-key sub-step against
    the same softmax; a half-tile is 4 KiB = 4 LDS-DMA pieces (2 per wave), so the loop issues 2 + 2 pieces per sub-step."""
    global D, KS, DT, HALF, VRING, NQK, NPV, NMF, PFX, DMA_GAPS, SALU_AT, VMC, GAPC
    D, KS, DT = d, d // 16, d // 32
    HALF = 32 * D * 2
    VRING = 4 * HALF
    NQK, NPV, NMF = 2 * KS, 4 * DT, 2 * KS + 4 * DT
    PFX = "K5W" if d == 128 else "K5V"
    if d == 128:
        DMA_GAPS = [4 * j + 1 for j in range(8)]
        SALU_AT = dict(k=22, k3=(22, 23, 24), v=30, h1=(2, 3))
        VMC, GAPC = 16, 24
    else:
        DMA_GAPS = [1, 3, 9, 11]        # K0 K1 | V0 V1: the wave's 2 + 2 pieces of the sub-step
        SALU_AT = dict(k=4, k3=(4, 5, 6), v=12, h1=(2, 3))
        VMC, GAPC = 8, int(os.environ.get("RSA_GEN64_GAPC", "36"))

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
VOK, VOV = 168, 172           # per-lane DMA source offsets of the wave's four K pieces / four V pieces of a half-tile
# scalar registers owned by the loop statement
S_KB, S_KL, S_VB, S_VL, S_BLK, S_CNT, S_T0, S_T2, S_K128, S_V128, S_KST, S_VST = 80, 82, 84, 86, 87, 88, 90, 92, 94, 95, 96, 97
S_CLOB = list(range(80, 98))
# schedule of the online body (round 6, profiles/r06_k5_forms.txt form 23: byte-identical outputs, -0.4 % R2 / -3 % dense): the row
# maxima go ahead of the row sums once S_nxt is complete, so the deferred-rescale test's two compares ride inside the block and
# only `s_or_b64` + the branch sit between two blocks
LOOP_XF = frozenset({"maxfirst", "earlytest"})
LOOP_PRE = 24                 # issue cycles of vector work the loop's blocks put in front of their first MFMA (the boundary blocks: 64)
DMA_GAPS = [4 * j + 1 for j in range(8)]      # the 8 gaps that carry the wave's LDS-DMA pieces: every fourth gap
SALU_AT = dict(k=22, k3=(22, 23, 24), v=30, h1=(2, 3))     # gaps of the loop's scalar bookkeeping (gen_loop, tight)
VMC = 16                      # LDS-DMA pieces of the last two sub-steps that may still fly at a sub-step boundary
GAPC = 24                     # issue cycles of vector work dealt into one MFMA gap (head dim 64: 36 -- 16 MFMAs per sub-step cannot hide
                              # the softmax of 64 x 32 scores, the stream is paced by the vector port there)
STATS = None


def AO(h, dt):
    return 16 * (DT * h + dt)


def AQ(h, ks):
    return 32 * DT + 4 * (KS * h + ks)


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


def sr(a, n=1):
    return f"s{a}" if n == 1 else f"s[{a}:{a + n - 1}]"


def gen_block(dt, U, dma, chain=None, xf=frozenset(), dma_cost=None, pre=64, salu=None):
    """Block U.  xf: experiment flags of the A/B forms (gen_forms; timing only, the results of most are garbage).  Block U.  dma: issue the wave's pieces from the scalar walkers.  chain (the loop): the block does not open with its K
    reads -- the previous block issued them in its tail -- and issues the NEXT block's first four K reads itself, behind the
    lines of `chain` (the sub-step boundary: vmcnt wait, barrier, ...), which sit in front of MFMA 30: behind the wait for the
    last V^T fragment, so every LDS read of this sub-step has returned when the barrier releases the slots."""
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    cv = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    SC, SN = (SA, SB) if U % 2 == 0 else (SB, SA)
    koff = ((U + 1) & 3) * HALF
    voff = VRING + U * HALF
    lines, lds_seq = [], []

    def k_read(ks):
        if "nolds" in xf:
            return
        lines.append(f"ds_read_b128 {vr(KF + 4 * (ks % AHEAD), 4)}, {vr(KA + ks)} offset:{koff}")
        lds_seq.append((("K", ks), 1))

    def v_read(p):
        if "nolds" in xf:
            return
        k2, d = divmod(p, DT)
        off = voff + k2 * 16 * D * 2
        b = VF + 4 * (p % AHEAD)
        lines.append(f"ds_read_b64_tr_b16 {vr(b, 2)}, {vr(VA + 2 * d)} offset:{off}")
        lines.append(f"ds_read_b64_tr_b16 {vr(b + 2, 2)}, {vr(VA + 2 * d + 1)} offset:{off}")
        lds_seq.append((("V", p), 2))

    def wait_for(tag):
        if "nolds" in xf:
            return
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
    cvq = []           # (kind, text, exponentials that must have been issued)
    for k2 in (0, 1):
        for j in range(4):
            for h in (0, 1):
                need = max(pos[(h, 8 * k2 + 2 * j)], pos[(h, 8 * k2 + 2 * j + 1)]) + 1
                cvq.append(("cvt", f"{cv} {vr(P[h] + 4 * k2 + j)}, {vr(SC[h] + 8 * k2 + 2 * j)}, {vr(SC[h] + 8 * k2 + 2 * j + 1)}", need))
    adds = []
    for h in (0, 1):
        adds.append([("add", f"v_add_f32 {vr(PS[h])}, {vr(SC[h])}, {vr(SC[h] + 1)}", pos[(h, 1)] + 1)]
                    + [("add", f"v_add_f32 {vr(PS[h])}, {vr(PS[h])}, {vr(SC[h] + i)}", pos[(h, i)] + 1) for i in range(2, 16)]
                    + [("add", f"v_add_f32 %[l{h}], %[l{h}], {vr(PS[h])}", 32)])
    addq = [x for pair in zip(*adds) for x in pair]      # the two halves' chains interleaved
    E = NQK + 1        # S_nxt[1]'s last MFMA is MFMA NQK - 1: its readers sit two or more MFMAs behind it
    maxq = []
    for h in (0, 1):
        maxq += [("max", f"v_max_f32 {vr(T[2 * h])}, {vr(SN[h])}, {vr(SN[h] + 1)}")]
    for h in (0, 1):
        maxq += [("max", f"v_max_f32 {vr(T[2 * h + 1])}, {vr(SN[h] + 2)}, {vr(SN[h] + 3)}")]
    for i in range(2, 8):
        for h in (0, 1):
            t = T[2 * h + (i & 1)]
            maxq += [("max", f"v_max3_f32 {vr(t)}, {vr(t)}, {vr(SN[h] + 2 * i)}, {vr(SN[h] + 2 * i + 1)}")]
    maxq += [("max", f"v_max_f32 {vr(T[2 * h])}, {vr(T[2 * h])}, {vr(T[2 * h + 1])}") for h in (0, 1)]
    maxq += [("mov", f"v_mov_b32 {vr(T[2 * h + 1])}, {vr(T[2 * h])}") for h in (0, 1)]
    # v_permlane32_swap: 2 wait states behind the VALU write of either operand and in front of a reader of its results: the
    # other half's mov / swap and one s_nop each way ("tail": emitted as one unit, never split by other vector work)
    tail = ["s_nop 0", f"v_permlane32_swap_b32 {vr(T[0])}, {vr(T[1])}", f"v_permlane32_swap_b32 {vr(T[2])}, {vr(T[3])}", "s_nop 0",
            f"v_max_f32 %[mx0], {vr(T[0])}, {vr(T[1])}", f"v_max_f32 %[mx1], {vr(T[2])}, {vr(T[3])}"]

    if "earlytest" in xf:      # the deferred-rescale test's compares ride in the block, behind the row maxima (the loop keeps the branch)
        tail = tail + ["v_cmp_gt_f32 vcc, %[mx0], %[th0]", f"v_cmp_gt_f32 {sr(S_T2, 2)}, %[mx1], %[th1]"]
    if "max8" in xf:           # one max3 chain per half: 8 instructions instead of 10
        maxq = []
        for h in (0, 1):
            maxq += [("max", f"v_max3_f32 {vr(T[2 * h])}, {vr(SN[h])}, {vr(SN[h] + 1)}, {vr(SN[h] + 2)}")]
        for i in range(6):
            for h in (0, 1):
                maxq += [("max", f"v_max3_f32 {vr(T[2 * h])}, {vr(T[2 * h])}, {vr(SN[h] + 3 + 2 * i)}, {vr(SN[h] + 4 + 2 * i)}")]
        maxq += [("max", f"v_max_f32 {vr(T[2 * h])}, {vr(T[2 * h])}, {vr(SN[h] + 15)}") for h in (0, 1)]
        maxq += [("mov", f"v_mov_b32 {vr(T[2 * h + 1])}, {vr(T[2 * h])}") for h in (0, 1)]
    if "novalu" in xf:
        EXP, cvq, addq, maxq, tail = [], [], [], [], []
    if "noadd" in xf:
        addq = []
    if "nomax" in xf:
        maxq, tail = [], []
    if "nocvt" in xf:
        cvq = []
    if "noexp" in xf:
        EXP = []
        cvq = [(k, t, 0) for k, t, _ in cvq]
        addq = [(k, t, 0) for k, t, _ in addq]
    dcost = (COST["dma"] if D == 128 else 4) if dma_cost is None else dma_cost
    dgaps = DMA_GAPS if dma else []
    salu = salu or {}
    ei = 0             # exponentials issued
    last_exp_line = -10
    exp_line = {}      # exponential n -> index of its line
    cvtlag = next((int(f[6:]) for f in xf if f.startswith("cvtlag")), 0)
    maxfirst = "maxfirst" in xf

    def emit_slot(cycles, gap, final=False):
        """Fill one slot: up to two exponentials while there are any, then whatever is ready, by issue cost."""
        nonlocal ei, last_exp_line
        used = 0
        nexp = 0
        progress = True
        while progress and (used < cycles or final):
            progress = False
            if cvq and cvq[0][2] <= ei and (len(lines) - exp_line.get(cvq[0][2] - 1, -99) >= cvtlag if cvtlag
                                             else len(lines) - last_exp_line >= 1 + (1 if cvq[0][2] == ei else 0)):
                k, t, need = cvq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            if ei < len(EXP) and (nexp < (2 if D == 128 else int(os.environ.get("RSA_GEN64_NEXP", "4"))) or gap < 0 or final):
                h, i = EXP[ei]
                lines.append(f"v_exp_f32 {vr(SC[h] + i)}, {vr(SC[h] + i)}")
                last_exp_line = len(lines) - 1
                exp_line[ei] = last_exp_line
                ei += 1; nexp += 1; used += COST["exp"]; progress = True
                continue
            if maxfirst and maxq and gap >= E:
                k, t = maxq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            if maxfirst and not maxq and tail and gap >= E:
                lines.extend(tail); used += len(tail) * 4; tail.clear(); progress = True
                continue
            if addq and addq[0][2] <= ei and len(lines) - last_exp_line >= 2:
                k, t, need = addq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            if maxq and gap >= E:
                k, t = maxq.pop(0)
                lines.append(t); used += COST[k]; progress = True
                continue
            if not maxq and not addq and not cvq and ei == len(EXP) and tail and gap >= E:
                lines.extend(tail); used += 6 * 4; tail.clear(); progress = True
        return used

    dma_j = 0

    def dma_piece():
        """Pieces in the order K0 K1 V0 V1 K2 K3 V2 V3.  One M0 value (the LDS destination) serves two pieces: the second one
        carries the instruction offset 2048, which moves BOTH its LDS destination and its source -- its lane offset register
        has the 2048 subtracted.  The source base is the half-tile's row 0 (re-based once per block); lane offset register j
        holds the rows of piece j.  One SALU instruction per two pieces instead of four per piece."""
        nonlocal dma_j
        if "nodma" in xf or ("halfdma" in xf and (dma_j & 1)):
            dma_j += 1
            return
        pair, sub = divmod(dma_j, 2)
        isv, hi = pair & 1, pair >> 1
        base, ldsw, vo = (S_VB, S_VL, VOV) if isv else (S_KB, S_KL, VOK)
        j = 2 * hi + sub
        if sub == 0:
            lines.append(f"s_add_u32 m0, {sr(ldsw)}, {4096 * hi}")
            lines.append("s_nop 0")      # (M0 write -> LDS-DMA: one wait state)
        lines.append(f"global_load_lds_dwordx4 {vr(vo + j)}, {sr(base, 2)}" + (" offset:2048" if sub else ""))
        dma_j += 1

    if chain is None:
        for ks in range(AHEAD):
            k_read(ks)
    else:
        lds_seq.extend(((("K", ks), 1)) for ks in range(AHEAD))      # in flight since the previous block's tail
    emit_slot(pre, -1)
    usage = []
    for i in range(NMF):
        if i < NQK:
            ks, h = divmod(i, 2)
            if h == 0:
                wait_for(("K", ks))
            c = vr(NM[h], 16) if ks == 0 else vr(SN[h], 16)
            if "m16" in xf:      # timing-only: the same FLOPs and operand registers as two 16x16x32 MFMAs (results are garbage)
                c0 = NM[h] if ks == 0 else SN[h]
                for q4 in (0, 4):
                    lines.append(f"v_mfma_f32_16x16x32_bf16 {vr(SN[h] + q4, 4)}, {vr(KF + 4 * (ks % AHEAD), 4)}, {ar(AQ(h, ks), 4)}, {vr(c0 + q4, 4)}")
            else:
                lines.append(f"{mf} {vr(SN[h], 16)}, {vr(KF + 4 * (ks % AHEAD), 4)}, {ar(AQ(h, ks), 4)}, {c}")
            fixed = 0
            if h == 1 and ks + AHEAD < KS:
                k_read(ks + AHEAD); fixed += COST["lds"]
            if h == 0 and ks >= KS - AHEAD:   # V^T fragments 0..3 ride the last QK^T shadows (head dim 128: gaps 8, 10, 12, 14)
                v_read(ks - (KS - AHEAD)); fixed += 2 * COST["lds"]
        else:
            p, h = divmod(i - NQK, 2)
            k2, d = divmod(p, DT)
            if h == 0:
                text = "\n".join(lines)
                for hh in (0, 1):
                    for jj in range(4):
                        assert "novalu" in xf or "nocvt" in xf or f"{cv} {vr(P[hh] + 4 * k2 + jj)}," in text, (dt, U, "P not packed before PV", p)
                wait_for(("V", p))
            if chain is not None and i == NMF - 2:
                lines.extend(chain)
                nk = ((U + 2) & 3) * HALF
                if "nolds" not in xf:
                    lines.extend(f"ds_read_b128 {vr(KF + 4 * ks, 4)}, {vr(KA + ks)} offset:{nk}" for ks in range(AHEAD))
            if "m16" in xf:
                for q4 in (0, 4):
                    lines.append(f"v_mfma_f32_16x16x32_bf16 {ar(AO(h, d) + q4, 4)}, {vr(VF + 4 * (p % AHEAD), 4)}, {vr(P[h] + 4 * k2, 4)}, {ar(AO(h, d) + q4, 4)}")
            else:
                lines.append(f"{mf} {ar(AO(h, d), 16)}, {vr(VF + 4 * (p % AHEAD), 4)}, {vr(P[h] + 4 * k2, 4)}, {ar(AO(h, d), 16)}")
            fixed = 0
            if h == 1 and p + AHEAD < 2 * DT:
                v_read(p + AHEAD); fixed += 2 * COST["lds"]
        lines.extend(salu.get(i, []))      # scalar bookkeeping of the loop riding in this gap (free beside an MFMA)
        if i in dgaps:
            dma_piece(); fixed += dcost
        usage.append(fixed + emit_slot(GAPC - fixed, i, final=(i == NMF - 1)))
    assert ei == len(EXP) and not cvq and not addq and not maxq and not tail, (dt, U, "vector work left over")
    assert dma_j == len(dgaps)
    if STATS is not None:
        STATS.append((dt, U, dma, usage))
    return lines


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
    """Prologue: S_A[h] = K(half-tile 0, slot 0) . Q[h]^T - m[h], row maxima -- the block's first half without a softmax."""
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    SN = SA
    lines = [f"ds_read_b128 {vr(KF + 4 * (ks % AHEAD), 4)}, {vr(KA + ks)}" for ks in range(AHEAD)]
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


def rescale_core(S, al, de, ng):
    """O[h] *= al[h], S[h] -= de[h], -m[h] = ng[h] for both halves (operand names given per half)."""
    lines = ["s_nop 11"]     # the last PV MFMA of the preceding block wrote O: 12 wait states before it is read
    for h in (0, 1):
        for g in range(0, 16 * DT, 8):
            base = AO(h, 0) + g
            lines += [f"v_accvgpr_read_b32 {vr(TMP0 + j)}, {ar(base + j)}" for j in range(8)]
            lines += [f"v_mul_f32 {vr(TMP0 + j)}, {vr(TMP0 + j)}, {al[h]}" for j in range(8)]
            lines += [f"v_accvgpr_write_b32 {ar(base + j)}, {vr(TMP0 + j)}" for j in range(8)]
        lines += [f"v_sub_f32 {vr(S[h] + i)}, {vr(S[h] + i)}, {de[h]}" for i in range(16)]
        lines += [f"v_mov_b32 {vr(NM[h] + i)}, {ng[h]}" for i in range(16)]
    return lines


def rescale_decide():
    """The deferred-rescale decision of rsa_attn_kernel64.hip::half in asm (both halves): from mx[h], thr[h], m_ref[h], l[h] to
    al = v144/v145, de = v146/v147, ng = v148/v149 (and the updated thr, m_ref, l).  A half moves iff ANY of its rows exceeds its
    threshold (wave-uniform, like the C++ side's ballot); first = the row has not seen a finite score yet (thr = -inf)."""
    L = []
    mv, fin = sr(S_T0, 2), sr(S_T2, 2)
    for h in (0, 1):
        al, de, ng = vr(144 + h), vr(146 + h), vr(148 + h)
        mx, th, mr, l = f"%[mx{h}]", f"%[th{h}]", f"%[mr{h}]", f"%[l{h}]"
        L += [f"v_cmp_gt_f32 vcc, {mx}, {th}",                     # rows above their threshold
              "s_cmp_lg_u64 vcc, 0",
              f"s_cselect_b64 {mv}, -1, 0",                        # move: any row of this half (all lanes or none)
              f"v_max_f32 {de}, 0, {mx}",                          # delta = first ? mx : max(mx, 0)
              f"v_cmp_eq_f32 vcc, {th}, %[ninf]",                  # first
              f"v_cndmask_b32 {de}, {de}, {mx}, vcc",
              f"v_cmp_neq_f32 {fin}, {de}, %[ninf]",               # delta finite (not "nothing but masked keys so far")
              f"s_and_b64 {mv}, {mv}, {fin}",                      # lanes that move their reference
              f"v_cndmask_b32 {de}, 0, {de}, {mv}",                # the others: delta = 0
              f"v_exp_f32 {al}, -{de}",
              f"v_cndmask_b32 {th}, {th}, %[eight], {mv}",
              f"v_add_f32 {mr}, {mr}, {de}",
              f"v_cndmask_b32 {al}, {al}, 1.0, vcc",               # first: alpha = 1 (O and l are still zero)
              f"v_xor_b32 {ng}, 0x80000000, {mr}",
              f"v_mul_f32 {l}, {l}, {al}",
              f"v_sub_f32 {mx}, {mx}, {de}"]                      # the row maximum follows its scores to the new reference
    return L


def gen_loop_head0(t0, t1):
    """K walker -> key 0 of the next kept block (S_BLK), K pieces to LDS slot 0.., V pieces to slot 3 (block 0's V(u+3))."""
    return [f"s_mul_i32 {t0}, {sr(S_BLK)}, {sr(S_K128)}", f"s_mul_hi_u32 {t1}, {sr(S_BLK)}, {sr(S_K128)}",
            f"s_mov_b64 {sr(S_KB, 2)}, %[kb]",
            f"s_add_u32 {sr(S_KB)}, {sr(S_KB)}, {t0}", f"s_addc_u32 {sr(S_KB + 1)}, {sr(S_KB + 1)}, {t1}",
            f"s_mov_b32 {sr(S_KL)}, %[ldsk]"]


def gen_loop(dt, diag=False, xf=frozenset(), dma_cost=None, pre=64, tight=False, static=False):
    """The steady-state loop, one asm statement (see the file docstring).  Operands: cnt (kept blocks to process, >= 0),
    blk0 / blk1 (block index of the first one and of its successor), la (VGPR: LDS byte address of the list entry two blocks
    ahead), kb / vb (64-bit bases of this head's K / V), krow / vrow (bytes per key row), ldsk / ldsv (LDS address of the wave's
    first piece in slot 0 of the K / V ring).
    static (round 6): the statement carries a SECOND body, taken when the scalar operand `stat` is non-zero: the same four blocks
    without the row maxima and without the deferred-rescale test -- the walk then keeps the softmax reference it entered the
    loop with (rsa_attn_kernel64.hip: "optimistic static reference"; the kernel checks l and O afterwards and redoes the walk
    through the first body if anything overflowed)."""
    t0, t1 = sr(S_T0), sr(S_T0 + 1)
    L = [f"s_mov_b32 {sr(S_CNT)}, %[cnt]",
         f"s_cmp_eq_u32 {sr(S_CNT)}, 0",
         "s_cbranch_scc1 .Lk5w_done_%=",
         f"s_lshl_b32 {sr(S_K128)}, %[krow], 7", f"s_lshl_b32 {sr(S_V128)}, %[vrow], 7",
         f"s_lshl_b32 {sr(S_KST)}, %[krow], 5", f"s_lshl_b32 {sr(S_VST)}, %[vrow], 5",
         f"s_mov_b32 {sr(S_BLK)}, %[blk1]",
         # V walker: the kept block being processed, its last half-tile (key 96): vb + blk0 * vrow128 + 96 * vrow
         f"s_mul_i32 {t0}, %[blk0], {sr(S_V128)}", f"s_mul_hi_u32 {t1}, %[blk0], {sr(S_V128)}",
         f"s_mov_b64 {sr(S_VB, 2)}, %[vb]",
         f"s_add_u32 {sr(S_VB)}, {sr(S_VB)}, {t0}", f"s_addc_u32 {sr(S_VB + 1)}, {sr(S_VB + 1)}, {t1}",
         f"s_mul_i32 {t0}, %[vrow], 96",
         f"s_add_u32 {sr(S_VB)}, {sr(S_VB)}, {t0}", f"s_addc_u32 {sr(S_VB + 1)}, {sr(S_VB + 1)}, 0"]
    if tight:      # the first iteration's K walker and V slot (every later one is set up in block 3's gaps)
        L += gen_loop_head0(t0, t1) + [f"s_add_u32 {sr(S_VL)}, %[ldsv], {3 * HALF}"]

    def boundary(diag):
        """end of a sub-step: the pieces of two blocks ago have landed, every wave has finished its LDS reads"""
        if not diag:
            return ([] if "novm" in xf else [f"s_waitcnt vmcnt({VMC // 2 if 'halfdma' in xf else VMC})"]) + ([] if "nobar" in xf else ["s_barrier"])
        tm, tt = sr(76, 2), sr(75)      # (not S_T2: the rescale test's second compare mask lives there from the block's tail to its end)
        return [f"s_memtime {tm}", "s_waitcnt lgkmcnt(0)", f"s_mov_b32 {tt}, s76", "s_waitcnt vmcnt(16)",
                f"s_memtime {tm}", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 {tt}, s76, {tt}", f"s_add_u32 s78, s78, {tt}",
                f"s_mov_b32 {tt}, s76", "s_barrier",
                f"s_memtime {tm}", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 {tt}, s76, {tt}", f"s_add_u32 s79, s79, {tt}"]
    # entry: the boundary in front of the first block and its first K reads (slot 1: U = 0 reads K(u+1))
    L += [f"s_waitcnt vmcnt({VMC // 2 if 'halfdma' in xf else VMC})", "s_barrier"] + [f"ds_read_b128 {vr(KF + 4 * ks, 4)}, {vr(KA + ks)} offset:{HALF}" for ks in range(AHEAD)]
    if diag:
        L += ["s_mov_b32 s78, 0", "s_mov_b32 s79, 0"]
    if static:
        L += ["s_cmp_lg_u32 %[stat], 0", "s_cbranch_scc1 .Lk5w_sloop_%="]
    bodies = [(xf, "loop")] + ([(frozenset(xf | {"nomax", "notest"}), "sloop")] if static else [])
    for xf, lname in bodies:
        L += [f".Lk5w_{lname}_%=:"]
        for U in range(4):
            head = []
            if U == 0:     # K walker: the next kept block, key 0; K pieces go to slot 0.., V pieces to slot 3
                head += [f"s_mul_i32 {t0}, {sr(S_BLK)}, {sr(S_K128)}", f"s_mul_hi_u32 {t1}, {sr(S_BLK)}, {sr(S_K128)}",
                         f"s_mov_b64 {sr(S_KB, 2)}, %[kb]",
                         f"s_add_u32 {sr(S_KB)}, {sr(S_KB)}, {t0}", f"s_addc_u32 {sr(S_KB + 1)}, {sr(S_KB + 1)}, {t1}",
                         f"s_mov_b32 {sr(S_KL)}, %[ldsk]", f"s_add_u32 {sr(S_VL)}, %[ldsv], {3 * HALF}"]
            if U == 1:     # V walker: the next kept block, key 0; V pieces to slot 0..
                head += [f"s_mul_i32 {t0}, {sr(S_BLK)}, {sr(S_V128)}", f"s_mul_hi_u32 {t1}, {sr(S_BLK)}, {sr(S_V128)}",
                         f"s_mov_b64 {sr(S_VB, 2)}, %[vb]",
                         f"s_add_u32 {sr(S_VB)}, {sr(S_VB)}, {t0}", f"s_addc_u32 {sr(S_VB + 1)}, {sr(S_VB + 1)}, {t1}",
                         f"s_mov_b32 {sr(S_VL)}, %[ldsv]"]
            tail = boundary(diag)
            if U == 1:     # list entry of the block after next: an LDS read older than every K read of block 2 (made scalar in block 3)
                tail = tail + ["ds_read_u16 %[lv], %[la]"]
            # the DMA walkers step to the next half-tile (re-based at U = 0 / 1 where a new kept block starts)
            kstep = [f"s_add_u32 {sr(S_KB)}, {sr(S_KB)}, {sr(S_KST)}", f"s_addc_u32 {sr(S_KB + 1)}, {sr(S_KB + 1)}, 0",
                     f"s_add_u32 {sr(S_KL)}, {sr(S_KL)}, {HALF}"]
            vstep = [f"s_add_u32 {sr(S_VB)}, {sr(S_VB)}, {sr(S_VST)}", f"s_addc_u32 {sr(S_VB + 1)}, {sr(S_VB + 1)}, 0",
                     f"s_add_u32 {sr(S_VL)}, {sr(S_VL)}, {HALF}"]
            nxt = [f"v_readfirstlane_b32 {sr(S_BLK)}, %[lv]", "v_add_u32 %[la], 2, %[la]"]   # (read two blocks ago) -> the next iteration's block index
            if not tight:
                blk = gen_block(dt, U, True, chain=tail, xf=xf, dma_cost=dma_cost, pre=pre)
                blk = head + blk + kstep[:2] + vstep[:2] + [kstep[2], vstep[2]] + (nxt if U == 3 else [])
            else:
                # the scalar bookkeeping rides in MFMA gaps (a scalar instruction beside an MFMA is free, between two blocks it is not:
                # one wave per SIMD).  The wave's K pieces sit in gaps 1, 5, 17, 21, its V pieces in gaps 9, 13, 25, 29: the K walker
                # steps (U = 3: is re-based on the next kept block) behind gap 21, the V walker steps behind gap 29 -- except in
                # front of block 1, which re-bases it in its own gaps 2..4, ahead of its first V piece.
                salu = {}
                g3, gh = SALU_AT["k3"], SALU_AT["h1"]
                if U == 3:
                    k0 = [l for l in gen_loop_head0(t0, t1)]
                    salu[g3[0]] = nxt[:1] + k0[:2]
                    salu[g3[1]] = k0[2:5]
                    salu[g3[2]] = k0[5:] + nxt[1:]
                else:
                    salu[SALU_AT["k"]] = kstep
                if U == 0:
                    pass                      # (block 1 re-bases the V walker itself)
                else:
                    salu[SALU_AT["v"]] = vstep
                if U == 1:
                    salu[gh[0]] = head[:3]
                    salu[gh[1]] = head[3:]
                blk = gen_block(dt, U, True, chain=tail, xf=xf, dma_cost=dma_cost, pre=pre, salu=salu)
            # deferred-rescale test on the scores the NEXT block consumes (S_nxt of this block)
            tst = ["v_cmp_gt_f32 vcc, %[mx0], %[th0]", f"v_cmp_gt_f32 {sr(S_T2, 2)}, %[mx1], %[th1]",
                   f"s_or_b64 vcc, vcc, {sr(S_T2, 2)}", f"s_cbranch_vccnz .Lk5w_resc{U}_%="]
            if "earlytest" in xf:
                tst = [f"s_or_b64 vcc, vcc, {sr(S_T2, 2)}", f"s_cbranch_vccnz .Lk5w_resc{U}_%="]
            if "notest" in xf:
                tst = []
            L += blk + tst + ([f".Lk5w_back{U}_%=:"] if lname == "loop" else [])
        L += [f"s_sub_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1", f"s_cmp_lg_u32 {sr(S_CNT)}, 0", f"s_cbranch_scc1 .Lk5w_{lname}_%=",
              "s_waitcnt lgkmcnt(0)",       # (the K reads the last block issued for its successor: nothing may land behind the statement)
              "s_branch .Lk5w_done_%="]
    xf = bodies[0][0]
    for U in range(4):
        S = SB if U % 2 == 0 else SA          # S_nxt of block U
        L += [f".Lk5w_resc{U}_%=:"] + (["s_add_u32 s79, s79, 0x1000000"] if diag else []) + rescale_decide()
        L += rescale_core(S, [vr(144), vr(145)], [vr(146), vr(147)], [vr(148), vr(149)])
        L += [f"s_branch .Lk5w_back{U}_%="]
    L += [".Lk5w_done_%=:"]
    if diag:
        L += ["s_mov_b32 %[d0], s78", "s_mov_b32 %[d1], s79"]
    return L


# A/B forms of the loop (librsa_hip_ab.so, tuning key "k5w_form"; rsa_attn_block64_forms.h).  Forms 1-6, 8 REMOVE work to price
# it (their outputs are garbage); form 7 is a candidate schedule with valid results.
FORMS = {
    1: dict(xf={"nodma"}),                                          # no staging instructions
    2: dict(xf={"nobar", "novm"}),                                  # no sub-step boundary (vmcnt wait + barrier)
    3: dict(xf={"noexp"}),                                          # no exponentials
    4: dict(xf={"novalu", "notest"}),                               # no vector work at all
    5: dict(xf={"novalu", "notest", "nodma", "nobar", "novm"}),     # MFMAs + LDS operand reads
    6: dict(xf={"novalu", "notest", "nodma", "nobar", "novm", "nolds"}),   # MFMAs
    7: dict(xf=set(), dma_cost=4),                                  # the pieces priced at 4 cycles: vector work rides in their gaps
    9: dict(xf=set(), pre=0),                                       # no vector work in front of a block's first MFMA
    10: dict(xf=set(), tight=True),                                 # the loop's scalar bookkeeping inside MFMA gaps
    11: dict(xf=set(), tight=True, pre=0),
    12: dict(xf=set(), tight=True, pre=0, dma_cost=4),
    13: dict(xf=set(), tight=True, pre=24),
    8: dict(xf={"nodma", "nobar", "novm"}),                         # MFMAs + LDS reads + all vector work, no memory side
    14: dict(xf={"m16"}, tight=True, pre=24),                       # round 6: every 32x32x16 MFMA as two 16x16x32 (same FLOPs, same operand traffic; garbage results)
    15: dict(xf={"halfdma"}, tight=True, pre=24),                   # round 6: every second LDS-DMA piece dropped (a 256-row workgroup's pieces per wave; garbage results)
    16: dict(xf={"m16", "halfdma"}, tight=True, pre=24),
    17: dict(xf={"m16", "novalu", "notest", "nodma", "nobar", "novm"}, tight=True, pre=24),   # form 5 on the 16x16x32 shape
    22: dict(xf={"cvtlag4"}, tight=True, pre=24),                   # VALID: a packing issues no sooner than 4 lines behind its second exponential
    23: dict(xf={"maxfirst", "earlytest"}, tight=True, pre=24),     # VALID: row maxima before the row sums, the rescale test's compares inside the block
    24: dict(xf={"max8"}, tight=True, pre=24),                      # VALID: one max3 chain per half
    25: dict(xf={"maxfirst", "earlytest", "max8", "cvtlag4"}, tight=True, pre=24),
    26: dict(xf={"maxfirst", "earlytest", "max8"}, tight=True, pre=24),
    # round 6, on top of the static body (= form 27: the online machinery removed; valid on ordinary data, no overflow check)
    27: dict(xf={"nomax", "notest"}, tight=True, pre=24),
    28: dict(xf={"nomax", "notest"}, tight=True, pre=24, dma_cost=4),     # vector work may share a gap with an LDS-DMA piece
    29: dict(xf={"nomax", "notest"}, tight=True, pre=0),                  # no vector work in front of a block's first MFMA
    30: dict(xf={"nomax", "notest"}, tight=True, pre=48),
    31: dict(xf={"nomax", "notest"}, tight=True, pre=24, dma_cost=12),
    18: dict(xf={"noadd"}, tight=True, pre=24),                     # no row-sum additions (what moving them to the matrix pipe could buy at most)
    19: dict(xf={"nomax", "notest"}, tight=True, pre=24),           # no row maxima, no rescale test
    20: dict(xf={"novalu", "notest"}, tight=True, pre=24),          # form 4 on the product's schedule
    21: dict(xf={"nocvt"}, tight=True, pre=24),                     # no packing of P
}


def gen_forms():
    out = ["// GENERATED by gen_k5_block64.py forms -- A/B forms of the 64-row loop (NOT the product).", "#pragma once", ""]
    for n, f in sorted(FORMS.items()):
        out.append(f"#define RSA_K5W_LOOP_BF16_X{n} \\")
        out.append(c_string(gen_loop("bf16", xf=frozenset(f["xf"]), dma_cost=f.get("dma_cost"), pre=f.get("pre", 64), tight=f.get("tight", False))))
        out.append("")
    print("\n".join(out))


def c_string(lines):
    return " \\\n".join(f'    "{l}\\n\\t"' for l in lines)


def main():
    print("// GENERATED by gen_k5_block64.py -- do not edit; edit the generator (its docstring says what this is).\n#pragma once\n")
    for d in (128, 64):      # RSA_K5W_* = head dim 128, RSA_K5V_* = head dim 64 (the same streams on 8 + 8 MFMAs per sub-step)
        configure(d)
        text = main_one()
        print(text if d == 128 else text.replace("RSA_K5W_", "RSA_K5V_"))
    configure(128)


def main_one():
    out = []
    for dt in ("bf16", "f16"):
        for U in range(4):
            out.append(f"#define RSA_K5W_BLOCK_{dt.upper()}_U{U} \\")
            out.append(c_string(gen_block(dt, U, False)))
            out.append("")
        out.append(f"#define RSA_K5W_LOOP_{dt.upper()} \\")
        out.append(c_string(gen_loop(dt, pre=LOOP_PRE, tight=True, static=(dt == "bf16"), xf=LOOP_XF)))   # (fp16 P overflows at 2^16: no static body)
        out.append("")
        # the 256-row dense form (four waves on one K/V ring): the same loop with every second LDS-DMA piece dropped -- each wave stages
        # 2 + 2 of a half-tile's 8 + 8 pieces (lane offset registers 0 and 2), vmcnt(8) at the sub-step boundary
        out.append(f"#define RSA_K5W_LOOP_{dt.upper()}_R256 \\")
        out.append(c_string(gen_loop(dt, pre=LOOP_PRE, tight=True, static=(dt == "bf16"), xf=LOOP_XF | {"halfdma"})))
        out.append("")
        out.append(f"#define RSA_K5W_LOOP_{dt.upper()}_DIAG \\")
        out.append(c_string(gen_loop(dt, diag=True, pre=LOOP_PRE, tight=True, static=(dt == "bf16"), xf=LOOP_XF)))
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
        out.append(f"#define RSA_K5W_RESCALE_{nmx} \\")
        out.append(c_string(rescale_core(S, ["%[al0]", "%[al1]"], ["%[de0]", "%[de1]"], ["%[ng0]", "%[ng1]"])))
        out.append("")
    out.append("#define RSA_K5W_NMZERO \\")
    out.append(c_string([f"v_mov_b32 {vr(NM[0] + i)}, 0" for i in range(32)]))
    out.append("")
    # accumulator-file housekeeping: zero O, write one Q fragment, read one O tile
    out.append("#define RSA_K5W_OZERO \\")
    out.append(c_string([f"v_accvgpr_write_b32 {ar(i)}, 0" for i in range(32 * DT)]))
    out.append("")
    # (k-steps / d tiles a head dim does not have get an empty body: the C++ side names all of them and discards by `if constexpr`)
    for h in (0, 1):
        for ks in range(8):
            out.append(f"#define RSA_K5W_QWRITE_H{h}_K{ks} \\")
            out.append(c_string([f"v_accvgpr_write_b32 {ar(AQ(h, ks) + j)}, {vr(TMP0 + j)}" for j in range(4)] if ks < KS else ["s_nop 0"]))
            out.append("")
    for h in (0, 1):
        for d in range(4):
            out.append(f"#define RSA_K5W_OREAD_H{h}_D{d} \\")
            out.append(c_string([f"v_accvgpr_read_b32 {vr(TMP0 + j)}, {ar(AO(h, d) + j)}" for j in range(16)] if d < DT else ["s_nop 0"]))
            out.append("")
    # operand lists
    souts = [f'"+{{{vr(SA[h], 16)}}}"(SA[{h}])' for h in (0, 1)] + [f'"+{{{vr(SB[h], 16)}}}"(SB[{h}])' for h in (0, 1)]
    outs = souts + ['[l0] "+v"(l[0])', '[l1] "+v"(l[1])', '[mx0] "=&v"(mx[0])', '[mx1] "=&v"(mx[1])']
    ins = [f'"{{{vr(NM[h], 16)}}}"(nm[{h}])' for h in (0, 1)] + [f'"{{{vr(KA, 8)}}}"(ka)', f'"{{{vr(VA, 8)}}}"(va)']
    out.append(f"#define RSA_K5W_OPS : {', '.join(outs)} : {', '.join(ins)}")
    nmio = [f'"+{{{vr(NM[h], 16)}}}"(nm[{h}])' for h in (0, 1)]
    louts = souts + nmio + ['[l0] "+v"(l[0])', '[l1] "+v"(l[1])', '[mx0] "+v"(mx[0])', '[mx1] "+v"(mx[1])',
                            '[th0] "+v"(thr[0])', '[th1] "+v"(thr[1])', '[mr0] "+v"(m_ref[0])', '[mr1] "+v"(m_ref[1])',
                            '[la] "+v"(la)', '[lv] "=&v"(lv)']
    lins = [f'"{{{vr(KA, 8)}}}"(ka)', f'"{{{vr(VA, 8)}}}"(va)', f'"{{{vr(VOK, 4)}}}"(vok)', f'"{{{vr(VOV, 4)}}}"(vov)',
            '[cnt] "s"(cnt)', '[blk0] "s"(blk0)', '[blk1] "s"(blk1)', '[kb] "s"(kb)', '[vb] "s"(vb)', '[krow] "s"(krow)',
            '[vrow] "s"(vrow)', '[ldsk] "s"(ldsk)', '[ldsv] "s"(ldsv)', '[ninf] "v"(ninf)', '[eight] "v"(eight)', '[stat] "s"(stat)']
    out.append(f"#define RSA_K5W_OPS_LOOP : {', '.join(louts)} : {', '.join(lins)}")
    dl = louts + ['[d0] "=s"(d0)', '[d1] "=s"(d1)']
    out.append(f"#define RSA_K5W_OPS_LOOP_DIAG : {', '.join(dl)} : {', '.join(lins)}")
    outs0 = [f'"+{{{vr(SA[h], 16)}}}"(SA[{h}])' for h in (0, 1)] + ['[mx0] "=&v"(mx[0])', '[mx1] "=&v"(mx[1])']
    out.append(f"#define RSA_K5W_OPS_QK0 : {', '.join(outs0)} : {', '.join(ins[:3])}")
    for nmx, S in (("A", SA), ("B", SB)):
        so = [f'"+{{{vr(S[h], 16)}}}"(S{nmx}[{h}])' for h in (0, 1)]
        out.append(f"#define RSA_K5W_OPS_ROWMAX_{nmx} : {', '.join(so)}, [mx0] \"=&v\"(mx[0]), [mx1] \"=&v\"(mx[1]) :")
        out.append(f"#define RSA_K5W_OPS_MASK_{nmx} : {', '.join(so)} : [kb0] \"v\"(kb0), [kb1] \"v\"(kb1), [sp0] \"v\"(sp0), "
                   f"[sp1] \"v\"(sp1), [ninf] \"v\"(ninf)")
        out.append(f"#define RSA_K5W_OPS_RESCALE_{nmx} : {', '.join(so + nmio)} : [al0] \"v\"(al0), [al1] \"v\"(al1), "
                   f"[de0] \"v\"(de0), [de1] \"v\"(de1), [ng0] \"v\"(ng0), [ng1] \"v\"(ng1)")
    out.append(f"#define RSA_K5W_OPS_NMZERO : \"={{{vr(NM[0], 16)}}}\"(nm[0]), \"={{{vr(NM[1], 16)}}}\"(nm[1])")
    tmp = ", ".join(f'"v{r}"' for r in range(TMP0, TMP1))
    acc_o = ", ".join(f'"a{r}"' for r in range(32 * DT))
    acc_q = ", ".join(f'"a{r}"' for r in range(32 * DT, 32 * DT + 8 * KS))
    out.append(f"#define RSA_K5W_CLOBBER_TMP {tmp}")
    out.append(f"#define RSA_K5W_CLOBBER_O {acc_o}")
    out.append(f"#define RSA_K5W_CLOBBER_Q {acc_q}")
    out.append("#define RSA_K5W_CLOBBER_LOOP " + ", ".join(f'"s{r}"' for r in S_CLOB) + ', "vcc", "scc"')
    out.append('#define RSA_K5W_CLOBBER_LOOP_DIAG "s75", "s76", "s77", "s78", "s79"')
    out.append(f"// head dim {D}: O a[0:{32 * DT - 1}], Q a[{32 * DT}:{32 * DT + 8 * KS - 1}]; SA v[0:31], SB v[32:63], -m v[64:95], temporaries v[{TMP0}:{TMP1 - 1}] "
               f"(P v[96:111], K ring v[112:127], V ring v[128:143]), K addresses v[{KA}:{KA + 7}], V addresses v[{VA}:{VA + 7}], "
               f"DMA lane offsets v[{VOK}:{VOV + 1}]; the loop statement owns s[{S_CLOB[0]}:{S_CLOB[-1]}]")
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "forms":
        gen_forms()
    elif len(sys.argv) > 1 and sys.argv[1] == "stats":
        STATS = []
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            main()
        for dt, U, dma, usage in STATS:
            if dt == "bf16" and U == 0:
                print(f"U{U} dma={dma}: per-gap issue cost {usage}  total {sum(usage)}")
    else:
        main()
