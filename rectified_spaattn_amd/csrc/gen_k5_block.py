#!/usr/bin/env python3
"""Generator of K5's hand-placed pipelined block (rsa_attn_block.h).

The block is one sub-step of the block-sparse attention main loop of rsa_attn_kernel.hip for ONE wave (32 query rows):

    S_nxt^T = K(sub-tile u+1) . Q^T - m    KS = D/16 MFMAs  (A = K rows from LDS by ds_read_b128, B = Q fragments, registers;
                                                             the chain starts from a 16-register block holding -m)
    P       = exp2(S_cur)                  in place, fp32; row sum into l; packed to the 2-byte type
    O^T    += V(sub-tile u)^T . P^T        2 * D/32 MFMAs   (A = V^T by ds_read_b64_tr_b16, B = packed P)
    mx      = row max of S_nxt             (two v_max3 chains + one v_permlane32_swap)

hipcc schedules that as it pleases (round 2: every operand read followed by lgkmcnt(0), profiles/r03_k5_r1_vs_head.md), so
the instruction stream is written here, once, with every register named: LDS operand reads run AHEAD MFMAs ahead of their
MFMA behind counted lgkmcnt waits, and the softmax's vector instructions are dealt into the MFMA shadows by issue cost
(cdna_hip_programming.md: 4 cycles per plain VALU, 8 per v_exp_f32, an MFMA leaves about 24 issue cycles of its 32).
Registers are pinned through physical-register constraints ("{v[96:111]}"); the operand lists are generated with the
streams (RSA_K5_OPS*), so the C++ side cannot drift from the register map.

Two forms are generated per (D, dtype, K/V slot parity, sub-step parity):
  * RSA_K5_BLOCKN_*: the product.  The kernel runs at the board's power cap (1.39 kW, 1.72 GHz under this loop,
    tools/clock_probe.py), so what it saves is instructions, not stalls: starting the score chain from -m removes the 16
    subtractions of a sub-step (-3.7 % time), the row maximum as two chains placed as soon as S_nxt is complete another
    0.7 % (profiles/r03_k5_block.md).  Nothing pads the statement behind its last MFMA: the only vector code that touches O
    outside the blocks (the rare rescale, the epilogue) carries the 12 wait states itself.
  * RSA_K5_BLOCK_*: the compiled block's arithmetic operation for operation (S, then S - m by v_sub), for the A/B build:
    bit-identical to the block as hipcc schedules it (tools/check_blk.py).

Measured and dropped: the sub-step's LDS-DMA pieces spread over the MFMA shadows (16.96 vs 16.42 ms: a piece stalls its wave
~70 cycles wherever it is issued, and inside the block that stall stops the wave's MFMA stream), 16x16x32 MFMAs (timing
probe: +-0 sparse, +3 % dense), packed f32 adds for the row sum (+0.8 % time).

usage: python3 gen_k5_block.py > rsa_attn_block.h
"""
import os

AHEAD = 4          # LDS operand buffers per operand kind (K fragments, V^T fragments)
COST = dict(sub=4, exp=8, cvt=4, cvt8=5, add=4, max=4, mov=4, nop=8, swap=4)


class Map:
    def __init__(self, D, negm):
        self.D, self.KS, self.DT = D, D // 16, D // 32
        r = 0
        self.O = r; r += 16 * self.DT
        self.Q = r; r += 4 * self.KS
        self.SA = r; r += 16
        self.SB = r; r += 16
        self.NM = r
        if negm: r += 16         # the reference maximum, negated, in all 16 registers: C operand of the first QK^T MFMA
        self.tmp0 = r
        self.P = r; r += 8
        self.KF = r; r += 4 * AHEAD
        self.VF = r; r += 4 * AHEAD
        self.PS = r; r += 1
        self.T0 = r; r += 1
        self.T1 = r; r += 1
        self.tmp1 = r            # clobbered temporaries: [tmp0, tmp1)
        r = (r + 3) & ~3
        self.KA = r; r += self.KS
        self.VA = r; r += 2 * self.DT
        self.RSM = rsm_form(D, negm)
        if self.RSM:             # row sums on the matrix pipe (head dim 64, round 5): l in four registers, the ones operand
            r = (r + 3) & ~3
            self.LACC = r; r += 4
            self.ONES = r; r += 4
        self.end = r


def rsm_form(D, negm):
    """Head dim 64 (both forms): 8 MFMAs per sub-step against the same softmax make the wave vector-bound
    (profiles/r05_d64_experiments.txt: without the vector work -29 %), so the 16 row-sum additions go to the matrix pipe as two
    v_mfma_f32_16x16x32 (ones . P^T, 4 passes each): l = the row sum of the ROUNDED P, complete over both lane halves."""
    import os
    return D == 64 and os.environ.get("RSA_GEN_NORSM", "") == ""     # (env: the A/B twin with the additions, tools/history/r5_d64x_build.sh)


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def gen_block(D, dt, VS, SUB, negm):
    """The asm lines of one block variant.  (env RSA_GEN_X: timing experiments of tools/history/r5_d64x_build.sh -- noexp = no exponentials,
    novalu = no vector work at all, nolds = no LDS operand reads; the results are garbage.)"""
    import os
    xf = set(filter(None, os.environ.get("RSA_GEN_X", "").split(",")))
    m = Map(D, negm)
    KS, DT = m.KS, m.DT
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    cv = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    TILE = 64 * D * 2
    SC, SN = (m.SA, m.SB) if SUB == 0 else (m.SB, m.SA)
    kslot, ksub = (VS, 1) if SUB == 0 else (VS ^ 1, 0)
    koff = kslot * TILE + ksub * 32 * D * 2
    vbase = (2 + VS) * TILE
    lines = []
    lds_seq = []          # issue order of LDS ops: (tag, count)

    def k_read(ks):
        if "nolds" in xf:
            lds_seq.append((("K", ks), 0))
            return
        lines.append(f"ds_read_b128 {vr(m.KF + 4 * (ks % AHEAD), 4)}, {vr(m.KA + ks)} offset:{koff}")
        lds_seq.append((("K", ks), 1))

    def v_read(p):
        if "nolds" in xf:
            lds_seq.append((("V", p), 0))
            return
        k2, d = divmod(p, DT)
        off = vbase + (2 * SUB + k2) * 16 * D * 2
        b = m.VF + 4 * (p % AHEAD)
        lines.append(f"ds_read_b64_tr_b16 {vr(b, 2)}, {vr(m.VA + 2 * d)} offset:{off}")
        lines.append(f"ds_read_b64_tr_b16 {vr(b + 2, 2)}, {vr(m.VA + 2 * d + 1)} offset:{off}")
        lds_seq.append((("V", p), 2))

    def wait_for(tag):
        idx = [i for i, (t, _) in enumerate(lds_seq) if t == tag][-1]
        after = sum(c for _, c in lds_seq[idx + 1:])
        lines.append(f"s_waitcnt lgkmcnt({after})")

    # ---- the vector work, in issue order ----
    def pair(i):   # two scores interleaved so that no instruction reads its predecessor's result
        sub = [("sub", f"v_sub_f32 {vr(SC + j)}, {vr(SC + j)}, %[m]") for j in (i, i + 1)]
        exp = [("exp", f"v_exp_f32 {vr(SC + j)}, {vr(SC + j)}") for j in (i, i + 1)]
        return exp if negm else sub + exp

    def C(j):
        return [("cvt", f"{cv} {vr(m.P + j)}, {vr(SC + 2 * j)}, {vr(SC + 2 * j + 1)}")]

    work = []
    for i in range(0, 8, 2): work += pair(i)
    work += pair(8)
    for j in range(0, 4): work += C(j)
    for i in range(10, 16, 2): work += pair(i)
    work += [("add", f"v_add_f32 {vr(m.PS)}, {vr(SC)}, {vr(SC + 1)}")]
    for i in range(2, 8): work += [("add", f"v_add_f32 {vr(m.PS)}, {vr(m.PS)}, {vr(SC + i)}")]
    for j in range(4, 8): work += C(j)
    for i in range(8, 16): work += [("add", f"v_add_f32 {vr(m.PS)}, {vr(m.PS)}, {vr(SC + i)}")]
    work += [("add", f"v_add_f32 %[l], %[l], {vr(m.PS)}")]
    if negm:   # two chains, merged by a v_max
        maxw = [("max", f"v_max_f32 {vr(m.T0)}, {vr(SN)}, {vr(SN + 1)}"), ("max", f"v_max_f32 {vr(m.T1)}, {vr(SN + 2)}, {vr(SN + 3)}")]
        for i in range(2, 8):
            t = m.T0 if i % 2 == 0 else m.T1
            maxw += [("max", f"v_max3_f32 {vr(t)}, {vr(t)}, {vr(SN + 2 * i)}, {vr(SN + 2 * i + 1)}")]
        maxw += [("max", f"v_max_f32 {vr(m.T0)}, {vr(m.T0)}, {vr(m.T1)}")]
    else:
        maxw = [("max", f"v_max_f32 {vr(m.T0)}, {vr(SN)}, {vr(SN + 1)}")]
        for i in range(1, 8): maxw += [("max", f"v_max3_f32 {vr(m.T0)}, {vr(m.T0)}, {vr(SN + 2 * i)}, {vr(SN + 2 * i + 1)}")]
    maxw += [("mov", f"v_mov_b32 {vr(m.T1)}, {vr(m.T0)}"), ("nop", "s_nop 1"),
             ("swap", f"v_permlane32_swap_b32 {vr(m.T0)}, {vr(m.T1)}"), ("nop", "s_nop 1"),
             ("max", f"v_max_f32 %[mx], {vr(m.T0)}, {vr(m.T1)}")]

    if "noexp" in xf:
        work = [w for w in work if w[0] != "exp"]
    if "novalu" in xf:
        work, maxw = [], [("mov", "v_mov_b32 %[mx], 0")]
    if m.RSM:
        work = [w for w in work if w[0] != "add"]
    mf_rs = "v_mfma_f32_16x16x32_bf16" if dt == "bf16" else "v_mfma_f32_16x16x32_f16"
    nm = KS + 2 * DT                      # MFMAs of the block
    total = sum(COST[k] for k, _ in work + maxw)
    pre = 72 if D == 128 else 48          # vector work issued while the first K fragments are in flight
    budget = max(24, -(-(total - pre) // nm) + 2)
    wi, mi = 0, 0

    def emit_work(cycles, allow_max):
        nonlocal wi, mi
        used = 0
        while used < cycles:
            if wi < len(work):
                k, t = work[wi]; wi += 1
            elif allow_max and mi < len(maxw):
                k, t = maxw[mi]; mi += 1
            else:
                break
            lines.append(t)
            used += COST[k]

    # (head dim 64: 8 MFMAs per sub-step against the same softmax, the wave is bound by vector issue, and raising its
    # priority over its partner on the SIMD costs 3-5 %: tools/perf_d64.py, profiles/r03_k5_block.md)
    if D == 128: lines.append("s_setprio 2")
    for ks in range(min(AHEAD, KS)): k_read(ks)
    emit_work(pre, False)
    for i in range(nm):
        if i < KS:
            wait_for(("K", i))
            c = (vr(m.NM, 16) if negm else "0") if i == 0 else vr(SN, 16)
            lines.append(f"{mf} {vr(SN, 16)}, {vr(m.KF + 4 * (i % AHEAD), 4)}, {vr(m.Q + 4 * i, 4)}, {c}")
            if i + AHEAD < KS: k_read(i + AHEAD)
            p = i - (KS - min(AHEAD, KS))          # V^T fragments of the first PV MFMAs ride the last QK shadows
            if 0 <= p < min(AHEAD, 2 * DT) and KS >= AHEAD: v_read(p)
            if KS < AHEAD and i == KS - 1:
                for pp in range(min(AHEAD, 2 * DT)): v_read(pp)
        else:
            p = i - KS
            k2, d = divmod(p, DT)
            text = "\n".join(lines)
            for jj in range(4 if "novalu" not in xf else 0):   # sanity: the half of P this MFMA reads has been packed
                assert f"{cv} {vr(m.P + 4 * k2 + jj)}," in text, (D, dt, VS, SUB, "P not packed before PV", p)
            if m.RSM and d == 0:      # l += ones . P^T over the 16 keys of half k2 (behind the conversions that packed it)
                lines.append("s_nop 1")
                lines.append(f"{mf_rs} {vr(m.LACC, 4)}, {vr(m.ONES, 4)}, {vr(m.P + 4 * k2, 4)}, {vr(m.LACC, 4)}")
            wait_for(("V", p))
            lines.append(f"{mf} {vr(m.O + 16 * d, 16)}, {vr(m.VF + 4 * (p % AHEAD), 4)}, {vr(m.P + 4 * k2, 4)}, {vr(m.O + 16 * d, 16)}")
            if p + AHEAD < 2 * DT: v_read(p + AHEAD)
        last = i == nm - 1
        # The row max reads S_nxt: two or more MFMAs behind the last QK MFMA (its 8 passes are over).  Classic form: it fills
        # the last two shadows and the tail, padded to the 12 wait states between the last PV MFMA and any VALU touching O.
        n0 = len(lines)
        emit_work(10 ** 6 if last else budget, i >= (KS + 1 if negm else max(KS + 1, nm - 2)))
        if last and not negm:
            tail = sum(int(l.split()[1]) + 1 if l.startswith("s_nop") else 1 for l in lines[n0:])
            if tail < 12: lines.append(f"s_nop {11 - tail}")
    assert wi == len(work) and mi == len(maxw)
    if D == 128: lines.append("s_setprio 0")
    return lines, m


def c_string(lines):
    return "\n".join(f'    "{l}\\n\\t"' for l in lines)


def main():
    out = ["// GENERATED by gen_k5_block.py -- do not edit; edit the generator (its docstring says what this is).", "#pragma once", ""]
    for D in (128, 64):
        for dt in ("bf16", "f16"):
            for negm in (False, True):
                for VS in (0, 1):
                    for SUB in (0, 1):
                        lines, m = gen_block(D, dt, VS, SUB, negm)
                        out.append(f"#define RSA_K5_BLOCK{'N' if negm else ''}_{D}_{dt.upper()}_V{VS}_S{SUB} \\")
                        out.append(" \\\n".join(c_string(lines).split("\n")))
                        out.append("")
    # operand lists (the constraint strings carry the register map) and the clobbered temporaries
    for D in (128, 64):
        for negm in (False, True):
            m = Map(D, negm)
            outs = [f'"+{{{vr(m.O + 16 * d, 16)}}}"(o[{d}])' for d in range(m.DT)]
            outs += [f'"+{{{vr(m.SA, 16)}}}"(SA)', f'"+{{{vr(m.SB, 16)}}}"(SB)',
                     f'"+{{{vr(m.LACC, 4)}}}"(lacc)' if m.RSM else '[l] "+v"(l)', '[mx] "=&v"(mx)']
            ins = [f'"{{{vr(m.Q + 4 * k, 4)}}}"(q[{k}])' for k in range(m.KS)]
            ins += [f'"{{{vr(m.NM, 16)}}}"(nm)'] if negm else ['[m] "v"(m)']
            ins += [f'"{{{vr(m.KA, m.KS)}}}"(ka)', f'"{{{vr(m.VA, 2 * m.DT)}}}"(va)']
            if m.RSM: ins += [f'"{{{vr(m.ONES, 4)}}}"(onesv)']
            tag = f"{'N' if negm else ''}_{D}"
            out.append(f"#define RSA_K5_OPS{tag} : {', '.join(outs)} : {', '.join(ins)}")
            out.append(f"#define RSA_K5_CLOBBER{tag} " + ", ".join(f'"v{r}"' for r in range(m.tmp0, m.tmp1)))
            out.append(f"// D = {D}{' (-m form)' if negm else ''}: O v[{m.O}:{m.Q - 1}], Q v[{m.Q}:{m.SA - 1}], SA v[{m.SA}:{m.SA + 15}], "
                       f"SB v[{m.SB}:{m.SB + 15}], " + (f"-m v[{m.NM}:{m.NM + 15}], " if negm else "")
                       + f"temporaries v[{m.tmp0}:{m.tmp1 - 1}], K addresses v[{m.KA}:{m.KA + m.KS - 1}], V addresses v[{m.VA}:{m.end - 1}]")
    main8(out)
    main8h(out)
    print("\n".join(out))



# =====================================================================================================================
# e4m3 kernel (rsa_attn_fp8_kernel.hip): one block = one 64-KEY TILE of one wave
#     S_nxt^T (two 32-key halves) = K8(tile+1) . Q8^T + (4 - m)   4 x v_mfma_scale_f32_32x32x64_f8f6f4 (chains start from MB)
#     P = exp2(S_cur) -> e4m3, packed in place into S_cur[0][0:7] 32 v_exp_f32 + 16 v_cvt_pk_fp8_f32
#     l += ones . P (row sum on the matrix pipe)                    1 x v_mfma_f32_16x16x128_f8f6f4
#     O^T += 2^ev V8^T(tile) . P                                    4 x v_mfma_scale_f32_32x32x64_f8f6f4 (V block scale: byte 1)
#     mx = row max of S_nxt
# Every A operand is 8 registers = 2 x ds_read_b128, read AHEAD8 MFMAs ahead into a ring of RING8 buffers.  K and V live in
# 4-slot LDS rings: the tile's slot (TS = tile & 3) is a compile-time constant of the block, so every LDS address is a
# loop-invariant VGPR plus an immediate (the compiled block spent ~16 v_add_u32 per tile on them).
# =====================================================================================================================
AHEAD8, RING8 = 3, 4
DMA8_GAPS = [0, 2, 4, 6]      # dma form: K piece 0, K piece 1, V piece 0, V piece 1 behind these MFMAs of the nine
if os.environ.get("RSA_GEN8_GAPS"):       # (placement A/B: profiles/r05_pv_hand_placed.txt)
    DMA8_GAPS = [int(x) for x in os.environ["RSA_GEN8_GAPS"].split(",")]


class Map8:
    def __init__(self, D8=128):
        r = 0
        self.D8, self.KS, self.DT = D8, D8 // 64, D8 // 32
        self.O = r; r += 16 * self.DT
        self.Q = r; r += 8 * self.KS
        self.SA = r; r += 32
        self.SB = r; r += 32
        self.MB = r; r += 16
        self.LACC = r; r += 4
        self.tmp0 = r
        self.OP = r; r += 8 * RING8
        self.T0 = r; r += 1
        self.T1 = r; r += 1
        self.tmp1 = r
        r = (r + 1) & ~1
        self.SC = r; r += 2          # E8M0 scale operands of the QK^T MFMAs
        self.KA = r; r += 2 * self.KS   # K read addresses [ks][chunk]
        self.VA = r; r += 2          # V read addresses [chunk]
        self.ON = r; r += 1          # address of this lane's ones / zeros pattern
        r = (r + 1) & ~1
        self.DK = r; r += 2          # LDS-DMA lane offsets of the wave's two K pieces (the second = the first + 4 096): dma form
        self.DV = r; r += 2          # ... of its two V pieces (head dim 64: one piece each)
        self.ONES = r; r += 8        # dma form: the row-sum product's ones operand in registers (as the pv block): 2 KiB fewer LDS reads per tile and wave
        self.end = r


def gen_block8(TS, codemap=False, D8=128, dma=False):
    """dma (round 5; head dim 128): the block also issues the wave's four LDS-DMA pieces -- K(tile + 3) into ring slot (TS + 3) & 3,
    V(tile + 2) into (TS + 2) & 3 -- one per MFMA shadow instead of as a burst behind the barrier (there the eight waves' pieces queue
    at the CU's one addresser: profiles/r05_pv_hand_placed.txt).  Scalar operands: %[ksrc] / %[vsrc] = first byte of the wave's
    first piece, %[ldsw] = LDS address of the wave's first piece in slot 0 of the K ring."""
    import os
    xf = set(filter(None, os.environ.get("RSA_GEN8_X", "").split(",")))   # timing experiments (tools/history/r5_pvx_build.sh): halfk, nolds
    m = Map8(D8)
    SC_, SN = (m.SA, m.SB) if TS % 2 == 0 else (m.SB, m.SA)
    TILE8 = 64 * D8
    kslot = (TS + 1) & 3
    lines, lds_seq = [], []
    # the A operands, in MFMA order (nine at head dim 128, five at 64): (kind, LDS reads as (address register, immediate) pairs)
    ops = []
    for sub in range(2):
        for ks in range(m.KS):
            off = kslot * TILE8 + sub * 32 * D8
            ops.append(("qk", sub, ks, [] if ("halfk" in xf and sub == 1) or "nolds" in xf else [(m.KA + 2 * ks, off), (m.KA + 2 * ks + 1, off)]))
    n_qk = len(ops)
    ones_reg = dma      # (round 5: -3 % at head dim 128, profiles/r05_pv_hand_placed.txt)
    ops.append(("rs", 0, 0, [] if "nolds" in xf or ones_reg else [(m.ON, 0), (m.ON, 16)]))
    for dt in range(m.DT):
        off = (4 + TS) * TILE8 + dt * 2048
        ops.append(("pv", dt, 0, [] if "nolds" in xf else [(m.VA, off), (m.VA + 1, off)]))

    def read(i):
        if not ops[i][3]:
            return
        b = m.OP + 8 * (i % RING8)
        for c2, (areg, off) in enumerate(ops[i][3]):
            lines.append(f"ds_read_b128 {vr(b + 4 * c2, 4)}, {vr(areg)}" + (f" offset:{off}" if off else ""))
        lds_seq.append((i, 2))

    def wait_for(i):
        idx = [k for k, (t, _) in enumerate(lds_seq) if t == i][-1]
        lines.append(f"s_waitcnt lgkmcnt({sum(c for _, c in lds_seq[idx + 1:])})")

    # vector work: exponentials in place, the e4m3 words of P built in place in S_cur[0][0:7]
    work = []
    def exps(sub, i0):
        return [("exp", f"v_exp_f32 {vr(SC_ + 16 * sub + i)}, {vr(SC_ + 16 * sub + i)}") for i in range(i0, i0 + 4)]
    def pack(j):   # P word j = 4 * sub + w4 <- exp values 4 w4 .. 4 w4 + 3 of half `sub`
        sub, w4 = divmod(j, 4)
        e = SC_ + 16 * sub + 4 * w4
        return [("cvt", f"v_cvt_pk_fp8_f32 {vr(SC_ + j)}, {vr(e)}, {vr(e + 1)}"),
                ("cvt", f"v_cvt_pk_fp8_f32 {vr(SC_ + j)}, {vr(e + 2)}, {vr(e + 3)} op_sel:[0,0,1]")]
    # order: a group's exponentials, then the PREVIOUS group's packing (a transcendental's result is not read by the next
    # instruction); word j overwrites register j of S_cur[0], whose value the words before it have consumed
    groups = [(s_, i0) for s_ in range(2) for i0 in range(0, 16, 4)]
    if not codemap:
        for g, (s_, i0) in enumerate(groups):
            work += exps(s_, i0)
            if g >= 1: work += pack(g - 1)
        work += pack(7)
    else:
        # code-map form: the accumulator holds 8 log2(P) + 56, the e4m3 CODE of P up to rounding: one v_cvt_pk_u8_f32 per score
        # (round to nearest even, saturating at 0: tools/probes/cvt_pk_u8_probe.hip), no exponential, no fp8 conversion.
        # Word j = register j of S_cur[0]; registers 0..7 are all sources of words 0 and 1, so those two go first (word 1
        # starts once word 0 has read register 1), the rest in interleaved pairs (no back-to-back dependent conversions).
        def byte(j, e):
            sub, w4 = divmod(j, 4)
            return ("cvt8", f"v_cvt_pk_u8_f32 {vr(SC_ + j)}, {vr(SC_ + 16 * sub + 4 * w4 + e)}, {e}, {vr(SC_ + j)}")
        work += [byte(0, 0), byte(0, 1), byte(1, 0), byte(0, 2), byte(1, 1), byte(0, 3), byte(1, 2), byte(1, 3)]
        for j in (2, 4, 6):
            for e in range(4):
                work += [byte(j, e), byte(j + 1, e)]
    maxw = [("max", f"v_max_f32 {vr(m.T0)}, {vr(SN)}, {vr(SN + 1)}"), ("max", f"v_max_f32 {vr(m.T1)}, {vr(SN + 2)}, {vr(SN + 3)}")]
    for i in range(2, 16):
        t = m.T0 if i % 2 == 0 else m.T1
        maxw += [("max", f"v_max3_f32 {vr(t)}, {vr(t)}, {vr(SN + 2 * i)}, {vr(SN + 2 * i + 1)}")]
    maxw += [("max", f"v_max_f32 {vr(m.T0)}, {vr(m.T0)}, {vr(m.T1)}"), ("mov", f"v_mov_b32 {vr(m.T1)}, {vr(m.T0)}"),
             ("nop", "s_nop 1"), ("swap", f"v_permlane32_swap_b32 {vr(m.T0)}, {vr(m.T1)}"), ("nop", "s_nop 1"),
             ("max", f"v_max_f32 %[mx], {vr(m.T0)}, {vr(m.T1)}")]
    wi, mi = 0, 0

    def emit_work(cycles, allow_max, force_all=False):
        nonlocal wi, mi
        used = 0
        while used < cycles or force_all:
            if wi < len(work):
                k, t = work[wi]; wi += 1
            elif allow_max and mi < len(maxw):
                k, t = maxw[mi]; mi += 1
            else:
                break
            lines.append(t)
            used += COST[k]

    lines.append("s_setprio 2")
    for i in range(AHEAD8): read(i)
    emit_work(64, False)
    n = len(ops)
    for i, (kind, x, y, _) in enumerate(ops):
        if kind != "qk" and wi < len(work):   # the row sum and PV read the whole packed P
            emit_work(0, False, force_all=True)
        if ops[i][3]:
            wait_for(i)
        a = vr(m.OP + 8 * ((i if ops[i][3] or "nolds" in xf else i - m.KS) % RING8), 8)
        if kind == "qk":
            c = vr(m.MB, 16) if y == 0 else vr(SN + 16 * x, 16)
            lines.append(f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(SN + 16 * x, 16)}, {a}, {vr(m.Q + 8 * y, 8)}, {c}, "
                         f"{vr(m.SC)}, {vr(m.SC + 1)} op_sel_hi:[0,0,0]")
        elif kind == "rs":
            lines.append("s_nop 1")   # the last word of P was packed just above
            lines.append(f"v_mfma_f32_16x16x128_f8f6f4 {vr(m.LACC, 4)}, {vr(m.ONES, 8) if ones_reg else a}, {vr(SC_, 8)}, {vr(m.LACC, 4)}")
        else:
            # (byte 1 of the scale registers: the V block's scale on A, 1.0 on the P operand)
            lines.append(f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(m.O + 16 * x, 16)}, {a}, {vr(SC_, 8)}, {vr(m.O + 16 * x, 16)}, "
                         f"{vr(m.SC)}, {vr(m.SC + 1)} op_sel:[1,1,0] op_sel_hi:[0,0,0]")
        if i + AHEAD8 < n: read(i + AHEAD8)
        gaps = DMA8_GAPS if D8 == 128 else [0, 2]      # head dim 64: one K piece, one V piece per wave and tile
        if dma and i in gaps:
            j = gaps.index(i)
            isv, hi = divmod(j, TILE8 // 4096)
            dst = (4 + ((TS + 2) & 3)) * TILE8 + hi * 4096 if isv else ((TS + 3) & 3) * TILE8 + hi * 4096
            lines.append(f"s_add_u32 m0, %[ldsw], {dst}")
            lines.append("s_nop 0")      # (M0 write -> LDS-DMA: one wait state)
            lines.append(f"global_load_lds_dwordx4 {vr((m.DV if isv else m.DK) + hi)}, " + ("%[vsrc]" if isv else "%[ksrc]"))
        # row max of S_nxt: two MFMAs behind the last QK^T MFMA (index n_qk - 1): from the shadow of PV 0 (index n_qk + 1) on
        emit_work(10 ** 6 if i == n - 1 else 48, i >= n_qk + 1)
    assert wi == len(work) and mi == len(maxw)
    lines.append("s_setprio 0")
    return lines, m


def main8(out):
    for codemap in (False, True):
        for TS in range(4):
            lines, m = gen_block8(TS, codemap)
            out.append(f"#define RSA_K5F8_BLOCK{'C' if codemap else ''}_T{TS} \\")
            out.append(" \\\n".join(c_string(lines).split("\n")))
            out.append("")
    for TS in range(4):   # the product at head dim 128: code map + the wave's LDS-DMA pieces inside the block
        lines, m = gen_block8(TS, True, 128, dma=True)
        out.append(f"#define RSA_K5F8_BLOCKCD_T{TS} \\")
        out.append(" \\\n".join(c_string(lines).split("\n")))
        out.append("")
    for TS in range(4):   # head dim 64 (CogVideoX): the code-map form, staging behind the barrier / inside the block (product)
        for dma in (False, True):
            lines, m = gen_block8(TS, True, 64, dma=dma)
            out.append(f"#define RSA_K5F8_BLOCKC{'D' if dma else ''}64_T{TS} \\")
            out.append(" \\\n".join(c_string(lines).split("\n")))
            out.append("")
    m = Map8(64)
    outs = [f'"+{{{vr(m.O + 16 * d, 16)}}}"(o[{d}])' for d in range(2)]
    outs += [f'"+{{{vr(m.SA, 16)}}}"(SA[0])', f'"+{{{vr(m.SA + 16, 16)}}}"(SA[1])', f'"+{{{vr(m.SB, 16)}}}"(SB[0])',
             f'"+{{{vr(m.SB + 16, 16)}}}"(SB[1])', f'"+{{{vr(m.LACC, 4)}}}"(lacc)', '[mx] "=&v"(mx)']
    ins = [f'"{{{vr(m.Q, 8)}}}"(q[0])', f'"{{{vr(m.MB, 16)}}}"(mblk)',
           f'"{{{vr(m.SC)}}}"(sca)', f'"{{{vr(m.SC + 1)}}}"(scb)', f'"{{{vr(m.KA, 2)}}}"(ka)', f'"{{{vr(m.VA, 2)}}}"(va)',
           f'"{{{vr(m.ON)}}}"(ona)']
    out.append(f"#define RSA_K5F8_OPS64 : {', '.join(outs)} : {', '.join(ins)}")
    insd = ins[:-1] + [f'"{{{vr(m.ONES, 8)}}}"(onesv)', f'"{{{vr(m.DK)}}}"(dk)', f'"{{{vr(m.DV)}}}"(dv)', '[ksrc] "s"(ksrc)', '[vsrc] "s"(vsrc)', '[ldsw] "s"(ldsw)']
    out.append(f"#define RSA_K5F8_OPS64D : {', '.join(outs)} : {', '.join(insd)}")
    out.append("#define RSA_K5F8_CLOBBER64 " + ", ".join(f'"v{r}"' for r in range(m.tmp0, m.tmp1)))
    out.append(f"// e4m3 kernel, head dim 64: O v[0:{m.Q - 1}], Q v[{m.Q}:{m.SA - 1}], SA v[{m.SA}:{m.SB - 1}], SB v[{m.SB}:{m.MB - 1}], "
               f"reference block v[{m.MB}:{m.LACC - 1}], l v[{m.LACC}:{m.LACC + 3}], temporaries v[{m.tmp0}:{m.tmp1 - 1}], "
               f"scales v[{m.SC}:{m.SC + 1}], K / V / ones addresses v[{m.KA}:{m.end - 1}]")
    m = Map8()
    outs = [f'"+{{{vr(m.O + 16 * d, 16)}}}"(o[{d}])' for d in range(4)]
    outs += [f'"+{{{vr(m.SA, 16)}}}"(SA[0])', f'"+{{{vr(m.SA + 16, 16)}}}"(SA[1])', f'"+{{{vr(m.SB, 16)}}}"(SB[0])',
             f'"+{{{vr(m.SB + 16, 16)}}}"(SB[1])', f'"+{{{vr(m.LACC, 4)}}}"(lacc)', '[mx] "=&v"(mx)']
    ins = [f'"{{{vr(m.Q, 8)}}}"(q[0])', f'"{{{vr(m.Q + 8, 8)}}}"(q[1])', f'"{{{vr(m.MB, 16)}}}"(mblk)',
           f'"{{{vr(m.SC)}}}"(sca)', f'"{{{vr(m.SC + 1)}}}"(scb)', f'"{{{vr(m.KA, 4)}}}"(ka)', f'"{{{vr(m.VA, 2)}}}"(va)',
           f'"{{{vr(m.ON)}}}"(ona)']
    out.append(f"#define RSA_K5F8_OPS : {', '.join(outs)} : {', '.join(ins)}")
    insd = ins[:-1] + [f'"{{{vr(m.ONES, 8)}}}"(onesv)', f'"{{{vr(m.DK, 2)}}}"(dk)', f'"{{{vr(m.DV, 2)}}}"(dv)', '[ksrc] "s"(ksrc)', '[vsrc] "s"(vsrc)', '[ldsw] "s"(ldsw)']
    out.append(f"#define RSA_K5F8_OPSD : {', '.join(outs)} : {', '.join(insd)}")
    out.append("#define RSA_K5F8_CLOBBER " + ", ".join(f'"v{r}"' for r in range(m.tmp0, m.tmp1)))
    out.append(f"// e4m3 kernel: O v[0:63], Q v[{m.Q}:{m.SA - 1}], SA v[{m.SA}:{m.SB - 1}], SB v[{m.SB}:{m.MB - 1}], 4 - m v[{m.MB}:{m.LACC - 1}], "
               f"l v[{m.LACC}:{m.LACC + 3}], temporaries v[{m.tmp0}:{m.tmp1 - 1}], scales v[{m.SC}:{m.SC + 1}], "
               f"K / V / ones addresses v[{m.KA}:{m.end - 1}]")


# =====================================================================================================================
# the "pv" form of the e4m3 kernel (round 5; rsa_attn_fp8_kernel.hip, HYB instances): one block = one 64-key tile of one wave
#     S_nxt^T (two 32-key halves) = K(tile+1) . Q^T + MB     16 x v_mfma_f32_32x32x16 on the 2-BYTE K rows (A = one ds_read_b128 per
#                                                            k-step from the 16 KiB K tile, image of rsa_attn_kernel.hip) and the
#                                                            2-byte Q fragments (32 registers; Q carries sm_scale log2e 8)
#     P codes, row sum, O^T += V8^T . P, mx                  exactly the e4m3 block above (code map form)
# Rings of THREE slots (K 3 x 16 KiB at LDS 0, V 3 x 8 KiB behind it): K(tile+1) sits in slot (tile + 1) % 3, V(tile) in
# tile % 3, S_cur is SA on even tiles -> six variants, T = tile % 6.  The V read addresses carry the V ring's base (49 152 is
# beyond what the 16-bit offset field could add to a K address with slot and d-tile offsets on top).
# The LDS operand registers are a pool of eight 4-register slots (QK^T fragments take one, the 8-register operands of the row
# sum and of P . V two, aligned): a read may re-target a slot once the MFMA AFTER its last consumer has been issued (the rule
# of the e4m3 block's ring); QK^T fragments are read six MFMAs ahead (a 2-byte MFMA is half as long as an e4m3 one), the rest three.
# =====================================================================================================================
class Map8H:
    def __init__(self, D8=128):
        self.D8, self.KS, self.DT = D8, D8 // 16, D8 // 32
        self.NPK, self.NPV = D8 // 32, D8 // 64          # LDS-DMA pieces per wave: K tile (64 rows x 2 D8 bytes), V tile (D8 x 64 bytes)
        r = 0
        self.O = r; r += 16 * self.DT
        self.Q = r; r += 4 * self.KS
        self.SA = r; r += 32
        self.SB = r; r += 32
        self.MB = r; r += 16
        self.LACC = r; r += 4
        self.tmp0 = r
        self.OP = r; r += 4 * NSLOT8H
        self.T0 = r; r += 1
        self.T1 = r; r += 1
        self.tmp1 = r
        r = (r + 1) & ~1
        self.SC = r; r += 2
        self.VA = r; r += 2          # V read addresses [chunk], V ring base included (an even-aligned pair)
        self.KB = r; r += 1          # K read address of k-step 0 (row r of slot 0); k-step ks = this ^ (ks << 5) (the tile's XOR
                                     # swizzle), built in T0 / T1 just before the reads; 32-key half and ring slot are immediates
        r = (r + 1) & ~1
        self.ONES = r; r += 8        # A operand of the row-sum product (e4m3 1.0 or 0 in all 32 bytes: a constant of the lane)
        r = (r + 3) & ~3
        self.DK = r; r += self.NPK   # dma form: LDS-DMA lane offsets of the wave's K pieces of tile + 3 (rows clamped: per tile)
        r = (r + 1) & ~1
        self.DV = r; r += self.NPV   # ... of its V pieces (a second one = the first + 4 096: constants of the lane)
        self.end = r


NSLOT8H = 6      # 4-register operand slots (256 registers per wave at two waves per SIMD: the block pins 212 of them at head dim 128)


DMA8H_GAPS = {128: [1, 4, 7, 10, 13, 16],     # dma form: K pieces 0..3, V pieces 0, 1 behind these MFMAs of the 21
              64: [1, 4, 7]}                  # head dim 64: K pieces 0, 1, V piece 0 of the 11
if os.environ.get("RSA_GEN8H_GAPS"):
    DMA8H_GAPS[128] = [int(x) for x in os.environ["RSA_GEN8H_GAPS"].split(",")]


def gen_block8h(T6, dt, dma=False, D8=128):
    """dma: the block also issues the wave's six LDS-DMA pieces -- K(tile + 3) into ring slot T6 % 3 (= K(tile)'s), V(tile + 2) into
    (T6 + 2) % 3 -- one per MFMA shadow (see gen_block8).  Scalar operands: %[kb16] = the head's K rows, %[vsrc] = first byte of the
    wave's first V piece, %[ldsw] = LDS address of the wave's first piece in slot 0 of the K ring.
    xf (env RSA_GEN8H_X, comma separated; timing experiments of tools/history/r5_pvx.sh, results are garbage): halfk = the second
    32-key half re-uses the first half's K fragment (8 instead of 16 K reads), nov = no V reads, novalu = no conversions / row max."""
    import os
    xf = set(filter(None, os.environ.get("RSA_GEN8H_X", "").split(",")))
    m = Map8H(D8)
    KTILE, VTILE = 128 * D8, 64 * D8          # bytes of a K tile (64 keys x 2 D8: 16 / 8 KiB) and of a V tile (D8 x 64: 8 / 4 KiB)
    mf = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
    SC_, SN = (m.SA, m.SB) if T6 % 2 == 0 else (m.SB, m.SA)
    kslot, vslot = (T6 + 1) % 3, T6 % 3
    lines, lds_seq = [], []
    ops = []      # (kind, x, y, reads [(address register, offset)], slots)
    for ks in range(m.KS):       # the two halves' chains alternate: no MFMA waits for the one just before it
        kreg = m.KB if ks == 0 else (m.T0 if ks % 2 == 0 else m.T1)
        for sub in range(2):
            if ("halfk" in xf and sub == 1) or "nok" in xf:
                ops.append(("qk", sub, ks, [], 0))
            else:
                ops.append(("qk", sub, ks, [(kreg, kslot * KTILE + sub * (KTILE // 2))], 1))
    n_qk = len(ops)
    ops.append(("rs", 0, 0, [], 0))
    for d in range(m.DT):
        off = vslot * VTILE + d * 2048
        ops.append(("pv", d, 0, [] if "nov" in xf else [(m.VA, off), (m.VA + 1, off)], 0 if "nov" in xf else 2))
    n = len(ops)
    slot_of, consumer = {}, [None] * NSLOT8H     # consumer[s] = index of the op that reads slot s (None: free)
    state = dict(issued=-1)

    def read(j, cur):
        need = ops[j][4]
        if need == 0:
            return
        for s0 in range(0, NSLOT8H, need):
            # (cur - 2: the MFMA AFTER the slot's last consumer has been issued, the rule of the e4m3 block's ring)
            if all(consumer[s0 + u] is None or consumer[s0 + u] <= cur - 2 for u in range(need)):
                break
        else:
            raise AssertionError(("no free operand slot", T6, j, cur, consumer))
        for u in range(need):
            consumer[s0 + u] = j
        slot_of[j] = s0
        b = m.OP + 4 * s0
        kind, sub, ks = ops[j][0], ops[j][1], ops[j][2]
        if kind == "qk" and sub == 0 and ks > 0 and "nok" not in xf:
            assert state["issued"] < n_qk, "T0 / T1 belong to the row maximum from P . V 0 on"
            lines.append(f"v_xor_b32 {vr(ops[j][3][0][0])}, {hex(ks << 5)}, {vr(m.KB)}")
        for c2, (areg, off) in enumerate(ops[j][3]):
            lines.append(f"ds_read_b128 {vr(b + 4 * c2, 4)}, {vr(areg)}" + (f" offset:{off}" if off else ""))
        lds_seq.append((j, len(ops[j][3])))

    def wait_for(i):
        idx = [k for k, (t, _) in enumerate(lds_seq) if t == i][-1]
        lines.append(f"s_waitcnt lgkmcnt({sum(c for _, c in lds_seq[idx + 1:])})")

    # vector work: the code-map conversions of S_cur in place (as gen_block8), then the row maximum of S_nxt
    def byte(j, e):
        sub, w4 = divmod(j, 4)
        return ("cvt8", f"v_cvt_pk_u8_f32 {vr(SC_ + j)}, {vr(SC_ + 16 * sub + 4 * w4 + e)}, {e}, {vr(SC_ + j)}")
    work = [byte(0, 0), byte(0, 1), byte(1, 0), byte(0, 2), byte(1, 1), byte(0, 3), byte(1, 2), byte(1, 3)]
    for j in (2, 4, 6):
        for e in range(4):
            work += [byte(j, e), byte(j + 1, e)]
    maxw = [("max", f"v_max_f32 {vr(m.T0)}, {vr(SN)}, {vr(SN + 1)}"), ("max", f"v_max_f32 {vr(m.T1)}, {vr(SN + 2)}, {vr(SN + 3)}")]
    for i in range(2, 16):
        t = m.T0 if i % 2 == 0 else m.T1
        maxw += [("max", f"v_max3_f32 {vr(t)}, {vr(t)}, {vr(SN + 2 * i)}, {vr(SN + 2 * i + 1)}")]
    maxw += [("max", f"v_max_f32 {vr(m.T0)}, {vr(m.T0)}, {vr(m.T1)}"), ("mov", f"v_mov_b32 {vr(m.T1)}, {vr(m.T0)}"),
             ("nop", "s_nop 1"), ("swap", f"v_permlane32_swap_b32 {vr(m.T0)}, {vr(m.T1)}"), ("nop", "s_nop 1"),
             ("max", f"v_max_f32 %[mx], {vr(m.T0)}, {vr(m.T1)}")]
    wi, mi = 0, 0
    if "novalu" in xf:
        work, maxw = [], [("mov", "v_mov_b32 %[mx], 0")]

    def emit_work(cycles, allow_max, force_all=False):
        nonlocal wi, mi
        used = 0
        while used < cycles or force_all:
            if wi < len(work):
                k, t = work[wi]; wi += 1
            elif allow_max and mi < len(maxw):
                k, t = maxw[mi]; mi += 1
            else:
                break
            lines.append(t)
            used += COST[k]

    nxt = 0           # next op whose operand read has not been issued

    def top_up(cur):
        """issue the reads of the next ops while an operand slot is free (`cur` = the op about to issue)"""
        nonlocal nxt
        while nxt < n:
            try:
                read(nxt, cur)
            except AssertionError:
                break
            nxt += 1

    lines.append("s_setprio 2")
    top_up(0)
    emit_work(48, False)
    for i, (kind, x, y, _, _) in enumerate(ops):
        if kind != "qk" and wi < len(work):   # the row sum and P . V read the whole packed P
            emit_work(0, False, force_all=True)
        assert i < nxt, ("operand never read", T6, i)
        if ops[i][4]:
            wait_for(i)
        b = m.OP + 4 * slot_of.get(i, slot_of.get(i - 1, 0))
        if kind == "qk":
            c = vr(m.MB, 16) if y == 0 else vr(SN + 16 * x, 16)
            lines.append(f"{mf} {vr(SN + 16 * x, 16)}, {vr(b, 4)}, {vr(m.Q + 4 * y, 4)}, {c}")
        elif kind == "rs":
            lines.append("s_nop 1")
            lines.append(f"v_mfma_f32_16x16x128_f8f6f4 {vr(m.LACC, 4)}, {vr(m.ONES, 8)}, {vr(SC_, 8)}, {vr(m.LACC, 4)}")
        else:
            lines.append(f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(m.O + 16 * x, 16)}, {vr(b, 8)}, {vr(SC_, 8)}, {vr(m.O + 16 * x, 16)}, "
                         f"{vr(m.SC)}, {vr(m.SC + 1)} op_sel:[1,1,0] op_sel_hi:[0,0,0]")
        state["issued"] = i
        top_up(i + 1)
        if dma and i in DMA8H_GAPS[D8]:
            j = DMA8H_GAPS[D8].index(i)
            if j < m.NPK:
                dst, vo, src = (T6 % 3) * KTILE + j * 4096, m.DK + j, "%[kb16]"
            else:
                dst, vo, src = 3 * KTILE + ((T6 + 2) % 3) * VTILE + (j - m.NPK) * 4096, m.DV + j - m.NPK, "%[vsrc]"
            lines.append(f"s_add_u32 m0, %[ldsw], {dst}")
            lines.append("s_nop 0")      # (M0 write -> LDS-DMA: one wait state)
            lines.append(f"global_load_lds_dwordx4 {vr(vo)}, {src}")
        # row max of S_nxt: two MFMAs behind the last QK^T MFMA
        emit_work(10 ** 6 if i == n - 1 else (20 if kind == "qk" else 48), i >= n_qk + 1)
    assert wi == len(work) and mi == len(maxw) and nxt == n
    lines.append("s_setprio 0")
    return lines, m


def main8h(out):
    for D8 in (128, 64):
        tag = "" if D8 == 128 else "64"
        for dt in ("bf16", "f16"):
            for dma in (False, True):
                for T6 in range(6):
                    lines, m = gen_block8h(T6, dt, dma, D8)
                    out.append(f"#define RSA_K5F8H{tag}_BLOCK{'D' if dma else ''}_{dt.upper()}_T{T6} \\")
                    out.append(" \\\n".join(c_string(lines).split("\n")))
                    out.append("")
        m = Map8H(D8)
        outs = [f'"+{{{vr(m.O + 16 * d, 16)}}}"(o[{d}])' for d in range(m.DT)]
        outs += [f'"+{{{vr(m.SA, 16)}}}"(SA[0])', f'"+{{{vr(m.SA + 16, 16)}}}"(SA[1])', f'"+{{{vr(m.SB, 16)}}}"(SB[0])',
                 f'"+{{{vr(m.SB + 16, 16)}}}"(SB[1])', f'"+{{{vr(m.LACC, 4)}}}"(lacc)', '[mx] "=&v"(mx)']
        ins = [f'"{{{vr(m.Q + 16 * i, 16)}}}"(qv[{i}])' for i in range(m.KS // 4)]     # Q fragments of four k-steps per 16-register value
        ins += [f'"{{{vr(m.MB, 16)}}}"(mblk)', f'"{{{vr(m.SC)}}}"(sca)', f'"{{{vr(m.SC + 1)}}}"(scb)', f'"{{{vr(m.VA, 2)}}}"(vah)',
                f'"{{{vr(m.KB)}}}"(kah)', f'"{{{vr(m.ONES, 8)}}}"(onesv)']
        out.append(f"#define RSA_K5F8H{tag}_OPS : {', '.join(outs)} : {', '.join(ins)}")
        insd = ins + [f'"{{{vr(m.DK, m.NPK)}}}"(dk)', f'"{{{vr(m.DV, m.NPV)}}}"(dv)', '[kb16] "s"(kb16)', '[vsrc] "s"(vsrc)', '[ldsw] "s"(ldsw)']
        out.append(f"#define RSA_K5F8H{tag}_OPSD : {', '.join(outs)} : {', '.join(insd)}")
        out.append(f"#define RSA_K5F8H{tag}_CLOBBER " + ", ".join(f'"v{r}"' for r in range(m.tmp0, m.tmp1)))
        out.append(f"// pv form, head dim {D8}: O v[0:{m.Q - 1}], Q v[{m.Q}:{m.SA - 1}], SA v[{m.SA}:{m.SB - 1}], SB v[{m.SB}:{m.MB - 1}], reference block "
                   f"v[{m.MB}:{m.LACC - 1}], l v[{m.LACC}:{m.LACC + 3}], temporaries v[{m.tmp0}:{m.tmp1 - 1}], scales v[{m.SC}:{m.SC + 1}], "
                   f"V / K addresses v[{m.VA}:{m.KB}], ones v[{m.ONES}:{m.ONES + 7}], DMA lane offsets v[{m.DK}:{m.end - 1}]")

if __name__ == "__main__":
    main()
