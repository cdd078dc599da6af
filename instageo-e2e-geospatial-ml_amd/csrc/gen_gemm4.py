#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loops of the 4-wave, one-wave-per-SIMD kernels: gemm4.hip (forward linears and data gradients,
pritvhi.py:446-456; plain and paired-split forms), gemm4w_kernel in gemm8w.hip (grouped weight gradients: the "w" / "wp" forms below) and
conv4_kernel in conv8.hip (the decode head's 3 x 3 convolutions, model.py:349-390: the "c" / "cp" forms).  The first section documents the
forward form; every other form is described where its functions start.

hipcc cannot hold 256 accumulators + 128 fragment registers of a 128 x 128-per-wave tile without spilling (rounds 2 and 5), so the K-loop of a
tile is ONE inline-asm block with hand-assigned registers; this script writes it (`gemm4_gen.inc`: string macros for the prologue and the tile
loop + the accumulator read-out helpers of the C++ epilogues).  Run by the Makefile; the output is not committed.

Structure of a tile (256 x 256 x K, BK = 64, 4 waves, wave (wr, wc) owns C[wr 128 ..][wc 128 ..] as 8 x 8 MFMA 16x16x32 accumulators = a0..a255):
  LDS (160 KiB): a ring of THREE A slots (256 rows x 128 B = 32 KiB each, at 0 / 32 / 64 KiB) and TWO B stages (96 / 128 KiB); rows swizzled
  (16-byte chunk ^= row & 7).  The slot of A K-tile g (counted over the workgroup's whole tile sequence) is g mod 3: three scalar registers
  hold the slot offsets of K-tiles kt, kt + 1, kt + 2 and rotate once per iteration (they travel through the asm block's in/out operands from
  tile to tile); B's stage is kt & 1 (K-tiles per tile are even), so the loop body is unrolled by two.  The slot whose K-tile has just been
  consumed is free until the next tile's first iteration: it is the epilogue's staging buffer.
  Fragments: two register sets of 16 x 4 VGPRs (8 A row blocks + 8 B column blocks of ONE 32-deep k-substep); while the 64 MFMAs of a
  substep run on one set the other set is read from LDS.  Iteration kt (B stage p = kt & 1):
      H1: 64 MFMAs on set 0 (K-tile kt, substep 0)      | 16 ds_read_b128 -> set 1 (K-tile kt, substep 1)
                                                        | 8 LDS-DMA issues: A of K-tile kt + 2 -> the slot K-tile kt - 1 left
      sync: lgkmcnt(0), vmcnt(8), s_barrier             (everything but the 8 A pieces just issued has landed, for every wave;
                                                         B stage p and A slot kt are dead for every wave)
      H2: 64 MFMAs on set 1                             | 16 ds_read_b128 -> set 0 (K-tile kt + 1, substep 0)
                                                        | 8 LDS-DMA issues: B of K-tile kt + 2 -> stage p
  One barrier per K-tile.  The 16 LDS-DMA issues of a wave are spread over the WHOLE iteration (one per 8 MFMAs = 128 cycles; four waves: one
  per 32 cycles on the CU's texture-address path, which takes ~16 cycles per 1-KiB piece) -- with two 64-KiB stages all 16 had to go out in
  H2 (one per 16 cycles CU-wide: saturated) -- and A, whose rows come from HBM when no other tile of the XCD has touched them yet, has 1.5
  iterations to land instead of 0.5-1.  The DMA stream crosses the tile boundary (the last two iterations fetch K-tiles 0 and 1 of the
  workgroup's next tile, whose row clamp has its own offset registers).
  History (profiles/r06_gemm4_*.txt): the first form had two 64-KiB stages, all 16 issues in H2 and an L2 prefetch of the activation rows one
  K-tile ahead (a global_load_dword per wave and K-tile: qkv 130 -> 122 us on that form); with the ring the prefetch LOSES (d_fc1 143 -> 161 us:
  vmcnt retires in order, the HBM-latency prefetch holds back the L2-hit pieces behind it) and is gone.
"""
import sys

# ---- register map (explicit; everything here is in the asm blocks' clobber lists) ----
V_TMP = 94           # v94, v95: scratch
V_FA = 96            # v96, v97: A fragment base of k-substep 0 / 1 (without the slot offset)
V_FB = 98            # v98..v101: B fragment bases [stage][substep]
V_AC = 102           # v102: A fragment base of the half being read (V_FA[s] + slot offset)
V_OFFA_N = 104       # v104..v111: DMA offsets of A, NEXT tile's row clamp
V_OFFA = 112         # v112..v119: DMA offsets of A, this tile
V_OFFB = 120         # v120..v127: DMA offsets of B
V_SET = 128          # v128..v255: set q at 128 + 64 q: A blocks [0..7] x 4, then B blocks [0..7] x 4
V_LO, V_HI = 94, 255
S_APTR, S_BPTR, S_CNT, S_LDSW = 70, 72, 74, 75   # s[70:71], s[72:73], s74, s75
S_A0, S_A1, S_A2, S_ADST, S_T = 78, 79, 80, 81, 82          # slot offsets of A K-tiles kt, kt + 1, kt + 2; DMA destination base; scratch
S_LO, S_HI = 70, 82

A_SLOT = 32768       # three A slots at 0, 32 KiB, 64 KiB
B_BASE = 98304       # two B stages at 96 KiB, 128 KiB
B_STAGE = 32768


def afrag(q, mi):
    b = V_SET + 64 * q + 4 * mi
    return f"v[{b}:{b + 3}]"


def bfrag(q, ni):
    b = V_SET + 64 * q + 32 + 4 * ni
    return f"v[{b}:{b + 3}]"


def acc(mi, ni):
    b = (mi * 8 + ni) * 4
    return f"a[{b}:{b + 3}]"


def fbase_b(stage, s):
    return f"v{V_FB + 2 * stage + s}"


class Stream:
    def __init__(self):
        self.lines = []

    def e(self, s):
        self.lines.append(s)

    def text(self):
        return "\n".join(f'    "{l}\\n\\t"' for l in self.lines)


def reads(bstage, s, q):
    """the 16 fragment reads of k-substep s into set q: A from the slot V_AC points into (set by the half's preamble), B from `bstage`"""
    out = []
    for i in range(8):
        out.append(f"ds_read_b128 {afrag(q, i)}, v{V_AC} offset:{i * 2048}")
    for i in range(8):
        out.append(f"ds_read_b128 {bfrag(q, i)}, {fbase_b(bstage, s)} offset:{i * 2048}")
    return out


def dmas_a(next_tile):
    """8 x (M0 write, LDS-DMA issue): the wave's 64 rows of an A K-tile (8 rows x 128 B per piece) into the slot S_ADST points at"""
    offa = V_OFFA_N if next_tile else V_OFFA
    return [(f"s_add_u32 m0, s{S_ADST}, {i * 1024}", f"global_load_lds_dwordx4 v{offa + i}, s[{S_APTR}:{S_APTR + 1}]") for i in range(8)]


def dmas_b(bstage):
    return [(f"s_add_u32 m0, s{S_LDSW}, {B_BASE + bstage * B_STAGE + i * 1024}", f"global_load_lds_dwordx4 v{V_OFFB + i}, s[{S_BPTR}:{S_BPTR + 1}]")
            for i in range(8)]


def advance(ptr):
    return [f"s_add_u32 s{ptr}, s{ptr}, 128", f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0"]


def half(st, q, first, rd, dm, cfg):
    """64 MFMAs on set q with the reads `rd` (16) and DMA pairs `dm` (0 or 16) woven in behind them"""
    rd_every, rd_at0 = cfg["rd_every"], cfg["rd_at"]
    dm_every, dm_at0 = cfg["dm_every"], cfg["dm_at"]
    extra = {j: [] for j in range(64)}
    for k, r in enumerate(rd):
        extra[min(63, rd_at0 + k * rd_every)].append(r)
    for k, (m0w, ld) in enumerate(dm):
        j = min(62, dm_at0 + k * dm_every)
        extra[j].append(m0w)       # M0 write behind MFMA j, the DMA behind MFMA j + 1 (one instruction between them: the wait state M0 needs)
        extra[j + 1].insert(0, ld)
    j = 0
    order = [(mi, ni) for ni in range(8) for mi in range(8)] if cfg["order"] == "ni" else [(mi, ni) for mi in range(8) for ni in range(8)]
    for mi, ni in order:
        c = "0" if first else acc(mi, ni)
        st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {bfrag(q, ni)}, {afrag(q, mi)}, {c}")
        for x in extra[j]:
            st.e(x)
        j += 1


def skew(st, cfg, tag):
    """optional start skew of the waves behind the barrier (wave w delays w x 16 cycles): spreads the 4 waves' DMA issues over the TA"""
    n = cfg.get("skew", 0)
    if not n:
        return
    for k in range(3):
        st.e(f"s_cmp_lt_u32 %[wave], {3 - k}")
        st.e(f"s_cbranch_scc1 L_skew_{tag}_{k}_%=")
        for _ in range(n):
            st.e("s_nop 15")
        st.e(f"L_skew_{tag}_{k}_%=:")


def iteration(st, p, first, next_tile, cfg, tag):
    # timing ablations (garbage results): abl_rd = no fragment reads, abl_dma = no LDS-DMA issues, abl_vmw = no vmcnt wait, abl_bar = no barrier
    no_rd, no_dm = cfg.get("abl_rd"), cfg.get("abl_dma")
    # ---- H1: K-tile kt substep 0 | reads of substep 1 | A pieces of K-tile kt + 2 -> slot S_A2
    st.e(f"v_add_u32 v{V_AC}, s{S_A0}, v{V_FA + 1}")
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    half(st, 0, first, [] if no_rd else reads(p, 1, 1), [] if no_dm else dmas_a(next_tile), cfg)
    for a in advance(S_APTR):
        st.e(a)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        st.e(f"s_waitcnt vmcnt({0 if no_dm else 8})")   # this half's 8 A pieces may stay in flight
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    skew(st, cfg, tag)
    # ---- H2: substep 1 | reads of K-tile kt + 1 substep 0 (slot S_A1, B stage p ^ 1) | B pieces of K-tile kt + 2 -> stage p | prefetch
    st.e(f"v_add_u32 v{V_AC}, s{S_A1}, v{V_FA}")
    half(st, 1, False, [] if no_rd else reads(p ^ 1, 0, 0), [] if no_dm else dmas_b(p), cfg)   # only H1 of a tile's first K-tile starts from 0
    for a in advance(S_BPTR):
        st.e(a)
    # rotate the A ring: (kt, kt + 1, kt + 2) <- (kt + 1, kt + 2, the slot kt has just left)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def setup(st):
    """offset / base registers from the block's inputs"""
    for i in range(8):
        st.e(f"v_add_u32 v{V_TMP}, {8 * i}, %[rowv]")
        st.e(f"v_min_u32 v{V_TMP + 1}, %[vrc], v{V_TMP}")
        st.e(f"v_mad_u32_u24 v{V_OFFA + i}, v{V_TMP + 1}, %[lda2], %[c16]")
        st.e(f"v_min_u32 v{V_TMP + 1}, %[vrn], v{V_TMP}")
        st.e(f"v_mad_u32_u24 v{V_OFFA_N + i}, v{V_TMP + 1}, %[lda2], %[c16]")
        st.e(f"v_mad_u32_u24 v{V_OFFB + i}, v{V_TMP}, %[ldb2], %[c16]")
    st.e(f"v_mov_b32 v{V_FA}, %[fa]")
    st.e(f"v_xor_b32 v{V_FA + 1}, 64, %[fa]")
    for b in (0, 1):
        st.e(f"v_add_u32 {fbase_b(b, 0)}, {B_BASE + b * B_STAGE}, %[fb]")
        st.e(f"v_xor_b32 {fbase_b(b, 1)}, 64, {fbase_b(b, 0)}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")


def gen_prologue():
    """first tile of a workgroup: A K-tiles 0 and 1 -> slots a0, a1; B K-tiles 0 and 1 -> stages 0 and 1 (inputs: aptr / bptr = the tile's
    operand bases, vrc = its row clamp)"""
    st = Stream()
    setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for m0w, ld in dmas_a(False) + dmas_b(k):
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in advance(S_APTR) + advance(S_BPTR):
            st.e(a)
    return st.text()


def gen_tile(cfg):
    st = Stream()
    setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")   # K-tile 2 of this tile
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")                  # middle pairs: nk / 2 - 2
    st.e("s_waitcnt vmcnt(0)")                             # K-tiles 0, 1 (and the previous epilogue's stores)
    st.e("s_barrier")                                      # ... for every wave; every wave has left the staging slot (= S_A2)
    st.e(f"v_add_u32 v{V_AC}, s{S_A0}, v{V_FA}")
    for r in reads(0, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    # first pair: the accumulators start from the inline constant 0
    iteration(st, 0, True, False, cfg, "f0")
    iteration(st, 1, False, False, cfg, "f1")
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    iteration(st, 0, False, False, cfg, "m0")
    iteration(st, 1, False, False, cfg, "m1")
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    # last pair: the DMA stream moves on to K-tiles 0, 1 of the next tile
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[anext]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bnext]")
    iteration(st, 0, False, True, cfg, "l0")
    iteration(st, 1, False, True, cfg, "l1")
    st.e(f"s_mov_b32 %[a0], s{S_A0}")   # ring state for the next tile; a2 = the slot this tile's last K-tile has left = the epilogue's staging
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")   # the last MFMAs retire before the epilogue's v_accvgpr_read (the compiler's hazard recognizer does not see into this block)
    st.e("s_nop 15")
    return st.text()


# =====================================================================================================================================
# PAIRED form (the split precision mode bf16x3: value = hi + lo, products hi hi + hi lo + lo hi).  A K-tile covers 32 reduction elements; its
# 128-byte LDS row is [hi k..k+31 | lo k..k+31] (the lo lanes of an LDS-DMA piece carry the hi -> lo tensor distance in their 32-bit offset, as
# gemm8.hip's NSEG = 2), so k-substep 0 of the fragment reads is hi and substep 1 is lo, and a K-tile is THREE products of 64 MFMAs:
#     P1  acc += Bhi x Ahi      | 8 reads: Blo of this K-tile                         | 8 LDS-DMA issues: A of K-tile kt + 2
#     sync lgkmcnt(0), vmcnt(8), s_barrier
#     P2  acc += Bhi x Alo      | 8 reads: Ahi of K-tile kt + 1 -> the OTHER Ahi buffer | 4 LDS-DMA issues: B of K-tile kt + 2 (first half)
#     P3  acc += Blo x Ahi      | 16 reads: Bhi, Alo of K-tile kt + 1                   | 4 LDS-DMA issues (second half)
# Five quarter sets of 8 x 4 VGPRs (Ahi twice -- it is live from P1 to P3, so the next K-tile's copy needs a buffer of its own; the two
# alternate with the K-tile parity, which the loop is unrolled by anyway), 192 MFMAs for the 64 KiB a K-tile moves: 21 B/clk/CU at full MFMA
# rate instead of the plain form's 32 -- the paired form is NOT bound by the L2 -> LDS feed.
PV_TMP, PV_FA, PV_FB, PV_AC = 54, 56, 58, 62
PV_OFFA_N, PV_OFFA, PV_OFFB = 64, 72, 80
PQ_AHI, PQ_ALO, PQ_BHI, PQ_BLO = (96, 128), 160, 192, 224
PV_LO = 54


def pq(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def p_fbase_b(stage, s):
    return f"v{PV_FB + 2 * stage + s}"


def p_reads_a(base):
    return [f"ds_read_b128 {pq(base, i)}, v{PV_AC} offset:{i * 2048}" for i in range(8)]


def p_reads_b(base, bstage, s):
    return [f"ds_read_b128 {pq(base, i)}, {p_fbase_b(bstage, s)} offset:{i * 2048}" for i in range(8)]


def p_dmas_a(next_tile):
    offa = PV_OFFA_N if next_tile else PV_OFFA
    return [(f"s_add_u32 m0, s{S_ADST}, {i * 1024}", f"global_load_lds_dwordx4 v{offa + i}, s[{S_APTR}:{S_APTR + 1}]") for i in range(8)]


def p_dmas_b(bstage):
    return [(f"s_add_u32 m0, s{S_LDSW}, {B_BASE + bstage * B_STAGE + i * 1024}", f"global_load_lds_dwordx4 v{PV_OFFB + i}, s[{S_BPTR}:{S_BPTR + 1}]")
            for i in range(8)]


def p_advance(ptr):
    return [f"s_add_u32 s{ptr}, s{ptr}, 64", f"s_addc_u32 s{ptr + 1}, s{ptr + 1}, 0"]


def product(st, bbase, abase, first, rd, dm, cfg, dm_every, dm_at0):
    """64 MFMAs acc += B[bbase] x A[abase] with reads and DMA pairs woven in"""
    rd_every, rd_at0 = cfg["rd_every"], cfg["rd_at"]
    extra = {j: [] for j in range(64)}
    for k, r in enumerate(rd):
        extra[min(63, rd_at0 + k * rd_every)].append(r)
    for k, (m0w, ld) in enumerate(dm):
        j = min(62, dm_at0 + k * dm_every)
        extra[j].append(m0w)
        extra[j + 1].insert(0, ld)
    j = 0
    for ni in range(8):
        for mi in range(8):
            c = "0" if first else acc(mi, ni)
            st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {pq(bbase, ni)}, {pq(abase, mi)}, {c}")
            for x in extra[j]:
                st.e(x)
            j += 1


def p_iteration(st, p, first, next_tile, cfg):
    ahi_cur, ahi_nxt = PQ_AHI[p], PQ_AHI[p ^ 1]
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    product(st, PQ_BHI, ahi_cur, first, p_reads_b(PQ_BLO, p, 1), p_dmas_a(next_tile), cfg, 8, cfg["dm_at"])
    for a in p_advance(S_APTR):
        st.e(a)
    st.e("s_waitcnt lgkmcnt(0)")
    st.e("s_waitcnt vmcnt(8)")
    st.e("s_barrier")
    st.e(f"v_add_u32 v{PV_AC}, s{S_A1}, v{PV_FA}")        # A of K-tile kt + 1, hi half
    bd = p_dmas_b(p)
    product(st, PQ_BHI, PQ_ALO, False, p_reads_a(ahi_nxt), bd[:4], cfg, 16, cfg["dm_at"])
    st.e(f"v_add_u32 v{PV_AC}, s{S_A1}, v{PV_FA + 1}")    # ... lo half (the reads above have been issued: the address is consumed at issue)
    product(st, PQ_BLO, ahi_cur, False, p_reads_b(PQ_BHI, p ^ 1, 0) + p_reads_a(PQ_ALO), bd[4:], cfg, 16, cfg["dm_at"])
    for a in p_advance(S_BPTR):
        st.e(a)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def p_setup(st):
    for i in range(8):
        st.e(f"v_add_u32 v{PV_TMP}, {8 * i}, %[rowv]")
        st.e(f"v_min_u32 v{PV_TMP + 1}, %[vrc], v{PV_TMP}")
        st.e(f"v_mad_u32_u24 v{PV_OFFA + i}, v{PV_TMP + 1}, %[lda2], %[c16a]")
        st.e(f"v_min_u32 v{PV_TMP + 1}, %[vrn], v{PV_TMP}")
        st.e(f"v_mad_u32_u24 v{PV_OFFA_N + i}, v{PV_TMP + 1}, %[lda2], %[c16a]")
        st.e(f"v_mad_u32_u24 v{PV_OFFB + i}, v{PV_TMP}, %[ldb2], %[c16b]")
    st.e(f"v_mov_b32 v{PV_FA}, %[fa]")
    st.e(f"v_xor_b32 v{PV_FA + 1}, 64, %[fa]")
    for b in (0, 1):
        st.e(f"v_add_u32 {p_fbase_b(b, 0)}, {B_BASE + b * B_STAGE}, %[fb]")
        st.e(f"v_xor_b32 {p_fbase_b(b, 1)}, 64, {p_fbase_b(b, 0)}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")


def gen_p_prologue():
    st = Stream()
    p_setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for m0w, ld in p_dmas_a(False) + p_dmas_b(k):
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in p_advance(S_APTR) + p_advance(S_BPTR):
            st.e(a)
    return st.text()


def gen_p_tile(cfg):
    st = Stream()
    p_setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")   # K-tile 2 of this tile
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    st.e(f"v_add_u32 v{PV_AC}, s{S_A0}, v{PV_FA}")
    for r in p_reads_a(PQ_AHI[0]):
        st.e(r)
    st.e(f"v_add_u32 v{PV_AC}, s{S_A0}, v{PV_FA + 1}")
    for r in p_reads_a(PQ_ALO) + p_reads_b(PQ_BHI, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    p_iteration(st, 0, True, False, cfg)
    p_iteration(st, 1, False, False, cfg)
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    p_iteration(st, 0, False, False, cfg)
    p_iteration(st, 1, False, False, cfg)
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[anext]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bnext]")
    p_iteration(st, 0, False, True, cfg)
    p_iteration(st, 1, False, True, cfg)
    st.e(f"s_mov_b32 %[a0], s{S_A0}")
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")
    st.e("s_nop 15")
    return st.text()


def p_clobbers():
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    used = set(range(PV_TMP, PV_TMP + 2)) | set(range(PV_FA, PV_FA + 2)) | set(range(PV_FB, PV_FB + 4)) | {PV_AC}
    used |= set(range(PV_OFFA_N, PV_OFFA_N + 8)) | set(range(PV_OFFA, PV_OFFA + 8)) | set(range(PV_OFFB, PV_OFFB + 8)) | set(range(96, 256))
    c += [f'"v{i}"' for i in sorted(used)]   # exactly the registers the block names: the compiler keeps the rest (v63, v88..v95 among them)
    c += [f'"s{i}"' for i in range(S_LO, S_HI + 1)]
    return ", ".join(c)


# =====================================================================================================================================
# WEIGHT-GRADIENT form (gemm4w, used by gemm8w.hip for the plain 256 x 256 linears): dW[n][k] += sum_t dy[t][n] x[t][k].  The reduction runs over
# tokens, so both operands are reduce-strided: an operand tile in LDS is two halves of [64 tokens][128 columns] (256-byte rows), filled by
# LDS-DMA in full 256-byte source rows through a buffer descriptor (tokens past the segment's end read as zeros), and the MFMA fragments come
# from ds_read_b64_tr_b16 (hardware transpose; the image, its chunk-pair swizzle and the fragment addressing are gemm8w.hip's).  Wave (wr, wc)
# reads A half wr (the n side: dy) and B half wc (the k side: x).  Same ring (three A slots, two B stages), same iteration shape as the
# forward form; a fragment is two 8-byte reads, so a half-iteration carries 32 reads.  The unit of work is a SEGMENT (output tile x token
# range) of the host-built plan; the DMA stream crosses segment boundaries the way the forward form crosses tiles.
WV_TMP = 62                     # v62, v63
WV_ABASE, WV_ACUR = 64, 72      # v64..v71: A fragment address of block mi without the slot; v72..v79: with the slot of the half being read
WV_B = 80                       # v80..v95: B fragment addresses [stage][ni]
WV_OFFA, WV_OFFA_N, WV_OFFB, WV_OFFB_N = 96, 104, 112, 120   # DMA offsets (h, i) of A / B, this and the next segment
WV_LO = 62
WS_RA, WS_RB = 84, 88           # s[84:87], s[88:91]: buffer descriptors of the A / B stream
WS_STEPA, WS_STEPB = 92, 93     # bytes per K-tile (64 tokens x row pitch)
WS_LO, WS_HI = 70, 93


def w_frag(q, is_b, i):
    b = V_SET + 64 * q + 32 * is_b + 4 * i
    return b


def w_reads(q, s, bstage):
    out = []
    for i in range(8):
        d = w_frag(q, 0, i)
        out.append(f"ds_read_b64_tr_b16 v[{d}:{d + 1}], v{WV_ACUR + i} offset:{s * 8192}")
        out.append(f"ds_read_b64_tr_b16 v[{d + 2}:{d + 3}], v{WV_ACUR + i} offset:{s * 8192 + 1024}")
    for i in range(8):
        d = w_frag(q, 1, i)
        out.append(f"ds_read_b64_tr_b16 v[{d}:{d + 1}], v{WV_B + 8 * bstage + i} offset:{s * 8192}")
        out.append(f"ds_read_b64_tr_b16 v[{d + 2}:{d + 3}], v{WV_B + 8 * bstage + i} offset:{s * 8192 + 1024}")
    return out


def w_dmas_a(next_seg):
    off = WV_OFFA_N if next_seg else WV_OFFA
    return [(f"s_add_u32 m0, s{S_ADST}, {h * 16384 + i * 1024}", f"buffer_load_dwordx4 v{off + h * 4 + i}, s[{WS_RA}:{WS_RA + 3}], 0 offen lds")
            for h in range(2) for i in range(4)]


def w_dmas_b(bstage, next_seg):
    off = WV_OFFB_N if next_seg else WV_OFFB
    return [(f"s_add_u32 m0, s{S_LDSW}, {B_BASE + bstage * B_STAGE + h * 16384 + i * 1024}",
             f"buffer_load_dwordx4 v{off + h * 4 + i}, s[{WS_RB}:{WS_RB + 3}], 0 offen lds") for h in range(2) for i in range(4)]


def w_advance(rs, step):
    """next K-tile: base += step, num_records -= step (clamped at 0: the tokens past the segment's end read as zeros)"""
    return [f"s_add_u32 s{rs}, s{rs}, s{step}", f"s_addc_u32 s{rs + 1}, s{rs + 1}, 0",
            f"s_sub_u32 s{rs + 2}, s{rs + 2}, s{step}", f"s_cselect_b32 s{rs + 2}, 0, s{rs + 2}"]


def w_half(st, q, first, rd, dm, cfg):
    extra = {j: [] for j in range(64)}
    if cfg.get("abl_rd"):   # timing ablations (garbage results), as in the forward form
        rd = []
    if cfg.get("abl_dma"):
        dm = []
    for k, r in enumerate(rd):
        extra[min(63, cfg["w_rd_at"] + k * cfg["w_rd_num"] // cfg["w_rd_den"])].append(r)
    for k, (m0w, ld) in enumerate(dm):
        j = min(62, cfg["w_dm_at"] + k * cfg["w_dm_every"])
        extra[j].append(m0w)
        extra[j + 1].insert(0, ld)
    j = 0
    for ni in range(8):
        for mi in range(8):
            a = w_frag(q, 0, mi)
            b = w_frag(q, 1, ni)
            c = "0" if first else acc(mi, ni)
            st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, v[{b}:{b + 3}], v[{a}:{a + 3}], {c}")
            for x in extra[j]:
                st.e(x)
            j += 1


def w_iteration(st, p, first, next_seg, cfg):
    for i in range(8):
        st.e(f"v_add_u32 v{WV_ACUR + i}, s{S_A0}, v{WV_ABASE + i}")
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    w_half(st, 0, first, w_reads(1, 1, p), w_dmas_a(next_seg), cfg)
    for a in w_advance(WS_RA, WS_STEPA):
        st.e(a)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        st.e("s_waitcnt vmcnt(8)")
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    for i in range(8):
        st.e(f"v_add_u32 v{WV_ACUR + i}, s{S_A1}, v{WV_ABASE + i}")
    w_half(st, 1, False, w_reads(0, 0, p ^ 1), w_dmas_b(p, next_seg), cfg)
    for a in w_advance(WS_RB, WS_STEPB):
        st.e(a)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def w_setup(st, with_next):
    # fragment addresses: toff[i] = tbase + ((i << 5) ^ rkey5); A: + the wave's half (fah); B: + the wave's half + stage base (fbh)
    for i in range(8):
        st.e(f"v_xor_b32 v{WV_TMP}, {i << 5}, %[rkey5]")
        st.e(f"v_add_u32 v{WV_TMP}, v{WV_TMP}, %[tbase]")
        st.e(f"v_add_u32 v{WV_ABASE + i}, v{WV_TMP}, %[fah]")
        st.e(f"v_add_u32 v{WV_TMP}, v{WV_TMP}, %[fbh]")
        st.e(f"v_add_u32 v{WV_B + i}, {B_BASE}, v{WV_TMP}")
        st.e(f"v_add_u32 v{WV_B + 8 + i}, {B_BASE + B_STAGE}, v{WV_TMP}")
    # DMA offsets: piece (h, i) of this wave = token rows 16 w + 4 i + (lane >> 4), half h: row x pitch + 256 h + 16 x source chunk
    # (the chunk-pair key of LDS rows 8 .. 15 of a 16-row group has bit 2 set: source chunk ^ 8, byte offset ^ 128)
    st.e(f"v_xor_b32 v{WV_TMP + 1}, 128, %[lch0]")
    sets = [(WV_OFFA, "%[lda2]"), (WV_OFFB, "%[ldb2]")]
    if with_next:
        sets += [(WV_OFFA_N, "%[lda2n]"), (WV_OFFB_N, "%[ldb2n]")]
    for base, ld in sets:
        for h in range(2):
            for i in range(4):
                st.e(f"v_add_u32 v{WV_TMP}, {4 * i}, %[rowv]")
                st.e(f"v_mad_u32_u24 v{base + h * 4 + i}, v{WV_TMP}, {ld}, {f'v{WV_TMP + 1}' if i >= 2 else '%[lch0]'}")
                if h:
                    st.e(f"v_add_u32 v{base + h * 4 + i}, 256, v{base + h * 4 + i}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")
    w_desc(st, "")


def w_desc(st, sfx, shift=6):
    """descriptors of the two streams: 64-bit base (stride 0), bytes, raw-buffer flags; bytes per K-tile = 64 (paired: 32) x row pitch"""
    st.e(f"s_mov_b64 s[{WS_RA}:{WS_RA + 1}], %[ra{sfx}]")
    st.e(f"s_mov_b32 s{WS_RA + 2}, %[na{sfx}]")
    st.e(f"s_mov_b32 s{WS_RA + 3}, 0x00020000")
    st.e(f"s_mov_b64 s[{WS_RB}:{WS_RB + 1}], %[rb{sfx}]")
    st.e(f"s_mov_b32 s{WS_RB + 2}, %[nb{sfx}]")
    st.e(f"s_mov_b32 s{WS_RB + 3}, 0x00020000")
    st.e(f"s_lshl_b32 s{WS_STEPA}, %[lda2{sfx}], {shift}")
    st.e(f"s_lshl_b32 s{WS_STEPB}, %[ldb2{sfx}], {shift}")


def gen_w_prologue():
    """first segment of a workgroup: K-tiles 0 and 1 (ra / rb: descriptors at the segment's first token)"""
    st = Stream()
    w_setup(st, False)
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for m0w, ld in w_dmas_a(False) + w_dmas_b(k, False):
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in w_advance(WS_RA, WS_STEPA) + w_advance(WS_RB, WS_STEPB):
            st.e(a)
    return st.text()


def gen_w_seg(cfg, short=False):
    """one segment: ra / rb stand on its K-tile 2, ran / rbn on the next segment's first token.  short: a segment of ONE K-tile pair (the
    walking remainder of a plan cuts such pieces): its only pair is the first and the last one"""
    st = Stream()
    w_setup(st, True)
    if not short:
        st.e(f"s_mov_b32 s{S_CNT}, %[npair]")
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    for i in range(8):
        st.e(f"v_add_u32 v{WV_ACUR + i}, s{S_A0}, v{WV_ABASE + i}")
    for r in w_reads(0, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    if not short:
        w_iteration(st, 0, True, False, cfg)
        w_iteration(st, 1, False, False, cfg)
        st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
        st.e("s_cbranch_scc1 L_last_%=")
        st.e("L_loop_%=:")
        w_iteration(st, 0, False, False, cfg)
        w_iteration(st, 1, False, False, cfg)
        st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        st.e("s_cbranch_scc1 L_loop_%=")
        st.e("L_last_%=:")
    w_desc(st, "n")   # the DMA stream moves on to the next segment
    w_iteration(st, 0, short, True, cfg)
    w_iteration(st, 1, False, True, cfg)
    st.e(f"s_mov_b32 %[a0], s{S_A0}")
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")
    st.e("s_nop 15")
    return st.text()


def w_clobbers():
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"v{i}"' for i in range(WV_LO, V_HI + 1)]
    c += [f'"s{i}"' for i in range(WS_LO, WS_HI + 1)]
    return ", ".join(c)


# ---- paired weight-gradient form (bf16x3): a K-tile covers 32 tokens, the 64 LDS rows of a half-tile are [hi tokens 0-31 | lo tokens 0-31]
# (waves 0, 1 fetch from the hi tensors, waves 2, 3 the same token rows from the lo tensors: wave-uniform, the descriptor base selects), so
# k-substep 0 of the fragment reads is hi and substep 1 is lo, and a K-tile is the three products of the forward paired form:
#     P1  acc += Bhi x Ahi      | 16 reads: Blo of this K-tile                              | 8 LDS-DMA issues: A of K-tile kt + 2
#     sync lgkmcnt(0), vmcnt(8), s_barrier
#     P2  acc += Bhi x Alo      | 16 reads: Ahi of K-tile kt + 1 -> the other Ahi buffer    | 4 LDS-DMA issues: B of K-tile kt + 2
#     P3  acc += Blo x Ahi      | 32 reads: Bhi, Alo of K-tile kt + 1                       | 4 LDS-DMA issues
# The next segment's DMA offsets are computed in place before the last pair (no second offset set: the five quarter sets take v96..v255).
WPV_TMP, WPV_ABASE, WPV_ACUR, WPV_B, WPV_OFFA, WPV_OFFB = 46, 48, 56, 64, 80, 88
WPV_LO = 46


def wp_reads_a(base, s):
    out = []
    for i in range(8):
        out.append(f"ds_read_b64_tr_b16 v[{base + 4 * i}:{base + 4 * i + 1}], v{WPV_ACUR + i} offset:{s * 8192}")
        out.append(f"ds_read_b64_tr_b16 v[{base + 4 * i + 2}:{base + 4 * i + 3}], v{WPV_ACUR + i} offset:{s * 8192 + 1024}")
    return out


def wp_reads_b(base, bstage, s):
    out = []
    for i in range(8):
        out.append(f"ds_read_b64_tr_b16 v[{base + 4 * i}:{base + 4 * i + 1}], v{WPV_B + 8 * bstage + i} offset:{s * 8192}")
        out.append(f"ds_read_b64_tr_b16 v[{base + 4 * i + 2}:{base + 4 * i + 3}], v{WPV_B + 8 * bstage + i} offset:{s * 8192 + 1024}")
    return out


def wp_dmas_a():
    return [(f"s_add_u32 m0, s{S_ADST}, {h * 16384 + i * 1024}", f"buffer_load_dwordx4 v{WPV_OFFA + h * 4 + i}, s[{WS_RA}:{WS_RA + 3}], 0 offen lds")
            for h in range(2) for i in range(4)]


def wp_dmas_b(bstage):
    return [(f"s_add_u32 m0, s{S_LDSW}, {B_BASE + bstage * B_STAGE + h * 16384 + i * 1024}",
             f"buffer_load_dwordx4 v{WPV_OFFB + h * 4 + i}, s[{WS_RB}:{WS_RB + 3}], 0 offen lds") for h in range(2) for i in range(4)]


def wp_product(st, bbase, abase, first, rd, dm, cfg, dm_every):
    extra = {j: [] for j in range(64)}
    if cfg.get("abl_rd"):
        rd = []
    if cfg.get("abl_dma"):
        dm = []
    for k, r in enumerate(rd):
        extra[min(63, k * 64 // max(len(rd), 1) * cfg["wp_rd_num"] // cfg["wp_rd_den"])].append(r)
    for k, (m0w, ld) in enumerate(dm):
        j = min(62, cfg["w_dm_at"] + k * dm_every)
        extra[j].append(m0w)
        extra[j + 1].insert(0, ld)
    j = 0
    for ni in range(8):
        for mi in range(8):
            c = "0" if first else acc(mi, ni)
            st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {pq(bbase, ni)}, {pq(abase, mi)}, {c}")
            for x in extra[j]:
                st.e(x)
            j += 1


def wp_iteration(st, p, first, cfg):
    ahi_cur, ahi_nxt = PQ_AHI[p], PQ_AHI[p ^ 1]
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    for i in range(8):
        st.e(f"v_add_u32 v{WPV_ACUR + i}, s{S_A1}, v{WPV_ABASE + i}")   # A of K-tile kt + 1
    wp_product(st, PQ_BHI, ahi_cur, first, wp_reads_b(PQ_BLO, p, 1), wp_dmas_a(), cfg, 8)
    for a in w_advance(WS_RA, WS_STEPA):
        st.e(a)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        st.e("s_waitcnt vmcnt(8)")
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    bd = wp_dmas_b(p)
    wp_product(st, PQ_BHI, PQ_ALO, False, wp_reads_a(ahi_nxt, 0), bd[:4], cfg, 16)
    wp_product(st, PQ_BLO, ahi_cur, False, wp_reads_b(PQ_BHI, p ^ 1, 0) + wp_reads_a(PQ_ALO, 1), bd[4:], cfg, 16)
    for a in w_advance(WS_RB, WS_STEPB):
        st.e(a)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def wp_offsets(st, sfx):
    st.e(f"v_xor_b32 v{WPV_TMP + 1}, 128, %[lch0]")
    for base, ld in ((WPV_OFFA, f"%[lda2{sfx}]"), (WPV_OFFB, f"%[ldb2{sfx}]")):
        for h in range(2):
            for i in range(4):
                st.e(f"v_add_u32 v{WPV_TMP}, {4 * i}, %[rowv]")
                st.e(f"v_mad_u32_u24 v{base + h * 4 + i}, v{WPV_TMP}, {ld}, {f'v{WPV_TMP + 1}' if i >= 2 else '%[lch0]'}")
                if h:
                    st.e(f"v_add_u32 v{base + h * 4 + i}, 256, v{base + h * 4 + i}")


def wp_setup(st):
    for i in range(8):
        st.e(f"v_xor_b32 v{WPV_TMP}, {i << 5}, %[rkey5]")
        st.e(f"v_add_u32 v{WPV_TMP}, v{WPV_TMP}, %[tbase]")
        st.e(f"v_add_u32 v{WPV_ABASE + i}, v{WPV_TMP}, %[fah]")
        st.e(f"v_add_u32 v{WPV_TMP}, v{WPV_TMP}, %[fbh]")
        st.e(f"v_add_u32 v{WPV_B + i}, {B_BASE}, v{WPV_TMP}")
        st.e(f"v_add_u32 v{WPV_B + 8 + i}, {B_BASE + B_STAGE}, v{WPV_TMP}")
    wp_offsets(st, "")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")
    w_desc(st, "", 5)


def gen_wp_prologue():
    st = Stream()
    wp_setup(st)
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for m0w, ld in wp_dmas_a() + wp_dmas_b(k):
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in w_advance(WS_RA, WS_STEPA) + w_advance(WS_RB, WS_STEPB):
            st.e(a)
    return st.text()


def gen_wp_seg(cfg):
    """one segment of the paired form (at least four 32-token K-tiles: a plan's shortest segment is one 128-token pair)"""
    st = Stream()
    wp_setup(st)
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    for i in range(8):
        st.e(f"v_add_u32 v{WPV_ACUR + i}, s{S_A0}, v{WPV_ABASE + i}")
    for r in wp_reads_a(PQ_AHI[0], 0) + wp_reads_a(PQ_ALO, 1) + wp_reads_b(PQ_BHI, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    wp_iteration(st, 0, True, cfg)
    wp_iteration(st, 1, False, cfg)
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    wp_iteration(st, 0, False, cfg)
    wp_iteration(st, 1, False, cfg)
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    w_desc(st, "n", 5)   # the DMA stream moves on to the next segment
    wp_offsets(st, "n")
    wp_iteration(st, 0, False, cfg)
    wp_iteration(st, 1, False, cfg)
    st.e(f"s_mov_b32 %[a0], s{S_A0}")
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")
    st.e("s_nop 15")
    return st.text()


def wp_clobbers():
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"v{i}"' for i in range(WPV_LO, V_HI + 1)]
    c += [f'"s{i}"' for i in range(WS_LO, WS_HI + 1)]
    return ", ".join(c)


# =====================================================================================================================================
# CONVOLUTION form (conv4_kernel<NI> in conv8.hip: the decode head's 3 x 3 convolutions as implicit GEMMs on a 256 x 32 NI tile, wave (wr, wc)
# owns 128 rows x 16 NI columns = 8 x NI accumulator blocks; NI = 6 or 3).  The loop is the forward form's with two changes:
#   * A is GATHERED (conv8.hip): piece i of a wave is `buffer_load_dwordx4 v_off, s[desc], 0 offen lds` with
#         v_off = rowoff_i + delta(K chunk) + (invalid ? 2^31 : 0)
#     where (delta, mask bit) of the lane's 16-byte K chunk come from ONE table entry per lane and K-tile (read from LDS one iteration ahead),
#     invalid = bit `mask bit` of the row's INVERTED tap mask, and an offset beyond the descriptor's bound makes the hardware write zeros:
#     three vector instructions per piece (v_bfe_u32, v_lshl_add_u32, v_add_u32) woven in front of its issue.  The table index runs
#     cyclically over the nk K-tiles of a tile and wraps to the NEXT tile's table (the four sub-pixel phases of a ConvTranspose forward have
#     their own tables, K lengths and packed row pitches); the row offsets / masks and B offsets of the next tile take over for the last pair.
#   * B (packed weights [N][Kpad]) has 32 ni rows (ni = 6: the 256 x 192 tile, 3: 256 x 96): ni pieces per wave and K-tile, two stages of
#     4 ni KiB behind the A slots; the chunk table sits at 144 KiB.
CV_E = 103            # table entry of the K-tile whose A pieces are issued next
CV_O = 104            # v104..v111: gathered offsets of the 8 A pieces
CV_OFFB = 112         # v112..v117: DMA offsets of the (up to 6) B pieces
CV_TA = 118           # table read address
CV_OFFB_N = 119       # v119..v124: DMA offsets of the B pieces of the NEXT tile (another phase of a ConvTranspose forward has another row pitch)
CS_RSRC = 84          # s[84:87]: buffer descriptor of the gathered tensor
CS_TOFF, CS_LEFT, CS_NK = 88, 89, 90   # table byte offset of the next entry to read, entries left before it wraps, K-tiles per tile
CS_LDSWB = 91         # LDS-DMA destination base of this wave's B pieces (8 ni rows per wave: wave * ni KiB)
CS_LO, CS_HI = 70, 91
CB_BASE = 98304       # B stages of ni * 4 KiB behind the three A slots


def c_reads(ni, bstage, s, q):
    out = [f"ds_read_b128 {afrag(q, i)}, v{V_AC} offset:{i * 2048}" for i in range(8)]
    out += [f"ds_read_b128 {bfrag(q, i)}, {fbase_b(bstage, s)} offset:{i * 2048}" for i in range(ni)]
    return out


def c_dmas_a(next_tile):
    """8 x (3 VALU, M0 write, issue): v94 = delta bytes, v95 = mask bit of this K-tile's entry (set at the top of the half)"""
    sfx = "n" if next_tile else ""
    out = []
    for i in range(8):
        pre = [f"v_bfe_u32 v{CV_O + i}, %[im{sfx}{i}], v{V_TMP + 1}, 1",
               f"v_lshl_add_u32 v{CV_O + i}, v{CV_O + i}, 31, v{V_TMP}",
               f"v_add_u32 v{CV_O + i}, v{CV_O + i}, %[ro{sfx}{i}]"]
        out.append((pre, f"s_add_u32 m0, s{S_ADST}, {i * 1024}", f"buffer_load_dwordx4 v{CV_O + i}, s[{CS_RSRC}:{CS_RSRC + 3}], 0 offen lds"))
    return out


def c_dmas_b(ni, bstage, next_tile):
    offb = CV_OFFB_N if next_tile else CV_OFFB
    return [([], f"s_add_u32 m0, s{CS_LDSWB}, {CB_BASE + bstage * ni * 4096 + i * 1024}", f"global_load_lds_dwordx4 v{offb + i}, s[{S_BPTR}:{S_BPTR + 1}]")
            for i in range(ni)]


def c_half(st, ni, q, first, rd, dm, tail, cfg):
    """8 ni MFMAs on set q with the reads, the DMA pieces (each with its address arithmetic in front) and `tail` spread over them"""
    n = 8 * ni
    extra = {j: [] for j in range(n)}
    if cfg.get("abl_rd"):
        rd = []
    if cfg.get("abl_dma"):
        dm = []
    for k, r in enumerate(rd):
        extra[k * (n - 4) // len(rd)].append(r)
    for k, (pre, m0w, ld) in enumerate(dm):
        j = 2 + k * (n - 4) // len(dm)
        extra[j - 1].extend(pre)
        extra[j].append(m0w)
        extra[j + 1].insert(0, ld)
    extra[n * 5 // 6].extend(tail)
    j = 0
    for b in range(ni):
        for mi in range(8):
            c = "0" if first else acc(mi, b)
            st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, b)}, {bfrag(q, b)}, {afrag(q, mi)}, {c}")
            for x in extra[j]:
                st.e(x)
            j += 1


def c_entry_decode(st):
    st.e(f"v_ashrrev_i32 v{V_TMP}, 8, v{CV_E}")          # displacement in 16-byte units (signed)
    st.e(f"v_lshlrev_b32 v{V_TMP}, 4, v{V_TMP}")
    st.e(f"v_and_b32 v{V_TMP + 1}, 31, v{CV_E}")          # bit of the row's inverted tap mask (31: padding chunk, always invalid)


def c_table_next():
    """advance the cyclic table offset and read the entry of the K-tile whose A pieces the NEXT iteration issues"""
    return [f"s_add_u32 s{CS_TOFF}, s{CS_TOFF}, 32", f"s_sub_u32 s{CS_LEFT}, s{CS_LEFT}, 1", f"s_cmp_eq_u32 s{CS_LEFT}, 0",
            f"s_cselect_b32 s{CS_TOFF}, %[toffn4], s{CS_TOFF}", f"s_cselect_b32 s{CS_LEFT}, s{CS_NK}, s{CS_LEFT}",
            f"v_add_u32 v{CV_TA}, s{CS_TOFF}, %[vtl]", f"ds_read_b32 v{CV_E}, v{CV_TA}"]


def c_iteration(st, ni, p, first, next_tile, cfg):
    st.e(f"v_add_u32 v{V_AC}, s{S_A0}, v{V_FA + 1}")
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    c_entry_decode(st)
    c_half(st, ni, 0, first, c_reads(ni, p, 1, 1), c_dmas_a(next_tile), [], cfg)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        st.e("s_waitcnt vmcnt(8)")
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    st.e(f"v_add_u32 v{V_AC}, s{S_A1}, v{V_FA}")
    c_half(st, ni, 1, False, c_reads(ni, p ^ 1, 0, 0), c_dmas_b(ni, p, next_tile), c_table_next(), cfg)
    for a in advance(S_BPTR):
        st.e(a)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def c_setup(st, ni, with_next):
    for i in range(ni):   # B rows past the tile's valid width (a ragged last column tile) re-read its last valid row: vrb = valid rows - 1
        st.e(f"v_add_u32 v{V_TMP}, {8 * i}, %[browv]")
        st.e(f"v_min_u32 v{V_TMP + 1}, %[vrb], v{V_TMP}")
        st.e(f"v_mad_u32_u24 v{CV_OFFB + i}, v{V_TMP + 1}, %[ldb2], %[c16]")
        if with_next:
            st.e(f"v_min_u32 v{V_TMP + 1}, %[vrbn], v{V_TMP}")
            st.e(f"v_mad_u32_u24 v{CV_OFFB_N + i}, v{V_TMP + 1}, %[ldb2n], %[c16]")
    st.e(f"v_mov_b32 v{V_FA}, %[fa]")
    st.e(f"v_xor_b32 v{V_FA + 1}, 64, %[fa]")
    for b in (0, 1):
        st.e(f"v_add_u32 {fbase_b(b, 0)}, {CB_BASE + b * ni * 4096}, %[fb]")
        st.e(f"v_xor_b32 {fbase_b(b, 1)}, 64, {fbase_b(b, 0)}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")
    st.e(f"s_mov_b64 s[{CS_RSRC}:{CS_RSRC + 1}], %[abase]")
    st.e(f"s_mov_b32 s{CS_RSRC + 2}, %[abytes]")
    st.e(f"s_mov_b32 s{CS_RSRC + 3}, 0x00020000")
    st.e(f"s_mov_b32 s{CS_NK}, %[nk]")
    st.e(f"s_mov_b32 s{CS_LDSWB}, %[ldswb]")


def gen_c_prologue(ni):
    """first tile of a workgroup: K-tiles 0 and 1 (table entries 0 and 1 of its phase)"""
    st = Stream()
    c_setup(st, ni, False)
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"v_add_u32 v{CV_TA}, %[toff4], %[vtl]")
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"ds_read_b32 v{CV_E}, v{CV_TA} offset:{32 * k}")
        st.e("s_waitcnt lgkmcnt(0)")
        c_entry_decode(st)
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for pre, m0w, ld in c_dmas_a(False) + c_dmas_b(ni, k, False):
            for x in pre:
                st.e(x)
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in advance(S_BPTR):
            st.e(a)
    return st.text()


def gen_c_tile(ni, cfg):
    """one tile: bptr on its K-tile 2, bnext on the next tile's K-tile 0; ro / im = this tile's rows, ron / imn = the next tile's"""
    st = Stream()
    c_setup(st, ni, True)
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")
    st.e(f"s_add_u32 s{CS_TOFF}, %[toff4], 64")        # the first iteration issues K-tile 2 of this tile's table (toff4: its byte offset)
    st.e(f"s_sub_u32 s{CS_LEFT}, s{CS_NK}, 2")
    st.e(f"v_add_u32 v{CV_TA}, s{CS_TOFF}, %[vtl]")
    st.e(f"ds_read_b32 v{CV_E}, v{CV_TA}")
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    st.e(f"v_add_u32 v{V_AC}, s{S_A0}, v{V_FA}")
    for r in c_reads(ni, 0, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    c_iteration(st, ni, 0, True, False, cfg)
    c_iteration(st, ni, 1, False, False, cfg)
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    c_iteration(st, ni, 0, False, False, cfg)
    c_iteration(st, ni, 1, False, False, cfg)
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bnext]")
    c_iteration(st, ni, 0, False, True, cfg)
    c_iteration(st, ni, 1, False, True, cfg)
    st.e(f"s_mov_b32 %[a0], s{S_A0}")
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")
    st.e("s_nop 15")
    return st.text()


def c_clobbers(ni):
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    used = set(range(V_TMP, CV_OFFB + ni)) | {CV_TA} | set(range(CV_OFFB_N, CV_OFFB_N + ni))
    for q in (0, 1):                              # fragment sets: 8 A blocks + ni B blocks of 4 registers
        used |= set(range(V_SET + 64 * q, V_SET + 64 * q + 32)) | set(range(V_SET + 64 * q + 32, V_SET + 64 * q + 32 + 4 * ni))
    c += [f'"v{i}"' for i in sorted(used)]       # exactly the registers the block names: the compiler keeps the rest
    c += [f'"s{i}"' for i in range(CS_LO, CS_HI + 1)]
    return ", ".join(c)


# ---- paired convolution form (conv4_kernel<6, PAIR>: the split precision mode bf16x3 of the wide head stages).  The forward paired form's
# K-tile (32 reduction elements, LDS row = [hi | lo], three products on five quarter fragment sets) with the convolution form's gathered A
# pieces: the lo lanes of a piece (source chunk >= 4) carry the hi -> lo tensor distance in `aloadd` / in their B offset, the table has four
# entries per K-tile (16 bytes), the descriptor spans both tensors.
CPV_TMP, CPV_FA, CPV_FB, CPV_AC, CPV_E = 54, 56, 58, 62, 63
CPV_O, CPV_OFFB, CPV_TA, CPV_OFFB_N = 64, 72, 78, 80


def cp_fbase_b(stage, s):
    return f"v{CPV_FB + 2 * stage + s}"


def cp_reads_a(base):
    return [f"ds_read_b128 {pq(base, i)}, v{CPV_AC} offset:{i * 2048}" for i in range(8)]


def cp_reads_b(ni, base, bstage, s):
    return [f"ds_read_b128 {pq(base, i)}, {cp_fbase_b(bstage, s)} offset:{i * 2048}" for i in range(ni)]


def cp_dmas_a(next_tile):
    sfx = "n" if next_tile else ""
    out = []
    for i in range(8):
        pre = [f"v_bfe_u32 v{CPV_O + i}, %[im{sfx}{i}], v{CPV_TMP + 1}, 1",
               f"v_lshl_add_u32 v{CPV_O + i}, v{CPV_O + i}, 31, v{CPV_TMP}",
               f"v_add_u32 v{CPV_O + i}, v{CPV_O + i}, %[ro{sfx}{i}]"]
        out.append((pre, f"s_add_u32 m0, s{S_ADST}, {i * 1024}", f"buffer_load_dwordx4 v{CPV_O + i}, s[{CS_RSRC}:{CS_RSRC + 3}], 0 offen lds"))
    return out


def cp_dmas_b(ni, bstage, next_tile):
    offb = CPV_OFFB_N if next_tile else CPV_OFFB
    return [([], f"s_add_u32 m0, s{CS_LDSWB}, {CB_BASE + bstage * ni * 4096 + i * 1024}", f"global_load_lds_dwordx4 v{offb + i}, s[{S_BPTR}:{S_BPTR + 1}]")
            for i in range(ni)]


def cp_product(st, ni, bbase, abase, first, rd, dm, tail, cfg):
    n = 8 * ni
    extra = {j: [] for j in range(n)}
    if cfg.get("abl_rd"):
        rd = []
    if cfg.get("abl_dma"):
        dm = []
    for k, r in enumerate(rd):
        extra[k * (n - 4) // len(rd)].append(r)
    for k, (pre, m0w, ld) in enumerate(dm):
        j = 2 + k * (n - 4) // len(dm)
        extra[j - 1].extend(pre)
        extra[j].append(m0w)
        extra[j + 1].insert(0, ld)
    extra[n * 5 // 6].extend(tail)
    j = 0
    for b in range(ni):
        for mi in range(8):
            c = "0" if first else acc(mi, b)
            st.e(f"v_mfma_f32_16x16x32_bf16 {acc(mi, b)}, {pq(bbase, b)}, {pq(abase, mi)}, {c}")
            for x in extra[j]:
                st.e(x)
            j += 1


def cp_entry_decode(st):
    st.e(f"v_ashrrev_i32 v{CPV_TMP}, 8, v{CPV_E}")
    st.e(f"v_lshlrev_b32 v{CPV_TMP}, 4, v{CPV_TMP}")
    st.e(f"v_add_u32 v{CPV_TMP}, v{CPV_TMP}, %[aloadd]")   # lanes that fetch lo: + the distance of the lo tensor
    st.e(f"v_and_b32 v{CPV_TMP + 1}, 31, v{CPV_E}")


def cp_table_next():
    return [f"s_add_u32 s{CS_TOFF}, s{CS_TOFF}, 16", f"s_sub_u32 s{CS_LEFT}, s{CS_LEFT}, 1", f"s_cmp_eq_u32 s{CS_LEFT}, 0",
            f"s_cselect_b32 s{CS_TOFF}, %[toffn4], s{CS_TOFF}", f"s_cselect_b32 s{CS_LEFT}, s{CS_NK}, s{CS_LEFT}",
            f"v_add_u32 v{CPV_TA}, s{CS_TOFF}, %[vtl]", f"ds_read_b32 v{CPV_E}, v{CPV_TA}"]


def cp_iteration(st, ni, p, first, next_tile, cfg):
    ahi_cur, ahi_nxt = PQ_AHI[p], PQ_AHI[p ^ 1]
    st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{S_A2}")
    cp_entry_decode(st)
    cp_product(st, ni, PQ_BHI, ahi_cur, first, cp_reads_b(ni, PQ_BLO, p, 1), cp_dmas_a(next_tile), [], cfg)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        st.e("s_waitcnt vmcnt(8)")
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    st.e(f"v_add_u32 v{CPV_AC}, s{S_A1}, v{CPV_FA}")        # A of K-tile kt + 1, hi half
    bd = cp_dmas_b(ni, p, next_tile)
    cp_product(st, ni, PQ_BHI, PQ_ALO, False, cp_reads_a(ahi_nxt), bd[:ni // 2], [], cfg)
    st.e(f"v_add_u32 v{CPV_AC}, s{S_A1}, v{CPV_FA + 1}")    # ... lo half (the reads above have been issued)
    cp_product(st, ni, PQ_BLO, ahi_cur, False, cp_reads_b(ni, PQ_BHI, p ^ 1, 0) + cp_reads_a(PQ_ALO), bd[ni // 2:], cp_table_next(), cfg)
    for a in p_advance(S_BPTR):
        st.e(a)
    st.e(f"s_mov_b32 s{S_T}, s{S_A0}")
    st.e(f"s_mov_b32 s{S_A0}, s{S_A1}")
    st.e(f"s_mov_b32 s{S_A1}, s{S_A2}")
    st.e(f"s_mov_b32 s{S_A2}, s{S_T}")
    st.e("s_waitcnt lgkmcnt(0)")


def cp_setup(st, ni, with_next):
    for i in range(ni):
        st.e(f"v_add_u32 v{CPV_TMP}, {8 * i}, %[browv]")
        st.e(f"v_min_u32 v{CPV_TMP + 1}, %[vrb], v{CPV_TMP}")
        st.e(f"v_mad_u32_u24 v{CPV_OFFB + i}, v{CPV_TMP + 1}, %[ldb2], %[c16b]")
        if with_next:
            st.e(f"v_min_u32 v{CPV_TMP + 1}, %[vrbn], v{CPV_TMP}")
            st.e(f"v_mad_u32_u24 v{CPV_OFFB_N + i}, v{CPV_TMP + 1}, %[ldb2n], %[c16b]")
    st.e(f"v_mov_b32 v{CPV_FA}, %[fa]")
    st.e(f"v_xor_b32 v{CPV_FA + 1}, 64, %[fa]")
    for b in (0, 1):
        st.e(f"v_add_u32 {cp_fbase_b(b, 0)}, {CB_BASE + b * ni * 4096}, %[fb]")
        st.e(f"v_xor_b32 {cp_fbase_b(b, 1)}, 64, {cp_fbase_b(b, 0)}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    st.e(f"s_mov_b32 s{S_A0}, %[a0]")
    st.e(f"s_mov_b32 s{S_A1}, %[a1]")
    st.e(f"s_mov_b32 s{S_A2}, %[a2]")
    st.e(f"s_mov_b64 s[{CS_RSRC}:{CS_RSRC + 1}], %[abase]")
    st.e(f"s_mov_b32 s{CS_RSRC + 2}, %[abytes]")
    st.e(f"s_mov_b32 s{CS_RSRC + 3}, 0x00020000")
    st.e(f"s_mov_b32 s{CS_NK}, %[nk]")
    st.e(f"s_mov_b32 s{CS_LDSWB}, %[ldswb]")


def gen_cp_prologue(ni):
    st = Stream()
    cp_setup(st, ni, False)
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"v_add_u32 v{CPV_TA}, %[toff4], %[vtl]")
    for k, slot in enumerate((S_A0, S_A1)):
        st.e(f"ds_read_b32 v{CPV_E}, v{CPV_TA} offset:{16 * k}")
        st.e("s_waitcnt lgkmcnt(0)")
        cp_entry_decode(st)
        st.e(f"s_add_u32 s{S_ADST}, s{S_LDSW}, s{slot}")
        for pre, m0w, ld in cp_dmas_a(False) + cp_dmas_b(ni, k, False):
            for x in pre:
                st.e(x)
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in p_advance(S_BPTR):
            st.e(a)
    return st.text()


def gen_cp_tile(ni, cfg):
    """one tile of the paired form: nk = its 32-element K-tiles (a multiple of 4), bptr on K-tile 2, toff4 = byte offset of its table"""
    st = Stream()
    cp_setup(st, ni, True)
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")
    st.e(f"s_add_u32 s{CS_TOFF}, %[toff4], 32")        # the first iteration issues K-tile 2 (16 bytes of table per K-tile)
    st.e(f"s_sub_u32 s{CS_LEFT}, s{CS_NK}, 2")
    st.e(f"v_add_u32 v{CPV_TA}, s{CS_TOFF}, %[vtl]")
    st.e(f"ds_read_b32 v{CPV_E}, v{CPV_TA}")
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    st.e(f"v_add_u32 v{CPV_AC}, s{S_A0}, v{CPV_FA}")
    for r in cp_reads_a(PQ_AHI[0]):
        st.e(r)
    st.e(f"v_add_u32 v{CPV_AC}, s{S_A0}, v{CPV_FA + 1}")
    for r in cp_reads_a(PQ_ALO) + cp_reads_b(ni, PQ_BHI, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    cp_iteration(st, ni, 0, True, False, cfg)
    cp_iteration(st, ni, 1, False, False, cfg)
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    cp_iteration(st, ni, 0, False, False, cfg)
    cp_iteration(st, ni, 1, False, False, cfg)
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bnext]")
    cp_iteration(st, ni, 0, False, True, cfg)
    cp_iteration(st, ni, 1, False, True, cfg)
    st.e(f"s_mov_b32 %[a0], s{S_A0}")
    st.e(f"s_mov_b32 %[a1], s{S_A1}")
    st.e(f"s_mov_b32 %[a2], s{S_A2}")
    st.e("s_nop 15")
    st.e("s_nop 15")
    return st.text()


def cp_clobbers(ni):
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    used = set(range(CPV_TMP, CPV_OFFB + ni)) | {CPV_TA} | set(range(CPV_OFFB_N, CPV_OFFB_N + ni))
    for base in (PQ_AHI[0], PQ_AHI[1], PQ_ALO):
        used |= set(range(base, base + 32))
    for base in (PQ_BHI, PQ_BLO):
        used |= set(range(base, base + 4 * ni))
    c += [f'"v{i}"' for i in sorted(used)]
    c += [f'"s{i}"' for i in range(CS_LO, CS_HI + 1)]
    return ", ".join(c)


def clobbers():
    c = ['"memory"', '"scc"', '"m0"']
    c += [f'"a{i}"' for i in range(256)]
    c += [f'"v{i}"' for i in range(V_LO, V_HI + 1)]
    c += [f'"s{i}"' for i in range(S_LO, S_HI + 1)]
    return ", ".join(c)


def gen_readout():
    """C++ helpers: g4_acc_row<MI>(f32x4 (&t)[8]) reads the 8 column blocks of row block MI out of the AGPRs"""
    out = []
    for mi in range(8):
        out.append(f"__device__ __forceinline__ void g4_acc_row{mi}(f32x4 (&t)[8]) {{")
        for ni in range(8):
            b = (mi * 8 + ni) * 4
            out.append(f'    asm volatile("v_accvgpr_read_b32 %0, a{b}\\n\\tv_accvgpr_read_b32 %1, a{b + 1}\\n\\tv_accvgpr_read_b32 %2, a{b + 2}\\n\\tv_accvgpr_read_b32 %3, a{b + 3}" '
                       f': "=v"(t[{ni}][0]), "=v"(t[{ni}][1]), "=v"(t[{ni}][2]), "=v"(t[{ni}][3]));')
        out.append("}")
    out.append("__device__ __forceinline__ void g4_acc_row(int mi, f32x4 (&t)[8]) {")
    out.append("    switch (mi) {")
    for mi in range(8):
        out.append(f"        case {mi}: g4_acc_row{mi}(t); break;")
    out.append("    }")
    out.append("}")
    return "\n".join(out)


def main():
    cfg = {"rd_every": 3, "rd_at": 0, "dm_every": 8, "dm_at": 3, "order": "ni", "skew": 0,
           "w_rd_at": 0, "w_rd_num": 2, "w_rd_den": 1, "w_dm_at": 3, "w_dm_every": 8, "wp_rd_num": 1, "wp_rd_den": 1}
    out_path = "gemm4_gen.inc"
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("=", 1)
            cfg[k] = v if k == "order" else int(v)
        else:
            out_path = a
    with open(out_path, "w") as f:
        f.write("// GENERATED by gen_gemm4.py -- do not edit.  cfg = %r\n" % (cfg,))
        f.write("#define G4_ASM_PROLOGUE \\\n" + gen_prologue().replace("\n", " \\\n") + "\n\n")
        f.write("#define G4_ASM_TILE \\\n" + gen_tile(cfg).replace("\n", " \\\n") + "\n\n")
        f.write("#define G4_CLOBBERS " + clobbers() + "\n\n")
        f.write("#define G4P_ASM_PROLOGUE \\\n" + gen_p_prologue().replace("\n", " \\\n") + "\n\n")
        f.write("#define G4P_ASM_TILE \\\n" + gen_p_tile(cfg).replace("\n", " \\\n") + "\n\n")
        f.write("#define G4P_CLOBBERS " + p_clobbers() + "\n\n")
        f.write("#define G4W_ASM_PROLOGUE \\\n" + gen_w_prologue().replace("\n", " \\\n") + "\n\n")
        f.write("#define G4W_ASM_SEG \\\n" + gen_w_seg(cfg).replace("\n", " \\\n") + "\n\n")
        f.write("#define G4W_ASM_SEG_SHORT \\\n" + gen_w_seg(cfg, True).replace("\n", " \\\n") + "\n\n")
        f.write("#define G4W_CLOBBERS " + w_clobbers() + "\n\n")
        f.write("#define G4WP_ASM_PROLOGUE \\\n" + gen_wp_prologue().replace("\n", " \\\n") + "\n\n")
        f.write("#define G4WP_ASM_SEG \\\n" + gen_wp_seg(cfg).replace("\n", " \\\n") + "\n\n")
        f.write("#define G4WP_CLOBBERS " + wp_clobbers() + "\n\n")
        for ni in (6, 3):   # conv4_kernel<NI>: 256 x 192 and 256 x 96 tiles
            f.write(f"#define G4C{ni}_ASM_PROLOGUE \\\n" + gen_c_prologue(ni).replace("\n", " \\\n") + "\n\n")
            f.write(f"#define G4C{ni}_ASM_TILE \\\n" + gen_c_tile(ni, cfg).replace("\n", " \\\n") + "\n\n")
            f.write(f"#define G4C{ni}_CLOBBERS " + c_clobbers(ni) + "\n\n")
        for ni in (6, 3):   # conv4_kernel<NI, true>
            f.write(f"#define G4CP{ni}_ASM_PROLOGUE \\\n" + gen_cp_prologue(ni).replace("\n", " \\\n") + "\n\n")
            f.write(f"#define G4CP{ni}_ASM_TILE \\\n" + gen_cp_tile(ni, cfg).replace("\n", " \\\n") + "\n\n")
            f.write(f"#define G4CP{ni}_CLOBBERS " + cp_clobbers(ni) + "\n\n")
        f.write(gen_readout() + "\n")


if __name__ == "__main__":
    main()
