#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of gemm4.hip (the 4-wave, one-wave-per-SIMD form of the forward linear GEMM, pritvhi.py:446-456).

hipcc cannot hold 256 accumulators + 128 fragment registers of a 128 x 128-per-wave tile without spilling (rounds 2 and 5), so the K-loop of a
tile is ONE inline-asm block with hand-assigned registers; this script writes it (`gemm4_gen.inc`: string macros for the prologue and the tile
loop + the accumulator read-out helpers of the C++ epilogues).  Run by the Makefile; the output is not committed.

Structure of a tile (256 x 256 x K, BK = 64, 4 waves, wave (wr, wc) owns C[wr 128 ..][wc 128 ..] as 8 x 8 MFMA 16x16x32 accumulators = a0..a255):
  LDS: 2 stages x {A 256 rows x 128 B, B 256 rows x 128 B} = 128 KiB (+ epilogue staging behind it), rows swizzled (16-byte chunk ^= row & 7).
  Fragments: two register sets of 16 x 4 VGPRs (8 A row blocks + 8 B column blocks of ONE 32-deep k-substep); while the 64 MFMAs of a
  substep run on one set the other set is read from LDS.  Iteration kt (stage p = kt & 1):
      H1: 64 MFMAs on set 0 (K-tile kt, substep 0)      | 16 ds_read_b128 -> set 1 (stage p, substep 1)
      sync: lgkmcnt(0), vmcnt(0), s_barrier             (stage p is dead for every wave, K-tile kt+1 has landed for every wave)
      H2: 64 MFMAs on set 1                             | 16 ds_read_b128 -> set 0 (stage p^1 = K-tile kt+1, substep 0)
                                                        | 16 LDS-DMA issues of K-tile kt+2 -> stage p
  One barrier per K-tile; the DMA stream runs two K-tiles ahead and crosses the tile boundary (the last two iterations fetch K-tiles 0 and 1 of
  the workgroup's next tile, whose row clamp has its own offset registers).
"""
import sys

# ---- register map (explicit; everything here is in the asm blocks' clobber lists) ----
V_PF = 90            # v90: L2-prefetch offset of A (one row per lane), v92: its dummy destination
V_TMP = 94           # v94, v95: scratch
V_FB = 96            # v96..v103: fragment base addresses  [stage][A_s0, A_s1, B_s0, B_s1]
V_OFFA_N = 104       # v104..v111: DMA offsets of A, NEXT tile's row clamp
V_OFFA = 112         # v112..v119: DMA offsets of A, this tile
V_OFFB = 120         # v120..v127: DMA offsets of B
V_SET = 128          # v128..v255: set q at 128 + 64 q: A blocks [0..7] x 4, then B blocks [0..7] x 4
V_LO, V_HI = 90, 255
S_APTR, S_BPTR, S_CNT, S_LDSW, S_PFP = 70, 72, 74, 75, 76   # s[70:71], s[72:73], s74, s75, s[76:77]
S_LO, S_HI = 70, 77

STAGE = 65536
B_OFF = 32768


def afrag(q, mi):
    b = V_SET + 64 * q + 4 * mi
    return f"v[{b}:{b + 3}]"


def bfrag(q, ni):
    b = V_SET + 64 * q + 32 + 4 * ni
    return f"v[{b}:{b + 3}]"


def acc(mi, ni):
    b = (mi * 8 + ni) * 4
    return f"a[{b}:{b + 3}]"


def fbase(stage, is_b, s):
    return f"v{V_FB + 4 * stage + 2 * is_b + s}"


class Stream:
    def __init__(self):
        self.lines = []

    def e(self, s):
        self.lines.append(s)

    def text(self):
        return "\n".join(f'    "{l}\\n\\t"' for l in self.lines)


def reads(stage, s, q):
    """the 16 fragment reads of k-substep s of `stage` into set q (A blocks first)"""
    out = []
    for i in range(8):
        out.append(f"ds_read_b128 {afrag(q, i)}, {fbase(stage, 0, s)} offset:{i * 2048}")
    for i in range(8):
        out.append(f"ds_read_b128 {bfrag(q, i)}, {fbase(stage, 1, s)} offset:{B_OFF + i * 2048}")
    return out


def dmas(stage, next_tile):
    """16 x (M0 write, LDS-DMA issue) of one K-tile into `stage`: 8 pieces of A (8 rows x 128 B each), 8 of B"""
    out = []
    offa = V_OFFA_N if next_tile else V_OFFA
    for i in range(8):
        out.append((f"s_add_u32 m0, s{S_LDSW}, {stage * STAGE + i * 1024}", f"global_load_lds_dwordx4 v{offa + i}, s[{S_APTR}:{S_APTR + 1}]"))
    for i in range(8):
        out.append((f"s_add_u32 m0, s{S_LDSW}, {stage * STAGE + B_OFF + i * 1024}", f"global_load_lds_dwordx4 v{V_OFFB + i}, s[{S_BPTR}:{S_BPTR + 1}]"))
    return out


def advance():
    return [f"s_add_u32 s{S_APTR}, s{S_APTR}, 128", f"s_addc_u32 s{S_APTR + 1}, s{S_APTR + 1}, 0",
            f"s_add_u32 s{S_BPTR}, s{S_BPTR}, 128", f"s_addc_u32 s{S_BPTR + 1}, s{S_BPTR + 1}, 0"]


def half(st, q, first, rd, dm, cfg, tail=()):
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
    for x in tail:   # behind the last DMA issue
        extra[min(63, dm_at0 + len(dm) * dm_every)].append(x)
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


def iteration(st, p, first, next_tile, cfg, tag, prev_pf=True, pf_guard=None):
    # timing ablations (garbage results): abl_rd = no fragment reads, abl_dma = no LDS-DMA issues, abl_vmw = no vmcnt wait, abl_bar = no barrier
    rd1 = [] if cfg.get("abl_rd") else reads(p, 1, 1)
    rd0 = [] if cfg.get("abl_rd") else reads(p ^ 1, 0, 0)
    dm = [] if cfg.get("abl_dma") else dmas(p, next_tile)
    if cfg.get("dma_half"):   # timing experiment: every other piece only
        dm = dm[::2]
    half(st, 0, first, rd1, [], cfg)
    st.e("s_waitcnt lgkmcnt(0)")
    if not cfg.get("abl_vmw"):
        npf = 1 if (cfg.get("pf", 0) and prev_pf) else 0   # the previous iteration's prefetch may stay in flight
        st.e(f"s_waitcnt vmcnt({npf})")
    if not cfg.get("abl_bar"):
        st.e("s_barrier")
    skew(st, cfg, tag)
    pf = []
    if cfg.get("pf", 0) and not next_tile:
        # L2 prefetch: one dword of every row of A's K-tile kt + pf (64 lanes = 64 lines = 8 KiB per instruction; B is L2-resident: prefetching
        # it cost more than it bought).  Younger than this iteration's LDS-DMA issues, so the next sync may leave it outstanding (vmcnt(1)).
        # In the last trip before the tile's final pair the target K-tile kt + pf would lie behind the row's K extent (behind the tensor for
        # its last row): pf_guard = the trip count value at which this iteration is that one -> the prefetch re-touches K-tile pf of the tile.
        d = (cfg["pf"] - 2) * 128
        base = f"s[{S_APTR}:{S_APTR + 1}]"
        if pf_guard is not None:
            pf.append(f"s_cmp_eq_u32 s{S_CNT}, {pf_guard}")
            pf.append(f"s_cselect_b64 s[{S_PFP}:{S_PFP + 1}], %[aptr], s[{S_APTR}:{S_APTR + 1}]")
            base = f"s[{S_PFP}:{S_PFP + 1}]"
        pf.append(f"global_load_dword v{V_PF + 2}, v{V_PF}, {base} offset:{d}")
    half(st, 1, False, rd0, dm, cfg, pf)   # only H1 of a tile's first K-tile starts from the constant 0
    for a in advance():
        st.e(a)
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
    for is_b, nm in ((0, "%[fa]"), (1, "%[fb]")):
        st.e(f"v_mov_b32 {fbase(0, is_b, 0)}, {nm}")
        st.e(f"v_xor_b32 {fbase(0, is_b, 1)}, 64, {nm}")
        st.e(f"v_add_u32 {fbase(1, is_b, 0)}, {STAGE}, {nm}")
        st.e(f"v_xor_b32 {fbase(1, is_b, 1)}, 64, {fbase(1, is_b, 0)}")
    st.e(f"s_mov_b32 s{S_LDSW}, %[ldsw]")
    # prefetch rows: lane l of wave w -> row 64 w + l of the tile
    st.e(f"v_lshrrev_b32 v{V_TMP}, 3, %[rowv]")             # rowv = 64 w + (l >> 3)  ->  8 w
    st.e(f"v_lshlrev_b32 v{V_TMP}, 3, v{V_TMP}")            # 64 w
    st.e(f"v_mbcnt_lo_u32_b32 v{V_TMP + 1}, -1, 0")
    st.e(f"v_mbcnt_hi_u32_b32 v{V_TMP + 1}, -1, v{V_TMP + 1}")   # lane id
    st.e(f"v_add_u32 v{V_TMP}, v{V_TMP}, v{V_TMP + 1}")
    st.e(f"v_min_u32 v{V_TMP}, %[vrc], v{V_TMP}")
    st.e(f"v_mul_u32_u24 v{V_PF}, v{V_TMP}, %[lda2]")


def gen_prologue():
    """first tile of a workgroup: K-tiles 0 and 1 -> stages 0 and 1 (inputs: aptr / bptr = the tile's operand bases, vrc = its row clamp)"""
    st = Stream()
    setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    for stage in (0, 1):
        for m0w, ld in dmas(stage, False):
            st.e(m0w)
            st.e("s_nop 0")
            st.e(ld)
        for a in advance():
            st.e(a)
    return st.text()


def gen_tile(cfg):
    st = Stream()
    setup(st)
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[aptr]")   # K-tile 2 of this tile
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bptr]")
    st.e(f"s_mov_b32 s{S_CNT}, %[npair]")                  # middle pairs: nk / 2 - 2
    st.e("s_waitcnt vmcnt(0)")
    st.e("s_barrier")
    for r in reads(0, 0, 0):
        st.e(r)
    st.e("s_waitcnt lgkmcnt(0)")
    # first pair: the accumulators start from the inline constant 0
    pf_ = cfg.get("pf", 0)
    assert pf_ in (0, 3, 4), "prefetch distance: 3 or 4 K-tiles ahead of the compute (1 or 2 ahead of the LDS-DMA stream)"
    g0, g1 = (pf_ >= 4), (pf_ >= 3)   # which iteration of the trip before the final pair runs past K
    iteration(st, 0, True, False, cfg, "f0", prev_pf=False, pf_guard=0 if g0 else None)   # first pair: that trip iff npair == 0
    iteration(st, 1, False, False, cfg, "f1", pf_guard=0 if g1 else None)
    st.e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_last_%=")
    st.e("L_loop_%=:")
    iteration(st, 0, False, False, cfg, "m0", pf_guard=1 if g0 else None)   # middle pairs: the trip with the count at 1
    iteration(st, 1, False, False, cfg, "m1", pf_guard=1 if g1 else None)
    st.e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    st.e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    st.e("s_cbranch_scc1 L_loop_%=")
    st.e("L_last_%=:")
    # last pair: the DMA stream moves on to K-tiles 0, 1 of the next tile
    st.e(f"s_mov_b64 s[{S_APTR}:{S_APTR + 1}], %[anext]")
    st.e(f"s_mov_b64 s[{S_BPTR}:{S_BPTR + 1}], %[bnext]")
    iteration(st, 0, False, True, cfg, "l0")
    iteration(st, 1, False, True, cfg, "l1", prev_pf=False)
    st.e("s_nop 15")   # the last MFMAs retire before the epilogue's v_accvgpr_read (the compiler's hazard recognizer does not see into this block)
    st.e("s_nop 15")
    return st.text()


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
    cfg = {"rd_every": 3, "rd_at": 0, "dm_every": 3, "dm_at": 1, "order": "ni", "skew": 0}
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
        f.write(gen_readout() + "\n")


if __name__ == "__main__":
    main()
