"""Generator of `gfx_corr_pipe`: the hand-scheduled form of corr1_kernel (csrc/fftconv.hip) -- the filter gradient of a
short-filter convolve() (autograd of processors/core/convolution.py:119-134, N <= 8193 taps):

    gh[r, c, k] = sum_n g[r, c_g, n] x[r, c_x, n - k],   k in [0, N)

as a tile-wise circular correlation summed in the frequency domain.  One workgroup per (row, channel) walks the row's
tiles; per tile the V-sample slice of x (zero-padded to 16384) and the 16384-sample window of g starting at the same
sample are transformed in the two register banks, conj(X) G is added onto 68 accumulator registers (the polyphase form
of fft_tile.hpp's pair product with he = conj(Xe), ho = conj(Xo) conj(W^k)), and one inverse transform per row yields the
N lags.  Same building blocks and register map as gen_fftconv_pipe.py: 2 x 66 (banks) + 68 (accumulators) + 32
(twiddles) + 12 (scratch) + 9 (addresses) VGPRs, no scratch memory -- the compiler-built kernel needs 256 VGPRs plus
176 B of scratch per lane for the same work.  The loads of the next tile are issued as soon as the product has released
their registers.

The library uses it for off = 0, even L and the same row grouping on x and g (what the equaliser's filter gradient
needs); corr1_kernel keeps everything else.
"""
import sys

from .gen_fftconv_pipe import (ARG0, BANK_A, BANK_B, CONST_TW_J, HQ, LO1, NUM_SGPR, NUM_VGPR, RSRC_FLAGS, S_C2, S_CTW, S_GEN_EXEC,
                               S_HUGE, S_OFF, S_ONE_NEG, S_WAVE0, SCR, TMP, TW_ROWS, V_OFF8, V_OFF16, V_P1HI, V_P2, V_P3, V_P3HI,
                               V_P4A, V_P4B, V_TID, PipeGen, magic)  # noqa: F401
from .isa import EXEC, Inst, Label, Lit, R, render, s, v
from .tilegen import S1_ROW, S2_ROW, TILE_LDS_BYTES, brev, const_tw_values, insert_waitcnt

ARG_NAMES = ["x_lo", "x_hi", "g_lo", "g_hi", "o_lo", "o_hi", "tw_lo", "tw_hi",
             "L_bytes", "Lg_bytes", "V_bytes", "N_even_bytes", "ntiles", "nblocks", "inner", "m_inner", "sh_inner",
             "cout_shift", "cout_mask", "cx_mask", "cg_mask", "out_row_bytes", "tail_row", "tail_lane", "tail_off", "scale",
             "xs_outer_lo", "xs_outer_hi", "xs_inner", "xs_ch", "gs_outer_lo", "gs_outer_hi", "gs_inner", "gs_ch",
             "pad0", "pad1", "pad2", "pad3", "pad4", "pad5", "pad6", "pad7", "pad8", "pad9", "pad10", "pad11", "pad12", "pad13"]
assert len(ARG_NAMES) == 48
KERNARG_BYTES = 4 * len(ARG_NAMES)


def A(name):
    return s(ARG0 + ARG_NAMES.index(name))


X_ROW = s(72, 2)      # row base addresses (64-bit), output descriptor, loop state
G_ROW = s(74, 2)
D_X, D_G, D_O = s(76, 4), s(80, 4), s(84, 4)
S_TILE, S_SB = s(88), s(89)
KERNEL_NAME = "gfx_corr_pipe"


class CorrGen(PipeGen):
    def __init__(self, **knobs):
        super().__init__(False, 0, **knobs)

    # ---- memory groups ------------------------------------------------------------------------------------------------
    def loads(self, bank, desc):
        out = []
        for a in range(32):
            out.append([Inst("s_mov_b32", S_OFF, (Lit(4096 * (a >> 1)),)),
                        Inst("buffer_load_dwordx2", bank.land[a], (V_OFF8, desc, S_OFF), dict(offset=2048 * (a & 1)))])
        return out

    def tile_descs(self):
        """descriptors of tile S_TILE: x slice [s, s + V) clipped to the row, g window [s, s + 16384) clipped to the row"""
        t0 = SCR[0]
        self.sop("s_mul_i32", S_SB, S_TILE, A("V_bytes"))
        for D, row, length, cap in ((D_X, X_ROW, "L_bytes", True), (D_G, G_ROW, "Lg_bytes", False)):
            self.sop("s_add_u32", D.sub(0), row.sub(0), S_SB)
            self.sop("s_addc_u32", D.sub(1), row.sub(1), Lit(0))
            self.sop("s_and_b32", D.sub(1), D.sub(1), Lit(0xFFFF))
            self.sop("s_sub_u32", t0, A(length), S_SB)
            self.sop("s_max_i32", t0, t0, Lit(0))
            if cap:
                self.sop("s_min_u32", t0, t0, A("V_bytes"))
            self.sop("s_mov_b32", D.sub(2), t0)
            self.mov_lit(D.sub(3), RSRC_FLAGS)

    # ---- conj(X) G accumulated onto (ye, yo) -----------------------------------------------------------------------------
    def cmacc(self, acc, a, w):
        """acc += conj(a) * w"""
        self.add("v_pk_fma_f32", acc, (a, w, acc), op_sel_hi=[1, 0, 1], neg_hi=[1, 0, 0])
        self.add("v_pk_fma_f32", acc, (a, w, acc), op_sel=[1, 1, 0], op_sel_hi=[0, 1, 1])

    def corr_pair(self, X, Y, ia, ib, slot, wk_idx, wj, after_split=None):
        xa, xb, ya, yb = X.nat[ia], X.nat[ib], Y.nat[ia], Y.nat[ib]
        xe, xo, ge, go, ho, wk = TMP
        ye, yo = HQ[slot].sub(0, 2), HQ[slot].sub(2, 2)
        self.add_conj(xe, xa, xb)
        self.sub_conj_mul_neg_i(xo, xa, xb)
        self.add_conj(ge, ya, yb)
        self.sub_conj_mul_neg_i(go, ya, yb)
        if after_split is not None:
            after_split((xa, xb), (ya, yb))
        # he = conj(xe), who = wk * ho = conj(xo), ho = conj(xo) conj(wk)  (the 1 / 4M scale is applied once, at the end)
        #   ye += he ge + who go ;  yo += ho ge + he go
        self.cmacc(ye, xe, ge)
        self.cmacc(ye, xo, go)
        self.cmacc(yo, xe, go)
        # conj(xo) conj(wk) ge = conj(xo wk) ge
        if wj is not None:
            if wk_idx % 32 == 0:
                self.cmul(ho, xo, wj)
            else:
                self.mul_const_any(wk, wj, wk_idx, False)
                self.cmul(ho, xo, wk)
        else:
            self.mul_const_any(ho, xo, wk_idx, False, tmp=wk)
        self.cmacc(yo, ho, ge)

    def corr_product(self, X, Y, next_loads=None):
        """thread 0's self-mirrored pairs first, then everybody else's; `next_loads(bank, reg)` -> the load group that
        refills `reg` for the next tile (or None): issued as soon as a pair has read its four spectrum values"""
        wj = LO1[1]
        skip = self.fresh("not_t0")
        self.add("s_cmp_eq_u32", None, (S_WAVE0, Lit(0)))
        self.add("s_cbranch_scc1", target=skip)
        self.add("s_mov_b64", EXEC, (Lit(1),))
        self.corr_pair(X, Y, 0, 0, 0, 0, None)
        self.corr_pair(X, Y, 8, 8, 8, 16, None)
        for k3 in range(1, 8):
            self.corr_pair(X, Y, k3, 16 - k3, k3, 2 * k3, None)
        for k3 in range(8):
            self.corr_pair(X, Y, 16 + k3, 16 + (15 - k3), 9 + k3, 1 + 2 * k3, None)
        self.label(skip)
        self.add("s_mov_b64", EXEC, (S_GEN_EXEC,))

        def refill(xs, ys):
            self.add("s_mov_b64", EXEC, (Lit(-1),))
            for bank, regs in ((X, xs), (Y, ys)):
                for reg in regs:
                    grp = next_loads(bank, reg)
                    if grp:
                        self.prog.extend(grp)
            self.add("s_mov_b64", EXEC, (S_GEN_EXEC,))

        if next_loads is not None:      # registers that hold no spectrum value can be refilled at once
            for bank in (X, Y):
                for reg in bank.land:
                    if reg not in bank.nat:
                        self.add("s_mov_b64", EXEC, (Lit(-1),))
                        self.prog.extend(next_loads(bank, reg))
                        self.add("s_mov_b64", EXEC, (S_GEN_EXEC,))
        for k3 in range(16):
            self.corr_pair(X, Y, k3, 16 + (15 - k3), k3, 2 * k3, wj, refill if next_loads is not None else None)
        self.add("s_mov_b64", EXEC, (Lit(-1),))

    def merge(self, X):
        """accumulators -> spectrum bins of the result in X.nat (pair_merge), scaled by 1 / (4 M)"""
        sc = s(SCR[0].idx & ~1, 2)
        self.sop("s_mov_b32", sc.sub(0), A("scale"))
        self.sop("s_mov_b32", sc.sub(1), A("scale"))
        for q in range(17):
            for part in (HQ[q].sub(0, 2), HQ[q].sub(2, 2)):
                self.add("v_pk_mul_f32", part, (part, sc))

        def put(ia, ib, slot, self_pair):
            ye, yo = HQ[slot].sub(0, 2), HQ[slot].sub(2, 2)
            self.add_mul_pos_i(X.nat[ia], ye, yo)
            if not self_pair:
                self.conj_sub_mul_pos_i(X.nat[ib], ye, yo)

        self.add("s_mov_b64", EXEC, (S_GEN_EXEC,))
        for k3 in range(16):
            put(k3, 16 + (15 - k3), k3, False)
        skip = self.fresh("merge_not_t0")
        self.add("s_cmp_eq_u32", None, (S_WAVE0, Lit(0)))
        self.add("s_cbranch_scc1", target=skip)
        self.add("s_mov_b64", EXEC, (Lit(1),))
        put(0, 0, 0, True)
        put(8, 8, 8, True)
        for k3 in range(1, 8):
            put(k3, 16 - k3, k3, False)
        for k3 in range(8):
            put(16 + k3, 16 + (15 - k3), 9 + k3, False)
        self.label(skip)
        self.add("s_mov_b64", EXEC, (Lit(-1),))

    def forward(self, bank):
        self.fwd_pass1(bank)
        self.barrier()
        self.fwd_read1(bank)
        self.barrier()
        self.fwd_pass2(bank)
        self.barrier()
        self.fwd_read2(bank)
        self.fwd_pass3(bank)

    # ---- kernel ---------------------------------------------------------------------------------------------------------
    def build(self):
        X, Y = BANK_A, BANK_B
        self.sub()
        for k in range(3):
            self.add("s_load_dwordx16", s(ARG0 + 16 * k, 16), (s(0, 2),), offset=64 * k)
        for j in (1, 2, 3, 4):
            c, sn = const_tw_values(j)
            self.mov_lit(S_CTW[j].sub(0), float(c))
            self.mov_lit(S_CTW[j].sub(1), float(sn))
        self.mov_lit(S_ONE_NEG.sub(0), 1.0)
        self.mov_lit(S_ONE_NEG.sub(1), -1.0)
        self.mov_lit(S_C2.sub(0), -2.0)
        self.mov_lit(S_C2.sub(1), 2.0)
        self.thread_addresses()
        self.add("s_waitcnt", lgkmcnt=0)
        # workgroup -> (row, channel): logical index xcd * (grid / 8) + w (grid padded to a multiple of 8)
        rco, c, r, q, rem, t0, t1 = SCR[1], SCR[2], SCR[3], SCR[4], SCR[5], SCR[6], SCR[7]
        self.sop("s_and_b32", t0, s(2), Lit(7))
        self.sop("s_lshr_b32", t1, s(2), Lit(3))
        self.sop("s_mul_i32", t0, t0, A("pad0"))          # pad0 carries grid / 8
        self.sop("s_add_u32", rco, t0, t1)
        self.add("s_cmp_ge_u32", None, (rco, A("nblocks")))
        self.add("s_cbranch_scc1", target=".Lnothing")
        self.sop("s_and_b32", c, rco, A("cout_mask"))
        self.sop("s_lshr_b32", r, rco, A("cout_shift"))
        self.udiv(q, rem, r, A("inner"), A("m_inner"), A("sh_inner"), t0)

        def row_base(dst, base, outer, inner, ch, mask):
            self.sop("s_mov_b32", dst.sub(0), A(base + "_lo"))
            self.sop("s_mov_b32", dst.sub(1), A(base + "_hi"))
            self.add64_mul_u32_u64(dst, q, A(outer + "_lo"), A(outer + "_hi"), t0, t1)
            self.add64_mul_u32_u64(dst, rem, A(inner), None, t0, t1)
            self.sop("s_and_b32", t0, c, A(mask))
            self.sop("s_mul_i32", t0, t0, A(ch))
            self.sop("s_add_u32", dst.sub(0), dst.sub(0), t0)
            self.sop("s_addc_u32", dst.sub(1), dst.sub(1), Lit(0))

        row_base(X_ROW, "x", "xs_outer", "xs_inner", "xs_ch", "cx_mask")
        row_base(G_ROW, "g", "gs_outer", "gs_inner", "gs_ch", "cg_mask")
        # output row: o + rco * out_row_bytes; the descriptor covers the even part of the N lags
        self.sop("s_mov_b32", D_O.sub(0), A("o_lo"))
        self.sop("s_mov_b32", D_O.sub(1), A("o_hi"))
        self.add64_mul_u32_u64(D_O.sub(0, 2), rco, A("out_row_bytes"), None, t0, t1)
        self.sop("s_and_b32", D_O.sub(1), D_O.sub(1), Lit(0xFFFF))
        self.sop("s_mov_b32", D_O.sub(2), A("N_even_bytes"))
        self.mov_lit(D_O.sub(3), RSRC_FLAGS)
        # resident twiddles
        tw = D_X
        self.sop("s_mov_b32", tw.sub(0), A("tw_lo"))
        self.sop("s_and_b32", tw.sub(1), A("tw_hi"), Lit(0xFFFF))
        self.mov_lit(tw.sub(2), 20 * 2048)
        self.mov_lit(tw.sub(3), RSRC_FLAGS)
        for row, reg in TW_ROWS.items():
            self.mov_lit(S_OFF, 2048 * row)
            self.add("buffer_load_dwordx2", reg, (V_OFF8, tw, S_OFF))
        for qd in HQ:
            for k in range(4):
                self.add("v_mov_b32", qd.sub(k), (Lit(0),))
        self.add(";touch", None, tuple(TW_ROWS.values()))     # (the twiddle descriptor's registers are reused right below)
        self.mov_lit(S_TILE, 0)
        self.tile_descs()
        for grp in self.loads(X, D_X) + self.loads(Y, D_G):
            self.prog.extend(grp)
        pro = self.sub()

        # one tile: transform x (bank X), transform g (bank Y), accumulate; the next tile's windows are requested from
        # inside the product, register by register (past the last tile the descriptors are empty: the loads return zeros)
        descs = {id(X): D_X, id(Y): D_G}

        def next_loads(bank, reg):
            if reg not in bank.land:
                return None
            a = bank.land.index(reg)
            return [Inst("s_mov_b32", S_OFF, (Lit(4096 * (a >> 1)),)),
                    Inst("buffer_load_dwordx2", reg, (V_OFF8, descs[id(bank)], S_OFF), dict(offset=2048 * (a & 1)))]

        self.label(".Lloop")
        self.forward(X)
        self.barrier()            # the x transform's S2 reads are done before the g transform's S1 writes
        self.forward(Y)
        self.add(";touch", None, tuple(X.nat) + tuple(Y.nat))
        self.sop("s_add_u32", S_TILE, S_TILE, Lit(1))
        self.tile_descs()
        self.corr_product(X, Y, next_loads)
        self.barrier()            # ... and the g transform's before the next tile's
        self.add("s_cmp_lt_u32", None, (S_TILE, A("ntiles")))
        self.add("s_cbranch_scc1", target=".Lloop")
        loop = self.sub()
        nxt = []

        # the row's lags: accumulators -> spectrum -> inverse transform -> the first N samples
        self.add(";touch", None, tuple(X.land) + tuple(Y.land))    # (the empty tile's zeros have landed)
        self.merge(X)
        self.inv_pass1(X)
        self.barrier()
        self.inv_read2(X)
        self.barrier()
        self.inv_pass2(X)
        self.barrier()
        self.inv_read3(X)
        self.inv_pass3(X)
        for a in range(17):      # rows past the N lags fall outside the descriptor and are dropped
            self.mov_lit(S_OFF, 2048 * a)
            self.add("buffer_store_dwordx2", None, (X.rows[a], V_OFF8, D_O, S_OFF))
        # odd N: lag N - 1 is the low half of a pair the descriptor cut off; its row / lane / offset come as arguments
        self.sop("s_add_u32", D_O.sub(2), D_O.sub(2), Lit(4))
        self.add("v_cmp_eq_u32", R("vcc", 0, 2), (A("tail_lane"), V_TID))
        self.add("s_nop", imm=1)
        self.add("s_and_b64", EXEC, (EXEC, R("vcc", 0, 2)))
        for a in range(17):
            lab = self.fresh("tail")
            self.add("s_cmp_lg_u32", None, (A("tail_row"), Lit(a)))
            self.add("s_cbranch_scc1", target=lab)
            self.add("buffer_store_dword", None, (X.rows[a].sub(0), V_OFF8, D_O, A("tail_off")))
            self.label(lab)
        self.add("s_endpgm")
        last = self.sub()
        blocks = {"pro": pro, "loop": loop, "next": nxt, "last": last, "end": [Label(".Lnothing"), Inst("s_endpgm")]}
        traces = [["pro", "loop", "last"], ["pro", "loop", "loop", "last"], ["pro", "loop", "loop", "loop", "last"]]
        blocks = insert_waitcnt(traces, blocks)
        return blocks["pro"] + blocks["loop"] + blocks["next"] + blocks["last"] + blocks["end"]

    def thread_addresses(self):
        """the per-thread address registers and the wave-0 flags (the same as the convolution kernel's prologue)"""
        self.mov_lit(S_HUGE, 0x7FFF0000)
        self.add("v_lshlrev_b32", V_OFF8, (Lit(3), V_TID))
        self.add("v_lshlrev_b32", V_OFF16, (Lit(4), V_TID))
        self.add("v_add_u32", V_P1HI, (Lit(8 * S1_ROW * 16), V_OFF8))
        kk, d = TMP[0].sub(0), TMP[0].sub(1)
        self.add("v_lshrrev_b32", kk, (Lit(4), V_TID))
        self.add("v_and_b32", d, (Lit(15), V_TID))
        self.add("v_mul_u32_u24", V_P2, (Lit(S1_ROW), kk))
        self.add("v_add_u32", V_P2, (V_P2, d))
        self.add("v_lshlrev_b32", V_P2, (Lit(3), V_P2))
        self.add("v_mul_u32_u24", V_P3, (Lit(S2_ROW), kk))
        self.add("v_add_u32", V_P3, (V_P3, d))
        self.add("v_lshlrev_b32", V_P3, (Lit(3), V_P3))
        self.add("v_add_u32", V_P3HI, (Lit(8 * S2_ROW * 8 * 32), V_P3))
        self.add("v_mul_u32_u24", V_P4A, (Lit(8 * S2_ROW), V_TID))
        jb = TMP[1].sub(0)
        self.add("v_sub_u32", jb, (Lit(512), V_TID))
        self.add("v_mov_b32", TMP[2].sub(0), (Lit(256),))
        self.add("v_cmp_ne_u32", R("vcc", 0, 2), (Lit(0), V_TID))
        self.add("s_nop", imm=1)
        self.add("v_cndmask_b32", jb, (TMP[2].sub(0), jb, R("vcc", 0, 2)))
        self.add("v_mul_u32_u24", V_P4B, (Lit(8 * S2_ROW), jb))
        self.add("v_lshrrev_b32", TMP[1].sub(1), (Lit(6), V_TID))
        self.add("s_nop", imm=0)
        self.add("v_readfirstlane_b32", SCR[6], (TMP[1].sub(1),))
        self.add("s_nop", imm=3)
        self.add("s_cmp_eq_u32", None, (SCR[6], Lit(0)))
        self.sop("s_cselect_b32", S_WAVE0, Lit(1), Lit(0))
        self.sop("s_cselect_b32", S_GEN_EXEC.sub(0), Lit(-2), Lit(-1))
        self.mov_lit(S_GEN_EXEC.sub(1), -1)


def pack_args(**kw):
    import struct

    return struct.pack(f"<{len(ARG_NAMES)}I", *[int(kw.get(n, 0)) & 0xFFFFFFFF for n in ARG_NAMES])


def kernel(**knobs):
    """(name, program) with kernel-unique labels, for gen_fftconv_pipe.generate() to put in the code object"""
    prog = CorrGen(**knobs).build()
    for i in prog:
        if isinstance(i, Label):
            i.name = i.name + "_" + KERNEL_NAME
        elif "target" in i.mods:
            i.mods["target"] = i.mods["target"] + "_" + KERNEL_NAME
    return KERNEL_NAME, prog
