"""TIMING-ONLY ablation of a 16-points-per-thread, 8-waves-per-tile form of `gfx_fftconv_pipe_t{0,1}_o8`.

    python tools/experiments/r4_tile16/gen_half_ablation.py --hsaco grafx_amd/lib/hs/half.hsaco
    GRAFX_PIPE_HSACO=grafx_amd/lib/hs/half.hsaco GRAFX_PIPE_THREADS=512 python tools/microbench.py eqbuf --rows 8192

What it is: the persistent kernel of csrc/asm/gen_fftconv_pipe.py launched with 512 threads per tile, every thread doing HALF
of what a thread of the shipped kernel does -- 16 window rows, one 16-point transform per pass, 8 mirrored bin pairs, 16
output rows -- on a register map of 127 VGPRs (four waves per SIMD: two resident 8-wave workgroups per CU), with the same LDS
images (every cell written and read exactly once, by the half-workgroup that owns it), the same bytes from and to memory and
the same barriers.  What a real kernel of this shape needs on top is present as stand-in instructions of the right kind
and count: the radix-2 layer that joins the two half-threads of a 32-point pass (copies + cross-lane swaps + 16 packed
adds, forward and inverse), the exchange of mirrored bins around the product (2 x 16 moves), the pass-2 twiddles fetched
from LDS (6 ds_read_b64 per pass), the eighth spectrum slot loaded late into a 7-slot ring.  What it does NOT do is compute
the convolution: the cross-half butterflies are not wired, the results are wrong by construction.  Its run time is an
optimistic bound for the rebuild: if this is not clearly faster than the shipped kernel, nothing with this shape will be.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

from grafx_amd.csrc.asm import gen_fftconv_pipe as base  # noqa: E402
from grafx_amd.csrc.asm.isa import EXEC, Inst, Label, Lit, R, render, s, v  # noqa: E402
from grafx_amd.csrc.asm.tilegen import (CONST_TW_J, S1_ROW, S2_ROW, TILE_LDS_BYTES, const_tw_values, insert_waitcnt,  # noqa: E402
                                        interleave)

A, A2 = base.A, base.A2
DEBUG = set(filter(None, os.environ.get("HALF_DEBUG", "").split(",")))   # bisecting aids: noload, nostore, noh

# ---- register map: 127 VGPRs ----------------------------------------------------------------------------------------
V_TID, V_G8, V_G16, V_S1W, V_P2, V_P3, V_P3HI, V_P4, V_SKIP8 = (v(i) for i in range(9))
TMP = [v(10 + 2 * i, 2) for i in range(6)]                                  # v[10:21]
LO1 = {i: v(22 + 2 * (i - 1), 2) for i in (1, 2, 3)}                          # v[22:27]
HI1 = {i: v(28 + 2 * (i - 1), 2) for i in (1, 2, 3)}                          # v[28:33]
LO2, HI2 = LO1, HI1                                                           # stand-ins: the real ones come from LDS
HQ = [v(34 + 4 * q, 4) for q in range(7)]                                     # v[34:61]: a ring of 7 spectrum slots


class Bank:
    def __init__(self, base_reg, spare):
        self.land = [v(base_reg + 2 * i, 2) for i in range(16)]
        self.spare = spare
        self.nat = self.free = self.rows = None


BANK_A = Bank(62, v(126, 2))          # v[62:93]; the spare pair is shared (only one transform runs at a time)
BANK_B = Bank(94, v(126, 2))          # v[94:125]
NUM_VGPR = 128
# (s0..s101 is all there is; the shipped map is full, so the three scalars of this experiment live in argument slots the
# kernel never reads: pad1, pad2, stamp_lo)
S_HALF, S_HROW, S_SKIPH = A("pad1"), A("pad2"), A("stamp_lo")
NUM_SGPR = base.NUM_SGPR
S_OFF, S_ALO, S_HUGE, S_VALID = base.S_OFF, base.S_ALO, base.S_HUGE, base.S_VALID
NX_X, NX_H, NX_T, NX_Y, CUR_Y, ST_Y = base.NX_X, base.NX_H, base.NX_T, base.NX_Y, base.CUR_Y, base.ST_Y


class HalfGen(base.PipeGen):
    def __init__(self, tee, a_lo, **knobs):
        base.TileGen.__init__(self, base.S_CTW, base.S_ONE_NEG, base.S_C2)
        self.tee, self.a_lo = tee, a_lo
        self.k = dict(base.KNOBS, **knobs)
        self.ablate = set()
        self.uid = 0
        for bank in (BANK_A, BANK_B):
            self.sub()
            bank.rows, _ = self.dft(bank.land, bank.spare, True)
            self.sub()

    # ---- memory groups: this half-workgroup's 16 of the window's 32 rows (row 16 h + i at S_HROW + 2048 i) -------------
    def g_window_loads(self, bank, carried=False):
        out = []
        for i in range(0 if "noload" not in DEBUG else 16, 16):
            grp = [Inst("s_add_u32", S_OFF, (S_HROW, Lit(2048 * i)))]
            # rows the lower half takes over from the previous window: a lane offset the range check rejects (no access)
            # (also in the prologue: a first tile's window starts O samples BEFORE its row -- the shipped kernel masks those
            # rows with S_ALO; reading them faulted at the first row of the allocation)
            lane = V_SKIP8 if i < self.a_lo else V_G8
            grp.append(Inst("buffer_load_dwordx2", bank[i], (lane, NX_X, S_OFF), {}))
            out.append(grp)
        return out

    def g_stores(self, regs_of_row, desc):
        out = []
        for i in range(0 if "nostore" not in DEBUG else 16, 16):
            grp = [Inst("s_add_u32", S_OFF, (S_HROW, Lit(2048 * i)))]
            # the overlap rows (the lower half's first 8) are not stored: their lane offset is one the range check rejects
            lane = V_SKIP8 if i < self.a_lo else V_G8
            grp.append(Inst("buffer_store_dwordx2", None, (regs_of_row(i), lane, desc, S_OFF), dict(nt=1)))
            out.append(grp)
        return out

    def g_h_loads(self):
        out = []
        for q in range(0 if "noh" not in DEBUG else 7, 7):
            out.append([Inst("s_add_u32", S_OFF, (S_HALF, Lit(4096 * q))),
                        Inst("buffer_load_dwordx4", HQ[q], (V_G16, NX_H, S_OFF), {})])
        return out

    # ---- stand-ins for what joins the two half-threads --------------------------------------------------------------
    def join_layer(self, X, fma):
        """copies + cross-lane swaps (32 moves) and the 16 packed operations of the radix-2 layer between the halves"""
        for i in range(16):
            self.add("v_mov_b32", TMP[i % 6].sub(0), (X.land[i].sub(0),))
            self.add("v_mov_b32", TMP[i % 6].sub(1), (X.land[i].sub(1),))
            if fma:
                self.add("v_pk_fma_f32", X.land[i], (TMP[i % 6], base.S_ONE_NEG, X.land[i]), op_sel_hi=[1, 0, 1])
            else:
                self.cadd(X.land[i], X.land[i], TMP[i % 6])

    def lds_twiddles(self):
        for q in range(6):
            self.add("ds_read_b64", TMP[q], (V_P2,), offset=8 * q)

    def fwd_pass1(self, X):
        t1, t2 = TMP[1], TMP[2]
        self.join_layer(X, fma=False)
        out, _ = self.dft(X.land, X.spare, False)
        for k1 in range(16):
            self.apply_tw(out[k1], LO1, HI1, k1 & 3, k1 >> 2, False, t1, t2)
            self.add("ds_write_b64", None, (V_S1W, out[k1]), offset=8 * S1_ROW * k1)

    def fwd_read1(self, X):
        for c in range(16):
            self.add("ds_read_b64", X.land[c], (V_P2,), offset=8 * 16 * c)

    def fwd_pass2(self, X):
        t1, t2 = TMP[1], TMP[2]
        self.lds_twiddles()
        out, _ = self.dft(X.land, X.spare, False)
        for k2 in range(16):
            self.apply_tw(out[k2], LO2, HI2, k2 & 3, k2 >> 2, False, t1, t2)
            off = 8 * S2_ROW * (k2 * 32)
            reg = V_P3
            if k2 >= 8:
                reg, off = V_P3HI, off - 8 * S2_ROW * 8 * 32
            self.add("ds_write_b64", None, (reg, out[k2]), offset=off)

    def fwd_read2(self, X):
        for q in range(8):
            self.add("ds_read_b128", R("v", X.land[2 * q].idx, 4), (V_P4,), offset=16 * q)

    def fwd_pass3(self, X):
        out, free = self.dft(X.land, X.spare, False)
        X.nat, X.free = out, free

    def pair(self, X, ia, ib, hq, wk_idx, wj, self_pair):
        za, zb = X.nat[ia], X.nat[ib]
        xe, xo, who, ye, yo, wk = TMP
        he, ho = hq.sub(0, 2), hq.sub(2, 2)
        self.add_conj(xe, za, zb)
        self.sub_conj_mul_neg_i(xo, za, zb)
        if wj is not None:
            if wk_idx % 32 == 0:
                self.cmul(who, wj, ho)
            else:
                self.mul_const_any(wk, wj, wk_idx, False)
                self.cmul(who, wk, ho)
        else:
            self.mul_const_any(who, ho, wk_idx, False, tmp=wk)
        self.cmul(ye, he, xe)
        self.cmac(ye, who, xo)
        self.cmul(yo, ho, xe)
        self.cmac(yo, he, xo)
        self.add_mul_pos_i(za, ye, yo)
        if not self_pair:
            self.conj_sub_mul_pos_i(zb, ye, yo)

    def exchange(self, X):
        for i in range(8, 16):      # the mirrored bins held by the partner lane: one move per dword each way
            self.add("v_mov_b32", X.nat[i].sub(0), (X.nat[i].sub(0),))
            self.add("v_mov_b32", X.nat[i].sub(1), (X.nat[i].sub(1),))

    def product(self, X):
        wj = LO1[1]
        self.add(";touch", None, tuple(HQ))
        self.exchange(X)
        self.add("s_mov_b64", EXEC, (base.S_GEN_EXEC,))
        for k3 in range(8):
            self.pair(X, k3, 15 - k3, HQ[k3 % 7], 2 * k3, wj, False)
            if k3 == 0 and "noh" not in DEBUG:   # the eighth slot arrives late, in the ring position pair 0 has just freed
                self.add("s_add_u32", S_OFF, (S_HALF, Lit(4096 * 7)))
                self.add("buffer_load_dwordx4", HQ[0], (V_G16, NX_H, S_OFF))
        skip = self.fresh("not_t0")
        self.add("s_cmp_eq_u32", None, (base.S_WAVE0, Lit(0)))
        self.add("s_cbranch_scc1", target=skip)
        self.add("s_mov_b64", EXEC, (Lit(1),))
        for k3 in range(9):       # the self-mirrored rows' extra pairs (lanes 0 and 32 of the first wave)
            self.pair(X, k3 % 8, 15 - (k3 % 8), HQ[1 + k3 % 6], 1 + 2 * (k3 % 8), None, k3 == 0)
        self.label(skip)
        self.add("s_mov_b64", EXEC, (Lit(-1),))
        self.exchange(X)

    def inv_pass1(self, X):
        out, free = self.dft(X.nat, X.free, True)
        for e in range(16):
            self.add("ds_write_b64", None, (V_P4, out[e]), offset=8 * e)

    def inv_read2(self, X):
        t1, t2 = TMP[1], TMP[2]
        for k2 in range(16):
            off = 8 * S2_ROW * (k2 * 32)
            reg = V_P3
            if k2 >= 8:
                reg, off = V_P3HI, off - 8 * S2_ROW * 8 * 32
            self.add("ds_read_b64", X.land[k2], (reg,), offset=off)
        self.lds_twiddles()
        for k2 in range(16):
            self.apply_tw(X.land[k2], LO2, HI2, k2 & 3, k2 >> 2, True, t1, t2)

    def inv_pass2(self, X):
        out, _ = self.dft(X.land, X.spare, True)
        for c in range(16):
            self.add("ds_write_b64", None, (V_P2, out[c]), offset=8 * 16 * c)

    def inv_read3(self, X):
        t1, t2 = TMP[1], TMP[2]
        for k1 in range(16):
            self.add("ds_read_b64", X.land[k1], (V_S1W,), offset=8 * S1_ROW * k1)
        for k1 in range(16):
            self.apply_tw(X.land[k1], LO1, HI1, k1 & 3, k1 >> 2, True, t1, t2)

    def inv_pass3(self, X):
        rows, _ = self.dft(X.land, X.spare, True)
        assert rows == X.rows
        self.join_layer(X, fma=True)

    # ---- one tile -----------------------------------------------------------------------------------------------------
    def iteration(self, X, Y, name):
        out = []
        self.sub()
        for k in range(4):
            self.sop("s_mov_b32", ST_Y.sub(k), CUR_Y.sub(k))
        for k in range(4):
            self.sop("s_mov_b32", CUR_Y.sub(k), NX_Y.sub(k))
        self.decode(first=False)
        out += self.sub()
        self.fwd_pass1(X)
        self.barrier()
        self.fwd_read1(X)
        self.barrier()
        self.fwd_pass2(X)
        self.barrier()
        self.fwd_read2(X)
        self.fwd_pass3(X)
        fwd = self.sub()
        stores = self.g_stores(lambda a: Y.rows[a], ST_Y)
        # the carried overlap rows (a cross-lane copy in a real kernel): the output rows parked in those registers leave first
        dest = Y.land[: self.a_lo]
        first = [g for g in stores if g[-1].src[0] in dest]
        stores = [g for g in stores if g[-1].src[0] not in dest]
        head = [i for g in first for i in g]
        head.append(Inst(";touch", None, tuple(X.land) + tuple(dest)))
        for m in range(self.a_lo):
            src = X.land[16 - self.a_lo + m]
            head += [Inst("v_mov_b32", dest[m].sub(0), (src.sub(0),)), Inst("v_mov_b32", dest[m].sub(1), (src.sub(1),))]
        out += head
        side = stores + self.g_window_loads(Y.land, carried=True)
        out += interleave(fwd, side, 0.0, 1.0)
        self.product(X)
        out += self.sub()
        self.inv_pass1(X)
        self.barrier()
        self.inv_read2(X)
        self.barrier()
        self.inv_pass2(X)
        self.barrier()
        self.inv_read3(X)
        self.inv_pass3(X)
        inv = self.sub()
        tee_st = self.g_stores(lambda a: Y.land[a], NX_T) if self.tee else []
        out += interleave(inv, tee_st + self.g_h_loads(), 0.0, 1.0)
        return out

    def prologue(self):
        self.sub()
        for k in range(3):
            self.add("s_load_dwordx16", s(base.ARG0 + 16 * k, 16), (s(0, 2),), offset=64 * k)
        for j in CONST_TW_J:
            c, sn = const_tw_values(j)
            self.mov_lit(base.S_CTW[j].sub(0), float(c))
            self.mov_lit(base.S_CTW[j].sub(1), float(sn))
        self.mov_lit(base.S_ONE_NEG.sub(0), 1.0)
        self.mov_lit(base.S_ONE_NEG.sub(1), -1.0)
        self.mov_lit(base.S_C2.sub(0), -2.0)
        self.mov_lit(base.S_C2.sub(1), 2.0)
        self.mov_lit(S_HUGE, 0x7FFF0000)
        # which half of the workgroup (waves 0-3 / 4-7), then t = tid & 255 for every address
        SCR = base.SCR
        self.add("v_lshrrev_b32", TMP[0].sub(0), (Lit(8), V_TID))
        self.add("s_nop", imm=0)
        self.add("v_readfirstlane_b32", S_HALF, (TMP[0].sub(0),))
        self.add("v_lshrrev_b32", TMP[1].sub(1), (Lit(6), V_TID))
        self.add("s_nop", imm=0)
        self.add("v_readfirstlane_b32", SCR[6], (TMP[1].sub(1),))
        self.add("v_and_b32", V_TID, (Lit(255), V_TID))
        self.add("s_nop", imm=3)
        self.sop("s_lshl_b32", S_HROW, S_HALF, Lit(15))                     # 16 rows of 2048 bytes
        self.add("s_cmp_eq_u32", None, (S_HALF, Lit(0)))
        self.sop("s_cselect_b32", S_SKIPH, Lit(0x7FFFF000), Lit(0))          # the lower half owns the overlap rows:
        self.add("v_lshlrev_b32", V_G8, (Lit(3), V_TID))
        self.add("v_lshlrev_b32", V_G16, (Lit(4), V_TID))
        self.add("v_add_u32", V_SKIP8, (S_SKIPH, V_G8))                       # ... 8 t there is pushed out of every range
        self.sop("s_mul_i32", SCR[7], S_HALF, Lit(8 * S1_ROW * 16))
        self.add("v_add_u32", V_S1W, (SCR[7], V_G8))
        kk, d = TMP[0].sub(0), TMP[0].sub(1)
        self.add("v_lshrrev_b32", kk, (Lit(4), V_TID))
        self.add("v_and_b32", d, (Lit(15), V_TID))
        self.add("v_mul_u32_u24", V_P2, (Lit(S1_ROW), kk))
        self.add("v_add_u32", V_P2, (V_P2, d))
        self.add("v_lshlrev_b32", V_P2, (Lit(3), V_P2))
        self.add("v_add_u32", V_P2, (SCR[7], V_P2))                           # sgrp = half
        self.add("v_mul_u32_u24", V_P3, (Lit(S2_ROW), kk))
        self.add("v_add_u32", V_P3, (V_P3, d))
        self.add("v_lshlrev_b32", V_P3, (Lit(3), V_P3))
        self.sop("s_mul_i32", SCR[7], S_HALF, Lit(8 * S2_ROW * 16))
        self.add("v_add_u32", V_P3, (SCR[7], V_P3))
        self.add("v_add_u32", V_P3HI, (Lit(8 * S2_ROW * 8 * 32), V_P3))
        # row j of the third pass: t (lower half) or 512 - t (upper half; thread 0: 256)
        jb = TMP[1].sub(0)
        lower = ".Llower_half"
        self.add("v_mov_b32", jb, (V_TID,))
        self.add("s_cmp_eq_u32", None, (S_HALF, Lit(0)))
        self.add("s_cbranch_scc1", target=lower)
        self.add("v_sub_u32", jb, (Lit(512), V_TID))
        self.add("v_mov_b32", TMP[2].sub(0), (Lit(256),))
        self.add("v_cmp_ne_u32", R("vcc", 0, 2), (Lit(0), V_TID))
        self.add("s_nop", imm=1)
        self.add("v_cndmask_b32", jb, (TMP[2].sub(0), jb, R("vcc", 0, 2)))
        self.label(lower)
        self.add("v_mul_u32_u24", V_P4, (Lit(8 * S2_ROW), jb))
        self.add("s_cmp_eq_u32", None, (SCR[6], Lit(0)))
        self.sop("s_cselect_b32", base.S_WAVE0, Lit(1), Lit(0))
        self.sop("s_cselect_b32", base.S_GEN_EXEC.sub(0), Lit(-2), Lit(-1))
        self.mov_lit(base.S_GEN_EXEC.sub(1), -1)
        self.sop("s_mul_i32", S_HALF, S_HALF, Lit(9 * 4096))                 # from here on: this half's first spectrum slot
        self.add("s_waitcnt", lgkmcnt=0)
        self.sop("s_mul_i32", base.S_LB, s(2), A("per_xcd"))
        self.sop("s_add_u32", SCR[5], base.S_LB, A("per_xcd"))
        self.sop("s_min_u32", base.S_END, SCR[5], A("nblocks"))
        self.mov_lit(base.S_STRIDE, 1)
        done = ".Lnothing"
        self.add("s_cmp_ge_u32", None, (base.S_LB, base.S_END))
        self.add("s_cbranch_scc1", target=done)
        self.sop("s_mov_b32", NX_X.sub(0), A("tw_lo"))
        self.sop("s_and_b32", NX_X.sub(1), A("tw_hi"), Lit(0xFFFF))
        self.mov_lit(NX_X.sub(2), 20 * 2048)
        self.mov_lit(NX_X.sub(3), base.RSRC_FLAGS)
        for row, reg in {**{i: LO1[i] for i in LO1}, **{4 + i: HI1[i] for i in HI1}}.items():
            self.mov_lit(S_OFF, 2048 * row)
            self.add("buffer_load_dwordx2", reg, (V_G8, NX_X, S_OFF))
        self.decode(first=True)
        for k in range(4):
            self.mov_lit(CUR_Y.sub(k), 0 if k < 3 else base.RSRC_FLAGS)
        for grp in self.g_window_loads(BANK_A.land):
            self.prog.extend(grp)
        if self.tee:
            for grp in self.g_stores(lambda a: BANK_A.land[a], NX_T):
                self.prog.extend(grp)
        for grp in self.g_h_loads():
            self.prog.extend(grp)
        return self.sub()

    def build(self):
        blocks = {"pro": self.prologue()}
        it0 = self.iteration(BANK_A, BANK_B, "it0")
        it1 = self.iteration(BANK_B, BANK_A, "it1")
        blocks["it0"] = [Label(".Lloop")] + it0 + [Inst("s_cmp_eq_u32", None, (S_VALID, Lit(0))),
                                                   Inst("s_cbranch_scc1", None, (), dict(target=".Lexit_a"))]
        blocks["it1"] = it1 + [Inst("s_cmp_eq_u32", None, (S_VALID, Lit(0))),
                               Inst("s_cbranch_scc1", None, (), dict(target=".Lexit_b")),
                               Inst("s_branch", None, (), dict(target=".Lloop"))]
        blocks["exa"] = [Label(".Lexit_a")] + self.exit_block(BANK_A)
        blocks["exb"] = [Label(".Lexit_b")] + self.exit_block(BANK_B)
        blocks["end"] = [Label(".Lnothing"), Inst("s_endpgm")]
        traces = [["pro", "it0", "exa"], ["pro", "it0", "it1", "exb"], ["pro", "it0", "it1", "it0", "exa"],
                  ["pro", "it0", "it1", "it0", "it1", "exb"], ["pro", "it0", "it1", "it0", "it1", "it0", "exa"]]
        blocks = insert_waitcnt(traces, blocks)
        prog = []
        for name in ("pro", "it0", "it1", "exa", "exb", "end"):
            prog += blocks[name]
        return prog


def kernel_text(name, prog):
    txt = base.kernel_text(name, prog)
    txt = txt.replace(f".amdhsa_next_free_vgpr {base.NUM_VGPR}", f".amdhsa_next_free_vgpr {NUM_VGPR}")
    txt = txt.replace(f".amdhsa_next_free_sgpr {base.NUM_SGPR}", f".amdhsa_next_free_sgpr {NUM_SGPR}")
    txt = txt.replace(f".amdhsa_accum_offset {(base.NUM_VGPR + 3) // 4 * 4}", f".amdhsa_accum_offset {NUM_VGPR}")
    return txt


def generate():
    """All kernel names of the shipped code object; the two o8 variants are the ablation, the rest the shipped kernels."""
    txt = '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.amdhsa_code_object_version 6\n'
    names, half = [], []
    for tee, a_lo in base.VARIANTS:
        name = base.kernel_name(tee, a_lo)
        is_half = a_lo == 8
        prog = (HalfGen(tee, 8) if is_half else base.PipeGen(tee, a_lo)).build()
        for i in prog:
            if isinstance(i, Label):
                i.name = i.name + "_" + name
            elif "target" in i.mods:
                i.mods["target"] = i.mods["target"] + "_" + name
        txt += kernel_text(name, prog) if is_half else base.kernel_text(name, prog)
        names.append(name)
        half.append(is_half)
    from grafx_amd.csrc.asm import gen_corr_pipe

    name, prog = gen_corr_pipe.kernel()
    txt += base.kernel_text(name, prog)
    names.append(name)
    half.append(False)
    meta = base.metadata(names)
    # per-kernel metadata of the ablation kernels: 512 threads, 128 VGPRs
    parts = meta.split("  - .agpr_count:")
    for k, is_half in enumerate(half):
        if is_half:
            parts[k + 1] = (parts[k + 1].replace(".max_flat_workgroup_size: 256", ".max_flat_workgroup_size: 512")
                            .replace(f".vgpr_count:     {base.NUM_VGPR}", f".vgpr_count:     {NUM_VGPR}")
                            .replace(f".sgpr_count:     {base.NUM_SGPR + 2}", f".sgpr_count:     {NUM_SGPR + 2}"))
    return txt + "  - .agpr_count:".join(parts)


def main(argv):
    import subprocess
    import tempfile

    text = generate()
    if "--hsaco" not in argv:
        sys.stdout.write(text)
        return
    hsaco = argv[argv.index("--hsaco") + 1]
    llvm = os.environ.get("GRAFX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "k.s"), "w") as f:
            f.write(text)
        subprocess.run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                        os.path.join(d, "k.s"), "-o", os.path.join(d, "k.o")], check=True)
        subprocess.run([os.path.join(llvm, "ld.lld"), "-shared", os.path.join(d, "k.o"), "-o", hsaco], check=True)
    n = sum(1 for i in HalfGen(True, 8).build() if isinstance(i, Inst) and i.op.startswith("v_"))
    print(f"{hsaco}: {NUM_VGPR} VGPRs, {n} vector instructions in the tee variant (prologue + 2 tiles + exits)")


if __name__ == "__main__":
    main(sys.argv[1:])
