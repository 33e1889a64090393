"""Generator of `gfx_fftconv_pipe_*`: the hand-scheduled, persistent form of fftconv1_kernel (csrc/fftconv.hip) for gfx950.

Same arithmetic and memory layout as the compiler-built kernel -- overlap-save FIR convolution on 16384-sample LDS FFT
tiles, N <= 8193 taps, replacing convolve() of the reference (processors/core/convolution.py:119-134) for even L + N - 1
-- with the register allocation and the issue order written out:

  * one persistent 256-thread workgroup per half CU walks a run of CONSECUTIVE tiles; the window overlap (the last
    O / 512 register rows of the running window) is copied into the next window's bank instead of being loaded again,
    so every input sample crosses HBM once and a tile issues 32 - O / 512 window loads;
  * two register banks of 32 complex values alternate: while tile i is transformed in one bank, the outputs of tile
    i-1 leave the other bank (buffer_store interleaved with the first forward passes) and the window of tile i+1 lands
    in it; the filter spectrum of tile i+1 is requested during the inverse transform of tile i; the per-thread twiddles
    stay in registers for the life of the workgroup;
  * radix-2 DIT butterflies with fused multiply-adds (3 instructions; values rotate through a spare pair per bank, the
    register map stays static): 2 x 66 (banks) + 68 (spectrum) + 32 (twiddles) + 12 (scratch) + 9 (addresses) VGPRs, no
    scratch memory, 2 workgroups per CU (73,728 B of LDS each).

Variants (kernel name suffix): `t<0|1>` = also store the window's valid part to a second destination (the render's
source rows), `o<k>` = k = O / 512 leading rows of every tile are overlap (not stored; N - 1 <= 512 k).

    python -m grafx_amd.csrc.asm.gen_fftconv_pipe out.s        # writes the assembly for all built variants
"""
import math
import struct
import sys

from .isa import EXEC, Inst, Label, Lit, R, render, s, v
from .tilegen import (CONST_TW_J, S1_ROW, S2_ROW, TILE_LDS_BYTES, TileGen, brev, cluster_at, const_tw_values,
                      insert_waitcnt, interleave)

# ---- kernel arguments: 48 dwords, loaded into s[ARG0 : ARG0 + 48) ---------------------------------------------------
ARG_NAMES = ["x_lo", "x_hi", "h_lo", "h_hi", "y_lo", "y_hi", "c_lo", "c_hi", "tw_lo", "tw_hi",
             "L_bytes", "Lout_bytes", "V_bytes", "O_bytes", "ntiles", "nblocks",
             "m_ntiles", "sh_ntiles", "inner", "m_inner", "sh_inner", "hrows", "m_hrows", "sh_hrows",
             "cout_shift", "cout_mask", "cin_mask", "cf_mask", "Cf", "per_xcd", "wgs_per_xcd", "pad0",
             "xs_outer_lo", "xs_outer_hi", "xs_inner", "xs_ch", "ys_outer_lo", "ys_outer_hi", "ys_inner", "ys_ch",
             "cs_outer_lo", "cs_outer_hi", "cs_inner", "cs_ch", "rm_lo", "rm_hi", "rm_rec", "rm_flags"]
# (rm_*: a ready buffer descriptor of one word per output row-channel -- bits of max |y| of the row, for the odd-length
# aliasing's pair scaling (czt_pair.hip) -- or zeros: with no records the atomics below are dropped by the range check)
ARG0 = 8
KERNARG_BYTES = 4 * len(ARG_NAMES)


def A(name):
    return s(ARG0 + ARG_NAMES.index(name))


def A2(name):   # 64-bit argument (name of its low dword)
    return s(ARG0 + ARG_NAMES.index(name), 2)


def magic(d):
    """(m, sh): q_est = mulhi(n, m) >> sh is floor(n / d) or one less, for every 32-bit n"""
    sh = d.bit_length() - 1
    m = min((1 << (32 + sh)) // d, (1 << 32) - 1)
    return m, sh


def pack_args(**kw):
    vals = []
    for n in ARG_NAMES:
        vals.append(int(kw.get(n, 0)) & 0xFFFFFFFF)
    return struct.pack(f"<{len(vals)}I", *vals)


# ---- register map ---------------------------------------------------------------------------------------------------
V_TID, V_OFF8, V_OFF16, V_P1HI, V_P2, V_P3, V_P3HI, V_P4A, V_P4B = (v(i) for i in range(9))
V_MAX = v(9)                                               # running max |output| of the row-channel being walked
TMP = [v(10 + 2 * i, 2) for i in range(6)]                  # v[10:21]
TW_BASE = 22
LO1 = {i: v(TW_BASE + 2 * (i - 1), 2) for i in (1, 2, 3)}                 # W_8192^(t i)
HI1 = {i: v(TW_BASE + 6 + 2 * (i - 1), 2) for i in range(1, 8)}           # W_8192^(4 t i)
LO2 = {i: v(TW_BASE + 20 + 2 * (i - 1), 2) for i in (1, 2, 3)}            # W_256^(d i)
HI2 = {i: v(TW_BASE + 26 + 2 * (i - 1), 2) for i in (1, 2, 3)}            # W_256^(4 d i)
TW_ROWS = {**{i: LO1[i] for i in LO1}, **{4 + i: HI1[i] for i in HI1}, **{12 + i: LO2[i] for i in LO2},
           **{16 + i: HI2[i] for i in HI2}}                               # table row -> register pair
H_BASE = 56
HQ = [v(H_BASE + 4 * q, 4) for q in range(17)]


class Bank:
    """32 complex values of a tile plus the spare pair the FMA butterflies rotate through.  `land`: where loads and LDS
    reads put logical element i (fixed, contiguous: element pairs form aligned quads); `nat[i]`: register of spectrum
    bin NAT i between the forward and the inverse transform; `rows[a]`: register of output time row a after the inverse
    transform -- both static, found by the generator."""

    def __init__(self, base, spare):
        self.land = [v(base + 2 * i, 2) for i in range(32)]
        self.spare = spare
        self.nat = self.free = self.rows = None


BANK_A = Bank(124, v(54, 2))
BANK_B = Bank(188, v(252, 2))
NUM_VGPR = 254

S_CTW = {j: s(56 + 2 * k, 2) for k, j in enumerate(CONST_TW_J)}    # s[56:63]
S_ONE_NEG = s(64, 2)
S_C2 = s(66, 2)
S_LB, S_END, S_VALID, S_STRIDE = s(68), s(69), s(70), s(71)
NX_X, NX_H, NX_T, NX_Y, CUR_Y, ST_Y = (s(72 + 4 * i, 4) for i in range(6))
S_ALO = s(96)
S_HUGE = s(97)
S_GEN_EXEC = s(98, 2)
S_WAVE0 = s(100)
S_OFF = s(101)
SCR = [s(i) for i in range(8)]      # s[0:7]: scratch once the arguments are loaded
RM_DESC = s(ARG0 + ARG_NAMES.index("rm_lo"), 4)            # s[52:55]: the rows' maxima (arguments rm_*)
NUM_SGPR = 102

H_TILE_BYTES = 17 * 256 * 16
RSRC_FLAGS = 0x00020000


# Schedule knobs (tuning and timing-only ablations; the defaults are what the library embeds)
KNOBS = dict(
    fwd_window=(0.0, 1.0),    # where in the forward transform the previous tile's stores / the next window's loads go
    inv_window=(0.0, 1.0),    # where in the inverse transform the tee stores / the next spectrum's loads go
    ablate="",                  # comma list out of: nostore, nowin, noh, notee  (timing only: results are wrong)
    prio=0,                     # s_setprio level while a tile's arithmetic runs (0: off)
    place="spread",             # spread: evenly inside the windows; barrier / after: in clusters before / after the barriers
    inv_order="tee_first",      # tee_first / h_first / mix: order of the tee stores and the spectrum loads in the inverse
    carry=1,                    # 1: a workgroup walks CONSECUTIVE tiles and copies the window overlap (the last a_lo rows)
                                # from the running tile's registers into the next window instead of re-reading it
    fake_epilogue=0,            # timing experiment: that many junk VALU instructions + 24 extra stores per tile
    rowmax=0,                   # 1: keep max |output| per row-channel and add it into the rm_* buffer (carry=1 only).  NOT in
                                # the build: the kernel needs an even output length (8-byte stores), and the full-length convolution
                                # in front of the odd-length aliasing -- the one consumer of the maxima -- has an odd one, so it
                                # always runs on fftconv1_kernel, which takes them itself; the emulator test keeps this alive
    codelet="dit",              # dit: radix-2 DIT with fused multiply-adds (3 instructions per general butterfly, values
                                # rotate through a spare pair); dif: the in-place DIF of fft_tile.hpp (4 instructions)
)


class PipeGen(TileGen):
    def __init__(self, tee, a_lo, **knobs):
        super().__init__(S_CTW, S_ONE_NEG, S_C2)
        self.tee, self.a_lo = tee, a_lo
        self.k = dict(KNOBS, **knobs)
        self.ablate = set(filter(None, self.k["ablate"].split(",")))
        self.uid = 0
        for bank in (BANK_A, BANK_B):       # where the inverse transform leaves the output rows (static per bank)
            self.sub()
            bank.rows, _ = self.dft(bank.land, bank.spare, True)
            self.sub()

    def dft(self, inputs, free, inv):
        """DFT of the pairs `inputs` (natural order) -> (out, free): out[k] = pair holding frequency k"""
        if self.k["codelet"] == "dit":
            return self.dit(list(inputs), free, inv)
        vals = list(inputs)
        self.dif(vals, TMP[0], inv)
        bits = len(vals).bit_length() - 1
        return [vals[brev(k, bits)] for k in range(len(vals))], free

    def sop(self, op, dst, *src):
        self.add(op, dst, src)

    def mov_lit(self, dst, val):
        self.add("s_mov_b32", dst, (Lit(val),))

    def fresh(self, stem):
        self.uid += 1
        return f".L{stem}_{self.uid}"

    # ---- scalar helpers --------------------------------------------------------------------------------------------
    def udiv(self, q, rem, n, d, m, sh, t):
        """q = n / d, rem = n % d (unsigned); q, rem, t: distinct scratch SGPRs, n may not be q / rem / t"""
        self.sop("s_mul_hi_u32", q, n, m)
        self.sop("s_lshr_b32", q, q, sh)
        self.sop("s_mul_i32", rem, q, d)
        self.sop("s_sub_u32", rem, n, rem)
        self.add("s_cmp_ge_u32", None, (rem, d))
        self.sop("s_cselect_b32", t, d, Lit(0))
        self.sop("s_addc_u32", q, q, Lit(0))
        self.sop("s_sub_u32", rem, rem, t)

    def add64_mul_u32_u64(self, acc, a, blo, bhi, t0, t1):
        """acc(64) += a(32) * b(64)"""
        self.sop("s_mul_i32", t0, a, blo)
        self.sop("s_mul_hi_u32", t1, a, blo)
        self.sop("s_add_u32", acc.sub(0), acc.sub(0), t0)
        self.sop("s_addc_u32", acc.sub(1), acc.sub(1), t1)
        if bhi is not None:
            self.sop("s_mul_i32", t0, a, bhi)
            self.sop("s_add_u32", acc.sub(1), acc.sub(1), t0)

    def add64_s32(self, acc, x, t0):
        """acc(64) += sign-extended x(32)"""
        self.sop("s_ashr_i32", t0, x, Lit(31))
        self.sop("s_add_u32", acc.sub(0), acc.sub(0), x)
        self.sop("s_addc_u32", acc.sub(1), acc.sub(1), t0)

    def decode(self, first):
        """Descriptors of tile S_LB into NX_X / NX_H / NX_T / NX_Y, S_ALO, S_VALID (0 past the workgroup's range; the
        descriptors then have zero records: loads return zeros, stores are dropped).  `first`: the prologue's call (the
        tile is known to be valid).  Clobbers SCR[0..7]; advances S_LB by S_STRIDE."""
        rco, tile, c, r, q, rem, t0, t1 = SCR
        lb = S_LB
        if not first:
            self.add("s_cmp_lt_u32", None, (S_LB, S_END))
            self.sop("s_cselect_b32", S_VALID, Lit(1), Lit(0))
            # an invalid index is decoded as the range's first tile... any valid one: keeps every address in bounds
            self.sop("s_cselect_b32", t1, S_LB, A("pad0"))    # pad0 holds 0: tile 0 is always valid to decode
            lb = t1
        else:
            self.mov_lit(S_VALID, 1)
        self.udiv(rco, tile, lb, A("ntiles"), A("m_ntiles"), A("sh_ntiles"), t0)
        self.sop("s_and_b32", c, rco, A("cout_mask"))
        self.sop("s_lshr_b32", r, rco, A("cout_shift"))
        # hr = r % hrows -> kept in `rco` (free from here on)
        self.udiv(t1, rco, r, A("hrows"), A("m_hrows"), A("sh_hrows"), t0)
        hr = rco
        # H descriptor: base + (hr * Cf + (c & cf_mask)) * 69632
        self.sop("s_mul_i32", t0, hr, A("Cf"))
        self.sop("s_and_b32", t1, c, A("cf_mask"))
        self.sop("s_add_u32", t0, t0, t1)
        self.sop("s_mov_b32", NX_H.sub(0), A("h_lo"))
        self.sop("s_mov_b32", NX_H.sub(1), A("h_hi"))
        self.mov_lit(hr, H_TILE_BYTES)
        self.add64_mul_u32_u64(NX_H.sub(0, 2), t0, hr, None, t1, q)
        self.sop("s_and_b32", NX_H.sub(1), NX_H.sub(1), Lit(0xFFFF))
        self.sop("s_mul_i32", NX_H.sub(2), S_VALID, hr)             # 69632 records, or none
        self.mov_lit(NX_H.sub(3), RSRC_FLAGS)
        # q, rem = divmod(r, inner)
        self.udiv(q, rem, r, A("inner"), A("m_inner"), A("sh_inner"), t0)
        # window start in bytes (signed): tile * V - O
        sb = tile
        self.sop("s_mul_i32", sb, tile, A("V_bytes"))
        self.sop("s_sub_u32", sb, sb, A("O_bytes"))
        # rows of the window that lie before the row start (first tile): (-sb) >> 11, else 0
        self.sop("s_sub_u32", t0, Lit(0), sb)
        self.sop("s_max_i32", t0, t0, Lit(0))
        self.sop("s_lshr_b32", S_ALO, t0, Lit(11))

        def desc(D, base, outer, inner, ch, chan, length):
            self.sop("s_mov_b32", D.sub(0), A(base + "_lo"))
            self.sop("s_mov_b32", D.sub(1), A(base + "_hi"))
            acc = D.sub(0, 2)
            self.add64_mul_u32_u64(acc, q, A(outer + "_lo"), A(outer + "_hi"), t0, t1)
            self.add64_mul_u32_u64(acc, rem, A(inner), None, t0, t1)
            self.sop("s_mul_i32", t0, chan, A(ch))
            self.sop("s_add_u32", acc.sub(0), acc.sub(0), t0)
            self.sop("s_addc_u32", acc.sub(1), acc.sub(1), Lit(0))
            self.add64_s32(acc, sb, t0)
            self.sop("s_and_b32", D.sub(1), D.sub(1), Lit(0xFFFF))
            self.sop("s_sub_u32", t0, A(length), sb)
            self.sop("s_max_i32", t0, t0, Lit(0))
            self.sop("s_mul_i32", D.sub(2), t0, S_VALID)
            self.mov_lit(D.sub(3), RSRC_FLAGS)

        self.sop("s_and_b32", r, c, A("cin_mask"))     # cx (r is free now)
        desc(NX_X, "x", "xs_outer", "xs_inner", "xs_ch", r, "L_bytes")
        desc(NX_Y, "y", "ys_outer", "ys_inner", "ys_ch", c, "Lout_bytes")
        if self.tee:
            desc(NX_T, "c", "cs_outer", "cs_inner", "cs_ch", c, "L_bytes")
        self.sop("s_add_u32", S_LB, S_LB, S_STRIDE)

    # ---- memory instruction groups (each a short list: the scalar offset, then the access) --------------------------
    def g_window_loads(self, bank, carried=False):
        """32 groups: row a of the next window -> bank[a]; rows before the row start (a < S_ALO, first tiles only) get an
        out-of-range offset and read as zero.  `carried`: rows below a_lo come from the previous window's registers
        (`carry_overlap`), only the rows from a_lo on are loaded."""
        out = []
        if carried and "wideload" in self.ablate:     # timing experiment (wrong results): the carried form as 16-byte loads
            for p in range(self.a_lo // 2, 16):
                out.append([Inst("s_mov_b32", S_OFF, (Lit(4096 * p),)),
                            Inst("buffer_load_dwordx4", R("v", bank[2 * p].idx, 4), (V_OFF16, NX_X, S_OFF), {})])
            return out
        if carried:
            for a in range(self.a_lo, 32):
                out.append([Inst("s_mov_b32", S_OFF, (Lit(2048 * a),)),
                            Inst("buffer_load_dwordx2", bank[a], (V_OFF8, NX_X, S_OFF), {})])
            return out
        if "widemem" in self.ablate:     # timing experiment (wrong results): the same bytes as 16-byte accesses
            for p in range(16):
                grp = [Inst("s_mov_b32", S_OFF, (Lit(4096 * p),))]
                if p < 8:
                    grp += [Inst("s_cmp_gt_u32", None, (S_ALO, Lit(2 * p))), Inst("s_cselect_b32", S_OFF, (S_HUGE, S_OFF))]
                grp.append(Inst("buffer_load_dwordx4", R("v", bank[2 * p].idx, 4), (V_OFF16, NX_X, S_OFF), {}))
                out.append(grp)
            return out
        for a in range(32):
            grp = []
            grp.append(Inst("s_mov_b32", S_OFF, (Lit(2048 * a),)))
            if a < 16:
                grp.append(Inst("s_cmp_gt_u32", None, (S_ALO, Lit(a))))
                grp.append(Inst("s_cselect_b32", S_OFF, (S_HUGE, S_OFF)))
            grp.append(Inst("buffer_load_dwordx2", bank[a], (V_OFF8, NX_X, S_OFF), {}))
            out.append(grp)
        return out

    def g_stores(self, regs_of_row, desc):
        """stores of the rows a >= a_lo (the tile's valid part): row a <- regs_of_row(a)"""
        out = []
        if "widemem" in self.ablate or ("wideload" in self.ablate and desc is NX_T):
            for p in range(self.a_lo // 2, 16):
                q = R("v", (BANK_A if regs_of_row(31).idx < BANK_B.land[0].idx else BANK_B).land[2 * p].idx, 4)
                out.append([Inst("s_mov_b32", S_OFF, (Lit(4096 * p),)),
                            Inst("buffer_store_dwordx4", None, (q, V_OFF16, desc, S_OFF), dict(nt=1))])
            return out
        for a in range(self.a_lo, 32):
            out.append([Inst("s_mov_b32", S_OFF, (Lit(4096 * (a >> 1)),)),
                        Inst("buffer_store_dwordx2", None, (regs_of_row(a), V_OFF8, desc, S_OFF),
                             dict(offset=2048 * (a & 1), nt=1))])
        return out

    def g_h_loads(self):
        out = []
        for q in range(17):
            out.append([Inst("s_mov_b32", S_OFF, (Lit(4096 * q),)),
                        Inst("buffer_load_dwordx4", HQ[q], (V_OFF16, NX_H, S_OFF), {})])
        return out

    # ---- the rows' maxima (round 6) ------------------------------------------------------------------------------------
    # Every lane keeps max |y| over the outputs it stores (one v_max3_f32 with |.| modifiers per stored register pair, the
    # overlap rows excluded); after the LAST tile of a row-channel -- and when the workgroup's run of tiles ends -- every
    # lane adds its word into the row's slot with a return-less atomic maximum (non-negative floats order like their bit
    # patterns) and starts again from zero.  No branch goes round the atomic (the wait-count pass counts memory
    # instructions statically): a tile that does not flush aims it past the end of the buffer, where the range check drops it.
    def rowmax_accumulate(self, regs_of_row):
        for a in range(self.a_lo, 32):
            r = regs_of_row(a)
            self.add("v_max3_f32", V_MAX, (r.sub(0), r.sub(1), V_MAX), abs=[1, 1, 0])

    def rowmax_flush(self, tiles_back, always):
        """tile S_LB - tiles_back is the one whose outputs were just accumulated.  always: the exit (flush whatever is there)."""
        q, rem, t0, lb, off, keep = SCR[0], SCR[1], SCR[2], SCR[3], SCR[4], SCR[5]
        self.sop("s_sub_u32", lb, S_LB, Lit(tiles_back))
        self.udiv(q, rem, lb, A("ntiles"), A("m_ntiles"), A("sh_ntiles"), t0)
        self.sop("s_lshl_b32", off, q, Lit(2))
        if not always:
            # flush = (this was the last tile of its row) and (the tile was a real one: its store descriptor has records)
            self.sop("s_add_u32", rem, rem, Lit(1))
            self.add("s_cmp_eq_u32", None, (rem, A("ntiles")))
            self.sop("s_cselect_b32", t0, Lit(1), Lit(0))
            self.add("s_cmp_lg_u32", None, (ST_Y.sub(2), Lit(0)))
            self.sop("s_cselect_b32", t0, t0, Lit(0))                  # flush
            self.sop("s_cselect_b32", keep, Lit(-1), Lit(0))           # a tile that was never there leaves garbage: dropped
            self.add("s_cmp_lg_u32", None, (t0, Lit(0)))
            self.sop("s_cselect_b32", off, off, S_HUGE)
            self.sop("s_cselect_b32", keep, Lit(0), keep)
        self.add("buffer_atomic_umax", None, (V_MAX, RM_DESC, off))
        self.add("s_nop", imm=0)
        if always:
            return
        self.add("v_and_b32", V_MAX, (keep, V_MAX))

    # ---- LDS exchanges + register passes -----------------------------------------------------------------------------
    def barrier(self):
        self.add("s_barrier")

    def sub(self):
        """start a fresh instruction list (the caller collects the pieces); returns the previous one"""
        prev, self.prog = self.prog, []
        return prev

    def fwd_pass1(self, X):
        t1, t2 = TMP[1], TMP[2]
        out, _ = self.dft(X.land, X.spare, False)
        for k1 in range(32):
            self.apply_tw(out[k1], LO1, HI1, k1 & 3, k1 >> 2, False, t1, t2)
            base, off = (V_OFF8, 8 * S1_ROW * k1) if k1 < 16 else (V_P1HI, 8 * S1_ROW * (k1 - 16))
            self.add("ds_write_b64", None, (base, out[k1]), offset=off)

    def fwd_read1(self, X):
        # u[s][c] = S1[kk + 16 s][16 c + d] -> land[16 s + c]
        for sgrp in range(2):
            for c in range(16):
                self.add("ds_read_b64", X.land[16 * sgrp + c], (V_P2,), offset=8 * (16 * sgrp * S1_ROW + 16 * c))

    def fwd_pass2(self, X):
        t1, t2 = TMP[1], TMP[2]
        free = X.spare
        for sgrp in range(2):
            out, free = self.dft(X.land[16 * sgrp: 16 * sgrp + 16], free, False)
            for k2 in range(16):
                self.apply_tw(out[k2], LO2, HI2, k2 & 3, k2 >> 2, False, t1, t2)
                off = 8 * S2_ROW * (k2 * 32 + 16 * sgrp)
                base = V_P3
                if k2 >= 8:
                    base, off = V_P3HI, off - 8 * S2_ROW * 8 * 32
                self.add("ds_write_b64", None, (base, out[k2]), offset=off)

    def fwd_read2(self, X):
        # w[bf][2q], w[bf][2q+1] = row_j[q] (float4), j = t (bf 0) and 512 - t (bf 1; thread 0: 256)
        for bf, base in ((0, V_P4A), (1, V_P4B)):
            for q in range(8):
                quad = R("v", X.land[16 * bf + 2 * q].idx, 4)
                self.add("ds_read_b128", quad, (base,), offset=16 * q)

    def fwd_pass3(self, X):
        free, nat = X.spare, []
        for bf in range(2):
            out, free = self.dft(X.land[16 * bf: 16 * bf + 16], free, False)
            nat += out
        X.nat, X.free = nat, free

    def pair(self, X, ia, ib, hq, wk_idx, wj, self_pair):
        """one mirrored bin pair (fft_tile.hpp: pair_split / pair_product / pair_merge).  wk = wj * W_32^wk_idx when wj
        is a register pair, else the constant W_32^wk_idx."""
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

    def product(self, X):
        wj = LO1[1]
        # (thread 0's branch below is skipped by three of the four waves: every spectrum slot is waited for up front,
        # so that no wait sits inside the skipped range)
        self.add(";touch", None, tuple(HQ))
        # every thread but thread 0 of the workgroup: pairs (k3, 16 + 15 - k3)
        self.add("s_mov_b64", EXEC, (S_GEN_EXEC,))
        for k3 in range(16):
            self.pair(X, k3, 16 + (15 - k3), HQ[k3], 2 * k3, wj, False)
        # thread 0 (wave 0, lane 0) owns the two self-mirrored butterflies j = 0 and j = 256
        skip = self.fresh("not_t0")
        if "not0" in self.ablate:          # timing experiment: what thread 0's extra pairs cost the workgroup
            self.add("s_mov_b64", EXEC, (Lit(-1),))
            return
        self.add("s_cmp_eq_u32", None, (S_WAVE0, Lit(0)))
        self.add("s_cbranch_scc1", target=skip)
        self.add("s_mov_b64", EXEC, (Lit(1),))
        self.pair(X, 0, 0, HQ[0], 0, None, True)
        self.pair(X, 8, 8, HQ[8], 16, None, True)
        for k3 in range(1, 8):
            self.pair(X, k3, 16 - k3, HQ[k3], 2 * k3, None, False)
        for k3 in range(8):
            self.pair(X, 16 + k3, 16 + (15 - k3), HQ[9 + k3], 1 + 2 * k3, None, False)
        self.label(skip)
        self.add("s_mov_b64", EXEC, (Lit(-1),))

    def inv_pass1(self, X):
        # rows j of S2 get the inverse 16-point transforms over k3 of the two butterfly sets (element e at row_j + 8 e)
        free = X.free
        for bf, base in ((0, V_P4A), (1, V_P4B)):
            out, free = self.dft(X.nat[16 * bf: 16 * bf + 16], free, True)
            for e in range(16):
                self.add("ds_write_b64", None, (base, out[e]), offset=8 * e)

    def inv_read2(self, X):
        t1, t2 = TMP[1], TMP[2]
        for sgrp in range(2):
            for k2 in range(16):
                off = 8 * S2_ROW * (k2 * 32 + 16 * sgrp)
                base = V_P3
                if k2 >= 8:
                    base, off = V_P3HI, off - 8 * S2_ROW * 8 * 32
                self.add("ds_read_b64", X.land[16 * sgrp + k2], (base,), offset=off)
        for sgrp in range(2):
            for k2 in range(16):
                self.apply_tw(X.land[16 * sgrp + k2], LO2, HI2, k2 & 3, k2 >> 2, True, t1, t2)

    def inv_pass2(self, X):
        free = X.spare
        for sgrp in range(2):
            out, free = self.dft(X.land[16 * sgrp: 16 * sgrp + 16], free, True)
            for c in range(16):
                self.add("ds_write_b64", None, (V_P2, out[c]), offset=8 * (16 * sgrp * S1_ROW + 16 * c))

    def inv_read3(self, X):
        t1, t2 = TMP[1], TMP[2]
        for k1 in range(32):
            base, off = (V_OFF8, 8 * S1_ROW * k1) if k1 < 16 else (V_P1HI, 8 * S1_ROW * (k1 - 16))
            self.add("ds_read_b64", X.land[k1], (base,), offset=off)
        for k1 in range(32):
            self.apply_tw(X.land[k1], LO1, HI1, k1 & 3, k1 >> 2, True, t1, t2)

    def inv_pass3(self, X):
        rows, _ = self.dft(X.land, X.spare, True)
        assert rows == X.rows

    # ---- one tile in bank X while bank Y drains / refills -----------------------------------------------------------
    def iteration(self, X, Y, name):
        out = []
        # top: rotate the output descriptors, decode the tile after this one
        self.sub()
        for k in range(4):
            self.sop("s_mov_b32", ST_Y.sub(k), CUR_Y.sub(k))
        for k in range(4):
            self.sop("s_mov_b32", CUR_Y.sub(k), NX_Y.sub(k))
        self.decode(first=False)
        if self.k["rowmax"]:
            # the previous tile's outputs are still in Y (they leave during this tile's forward transform): tile S_LB - 3
            self.rowmax_accumulate(lambda a: Y.rows[a])
            self.rowmax_flush(3, always=False)
        out += self.sub()
        # forward transform of X; meanwhile: outputs of the previous tile leave Y, then the next window lands in Y
        self.fwd_pass1(X)
        self.barrier()
        self.fwd_read1(X)
        self.barrier()
        self.fwd_pass2(X)
        self.barrier()
        self.fwd_read2(X)
        self.fwd_pass3(X)
        fwd = self.sub()
        stores = [] if "nostore" in self.ablate else self.g_stores(lambda a: Y.rows[a], ST_Y)
        if self.k["carry"]:
            # The next window starts V = 16384 - 512 a_lo samples later: its first a_lo rows are this window's last a_lo
            # rows, still untouched in X at this point.  The output rows of the previous tile that sit in those
            # registers of Y leave first, then the rows are copied (or zeroed when the next tile starts a new row).
            dest = Y.land[: self.a_lo]
            first = [g for g in stores if g[-1].src[0] in dest]
            stores = [g for g in stores if g[-1].src[0] not in dest]
            head = [i for g in first for i in g]
            zero, done = self.fresh("carry_zero"), self.fresh("carry_done")
            # (the wait for this window's registers has to sit in front of the branch, not inside one of its arms)
            head.append(Inst(";touch", None, tuple(X.land) + tuple(dest)))
            head.append(Inst("s_cmp_lg_u32", None, (S_ALO, Lit(0))))
            head.append(Inst("s_cbranch_scc1", None, (), dict(target=zero)))
            for m in range(self.a_lo):
                src = X.land[32 - self.a_lo + m]
                head += [Inst("v_mov_b32", dest[m].sub(0), (src.sub(0),)), Inst("v_mov_b32", dest[m].sub(1), (src.sub(1),))]
            head.append(Inst("s_branch", None, (), dict(target=done)))
            head.append(Label(zero))
            for m in range(self.a_lo):
                head += [Inst("v_mov_b32", dest[m].sub(0), (Lit(0),)), Inst("v_mov_b32", dest[m].sub(1), (Lit(0),))]
            head.append(Label(done))
            out += head
        side = stores + ([] if "nowin" in self.ablate else self.g_window_loads(Y.land, carried=bool(self.k["carry"])))
        out += self.place(fwd, side, self.k["fwd_window"])
        self.product(X)
        out += self.sub()
        # inverse transform; meanwhile: the next window's valid part is copied out (tee), the next spectrum is requested
        self.inv_pass1(X)
        self.barrier()
        self.inv_read2(X)
        self.barrier()
        self.inv_pass2(X)
        self.barrier()
        self.inv_read3(X)
        self.inv_pass3(X)
        if self.k["fake_epilogue"]:
            n = int(self.k["fake_epilogue"])
            for i in range(n):
                t = TMP[i % 6]
                if i % 12 == 5:
                    self.add("v_exp_f32", t.sub(0), (t.sub(0),))
                else:
                    self.add("v_pk_fma_f32", t, (t, S_ONE_NEG, t))
            for grp in self.g_stores(lambda a: X.rows[a], CUR_Y):
                self.prog.extend(grp)
        inv = self.sub()
        tee_st = self.g_stores(lambda a: Y.land[a], NX_T) if self.tee and "notee" not in self.ablate else []
        h_ld = [] if "noh" in self.ablate else self.g_h_loads()
        if self.k["inv_order"] == "h_first":
            side = h_ld + tee_st
        elif self.k["inv_order"] == "mix":
            side, a, b = [], list(tee_st), list(h_ld)
            while a or b:
                if a and (not b or len(a) * len(h_ld) >= len(b) * max(len(tee_st), 1)):
                    side.append(a.pop(0))
                else:
                    side.append(b.pop(0))
        else:
            side = tee_st + h_ld
        out += self.place(inv, side, self.k["inv_window"])
        return out

    def place(self, main, side, window):
        mode = self.k["place"]
        if mode == "spread":
            return interleave(main, side, *window)
        return cluster_at(main, side, lambda i: i.op == "s_barrier", before=(mode == "barrier"))

    # ---- whole kernel -----------------------------------------------------------------------------------------------
    def prologue(self):
        self.sub()
        for k in range(3):
            self.add("s_load_dwordx16", s(ARG0 + 16 * k, 16), (s(0, 2),), offset=64 * k)
        for j in CONST_TW_J:
            c, sn = const_tw_values(j)
            self.mov_lit(S_CTW[j].sub(0), float(c))
            self.mov_lit(S_CTW[j].sub(1), float(sn))
        self.mov_lit(S_ONE_NEG.sub(0), 1.0)
        self.mov_lit(S_ONE_NEG.sub(1), -1.0)
        self.mov_lit(S_C2.sub(0), -2.0)
        self.mov_lit(S_C2.sub(1), 2.0)
        self.mov_lit(S_HUGE, 0x7FFF0000)
        # per-thread addresses
        self.add("v_mov_b32", V_MAX, (Lit(0),))
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
        # j_b = 512 - t, thread 0: 256
        jb = TMP[1].sub(0)
        self.add("v_sub_u32", jb, (Lit(512), V_TID))
        self.add("v_mov_b32", TMP[2].sub(0), (Lit(256),))
        self.add("v_cmp_ne_u32", R("vcc", 0, 2), (Lit(0), V_TID))
        self.add("s_nop", imm=1)
        self.add("v_cndmask_b32", jb, (TMP[2].sub(0), jb, R("vcc", 0, 2)))    # t == 0 ? 256 : 512 - t
        self.add("v_mul_u32_u24", V_P4B, (Lit(8 * S2_ROW), jb))
        # wave index -> S_WAVE0 (1 for the wave that holds thread 0), exec mask of the generic product
        self.add("v_lshrrev_b32", TMP[1].sub(1), (Lit(6), V_TID))
        self.add("s_nop", imm=0)
        self.add("v_readfirstlane_b32", SCR[6], (TMP[1].sub(1),))     # (s2 still holds the workgroup id)
        self.add("s_nop", imm=3)
        self.add("s_cmp_eq_u32", None, (SCR[6], Lit(0)))
        self.sop("s_cselect_b32", S_WAVE0, Lit(1), Lit(0))
        self.sop("s_cselect_b32", S_GEN_EXEC.sub(0), Lit(-2), Lit(-1))
        self.mov_lit(S_GEN_EXEC.sub(1), -1)
        # this workgroup's tiles: logical blocks xcd * per_xcd + w, step wgs_per_xcd, below min((xcd + 1) per_xcd, nblocks)
        self.add("s_waitcnt", lgkmcnt=0)
        if self.k["carry"]:
            # consecutive tiles: workgroup g takes [g * per_wg, (g + 1) * per_wg) (per_wg arrives in `per_xcd`)
            self.sop("s_mul_i32", S_LB, s(2), A("per_xcd"))
            self.sop("s_add_u32", SCR[5], S_LB, A("per_xcd"))
            self.sop("s_min_u32", S_END, SCR[5], A("nblocks"))
            self.mov_lit(S_STRIDE, 1)
        else:
            self.sop("s_and_b32", SCR[3], s(2), Lit(7))
            self.sop("s_lshr_b32", SCR[4], s(2), Lit(3))
            self.sop("s_mul_i32", SCR[5], SCR[3], A("per_xcd"))
            self.sop("s_add_u32", S_LB, SCR[5], SCR[4])
            self.sop("s_add_u32", SCR[5], SCR[5], A("per_xcd"))
            self.sop("s_min_u32", S_END, SCR[5], A("nblocks"))
            self.sop("s_mov_b32", S_STRIDE, A("wgs_per_xcd"))
        done = ".Lnothing"
        self.add("s_cmp_ge_u32", None, (S_LB, S_END))
        self.add("s_cbranch_scc1", target=done)
        # resident twiddles (row-major table: row r at tw + 2048 r + 8 t)
        self.sop("s_mov_b32", NX_X.sub(0), A("tw_lo"))
        self.sop("s_and_b32", NX_X.sub(1), A("tw_hi"), Lit(0xFFFF))
        self.mov_lit(NX_X.sub(2), 20 * 2048)
        self.mov_lit(NX_X.sub(3), RSRC_FLAGS)
        for row, reg in TW_ROWS.items():
            self.mov_lit(S_OFF, 2048 * row)
            self.add("buffer_load_dwordx2", reg, (V_OFF8, NX_X, S_OFF))
        # tile 0: descriptors, window into bank A, tee, spectrum; no previous output
        self.decode(first=True)
        for k in range(4):
            self.mov_lit(CUR_Y.sub(k), 0 if k < 3 else RSRC_FLAGS)
        for grp in self.g_window_loads(BANK_A.land):
            self.prog.extend(grp)
        if self.tee:
            for grp in self.g_stores(lambda a: BANK_A.land[a], NX_T):
                self.prog.extend(grp)
        for grp in self.g_h_loads():
            self.prog.extend(grp)
        return self.sub()

    def exit_block(self, X):
        self.sub()
        if self.k["rowmax"]:
            self.rowmax_accumulate(lambda a: X.rows[a])       # the run's last tile: S_LB - 2 (the decode went one past it)
            self.rowmax_flush(2, always=True)
        for grp in self.g_stores(lambda a: X.rows[a], CUR_Y):
            self.prog.extend(grp)
        self.add("s_endpgm")
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


def kernel_name(tee, a_lo):
    return f"gfx_fftconv_pipe_t{int(tee)}_o{a_lo}"


# (tee, a_lo): overlap O = 512 a_lo covers filters of 512 (a_lo - 1) + 2 .. 512 a_lo + 1 taps; 8 is the equalisers' 4001
VARIANTS = [(tee, a_lo) for a_lo in (1, 2, 4, 8, 16) for tee in (False, True)]


def kernel_text(name, prog):
    body = render(prog)
    return f"""
\t.text
\t.protected\t{name}
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
{name}:
{body}
.Lfunc_end_{name}:
\t.size\t{name}, .Lfunc_end_{name}-{name}

\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {TILE_LDS_BYTES}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {KERNARG_BYTES}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr {NUM_VGPR}
\t\t.amdhsa_next_free_sgpr {NUM_SGPR}
\t\t.amdhsa_accum_offset {(NUM_VGPR + 3) // 4 * 4}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
"""


def metadata(names):
    ks = []
    for n in names:
        ks.append(f"""  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           {KERNARG_BYTES}
        .value_kind:     by_value
    .group_segment_fixed_size: {TILE_LDS_BYTES}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {KERNARG_BYTES}
    .max_flat_workgroup_size: 256
    .name:           {n}
    .private_segment_fixed_size: 0
    .sgpr_count:     {NUM_SGPR + 2}
    .sgpr_spill_count: 0
    .symbol:         {n}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {NUM_VGPR}
    .vgpr_spill_count: 0
    .wavefront_size: 64
""")
    return ("\t.amdgpu_metadata\n---\namdhsa.kernels:\n" + "".join(ks) +
            "amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\n\t.end_amdgpu_metadata\n")


def generate(**knobs):
    txt = '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.amdhsa_code_object_version 6\n'
    names = []
    for tee, a_lo in VARIANTS:
        name = kernel_name(tee, a_lo)
        prog = PipeGen(tee, a_lo, **knobs).build()
        # labels are local to a kernel: make them unique per kernel
        for i in prog:
            if isinstance(i, Label):
                i.name = i.name + "_" + name
            elif "target" in i.mods:
                i.mods["target"] = i.mods["target"] + "_" + name
        txt += kernel_text(name, prog)
        names.append(name)
    from . import gen_corr_pipe       # the filter-gradient kernel shares the code object (and most of this generator)

    name, prog = gen_corr_pipe.kernel(**{k: v for k, v in knobs.items() if k == "codelet"})
    txt += kernel_text(name, prog)
    names.append(name)
    return txt + metadata(names)


def main(argv):
    """gen_fftconv_pipe.py out.s | --hsaco out.hsaco  [knob=value ...]   (knobs: see KNOBS; tuples as a,b)"""
    import os
    import subprocess
    import tempfile

    knobs, pos, hsaco = {}, [], None
    it = iter(argv)
    for a in it:
        if a == "--hsaco":
            hsaco = next(it)
        elif "=" in a:
            k, val = a.split("=", 1)
            cur = KNOBS[k]
            knobs[k] = tuple(float(x) for x in val.split(",")) if isinstance(cur, tuple) else type(cur)(val)
        else:
            pos.append(a)
    text = generate(**knobs)
    if hsaco is None:
        with open(pos[0] if pos else "/dev/stdout", "w") as f:
            f.write(text)
        return
    llvm = os.environ.get("GRAFX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "k.s"), "w") as f:
            f.write(text)
        subprocess.run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                        os.path.join(d, "k.s"), "-o", os.path.join(d, "k.o")], check=True)
        subprocess.run([os.path.join(llvm, "ld.lld"), "-shared", os.path.join(d, "k.o"), "-o", hsaco], check=True)


if __name__ == "__main__":
    main(sys.argv[1:])
