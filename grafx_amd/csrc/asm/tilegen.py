"""Building blocks of the hand-scheduled FFT-tile kernels (gfx950): the arithmetic of csrc/fft_tile.hpp emitted as explicit
instructions on explicit registers.

Same transform, same LDS images, same thread layout as the compiler-built tile (8192 complex points = 32 x 16 x 16 per
256-thread workgroup, DIF forward, mirrored inverse, every bin next to its mirror in one thread) -- what changes is who
decides where a value lives and when a load or a store is issued.  A complex value is an even-aligned VGPR pair; every
function here appends `isa.Inst` objects to a list and never allocates behind the caller's back: the register map is an
argument.

Stages are emitted as lists so that the kernel generator can interleave memory instructions between them
(`interleave`), and an automatic pass (`insert_waitcnt`) places the s_waitcnt instructions.
"""
import math

from .isa import EXEC, Inst, Label, Lit, R, s, v  # noqa: F401

S1_ROW = 272      # float2 units, see fft_tile.hpp
S2_ROW = 18
TILE_LDS_BYTES = 512 * S2_ROW * 8


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


class Emit:
    """An instruction list with small helpers.  `const` maps names to SGPR pairs prepared by the kernel prologue."""

    def __init__(self):
        self.prog = []

    def add(self, op, dst=None, src=(), comment=None, **mods):
        self.prog.append(Inst(op, dst, src, mods, comment))
        return self.prog[-1]

    def label(self, name):
        self.prog.append(Label(name))

    # ---- complex arithmetic on packed fp32 (see fft_tile.hpp for the derivations)
    def cadd(self, d, a, b):
        self.add("v_pk_add_f32", d, (a, b))

    def csub(self, d, a, b):
        self.add("v_pk_add_f32", d, (a, b), neg_lo=[0, 1], neg_hi=[0, 1])

    def sub_mul_neg_i(self, d, a, b):     # (a - b) * (-i)
        self.add("v_pk_add_f32", d, (a, b), op_sel=[1, 1], op_sel_hi=[0, 0], neg_lo=[0, 1], neg_hi=[1, 0])

    def sub_mul_pos_i(self, d, a, b):     # (a - b) * (+i)
        self.add("v_pk_add_f32", d, (a, b), op_sel=[1, 1], op_sel_hi=[0, 0], neg_lo=[1, 0], neg_hi=[0, 1])

    def add_conj(self, d, a, b):          # a + conj(b)
        self.add("v_pk_add_f32", d, (a, b), neg_hi=[0, 1])

    def sub_conj_mul_neg_i(self, d, a, b):  # (a - conj(b)) * (-i) = (a.y + b.y, b.x - a.x)
        self.add("v_pk_add_f32", d, (a, b), op_sel=[1, 1], op_sel_hi=[0, 0], neg_hi=[1, 0])

    def add_mul_pos_i(self, d, a, b):     # a + i*b = (a.x - b.y, a.y + b.x)
        self.add("v_pk_add_f32", d, (a, b), op_sel=[0, 1], op_sel_hi=[1, 0], neg_lo=[0, 1])

    def conj_sub_mul_pos_i(self, d, a, b):  # conj(a - i*b) = (a.x + b.y, b.x - a.y)
        self.add("v_pk_add_f32", d, (a, b), op_sel=[0, 1], op_sel_hi=[1, 0], neg_hi=[1, 0])

    def cmul(self, d, a, w, conj=False, tmp=None):
        """d = a * w (or a * conj(w)); w may be an SGPR pair.  d may equal a only if `tmp` is given."""
        neg = dict(neg_hi=[0, 1, 0]) if conj else dict(neg_lo=[0, 1, 0])
        if d == a:
            assert tmp is not None and tmp != a
            self.add("v_pk_mul_f32", tmp, (a, w), op_sel_hi=[1, 0])
            self.add("v_pk_fma_f32", d, (a, w, tmp), op_sel=[1, 1, 0], op_sel_hi=[0, 1, 1], **neg)
        else:
            assert d != w
            self.add("v_pk_mul_f32", d, (a, w), op_sel_hi=[1, 0])
            self.add("v_pk_fma_f32", d, (a, w, d), op_sel=[1, 1, 0], op_sel_hi=[0, 1, 1], **neg)

    def cmac(self, acc, a, w):
        """acc += a * w (in place)"""
        self.add("v_pk_fma_f32", acc, (a, w, acc), op_sel_hi=[1, 0, 1])
        self.add("v_pk_fma_f32", acc, (a, w, acc), op_sel=[1, 1, 0], op_sel_hi=[0, 1, 1], neg_lo=[0, 1, 0])


# ---- constant twiddles W_32^idx = exp(-+ 2 pi i idx / 32) from four SGPR pairs ------------------------------------
# The prologue stores P_j = (cos(2 pi j / 32), -sin(2 pi j / 32)), j = 1..4, in SGPR pairs.  Every other W_32^idx is one
# of those two numbers per component up to sign, picked with op_sel and negated with neg_lo / neg_hi: no more registers.
CONST_TW_J = (1, 2, 3, 4)


def const_tw_values(j):
    return math.cos(2 * math.pi * j / 32), -math.sin(2 * math.pi * j / 32)


def _match_const(idx32, inv):
    """(j, ir, sr, ii, si): Re w = sr * P_j[ir], Im w = si * P_j[ii] for w = exp(-+ 2 pi i idx32 / 32)"""
    ang = 2 * math.pi * (idx32 % 32) / 32
    wr, wi = math.cos(ang), (math.sin(ang) if inv else -math.sin(ang))
    for j in CONST_TW_J:
        p = const_tw_values(j)
        for ir in (0, 1):
            for sr in (1, -1):
                if abs(sr * p[ir] - wr) > 1e-12:
                    continue
                for ii in (0, 1):
                    for si in (1, -1):
                        if abs(si * p[ii] - wi) < 1e-12:
                            return j, ir, sr, ii, si
    raise ValueError(f"W_32^{idx32} is not a general twiddle")


class TileGen(Emit):
    def __init__(self, sgpr_const_tw, sgpr_one_neg=None, sgpr_c2=None):
        """sgpr_const_tw: dict j -> SGPR pair holding P_j; sgpr_one_neg: SGPR pair (1.0, -1.0); sgpr_c2: (-2.0, 2.0)"""
        super().__init__()
        self.ctw = sgpr_const_tw
        self.one_neg = sgpr_one_neg
        self.c2 = sgpr_c2

    def cmul_const(self, d, a, idx32, inv, tmp=None):
        """d = a * W_32^(+-idx32) for a general (non-trivial) constant twiddle: two packed instructions"""
        j, ir, sr, ii, si = _match_const(idx32, inv)
        P = self.ctw[j]
        nr = int(sr < 0)
        t = d
        if d == a:
            assert tmp is not None
            t = tmp
        self.add("v_pk_mul_f32", t, (a, P), op_sel=[0, ir], op_sel_hi=[1, ir], neg_lo=[0, nr], neg_hi=[0, nr])
        # lo = -a.y * wi + t.x ; hi = a.x * wi + t.y
        self.add("v_pk_fma_f32", d, (a, P, t), op_sel=[1, ii, 0], op_sel_hi=[0, ii, 1],
                 neg_lo=[0, int(si > 0), 0], neg_hi=[0, int(si < 0), 0])

    def mul_const_any(self, d, a, idx32, inv, tmp=None):
        """d = a * W_32^(+-idx32), any idx32 (trivial ones are one instruction or a copy)"""
        k = idx32 % 32
        if k == 0:
            if d != a:
                self.add("v_pk_mul_f32", d, (a, self.one_neg), op_sel=[0, 0], op_sel_hi=[1, 0])     # * (1, 1)
            return
        if k == 16:
            self.add("v_pk_mul_f32", d, (a, self.one_neg), op_sel=[0, 1], op_sel_hi=[1, 1])         # * (-1, -1)
            return
        if k in (8, 24):
            neg_i = (k == 8) != bool(inv)    # forward idx 8 = -i ; inverse idx 8 = +i
            # a * (-i) = (a.y, -a.x) ; a * (+i) = (-a.y, a.x)
            if neg_i:
                self.add("v_pk_mul_f32", d, (a, self.one_neg), op_sel=[1, 0], op_sel_hi=[0, 1])
            else:
                self.add("v_pk_mul_f32", d, (a, self.one_neg), op_sel=[1, 1], op_sel_hi=[0, 0])
            return
        self.cmul_const(d, a, k, inv, tmp)

    # ---- in-register radix-2 DIF DFT of n = 16 or 32 points; result for frequency k ends up at vals[brev(k)] ----------
    # Every butterfly is IN PLACE, so a value never changes its register (the register map of a whole tile is static):
    #   trivial twiddle (1):   A' = A + B ;  B' = A' - 2 B                      (= A - B, one rounding more than a - b)
    #   twiddle -+i:           A' = A + B ;  B' = (A' - 2 B) * (-+i)            (one packed FMA with op_sel / neg)
    #   general twiddle:       T = A - B ; A' = A + B ; B' = T * w              (T: the caller's scratch pair)
    def dif(self, vals, tmp, inv):
        n = len(vals)
        C2 = self.c2        # SGPR pair (-2.0, 2.0)
        length = n
        while length >= 2:
            half = length // 2
            for base in range(0, n, length):
                for j in range(half):
                    a, b = vals[base + j], vals[base + j + half]
                    idx = j * (32 // length)
                    if idx == 0:
                        self.cadd(a, a, b)
                        self.add("v_pk_fma_f32", b, (b, C2, a), op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1])
                    elif idx == 8 and not inv:   # (A - B)(-i) = (A'.y - 2 B.y, 2 B.x - A'.x)
                        self.cadd(a, a, b)
                        self.add("v_pk_fma_f32", b, (b, C2, a), op_sel=[1, 0, 1], op_sel_hi=[0, 1, 0], neg_hi=[0, 0, 1])
                    elif idx == 8:               # (A - B)(+i) = (2 B.y - A'.y, A'.x - 2 B.x)
                        self.cadd(a, a, b)
                        self.add("v_pk_fma_f32", b, (b, C2, a), op_sel=[1, 1, 1], op_sel_hi=[0, 0, 0], neg_lo=[0, 0, 1])
                    else:
                        self.csub(tmp, a, b)
                        self.cadd(a, a, b)
                        self.cmul_const(b, tmp, idx, inv)
            length //= 2

    # ---- in-register radix-2 DIT DFT with fused multiply-adds ---------------------------------------------------------
    # A general butterfly (a, b) -> (a + w b, a - w b) is THREE packed instructions: T = a + w b as two FMAs, then
    # b' = 2 a - T.  The sum lands in a scratch pair and a's old register becomes the scratch of the next butterfly, so
    # values move between registers -- statically: the generator tracks where every value lives.  Trivial twiddles
    # (1, -+i) stay in place with two instructions as in `dif`.
    def dit(self, inputs, free, inv):
        """DFT of `inputs` (register pairs in NATURAL input order, n = 16 or 32).  Returns (out, free): out[k] = the pair
        that holds frequency k, free = the pair left over.  Registers used: exactly inputs + [free]."""
        n = len(inputs)
        bits = n.bit_length() - 1
        pos = [inputs[brev(i, bits)] for i in range(n)]
        C2 = self.c2
        h = 1
        while h < n:
            for base in range(0, n, 2 * h):
                for j in range(h):
                    ia, ib = base + j, base + j + h
                    a, b = pos[ia], pos[ib]
                    idx = j * (32 // (2 * h))          # w = W_{2h}^j = W_32^idx
                    if idx == 0:
                        self.cadd(a, a, b)
                        self.add("v_pk_fma_f32", b, (b, C2, a), op_sel=[0, 0, 0], op_sel_hi=[1, 0, 1])          # a' - 2 b
                    elif idx == 8:
                        # w b = -+i b.  forward (-i): a' = a - i b = (a.x + b.y, a.y - b.x); b' = 2 a - a'
                        # in place: a' first needs old a for b' -> compute b' = a + i b from (a, b), then a' = 2 a - b'
                        if not inv:
                            self.add("v_pk_add_f32", free, (a, b), op_sel=[0, 1], op_sel_hi=[1, 0], neg_hi=[0, 1])   # a - i b
                        else:
                            self.add("v_pk_add_f32", free, (a, b), op_sel=[0, 1], op_sel_hi=[1, 0], neg_lo=[0, 1])   # a + i b
                        self.add("v_pk_fma_f32", b, (a, C2, free), op_sel=[0, 1, 0], op_sel_hi=[1, 1, 1],
                                 neg_lo=[0, 0, 1], neg_hi=[0, 0, 1])                                                  # 2 a - a'
                        pos[ia], free = free, a
                    else:
                        j_, ir, sr, ii, si = _match_const(idx, inv)
                        P = self.ctw[j_]
                        nr = int(sr < 0)
                        self.add("v_pk_fma_f32", free, (b, P, a), op_sel=[0, ir, 0], op_sel_hi=[1, ir, 1],
                                 neg_lo=[0, nr, 0], neg_hi=[0, nr, 0])
                        self.add("v_pk_fma_f32", free, (b, P, free), op_sel=[1, ii, 0], op_sel_hi=[0, ii, 1],
                                 neg_lo=[0, int(si > 0), 0], neg_hi=[0, int(si < 0), 0])
                        self.add("v_pk_fma_f32", b, (a, C2, free), op_sel=[0, 1, 0], op_sel_hi=[1, 1, 1],
                                 neg_lo=[0, 0, 1], neg_hi=[0, 0, 1])
                        pos[ia], free = free, a
            h *= 2
        return pos, free

    # ---- per-thread twiddles (two-level, as TileTw in fft_tile.hpp) -----------------------------------------------
    def apply_tw(self, e, lo, hi, il, ih, conj, t1, t2):
        """e *= lo[il] * hi[ih] (conjugated for the inverse); lo / hi: dicts index -> VGPR pair (index 0 = one, absent)"""
        if il == 0 and ih == 0:
            return
        if il == 0:
            w = hi[ih]
        elif ih == 0:
            w = lo[il]
        else:
            self.cmul(t2, lo[il], hi[ih])
            w = t2
        self.cmul(e, e, w, conj=conj, tmp=t1)


def interleave(main, side, first=0.0, last=1.0):
    """Spread the instructions of `side` evenly through main[first*len : last*len]; order within both lists is kept."""
    if not side:
        return list(main)
    n = len(main)
    lo, hi = int(first * n), max(int(last * n), int(first * n) + 1)
    out = list(main[:lo])
    span = hi - lo
    k = 0
    for idx in range(span):
        # side instruction k goes in front of main instruction lo + idx once idx passes its slot
        while k < len(side) and (k + 0.5) * span / len(side) <= idx:
            out.extend(side[k] if isinstance(side[k], list) else [side[k]])
            k += 1
        out.append(main[lo + idx])
    for grp in side[k:]:
        out.extend(grp if isinstance(grp, list) else [grp])
    out.extend(main[hi:])
    return out


def cluster_at(main, side, anchors, before=True, weights=None):
    """Put the groups of `side` in clusters next to the anchor instructions of `main` (predicate `anchors(inst)`), e.g. in
    front of every s_barrier: a wave that has to wait there anyway can spend the wait in the memory pipeline's queue."""
    pos = [k for k, i in enumerate(main) if not isinstance(i, Label) and anchors(i)]
    if not pos or not side:
        return interleave(main, side)
    w = weights or [1.0] * len(pos)
    tot = float(sum(w[:len(pos)]))
    counts, acc, given = [], 0.0, 0
    for k in range(len(pos)):
        acc += w[k] / tot * len(side)
        c = int(round(acc)) - given
        counts.append(c)
        given += c
    counts[-1] += len(side) - given
    out, k0, sidx = [], 0, 0
    for p, c in zip(pos, counts):
        cut = p if before else p + 1
        out.extend(main[k0:cut])
        for grp in side[sidx:sidx + c]:
            out.extend(grp if isinstance(grp, list) else [grp])
        sidx += c
        k0 = cut
    out.extend(main[k0:])
    return out


# ---- automatic s_waitcnt placement --------------------------------------------------------------------------------
_VM_LOAD = ("buffer_load_dword", "buffer_load_dwordx2", "buffer_load_dwordx4")
_VM_STORE = ("buffer_store_dword", "buffer_store_dwordx2", "buffer_store_dwordx4", "buffer_atomic_umax")
_DS_READ = ("ds_read_b32", "ds_read_b64", "ds_read_b128")
_DS_WRITE = ("ds_write_b32", "ds_write_b64", "ds_write_b128")
_SMEM = ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16", "s_memrealtime")


def _regs_read(i):
    out = []
    for x in i.src:
        if isinstance(x, R) and x.kind in ("v", "s"):
            out.extend(x.regs())
    return out


def _regs_written(i):
    if i.dst is not None and isinstance(i.dst, R) and i.dst.kind in ("v", "s"):
        return i.dst.regs()
    return []


class _Counter:
    def __init__(self, cap):
        self.q = []      # oldest first: set of destination registers (empty for stores)
        self.cap = cap

    def issue(self, regs):
        # (the queue is not capped: whether the hardware stalls at `cap` outstanding operations or not, waiting for
        # min(N, cap) is safe for an operation with N younger ones)
        self.q.append(set(regs))

    def need(self, regs):
        """largest count N such that `s_waitcnt cnt(N)` guarantees none of `regs` is pending; None if nothing pending"""
        regs = set(regs)
        last = -1
        for k, dst in enumerate(self.q):
            if dst & regs:
                last = k
        if last < 0:
            return None
        return min(len(self.q) - 1 - last, self.cap)

    def wait(self, n):
        while len(self.q) > n:
            self.q.pop(0)

    def has_writes(self):
        return any(len(x) == 0 for x in self.q)


def insert_waitcnt(trace_blocks, blocks):
    """Place s_waitcnt instructions.

    `blocks`: dict name -> list of Inst/Label (straight-line code, branches only at the end of a block or skipping
    forward inside it without memory instructions in the skipped range).  `trace_blocks`: lists of block names, each a
    possible execution order (prologue, loop bodies repeated, an exit).  Every instruction gets the strictest wait any
    trace needs in front of it; the result is a dict name -> new instruction list.  LDS writes are completed before
    every s_barrier.  (The emulator re-checks the result by execution.)"""
    need = {name: [dict() for _ in blk] for name, blk in blocks.items()}
    for trace in trace_blocks:
        vm, lg = _Counter(63), _Counter(15)
        for name in trace:
            for pos, i in enumerate(blocks[name]):
                if isinstance(i, Label):
                    continue
                w = need[name][pos]
                if i.op == "s_waitcnt":
                    if i.mods.get("vmcnt") is not None:
                        vm.wait(i.mods["vmcnt"])
                    if i.mods.get("lgkmcnt") is not None:
                        lg.wait(i.mods["lgkmcnt"])
                    continue
                touched = _regs_read(i) + _regs_written(i)
                if i.op in _VM_STORE:
                    touched = _regs_read(i)
                for cnt, key in ((vm, "vmcnt"), (lg, "lgkmcnt")):
                    n = cnt.need(touched)
                    if n is not None:
                        w[key] = n if key not in w else min(w[key], n)
                        cnt.wait(n)
                if i.op == "s_barrier" and lg.q:
                    w["lgkmcnt"] = 0
                    lg.wait(0)
                if i.op == "s_endpgm":
                    pass
                if i.op in _VM_LOAD:
                    vm.issue(_regs_written(i))
                elif i.op in _VM_STORE:
                    vm.issue([])
                elif i.op in _DS_READ or i.op in _SMEM:
                    lg.issue(_regs_written(i))
                elif i.op in _DS_WRITE:
                    lg.issue([])
    out = {}
    for name, blk in blocks.items():
        new = []
        for pos, i in enumerate(blk):
            w = need[name][pos]
            if w:
                new.append(Inst("s_waitcnt", None, (), dict(vmcnt=w.get("vmcnt"), lgkmcnt=w.get("lgkmcnt"))))
            new.append(i)
        out[name] = new
    return out
