"""A small gfx950 instruction layer for hand-scheduled kernels: an IR, its assembly text, and an emulator.

The kernels under csrc/asm are GENERATED: a Python schedule emits a list of `Inst` objects with explicit
physical registers; `render()` prints the .s file the ROCm assembler takes, and `Emulator` executes the very
same list wave by wave on numpy arrays (4 waves of 64 lanes, shared LDS, buffer descriptors with the hardware
range check), so that a schedule is checked against the reference convolution on the CPU before it ever
reaches a GPU.  The emulator also polices what an assembler does not:

  * vmcnt / lgkmcnt: reading (or overwriting) a register whose load has not been waited for is an error;
  * LDS hand-offs: a read of data another wave wrote in the same barrier interval is an error (missing s_barrier),
    and waves are run one after the other between barriers, so a missing barrier also shows as wrong numbers;
  * operand rules of the packed-FP32 instructions (even-aligned pairs, one SGPR pair per instruction).

Only the instructions the generators use are modelled (see `_EXEC`).  This is build tooling and test
infrastructure, not part of the shipped library.
"""
import numpy as np

WAVE = 64


# ---------------------------------------------------------------------------------------------- operands
class R:
    """A register operand: kind 'v' or 's', first index, width in dwords."""
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        self.kind, self.idx, self.n = kind, int(idx), int(n)

    def __repr__(self):
        if self.kind in ("vcc", "exec"):
            return self.kind
        if self.n == 1:
            return f"{self.kind}{self.idx}"
        return f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def __eq__(self, o):
        return isinstance(o, R) and (self.kind, self.idx, self.n) == (o.kind, o.idx, o.n)

    def __hash__(self):
        return hash((self.kind, self.idx, self.n))

    def sub(self, i, n=1):
        assert 0 <= i and i + n <= self.n
        return R(self.kind, self.idx + i, n)

    def regs(self):
        return [(self.kind, self.idx + i) for i in range(self.n)]


def v(i, n=1):
    return R("v", i, n)


def s(i, n=1):
    return R("s", i, n)


VCC = R("vcc", 0, 2)
EXEC = R("exec", 0, 2)


class Lit:
    """A 32-bit literal / inline constant (int or float)."""
    __slots__ = ("val",)

    def __init__(self, val):
        self.val = val

    def __repr__(self):
        if isinstance(self.val, float):
            return repr(float(self.val))
        return hex(self.val) if abs(self.val) > 64 else str(self.val)

    def bits(self):
        if isinstance(self.val, float):
            return np.float32(self.val).view(np.uint32)
        return np.uint32(self.val & 0xFFFFFFFF)


class Inst:
    __slots__ = ("op", "dst", "src", "mods", "comment")

    def __init__(self, op, dst=None, src=(), mods=None, comment=None):
        self.op, self.dst, self.src, self.mods, self.comment = op, dst, tuple(src), mods or {}, comment


class Label:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name


# ---------------------------------------------------------------------------------------------- text
def _fmt_mods(m, keys):
    out = []
    for k in keys:
        if k in m and m[k] is not None:
            out.append(f"{k}:[{','.join(str(int(b)) for b in m[k])}]")
    return " ".join(out)


def render_inst(i):
    if isinstance(i, Label):
        return f"{i.name}:"
    op, m = i.op, i.mods
    c = f"  ; {i.comment}" if i.comment else ""
    if op.startswith("v_pk_"):
        ops = ", ".join(map(repr, (i.dst,) + i.src))
        mods = _fmt_mods(m, ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"))
        return f"\t{op} {ops} {mods}".rstrip() + c
    if op == "v_max3_f32":      # |x| on the sources named by the `abs` modifier list
        ab = m.get("abs") or [0, 0, 0]
        srcs = ", ".join(f"|{x!r}|" if ab[k] else repr(x) for k, x in enumerate(i.src))
        return f"\t{op} {i.dst!r}, {srcs}" + c
    if op == "buffer_atomic_umax":   # (data, rsrc, soffset): no vector address ("off"), every lane hits rsrc + soffset
        data, rsrc, soff = i.src
        return f"\t{op} {data!r}, off, {rsrc!r}, {soff!r}" + c
    if op.startswith("buffer_load") or op.startswith("buffer_store"):
        data = i.dst if op.startswith("buffer_load") else i.src[0]
        vaddr, rsrc, soff = (i.src if op.startswith("buffer_load") else i.src[1:])
        txt = f"\t{op} {data!r}, {vaddr!r}, {rsrc!r}, {soff!r} offen"
        if m.get("offset"):
            txt += f" offset:{m['offset']}"
        for flag in ("sc0", "sc1", "nt"):
            if m.get(flag):
                txt += f" {flag}"
        return txt + c
    if op.startswith("ds_read"):
        txt = f"\t{op} {i.dst!r}, {i.src[0]!r}"
        if m.get("offset"):
            txt += f" offset:{m['offset']}"
        return txt + c
    if op.startswith("ds_write"):
        txt = f"\t{op} {i.src[0]!r}, {i.src[1]!r}"
        if m.get("offset"):
            txt += f" offset:{m['offset']}"
        return txt + c
    if op == "s_waitcnt":
        parts = []
        if m.get("vmcnt") is not None:
            parts.append(f"vmcnt({m['vmcnt']})")
        if m.get("lgkmcnt") is not None:
            parts.append(f"lgkmcnt({m['lgkmcnt']})")
        return "\ts_waitcnt " + " ".join(parts) + c
    if op.startswith("s_load_dword"):
        return f"\t{op} {i.dst!r}, {i.src[0]!r}, {hex(m.get('offset', 0))}" + c
    if op == ";touch":
        return "\t; (registers needed from here on: " + ", ".join(map(repr, i.src)) + ")"
    if op in ("s_barrier", "s_endpgm"):
        return f"\t{op}" + c
    if op in ("s_nop", "s_setprio", "s_sleep"):
        return f"\t{op} {m['imm']}" + c
    if op.startswith("s_cbranch") or op == "s_branch":
        return f"\t{op} {m['target']}" + c
    if op.startswith("s_cmp"):
        return f"\t{op} {i.src[0]!r}, {i.src[1]!r}" + c
    if op.startswith("v_cmp"):
        return f"\t{op} {i.dst!r}, {i.src[0]!r}, {i.src[1]!r}" + c
    if op == "v_cndmask_b32":
        return f"\t{op} {i.dst!r}, {i.src[0]!r}, {i.src[1]!r}, {i.src[2]!r}" + c
    ops = ", ".join(map(repr, ((i.dst,) if i.dst is not None else ()) + i.src))
    return f"\t{op} {ops}" + c


def render(prog):
    return "\n".join(render_inst(i) for i in prog) + "\n"


# ---------------------------------------------------------------------------------------------- emulator
class EmuError(Exception):
    pass


class Buffer:
    """Flat simulated device memory: named allocations at 256-byte aligned fake addresses."""

    def __init__(self):
        self.chunks = []   # (base, np.uint32 array)
        self.next = 0x1000_0000

    def alloc(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint32).reshape(-1).copy()
        base = self.next
        self.chunks.append((base, a))
        self.next = (base + a.nbytes + 0xFFFF) & ~0xFFF
        return base

    def find(self, addr):
        for base, a in self.chunks:
            if base <= addr < base + a.nbytes:
                return a, (addr - base) // 4
        raise EmuError(f"access to unmapped address {hex(addr)}")

    def read_back(self, base, dtype=np.float32):
        for b, a in self.chunks:
            if b == base:
                return a.view(dtype)
        raise KeyError(base)


class Wave:
    def __init__(self, wid, nv=256, ns=104):
        self.wid = wid
        self.v = np.zeros((nv, WAVE), dtype=np.uint32)
        self.s = np.zeros(ns + 8, dtype=np.uint32)
        self.vcc = np.zeros(WAVE, dtype=bool)
        self.exec = np.ones(WAVE, dtype=bool)
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        # outstanding memory operations, oldest first: (kind, set of ('v'|'s', idx) destination registers)
        self.vm = []
        self.lgkm = []
        self.pending = {}   # (kind, idx) -> 'vm' | 'lgkm'
        self.n_inst = {}


def _mask64(m):
    return int(sum(1 << i for i in range(WAVE) if m[i]))


class Emulator:
    """Executes a program (list of Inst / Label) for one workgroup of `nwaves` waves."""

    def __init__(self, prog, mem, lds_bytes, nwaves=4, kernarg=b"", wg_id=0, rng=None, check_lds_races=True):
        self.prog = list(prog)
        self.labels = {p.name: i for i, p in enumerate(self.prog) if isinstance(p, Label)}
        self.mem = mem
        self.lds = np.zeros(lds_bytes // 4, dtype=np.uint32)
        # LDS race bookkeeping: per dword, (epoch, wave) of the last write
        self.lds_wepoch = np.full(lds_bytes // 4, -1, dtype=np.int64)
        self.lds_wwave = np.full(lds_bytes // 4, -1, dtype=np.int64)
        self.lds_repoch = np.full(lds_bytes // 4, -1, dtype=np.int64)
        self.lds_rwave = np.full(lds_bytes // 4, -1, dtype=np.int64)
        self.epoch = 0
        self.check_lds_races = check_lds_races
        self.kernarg_base = mem.alloc(np.frombuffer(kernarg + b"\0" * (-len(kernarg) % 4), dtype=np.uint32))
        self.waves = []
        for w in range(nwaves):
            wv = Wave(w)
            wv.v[0] = np.arange(WAVE, dtype=np.uint32) + WAVE * w     # workitem id x
            wv.s[0] = np.uint32(self.kernarg_base & 0xFFFFFFFF)
            wv.s[1] = np.uint32(self.kernarg_base >> 32)
            wv.s[2] = np.uint32(wg_id)
            self.waves.append(wv)
        self.rng = rng or np.random.default_rng(0)

    # ---- register access with hazard checks
    def _check_ready(self, w, reg, what):
        for key in reg.regs():
            if key in w.pending:
                raise EmuError(f"wave {w.wid} pc {w.pc}: {what} {reg!r} while its {w.pending[key]} load is outstanding "
                               f"(missing s_waitcnt) -- {render_inst(self.prog[w.pc]).strip()}")

    def rd(self, w, x, lanes=True):
        """dword value(s) of a 1-dword operand: array of 64 (vector context)."""
        if isinstance(x, Lit):
            return np.full(WAVE, x.bits(), dtype=np.uint32)
        if x.kind == "v":
            self._check_ready(w, x, "read of")
            return w.v[x.idx]
        if x.kind == "s":
            self._check_ready(w, x, "read of")
            return np.full(WAVE, w.s[x.idx], dtype=np.uint32)
        raise EmuError(f"bad operand {x!r}")

    def rds(self, w, x):
        if isinstance(x, Lit):
            return np.uint32(x.bits())
        if x.kind == "s":
            self._check_ready(w, x, "read of")
            return w.s[x.idx]
        if x.kind == "vcc":
            return None
        raise EmuError(f"scalar read of {x!r}")

    def rd64s(self, w, x):
        if x.kind == "s":
            self._check_ready(w, x, "read of")
            return int(w.s[x.idx]) | (int(w.s[x.idx + 1]) << 32)
        if x.kind == "vcc":
            return _mask64(w.vcc)
        if x.kind == "exec":
            return _mask64(w.exec)
        raise EmuError(f"64-bit scalar read of {x!r}")

    def wr64s(self, w, x, val):
        if x.kind == "s":
            self._check_ready(w, x, "write of")
            w.s[x.idx] = np.uint32(val & 0xFFFFFFFF)
            w.s[x.idx + 1] = np.uint32((val >> 32) & 0xFFFFFFFF)
        elif x.kind == "vcc":
            w.vcc = np.array([(val >> i) & 1 for i in range(WAVE)], dtype=bool)
        elif x.kind == "exec":
            w.exec = np.array([(val >> i) & 1 for i in range(WAVE)], dtype=bool)
        else:
            raise EmuError(f"64-bit scalar write of {x!r}")

    def wrv(self, w, reg, i, val):
        """write dword i of vector register tuple `reg` under EXEC"""
        self._check_ready(w, reg.sub(i), "write of")
        row = w.v[reg.idx + i]
        row[w.exec] = np.asarray(val, dtype=np.uint32)[w.exec] if np.ndim(val) else np.uint32(val)

    def wrs(self, w, reg, val):
        self._check_ready(w, reg, "write of")
        w.s[reg.idx] = np.uint32(int(val) & 0xFFFFFFFF)

    # ---- packed fp32
    def _pk_src(self, w, x, sel, neg):
        """float64 array of the selected half (sel 0 = low dword, 1 = high) of a 64-bit operand"""
        if isinstance(x, Lit):
            val = np.full(WAVE, x.bits(), dtype=np.uint32)
        elif x.kind == "v":
            if x.n != 2 or x.idx % 2:
                raise EmuError(f"packed operand {x!r} must be an even-aligned VGPR pair")
            self._check_ready(w, x, "read of")
            val = w.v[x.idx + sel]
        elif x.kind == "s":
            if x.n != 2 or x.idx % 2:
                raise EmuError(f"packed operand {x!r} must be an even-aligned SGPR pair")
            self._check_ready(w, x, "read of")
            val = np.full(WAVE, w.s[x.idx + sel], dtype=np.uint32)
        else:
            raise EmuError(f"bad packed operand {x!r}")
        f = val.view(np.float32).astype(np.float64)
        return -f if neg else f

    def _pk(self, w, i, nsrc, fn):
        m = i.mods
        if sum(1 for x in i.src if isinstance(x, R) and x.kind == "s") > 1 and len({x for x in i.src if isinstance(x, R) and x.kind == "s"}) > 1:
            raise EmuError(f"more than one SGPR operand in {render_inst(i).strip()}")
        op_sel = m.get("op_sel") or [0] * nsrc
        op_sel_hi = m.get("op_sel_hi") or [1] * nsrc
        neg_lo = m.get("neg_lo") or [0] * nsrc
        neg_hi = m.get("neg_hi") or [0] * nsrc
        lo = fn(*[self._pk_src(w, i.src[k], op_sel[k], neg_lo[k]) for k in range(nsrc)])
        hi = fn(*[self._pk_src(w, i.src[k], op_sel_hi[k], neg_hi[k]) for k in range(nsrc)])
        if i.dst.kind != "v" or i.dst.n != 2 or i.dst.idx % 2:
            raise EmuError(f"packed destination {i.dst!r} must be an even-aligned VGPR pair")
        self.wrv(w, i.dst, 0, lo.astype(np.float32).view(np.uint32))
        self.wrv(w, i.dst, 1, hi.astype(np.float32).view(np.uint32))

    # ---- memory counters
    def _issue(self, w, kind, dst_regs):
        q = w.vm if kind == "vm" else w.lgkm
        q.append(dst_regs)     # (no cap: nothing may rely on the hardware stalling at a full counter)
        for key in dst_regs:
            if key in w.pending:
                raise EmuError(f"wave {w.wid} pc {w.pc}: load into {key} which is still the target of an outstanding load")
            w.pending[key] = kind

    def _retire(self, w, kind, n):
        q = w.vm if kind == "vm" else w.lgkm
        for _ in range(n):
            for key in q.pop(0):
                w.pending.pop(key, None)

    def _waitcnt(self, w, m):
        if m.get("vmcnt") is not None and len(w.vm) > m["vmcnt"]:
            self._retire(w, "vm", len(w.vm) - m["vmcnt"])
        if m.get("lgkmcnt") is not None and len(w.lgkm) > m["lgkmcnt"]:
            self._retire(w, "lgkm", len(w.lgkm) - m["lgkmcnt"])

    # ---- buffer addressing (raw buffer, stride 0): range check on voffset + soffset + imm against num_records
    def _buf(self, w, vaddr, rsrc, soff, imm, nd):
        self._check_ready(w, rsrc, "read of")
        base = int(w.s[rsrc.idx]) | ((int(w.s[rsrc.idx + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[rsrc.idx + 2])
        so = int(self.rds(w, soff)) if not isinstance(soff, Lit) else int(soff.val)
        voff = self.rd(w, vaddr).astype(np.int64)
        off = voff + so + imm
        return base, nrec, off

    def _buffer_load(self, w, i, nd):
        vaddr, rsrc, soff = i.src
        base, nrec, off = self._buf(w, vaddr, rsrc, soff, i.mods.get("offset", 0), nd)
        out = np.zeros((nd, WAVE), dtype=np.uint32)
        for l in range(WAVE):
            if not w.exec[l]:
                continue
            for d in range(nd):
                o = int(off[l]) + 4 * d
                if 0 <= o and o + 4 <= nrec and o < (1 << 32):
                    arr, k = self.mem.find(base + o)
                    out[d, l] = arr[k]
        # the data lands later: record the values now (memory is not modified behind our back), mark registers pending
        for d in range(nd):
            row = w.v[i.dst.idx + d]
            self._check_ready(w, i.dst.sub(d), "load into")
            row[w.exec] = out[d][w.exec]
        self._issue(w, "vm", [("v", i.dst.idx + d) for d in range(nd)])

    def _buffer_store(self, w, i, nd):
        data, vaddr, rsrc, soff = i.src
        self._check_ready(w, data, "store of")
        base, nrec, off = self._buf(w, vaddr, rsrc, soff, i.mods.get("offset", 0), nd)
        for l in range(WAVE):
            if not w.exec[l]:
                continue
            for d in range(nd):
                o = int(off[l]) + 4 * d
                if 0 <= o and o + 4 <= nrec and o < (1 << 32):
                    arr, k = self.mem.find(base + o)
                    arr[k] = w.v[data.idx + d, l]
        self._issue(w, "vm", [])

    # ---- LDS
    def _lds_addr(self, w, vaddr, imm, nd):
        a = self.rd(w, vaddr).astype(np.int64) + imm
        if (a % (4 if nd < 2 else 8) != 0).any():
            raise EmuError(f"misaligned LDS address in {render_inst(self.prog[w.pc]).strip()}")
        if (a[w.exec] < 0).any() or (a[w.exec] + 4 * nd > self.lds.nbytes).any():
            raise EmuError(f"LDS address out of range in wave {w.wid}: {render_inst(self.prog[w.pc]).strip()}")
        return a // 4

    def _ds_read(self, w, i, nd):
        k = self._lds_addr(w, i.src[0], i.mods.get("offset", 0), nd)
        for d in range(nd):
            idx = (k + d)[w.exec]
            if self.check_lds_races:
                bad = (self.lds_wepoch[idx] == self.epoch) & (self.lds_wwave[idx] != w.wid)
                if bad.any():
                    raise EmuError(f"wave {w.wid} pc {w.pc}: LDS read of data written by another wave in the same barrier "
                                   f"interval (missing s_barrier) -- {render_inst(i).strip()}")
                fresh = self.lds_repoch[idx] != self.epoch          # first read of this dword in this interval
                self.lds_rwave[idx] = np.where(fresh | (self.lds_rwave[idx] == w.wid), w.wid, -2)
                self.lds_repoch[idx] = self.epoch
            vals = np.zeros(WAVE, dtype=np.uint32)
            vals[w.exec] = self.lds[idx]
            self.wrv(w, i.dst, d, vals)
        self._issue(w, "lgkm", [("v", i.dst.idx + d) for d in range(nd)])

    def _ds_write(self, w, i, nd):
        k = self._lds_addr(w, i.src[0], i.mods.get("offset", 0), nd)
        self._check_ready(w, i.src[1], "LDS store of")
        for d in range(nd):
            idx = (k + d)[w.exec]
            if self.check_lds_races:
                # write after another wave's read / write in the same interval
                bad = ((self.lds_repoch[idx] == self.epoch) & (self.lds_rwave[idx] != w.wid)) | \
                      ((self.lds_wepoch[idx] == self.epoch) & (self.lds_wwave[idx] != w.wid))
                if bad.any():
                    raise EmuError(f"wave {w.wid} pc {w.pc}: LDS write over data another wave read or wrote in the same "
                                   f"barrier interval (missing s_barrier) -- {render_inst(i).strip()}")
                self.lds_wepoch[idx] = self.epoch
                self.lds_wwave[idx] = w.wid
            self.lds[idx] = w.v[i.src[1].idx + d][w.exec]
        self._issue(w, "lgkm", [])

    # ---- one instruction
    def step(self, w):
        i = self.prog[w.pc]
        if isinstance(i, Label):
            w.pc += 1
            return
        op, m = i.op, i.mods
        w.n_inst[op] = w.n_inst.get(op, 0) + 1
        nxt = w.pc + 1
        u32 = lambda a: np.asarray(a).astype(np.uint64) & 0xFFFFFFFF  # noqa: E731
        if op == "v_pk_add_f32":
            self._pk(w, i, 2, lambda a, b: a + b)
        elif op == "v_pk_mul_f32":
            self._pk(w, i, 2, lambda a, b: a * b)
        elif op == "v_pk_fma_f32":
            self._pk(w, i, 3, lambda a, b, c: a * b + c)
        elif op == "v_mov_b32":
            self.wrv(w, i.dst, 0, self.rd(w, i.src[0]))
        elif op in ("v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32",
                    "v_mul_u32_u24", "v_mul_lo_u32", "v_subrev_u32"):
            a, b = u32(self.rd(w, i.src[0])), u32(self.rd(w, i.src[1]))
            r = {"v_add_u32": lambda: a + b, "v_sub_u32": lambda: a - b, "v_subrev_u32": lambda: b - a,
                 "v_and_b32": lambda: a & b, "v_or_b32": lambda: a | b, "v_xor_b32": lambda: a ^ b,
                 "v_lshlrev_b32": lambda: b << (a & 31), "v_lshrrev_b32": lambda: b >> (a & 31),
                 "v_mul_u32_u24": lambda: (a & 0xFFFFFF) * (b & 0xFFFFFF), "v_mul_lo_u32": lambda: a * b}[op]()
            self.wrv(w, i.dst, 0, (r & 0xFFFFFFFF).astype(np.uint32))
        elif op == "v_mad_u32_u24":
            a, b, c = (u32(self.rd(w, x)) for x in i.src)
            self.wrv(w, i.dst, 0, (((a & 0xFFFFFF) * (b & 0xFFFFFF) + c) & 0xFFFFFFFF).astype(np.uint32))
        elif op == "v_lshl_add_u32":
            a, b, c = (u32(self.rd(w, x)) for x in i.src)
            self.wrv(w, i.dst, 0, (((a << (b & 31)) + c) & 0xFFFFFFFF).astype(np.uint32))
        elif op.startswith("v_cmp_"):
            a, b = u32(self.rd(w, i.src[0])), u32(self.rd(w, i.src[1]))
            rel = op.split("_")[2]
            r = {"eq": a == b, "ne": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[rel]
            r = r & w.exec
            if i.dst.kind == "vcc":
                w.vcc = r
            else:
                self.wr64s(w, i.dst, _mask64(r))
        elif op == "v_cndmask_b32":
            a, b = self.rd(w, i.src[0]), self.rd(w, i.src[1])
            sel = w.vcc if i.src[2].kind == "vcc" else np.array([(self.rd64s(w, i.src[2]) >> k) & 1 for k in range(WAVE)], dtype=bool)
            self.wrv(w, i.dst, 0, np.where(sel, b, a))
        elif op == "v_readfirstlane_b32":
            lane = int(np.argmax(w.exec)) if w.exec.any() else 0
            self.wrs(w, i.dst, self.rd(w, i.src[0])[lane])
        elif op in ("s_mov_b32", "s_movk_i32"):
            self.wrs(w, i.dst, self.rds(w, i.src[0]))
        elif op == "s_mov_b64":
            self.wr64s(w, i.dst, self.rd64s(w, i.src[0]) if isinstance(i.src[0], R) else int(i.src[0].val) & 0xFFFFFFFFFFFFFFFF)
        elif op in ("s_and_b64", "s_andn2_b64", "s_or_b64"):
            a, b = self.rd64s(w, i.src[0]), self.rd64s(w, i.src[1])
            r = {"s_and_b64": a & b, "s_andn2_b64": a & ~b, "s_or_b64": a | b}[op] & 0xFFFFFFFFFFFFFFFF
            self.wr64s(w, i.dst, r)
            w.scc = int(r != 0)
        elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32", "s_mul_i32", "s_mul_hi_u32", "s_lshl_b32",
                    "s_lshr_b32", "s_and_b32", "s_or_b32", "s_min_u32", "s_max_u32", "s_andn2_b32", "s_ashr_i32", "s_xor_b32",
                    "s_max_i32", "s_min_i32"):
            a, b = int(self.rds(w, i.src[0])), int(self.rds(w, i.src[1]))
            if op == "s_add_u32":
                r = a + b
                w.scc = int(r >> 32 != 0)
            elif op == "s_addc_u32":
                r = a + b + w.scc
                w.scc = int(r >> 32 != 0)
            elif op == "s_sub_u32":
                r = a - b
                w.scc = int(b > a)
            elif op == "s_subb_u32":
                r = a - b - w.scc
                w.scc = int(b + w.scc > a)
            elif op == "s_mul_i32":
                r = a * b
            elif op == "s_mul_hi_u32":
                r = (a * b) >> 32
            elif op == "s_lshl_b32":
                r = a << (b & 31)
                w.scc = int(r & 0xFFFFFFFF != 0)
            elif op == "s_lshr_b32":
                r = a >> (b & 31)
                w.scc = int(r != 0)
            elif op == "s_ashr_i32":
                sa = a - (1 << 32) if a >> 31 else a
                r = sa >> (b & 31)
                w.scc = int(r & 0xFFFFFFFF != 0)
            elif op == "s_and_b32":
                r = a & b
                w.scc = int(r != 0)
            elif op == "s_andn2_b32":
                r = a & ~b
                w.scc = int(r & 0xFFFFFFFF != 0)
            elif op == "s_or_b32":
                r = a | b
                w.scc = int(r != 0)
            elif op == "s_xor_b32":
                r = a ^ b
                w.scc = int(r != 0)
            elif op in ("s_max_i32", "s_min_i32"):
                sa = a - (1 << 32) if a >> 31 else a
                sb = b - (1 << 32) if b >> 31 else b
                r = max(sa, sb) if op == "s_max_i32" else min(sa, sb)
                w.scc = int((sa >= sb) if op == "s_max_i32" else (sa <= sb))
            elif op == "s_min_u32":
                r = min(a, b)
                w.scc = int(a <= b)
            else:
                r = max(a, b)
                w.scc = int(a >= b)
            self.wrs(w, i.dst, r)
        elif op.startswith("s_cmp_"):
            a, b = int(self.rds(w, i.src[0])), int(self.rds(w, i.src[1]))
            rel, ty = op.split("_")[2], op.split("_")[3]
            if ty == "i32":
                a = a - (1 << 32) if a >> 31 else a
                b = b - (1 << 32) if b >> 31 else b
            w.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[rel])
        elif op == "s_cselect_b32":
            self.wrs(w, i.dst, self.rds(w, i.src[0]) if w.scc else self.rds(w, i.src[1]))
        elif op == "s_branch":
            nxt = self.labels[m["target"]]
        elif op == "s_cbranch_scc0":
            if not w.scc:
                nxt = self.labels[m["target"]]
        elif op == "s_cbranch_scc1":
            if w.scc:
                nxt = self.labels[m["target"]]
        elif op == "s_cbranch_execz":
            if not w.exec.any():
                nxt = self.labels[m["target"]]
        elif op.startswith("s_load_dword"):
            nd = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[op]
            base = self.rd64s(w, i.src[0]) + m.get("offset", 0)
            arr, k = self.mem.find(base)
            for d in range(nd):
                self._check_ready(w, i.dst.sub(d), "load into")
                w.s[i.dst.idx + d] = arr[k + d]
            self._issue(w, "lgkm", [("s", i.dst.idx + d) for d in range(nd)])
        elif op == "s_waitcnt":
            self._waitcnt(w, m)
        elif op in ("s_nop", "s_setprio", "s_sleep", ";touch"):
            if op == ";touch":
                for x in i.src:
                    self._check_ready(w, x, "use of")
        elif op == "s_barrier":
            if w.lgkm and any(len(x) == 0 for x in w.lgkm):
                raise EmuError(f"wave {w.wid} pc {w.pc}: s_barrier with LDS writes still outstanding (needs s_waitcnt lgkmcnt(0))")
            w.at_barrier = True
        elif op == "s_endpgm":
            w.done = True
        elif op == "v_max3_f32":
            ab = m.get("abs") or [0, 0, 0]
            vals = []
            for k, x in enumerate(i.src):
                f = self.rd(w, x).astype(np.uint32).view(np.float32)
                vals.append(np.abs(f) if ab[k] else f)
            # IEEE maximum that prefers numbers over NaN (fmax), as the hardware's max3
            r = np.fmax(np.fmax(vals[0], vals[1]), vals[2]).astype(np.float32)
            self.wrv(w, i.dst, 0, r.view(np.uint32))
        elif op == "buffer_atomic_umax":
            data, rsrc, soff = i.src
            self._check_ready(w, data, "atomic of")
            self._check_ready(w, rsrc, "read of")
            base = int(w.s[rsrc.idx]) | ((int(w.s[rsrc.idx + 1]) & 0xFFFF) << 32)
            nrec = int(w.s[rsrc.idx + 2])
            so = int(self.rds(w, soff)) if not isinstance(soff, Lit) else int(soff.val)
            if 0 <= so and so + 4 <= nrec:
                arr, k = self.mem.find(base + so)
                for l in range(WAVE):
                    if w.exec[l]:
                        arr[k] = max(np.uint32(arr[k]), np.uint32(w.v[data.idx, l]))
            self._issue(w, "vm", [])
        elif op.startswith("buffer_load_dword"):
            self._buffer_load(w, i, {"buffer_load_dword": 1, "buffer_load_dwordx2": 2, "buffer_load_dwordx4": 4}[op])
        elif op.startswith("buffer_store_dword"):
            self._buffer_store(w, i, {"buffer_store_dword": 1, "buffer_store_dwordx2": 2, "buffer_store_dwordx4": 4}[op])
        elif op in ("ds_read_b32", "ds_read_b64", "ds_read_b128"):
            self._ds_read(w, i, {"ds_read_b32": 1, "ds_read_b64": 2, "ds_read_b128": 4}[op])
        elif op in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
            self._ds_write(w, i, {"ds_write_b32": 1, "ds_write_b64": 2, "ds_write_b128": 4}[op])
        elif op == "s_memrealtime":
            self._check_ready(w, i.dst, "write of")
            w.s[i.dst.idx] = 0
            w.s[i.dst.idx + 1] = 0
            self._issue(w, "lgkm", [("s", i.dst.idx), ("s", i.dst.idx + 1)])
        else:
            raise EmuError(f"instruction not modelled: {op}")
        w.pc = nxt

    def run(self, max_steps=50_000_000):
        """Run all waves to completion.  Between barriers the waves run one after the other in a random order."""
        steps = 0
        while not all(w.done for w in self.waves):
            order = list(self.rng.permutation(len(self.waves)))
            for k in order:
                w = self.waves[k]
                while not w.done and not w.at_barrier:
                    self.step(w)
                    steps += 1
                    if steps > max_steps:
                        raise EmuError("step limit exceeded (endless loop?)")
            live = [w for w in self.waves if not w.done]
            if live and all(w.at_barrier for w in live):
                if len(live) != len(self.waves):
                    raise EmuError("s_barrier reached by some waves after others ended")
                for w in live:
                    w.at_barrier = False
                    w.pc += 0
                self.epoch += 1
        return steps
