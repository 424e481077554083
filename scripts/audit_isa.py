"""Static hazard audit of the hand-scheduled kernels, on the gfx950 assembly hipcc emits for them.

Why: the hot kernels issue memory instructions from inline asm and wait for them with hand-counted `s_waitcnt`s.  The compiler
knows nothing about what an asm statement has in flight: it neither inserts waits for it nor keeps the registers involved
alive, and its hazard recogniser does not look inside an asm block.  Two failures of round 5 were of exactly that class (an
ablation build without its `lgkmcnt` wait: a late LDS return landed in a register re-used as an LDS-DMA address -> aperture
violation; an asm `global_store_dwordx4` whose data registers the compiler re-wrote inside the store-data hazard window -> a
parity failure).  A passing numerical test cannot show that such code is right by construction; this audit does, per
instantiation, on every path of the control-flow graph:

  (a) asm global / buffer loads to registers: from issue until a covering `s_waitcnt vmcnt(N)` (N <= the number of younger
      vector-memory operations issued on that path: loads, stores, atomics and LDS-DMA count together, in order) no
      instruction may read or write the destination registers;
  (b) asm `ds_read_*` / `ds_bpermute` ...: the same with `lgkmcnt`; only younger LDS operations count towards N (scalar loads
      share the counter but return out of order, so they can only make a wait stricter than counted, never weaker);
  (c) asm stores of more than 8 bytes per lane (`global_store_dwordx3/x4`, `buffer_store_dwordx3/x4`): no instruction may
      write a data register in the 2 wait states behind the store (the gfx940+ store-data hazard, which the compiler pads
      only for stores it emitted itself);
  (d) asm MFMAs: the result registers may not be read or written by anything but an MFMA accumulating into exactly the same
      registers for `passes + 2` wait states (16 passes assumed for 32x32 shapes, 8 for 16x16 -- an upper bound on gfx950),
      and no MFMA may take them as A / B operand inside that window;
  (e) register spills / scratch per kernel against an allow-list (`ALLOW_SPILLS`): a spill reload in a tile loop costs a
      `vmcnt(0)`, i.e. drains the LDS-DMA ring.

    python scripts/audit_isa.py [file.hip ...]      -> exit code 0 / 1, one line per kernel (only problems with -q)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "moma_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the three hand-scheduled files (inline asm: checks a-d) and, for the spill budget (e) alone, the other kernel files of the KD path
# (K3 / K4, the generic GEMM and the row kernels of the staged paths; the backbone helpers bn / dwconv / se can be named on the
# command line: 600 more instantiations, all clean, a minute more)
FILES = ["infonce_fused.hip", "infonce_f32.hip", "k1_fast.hip", "gemm.hip", "queue.hip", "rowops.hip"]
SHOW = int(os.environ.get("AUDIT_SHOW", "3"))            # problems printed per kind and kernel

# (e) kernels that may spill: demangled-name regex -> (max VGPR spills, max scratch bytes, max SGPR spills -- those go to the
# lanes of a reserved VGPR: no memory traffic, no wait).  Everything else: 0 / 0 / 0.
ALLOW_SPILLS = {
    r"^infonce_combine_multi_kernel$": (0, 256, 0),          # a by-value MultiArgs table indexed at run time lives in scratch
    r"^infonce_slab_kernel<512, 2>$": (1, 8, 0),             # one dword stored in front of the tile loop, reloaded behind it
    r"^k1_core_bwd_kernel<false>$": (1, 8, 6),               # (the same; head widths below 128: ViT-S, feat_dim 256)
    r"^infonce_f32_flash_kernel<2, 2, false>$": (0, 0, 2),   # forward-only exact-fp32 passes: scalar spills only
    r"^infonce_f32_flash_kernel<4, 4, false>$": (0, 0, 26),
    r"^gemm_kernel<1, (unsigned short|float), false, false, 32>$": (0, 0, 2),   # generic GEMM (staged paths), unaligned operands
}

VM_RE = re.compile(r"^(global_|buffer_|scratch_|flat_)")
DS_RE = re.compile(r"^ds_")
BRANCH_RE = re.compile(r"^s_c?branch")
REG_RE = re.compile(r"(?<![\w.])([va])(\d+)\b|(?<![\w.])([va])\[(\d+):(\d+)\]")


def regs_of(text: str) -> frozenset:
    """vector registers mentioned: v_n -> n, a_n -> 1000 + n"""
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add((1000 if m.group(1) == "a" else 0) + int(m.group(2)))
        else:
            base = 1000 if m.group(3) == "a" else 0
            out.update(range(base + int(m.group(4)), base + int(m.group(5)) + 1))
    return frozenset(out)


class Ins:
    __slots__ = ("op", "args", "asm", "regs", "line", "text", "wr")

    def __init__(self, op, args, asm, line, text):
        self.op, self.args, self.asm, self.line, self.text = op, args, asm, line, text
        self.regs = regs_of(args)
        self.wr = written(op, args)


def written(op: str, args: str) -> frozenset:
    """vector registers an instruction writes (first operand; both for the swaps; none for stores / compares / waits)"""
    if not args or op.startswith(("s_", "global_store", "buffer_store", "scratch_store", "flat_store", "ds_write", "ds_append",
                                  "v_cmp", "global_load_lds", "buffer_wbl2", "buffer_inv")):
        return frozenset()
    if (op.startswith(("global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min", "ds_or", "ds_and")) and "rtn" not in op
            and " sc0" not in args and "glc" not in args):
        return frozenset()
    if " lds" in args and op.startswith(("buffer_load", "global_load")):
        return frozenset()
    parts = split_operands(args)
    w = set(regs_of(parts[0]))
    if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) and len(parts) > 1:
        w |= regs_of(parts[1])
    return frozenset(w)


def split_operands(args: str):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def parse(asm_text: str):
    """-> {mangled name: (instructions, labels {name: index}, meta)}"""
    funcs, cur, labels, name, inasm = {}, None, None, None, False
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        if cur is None:
            m = re.match(r"^(_Z\w+):", raw)
            if m:
                name, cur, labels, inasm = m.group(1), [], {}, False
            continue
        if raw.startswith(".Lfunc_end"):
            funcs[name] = (cur, labels)
            cur = None
            continue
        m = re.match(r"^(\.L\w+):", raw)
        if m:
            labels[m.group(1)] = len(cur)
            continue
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        op, _, args = s.partition(" ")
        cur.append(Ins(op, args.strip(), inasm, ln, s))
    return funcs


def metadata(asm_text: str):
    """.amdgpu_metadata per kernel: name -> dict(vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, vgpr_count, agpr_count)"""
    res, cur = {}, {}
    for line in asm_text.split("\n"):
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "agpr_count" and cur.get("name"):          # first key of the next kernel's record
            res[cur["name"]] = cur
            cur = {}
        if k in ("name", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "agpr_count"):
            cur[k] = v if k == "name" else int(v)
    if cur.get("name"):
        res[cur["name"]] = cur
    return res


def successors(ins, labels, i):
    op = ins[i].op
    if op in ("s_endpgm", "s_setpc_b64", "s_trap"):
        return []
    if op == "s_branch":
        return [labels[ins[i].args]]
    nxt = [i + 1] if i + 1 < len(ins) else []
    if BRANCH_RE.match(op):
        nxt.append(labels[ins[i].args])
    return nxt


def waitcnt(ins_i, which):
    """the count an s_waitcnt leaves for counter `which` ('vmcnt' / 'lgkmcnt'), or None"""
    if ins_i.op != "s_waitcnt":
        return None
    m = re.search(which + r"\((\d+)\)", ins_i.args)
    if m:
        return int(m.group(1))
    if re.fullmatch(r"(0x)?[0-9a-fA-F]+", ins_i.args):      # raw immediate: vmcnt = bits 3:0 + 15:14, lgkmcnt = bits 11:8
        v = int(ins_i.args, 0)
        return ((v & 0xF) | ((v >> 14) & 3) << 4) if which == "vmcnt" else (v >> 8) & 0xF
    return None


SREG_RE = re.compile(r"(?<![\w.])s(\d+)\b|(?<![\w.])s\[(\d+):(\d+)\]|\b(vcc|exec)(?:_lo|_hi)?\b")
VCC, EXEC, SCC = frozenset((106, 107)), frozenset((126, 127)), frozenset((128,))
# scalar instructions that leave SCC alone (everything else on the scalar unit is taken to overwrite it)
SCC_KEEPERS = ("s_mov_", "s_movk_", "s_cselect_", "s_cmov_", "s_mul_i32", "s_mul_hi_", "s_load_", "s_buffer_load_", "s_waitcnt", "s_nop",
               "s_barrier", "s_branch", "s_cbranch_", "s_setprio", "s_sleep", "s_sext_", "s_brev_", "s_ff1_", "s_flbit_", "s_getreg_",
               "s_setreg_", "s_pack_", "s_memtime", "s_memrealtime", "s_sendmsg", "s_endpgm", "s_bfm_", "s_getpc_", "s_setpc_",
               "s_dcache_", "s_icache_", "s_ttrace", "s_sethalt", "s_set_gpr_idx", "s_movrel", "s_bitset", "s_store_", "s_atc_")
S_NO_DEST = ("s_cmp_", "s_cmpk_", "s_bitcmp", "s_cbranch_", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep",
             "s_sendmsg", "s_store_", "s_setreg_", "s_endpgm", "s_dcache_", "s_icache_", "s_ttrace", "s_sethalt", "s_setpc_", "s_trap")


def sregs_of(text: str) -> frozenset:
    out = set()
    for m in SREG_RE.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        elif m.group(2):
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.update(VCC if m.group(4) == "vcc" else EXEC)
    return frozenset(out)


def scalar_writes(x) -> frozenset:
    """scalar state an instruction may overwrite: SGPRs, vcc (106, 107), exec (126, 127), SCC (128)"""
    o = split_operands(x.args) if x.args else []
    w = set()
    if x.op.startswith("s_"):
        if not x.op.startswith(S_NO_DEST) and o:
            w |= sregs_of(o[0])
        if "saveexec" in x.op or x.op.startswith(("s_wqm", "s_wrexec")):
            w |= EXEC
        if not x.op.startswith(SCC_KEEPERS):
            w |= SCC
    elif x.op.startswith("v_"):
        for t in o[:2]:                       # sdst of compares / readlane / carry-outs / v_mad_u64 / v_div_scale
            w |= sregs_of(t)
        if x.op.startswith("v_cmpx"):
            w |= EXEC
        if x.op.endswith("_e32") and x.op.startswith(("v_cmp", "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_subrev_co", "v_div_fmas")):
            w |= VCC
    return frozenset(w)


CMP_RE = re.compile(r"s_cmp_(lt|le|gt|ge|eq|lg)_(i32|u32) s(\d+), (-?(?:0x)?[0-9a-fA-F]+)$")


def parse_cmp_imm(text):
    """`s_cmp_<rel>_<i32|u32> sN, imm` -> (rel, N, unsigned, imm)"""
    m = CMP_RE.match(text)
    if not m:
        return None
    return m.group(1), int(m.group(3)), m.group(2) == "u32", int(m.group(4), 0)


def cmp_decided(rel, unsigned, imm, lo, hi):
    """True / False when every value of [lo, hi] gives the compare the same outcome, else None"""
    if unsigned and (lo < 0 or imm < 0):
        return None
    if rel in ("lt", "ge"):
        r = True if hi < imm else False if lo >= imm else None
        return r if rel == "lt" or r is None else not r
    if rel in ("le", "gt"):
        r = True if hi <= imm else False if lo > imm else None
        return r if rel == "le" or r is None else not r
    r = True if lo == hi == imm else False if (imm < lo or imm > hi) else None
    return r if rel == "eq" or r is None else not r


class Facts:
    """What a path knows about the scalar conditions its branches test -- so that the walk does not follow paths the program
    rules out.  Only true facts: (1) a 64-bit scalar pair set to the constant 0 / -1 (the structuriser's flags); (2) the outcome
    of a condition already branched on, keyed by the TEXT of the scalar instruction that produced it while none of its source
    registers has been overwritten (the same comparison of the same values gives the same answer), or by the LINE of any other
    producer while vcc / SCC still holds that very result.  EXEC is taken as non-zero (code with EXEC = 0 is branched around)."""
    __slots__ = ("consts", "src", "known", "ranges", "derived", "bools")

    def __init__(self, consts=frozenset(), src=frozenset(), known=frozenset(), ranges=frozenset(), derived=frozenset(),
                 bools=frozenset()):
        self.bools = bools        # {regs}: scalar pairs that hold 0 or -1 (selected, or combined from such): branching on one
        #                            of them against EXEC tells its VALUE, which outlives a change of EXEC
        self.derived = derived    # {(regs, key, polarity, deps)}: pair = (outcome of `key` == polarity) ? -1 : 0  (s_cselect_b64)
        self.consts = consts      # {(regs, 0 | -1)}
        self.src = src            # {(VCC | SCC, key, deps)}: who produced the current value of vcc / SCC
        self.known = known        # {(key, outcome, deps)}: outcome = the produced value is non-zero
        self.ranges = ranges      # {(sgpr, lo, hi)}: signed interval learned from compares with immediates (hipcc refines a
        #                            value step by step -- `< 1`, `== 1`, `<u 3` ... -- and tests it differently later)

    def key(self):
        return (self.consts, self.src, self.known, self.ranges, self.derived, self.bools)

    def step(self, x):
        if x.op in ("s_waitcnt", "s_nop", "s_barrier") or x.op.startswith("s_cbranch") or x.op == "s_branch":
            return self
        w = scalar_writes(x)
        if not w and not x.op.startswith("s_"):
            return self
        consts = frozenset(c for c in self.consts if not (c[0] & w))
        src = frozenset(t for t in self.src if not (t[0] & w))
        known = frozenset(k for k in self.known if not (k[2] & w))
        ranges = frozenset(r for r in self.ranges if r[0] not in w)
        derived = frozenset(d for d in self.derived if not ((d[0] | d[3]) & w))
        bools = frozenset(b for b in self.bools if not (b & w))
        o = split_operands(x.args) if x.args else []
        cmap = dict(self.consts)

        def bval(t):                                  # ('c', 0 | -1) / ('b',) for a 0-or-minus-one operand, else None
            if t in ("0", "-1"):
                return ("c", int(t))
            r = sregs_of(t)
            if r in cmap:
                return ("c", cmap[r])
            return ("b",) if r in self.bools else None
        if x.op == "s_mov_b64" and len(o) == 2 and bval(o[1]) and sregs_of(o[0]) not in (VCC, EXEC):
            bools |= {sregs_of(o[0])}
            if bval(o[1])[0] == "c":
                consts |= {(sregs_of(o[0]), bval(o[1])[1])}
        elif (x.op in ("s_or_b64", "s_and_b64", "s_xor_b64", "s_andn2_b64", "s_orn2_b64", "s_not_b64") and len(o) >= 2
              and all(bval(t) for t in o[1:]) and sregs_of(o[0]) not in (VCC, EXEC)):
            bools |= {sregs_of(o[0])}
            v = [bval(t)[1] if bval(t)[0] == "c" else None for t in o[1:]]
            res = None
            if x.op == "s_not_b64" and v[0] is not None:
                res = ~v[0]
            elif x.op == "s_or_b64":
                res = -1 if -1 in v else 0 if v == [0, 0] else None
            elif x.op == "s_and_b64":
                res = 0 if 0 in v else -1 if v == [-1, -1] else None
            elif x.op == "s_andn2_b64":
                res = 0 if (v[0] == 0 or v[1] == -1) else -1 if (v[0] == -1 and v[1] == 0) else None
            elif None not in v:
                res = (v[0] ^ v[1]) if x.op == "s_xor_b64" else (v[0] | ~v[1])
            if res is not None:
                consts |= {(sregs_of(o[0]), res)}
        elif x.op == "s_cselect_b64" and len(o) == 3 and (o[1], o[2]) in (("-1", "0"), ("0", "-1")):
            val, sc = self._value(SCC)                    # (SCC as it stands BEFORE this instruction, which does not write it)
            pol = o[1] == "-1"
            bools |= {sregs_of(o[0])}
            if val is not None:
                consts |= {(sregs_of(o[0]), -1 if val == pol else 0)}
            elif sc is not None and isinstance(sc[1], str) and not (sc[2] & sregs_of(o[0])):
                derived |= {(sregs_of(o[0]), sc[1], pol, sc[2])}
        # who produced vcc / SCC: a text key when every source is scalar (SGPR / exec / immediate), else the line
        for reg in (VCC, SCC):
            if reg & w and (reg is SCC or (o and sregs_of(o[0]) == VCC)):
                srcs = sregs_of(",".join(o[1:] if reg is VCC else o[(0 if x.op.startswith(("s_cmp", "s_bitcmp")) else 1):]))
                scalar_only = x.op.startswith("s_") and not regs_of(x.args)
                if scalar_only and not (srcs & w):
                    src |= {(reg, x.text, srcs)}
                else:
                    src |= {(reg, ("line", x.line), reg)}
        return Facts(consts, src, known, ranges, derived, bools)

    def _value(self, reg):
        """True / False / None: is the current vcc (as a 64-bit value) or SCC non-zero"""
        s = next((t for t in self.src if t[0] == reg), None)
        if s is None:
            return None, None
        key, deps = s[1], s[2]
        if isinstance(key, str):
            o = split_operands(key.split(" ", 1)[1])
            op = key.split(" ", 1)[0]
            if op in ("s_and_b64", "s_andn2_b64") and len(o) == 3 and o[1] == "exec":
                c = dict(self.consts).get(sregs_of(o[2]))
                if c is not None:
                    return ((c == -1) if op == "s_and_b64" else (c == 0)), s
                d = next((d for d in self.derived if d[0] == sregs_of(o[2])), None)
                if d is not None:
                    v = self._outcome(d[1])
                    if v is not None:
                        return ((v == d[2]) == (op == "s_and_b64")), s
        return self._outcome(key), s

    def _outcome(self, key):
        for k in self.known:
            if k[0] == key:
                return k[1]
        c = parse_cmp_imm(key) if isinstance(key, str) else None
        if c:
            lo, hi = self._range(c[1])
            return cmp_decided(c[0], c[2], c[3], lo, hi)
        return None

    def _range(self, reg):
        for r in self.ranges:
            if r[0] == reg:
                return r[1], r[2]
        return -(1 << 31), (1 << 31) - 1

    def _refined(self, key, outcome):
        """ranges after learning that compare `key` came out `outcome`"""
        c = parse_cmp_imm(key) if isinstance(key, str) else None
        if not c:
            return self.ranges
        rel, reg, unsigned, imm = c
        lo, hi = self._range(reg)
        if unsigned and lo < 0:
            if outcome and rel in ("lt", "le", "eq"):         # below a small unsigned bound: non-negative as a signed value too
                lo = 0
            else:
                return self.ranges
        if not outcome:
            rel = {"lt": "ge", "ge": "lt", "le": "gt", "gt": "le", "eq": "lg", "lg": "eq"}[rel]
        if rel == "lt":
            hi = min(hi, imm - 1)
        elif rel == "le":
            hi = min(hi, imm)
        elif rel == "gt":
            lo = max(lo, imm + 1)
        elif rel == "ge":
            lo = max(lo, imm)
        elif rel == "eq":
            lo, hi = max(lo, imm), min(hi, imm)
        elif rel == "lg":
            if imm == lo:
                lo += 1
            elif imm == hi:
                hi -= 1
        return frozenset(r for r in self.ranges if r[0] != reg) | {(reg, lo, hi)}

    def branch(self, x, taken_target, fall):
        """[(successor, facts)] of a conditional branch"""
        reg = VCC if "vcc" in x.op else SCC if "scc" in x.op else None
        if reg is None:
            return [(t, self) for t in (fall, taken_target) if t is not None]
        want_nz = x.op.endswith(("vccnz", "scc1"))
        val, s = self._value(reg)
        out = []
        for nz in (True, False):
            if val is not None and val != nz:
                continue
            tgt = taken_target if nz == want_nz else fall
            if tgt is None:
                continue
            f = self
            if val is None and s is not None:
                key, res, deps = s[1], nz, s[2]
                consts = self.consts
                if isinstance(key, str) and key.startswith(("s_and_b64 vcc, exec, ", "s_andn2_b64 vcc, exec, ")):
                    pair, is_and = sregs_of(key.split(", ")[2]), key.startswith("s_and_b64")
                    if pair in self.bools:                     # the VALUE of a 0-or-minus-one pair
                        consts = consts | {(pair, -1 if nz == is_and else 0)}
                    d = next((d for d in self.derived if d[0] == pair), None)
                    if d is not None:                          # ... and the compare the pair was selected from
                        key, res, deps = d[1], (nz == is_and) == d[2], d[3]
                f = Facts(consts, self.src, self.known | {(s[1], nz, s[2]), (key, res, deps)}, self._refined(key, res), self.derived,
                          self.bools)
            out.append((tgt, f))
        return out


def edges(ins, labels, i, facts):
    op = ins[i].op
    if op in ("s_endpgm", "s_setpc_b64", "s_trap"):
        return []
    if op == "s_branch":
        return [(labels[ins[i].args], facts)]
    fall = i + 1 if i + 1 < len(ins) else None
    if BRANCH_RE.match(op):
        return facts.branch(ins[i], labels[ins[i].args], fall)
    return [(fall, facts)] if fall is not None else []


TRACE = os.environ.get("AUDIT_TRACE") == "1"          # print the path (branches, waits, counted operations) to each hit


def window_until_wait(ins, labels, i, dest, which, counts):
    """(a)/(b): first instruction on any path from i that touches `dest` before a covering wait, else None.  A YOUNGER memory
    operation of the same in-order class that only overwrites destination registers retires them from the window: results
    return in issue order, so the older data lands first (and what the younger load's own registers need is its own window)."""
    seen, stack, parent = set(), [(s, 0, dest, f, None) for s, f in edges(ins, labels, i, Facts())], {}
    while stack:
        j, n, pend, facts, par = stack.pop()
        k = (j, n, pend, facts.key())
        if k in seen:
            continue
        seen.add(k)
        if TRACE:
            parent[k] = par
        x = ins[j]
        w = waitcnt(x, which)
        if w is not None and w <= n:
            continue
        if x.regs & pend:
            if counts(x) and not ((x.regs - x.wr) & pend):
                pend = pend - x.wr
                if not pend:
                    continue
            else:
                if TRACE:
                    path, q = [], k
                    while q is not None:
                        y = ins[q[0]]
                        if BRANCH_RE.match(y.op) or y.op == "s_waitcnt" or counts(y) or q == k:
                            path.append(f"          n={q[1]:2d} line {y.line}: {y.text}")
                        q = parent[q]
                    print(f"      path from line {ins[i].line} `{ins[i].text}`:\n" + "\n".join(reversed(path)))
                return x
        if counts(x):
            n = min(n + 1, 64)
        facts = facts.step(x)
        for s, f in edges(ins, labels, j, facts):
            stack.append((s, n, pend, f, k))
    return None


def wait_states(x):
    if x.op == "s_nop":
        return int(x.args, 0) + 1
    return 1


def window_states(ins, labels, i, limit, bad):
    """first instruction within `limit` wait states behind i (on any path) for which bad(x, n) holds, else None"""
    seen, stack = set(), [(s, 0, f) for s, f in edges(ins, labels, i, Facts())]
    while stack:
        j, n, facts = stack.pop()
        k = (j, n, facts.key())
        if n >= limit or k in seen:
            continue
        seen.add(k)
        x = ins[j]
        if bad(x, n):
            return x
        n += wait_states(x)
        facts = facts.step(x)
        for s, f in edges(ins, labels, j, facts):
            stack.append((s, n, f))
    return None


def is_vm(x):
    return bool(VM_RE.match(x.op))


def is_ds(x):
    return bool(DS_RE.match(x.op))


def mfma_passes(op):
    return 16 if "32x32" in op else 8 if "16x16" in op else 4


def audit_function(ins, labels):
    problems = []
    counts = {"asm_vm_loads": 0, "asm_ds_reads": 0, "asm_wide_stores": 0, "asm_mfma": 0}
    for i, x in enumerate(ins):
        if not x.asm:
            continue
        if is_vm(x) and x.wr:                                               # (a)
            counts["asm_vm_loads"] += 1
            if x.op.startswith("flat_"):
                problems.append(f"line {x.line}: flat load in asm (counts in vmcnt AND lgkmcnt, out of order): {x.text}")
                continue
            hit = window_until_wait(ins, labels, i, x.wr, "vmcnt", is_vm)
            if hit:
                problems.append(f"line {x.line}: `{x.text}` -- destination touched before a covering vmcnt wait by line {hit.line}: `{hit.text}`")
        elif is_ds(x) and x.wr:                                             # (b)
            counts["asm_ds_reads"] += 1
            hit = window_until_wait(ins, labels, i, x.wr, "lgkmcnt", is_ds)
            if hit:
                problems.append(f"line {x.line}: `{x.text}` -- destination touched before a covering lgkmcnt wait by line {hit.line}: `{hit.text}`")
        elif re.match(r"(global|buffer|scratch)_store_dwordx[34]", x.op):   # (c)
            counts["asm_wide_stores"] += 1
            ops = split_operands(x.args)
            data = regs_of(ops[1] if x.op.startswith("global") or x.op.startswith("scratch") else ops[0])
            hit = window_states(ins, labels, i, 2, lambda y, n: bool(y.wr & data))
            if hit:
                problems.append(f"line {x.line}: `{x.text}` -- data register written inside the store-data hazard window by line {hit.line}: `{hit.text}`")
        elif x.op.startswith("v_mfma") or x.op.startswith("v_smfmac"):      # (d)
            counts["asm_mfma"] += 1
            dst = x.wr
            exact = split_operands(x.args)[0]

            npass = mfma_passes(x.op)

            def bad(y, n, dst=dst, exact=exact, npass=npass):
                if not (y.regs & dst):
                    return False
                if y.op.startswith("v_mfma") or y.op.startswith("v_smfmac"):
                    o = split_operands(y.args)
                    if regs_of(o[1]) & dst or regs_of(o[2]) & dst:
                        return True                  # the result as A / B operand of a following MFMA: passes + 5
                    # as accumulator input: only the very same register tuple (the dependent chain the hardware handles)
                    return n < npass + 2 and len(o) > 3 and bool(regs_of(o[3]) & dst) and o[3] != exact
                return n < npass + 2                 # VALU / memory / export read or write: passes + 2
            hit = window_states(ins, labels, i, npass + 5, bad)
            if hit:
                problems.append(f"line {x.line}: `{x.text}` -- result touched inside the MFMA hazard window by line {hit.line}: `{hit.text}`")
    return problems, counts


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.split("\n") if r.returncode == 0 else names
    short = []
    for d in out[:len(names)]:
        d = re.sub(r"^void ", "", d)
        d = d.replace("moma::(anonymous namespace)::", "").replace("moma::", "")
        m = re.match(r"^([\w:]+(?:<[^(]*>)?)\(", d)
        short.append(m.group(1) if m else d[:60])
    return dict(zip(names, short))


def compile_s(src: str, tmp: str) -> str:
    base = os.path.basename(src).replace(".hip", "")
    d = os.path.join(tmp, base)
    os.makedirs(d, exist_ok=True)
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-save-temps", "-c", src, "-o", base + ".o"],
                   cwd=d, check=True, capture_output=True)
    return open(os.path.join(d, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def audit_text(asm_text: str, quiet=False, label=""):
    funcs, meta = parse(asm_text), metadata(asm_text)
    names = demangle(list(funcs))
    ok, report = True, []
    for name, (ins, labels) in funcs.items():
        short = names[name]
        problems, counts = audit_function(ins, labels)
        md = meta.get(name, {})
        sp, ssp, scratch = md.get("vgpr_spill_count", 0), md.get("sgpr_spill_count", 0), md.get("private_segment_fixed_size", 0)
        lim = next((v for k, v in ALLOW_SPILLS.items() if re.search(k, short)), (0, 0, 0))
        if sp > lim[0] or scratch > lim[1] or ssp > lim[2]:
            problems.append(f"{sp} VGPR spills, {scratch} B of scratch, {ssp} SGPR spills (allowed {lim[0]} / {lim[1]} / {lim[2]})")
        ok = ok and not problems
        line = (f"{label}{short:58s} regs {md.get('vgpr_count', 0):3d}+{md.get('agpr_count', 0):3d} spills {sp:3d}v {ssp:2d}s scratch {scratch:4d}  "
                f"asm: {counts['asm_vm_loads']:4d} loads {counts['asm_ds_reads']:5d} lds reads {counts['asm_wide_stores']:3d} stores "
                f"{counts['asm_mfma']:5d} mfma  {'OK' if not problems else 'PROBLEMS'}")
        if problems or not quiet:
            report.append(line)
            kinds = {}
            for p in problems:
                kinds.setdefault(p.split(" -- ")[1].split(" by line")[0] if " -- " in p else "spills", []).append(p)
            for k, ps in kinds.items():
                report += ["      " + p for p in ps[:SHOW]]
                if len(ps) > SHOW:
                    report.append(f"      ... {len(ps) - SHOW} more of this kind")
    return ok, report


def audit(files=None, quiet=False):
    files = files or [os.path.join(CSRC, f) for f in FILES]
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(max_workers=3) as ex:
            texts = list(ex.map(lambda f: compile_s(f, tmp), files))
    ok = True
    for f, t in zip(files, texts):
        o, rep = audit_text(t, quiet, label=os.path.basename(f).replace(".hip", "") + ": ")
        if rep:
            print("\n".join(rep))
        ok = ok and o
    if quiet and ok:
        print("audit_isa: every kernel clean")
    return ok


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-q"]
    sys.exit(0 if audit([os.path.abspath(a) for a in args] or None, quiet="-q" in sys.argv) else 1)
