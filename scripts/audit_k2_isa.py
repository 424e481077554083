"""Audit of the compiled K2 kernels (csrc/infonce_fused.hip): things a passing numerical test cannot see.

  1. no register spills in the one-pass kernels (a spill reload inside the tile loop costs a vmcnt(0): the LDS-DMA ring drains);
  2. the Q fragments are loaded by inline asm and counted by hand: between the first asm load and the barrier after the
     hand-placed wait NO instruction may read or write a Q register (the compiler does not know the loads are in flight);
  3. between the first Q load and the first score MFMA there is no compiler-inserted `s_waitcnt vmcnt(0)` (the prologue then
     waits for the whole ring instead of Q + the first tile).

    python scripts/audit_k2_isa.py          -> exit code 0 / 1, one line per kernel
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "moma_amd", "csrc", "infonce_fused.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_s(tmp):
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-save-temps", "-c", SRC, "-o", "k2.o"],
                   cwd=tmp, check=True, capture_output=True)
    return open(os.path.join(tmp, "infonce_fused-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def kernels(asm):
    """name -> list of instruction lines"""
    out, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN4moma\S*infonce_(flash|slab)_kernel\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if line.startswith(".Lfunc_end"):
                out[name] = cur
                cur = None
    return out


def spills(asm):
    res, name = {}, None
    for line in asm.split("\n"):
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\.vgpr_spill_count:\s+(\d+)", line)
        if m and name:
            res[name] = int(m.group(1))
    return res


def regs_of(text):
    rs = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", text.split(";")[0]):
        if m.group(1):
            rs.add(int(m.group(1)))
        else:
            rs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return rs


def audit():
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_s(tmp)
    sp = spills(asm)
    ok = True
    for name, lines in sorted(kernels(asm).items()):
        short = re.sub(r"^_ZN4moma12_GLOBAL__N_1\d+", "", name)[:34]
        problems = []
        one_pass_dq = "infonce_flash_kernel" in name and "ELb1E" in name
        if one_pass_dq and sp.get(name, 0) != 0:
            problems.append(f"{sp[name]} VGPR spills")
        loads = [i for i, l in enumerate(lines) if re.search(r"global_load_dwordx4 v\[", l) and " lds" not in l]
        if "infonce_slab_kernel" in name:
            loads = []          # (the slab passes also load their score scratch with ordinary loads: not audited here)
        if loads:
            qregs = set()
            for i in loads:
                qregs |= regs_of(lines[i].split(",")[0])
            first = loads[0]
            mfma = next((i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l and i > loads[-1]
                         and regs_of(l) & qregs), None)
            if mfma is None:
                print(f"{short:36s} no score MFMA reading a Q register found"); ok = False
                continue
            bars = [i for i, l in enumerate(lines) if "s_barrier" in l and first < i < mfma]
            if not bars:
                problems.append("no barrier between the Q loads and the first score MFMA")
            else:
                for i in range(first, bars[-1]):
                    if i in loads:
                        continue
                    if regs_of(lines[i]) & qregs and not lines[i].strip().startswith(";"):
                        problems.append(f"Q register touched before the wait: {lines[i].strip()}")
                        break
            inasm = False
            for i in range(first, mfma):
                if "#ASMSTART" in lines[i]:
                    inasm = True
                elif "#ASMEND" in lines[i]:
                    inasm = False
                elif not inasm and re.search(r"s_waitcnt\s+vmcnt\(0\)", lines[i]):
                    problems.append("compiler-inserted s_waitcnt vmcnt(0) before the first score MFMA")
                    break
        print(f"{short:36s} spills={sp.get(name, 0):4d}  {'OK' if not problems else '; '.join(problems)}")
        ok = ok and not problems
    return ok


if __name__ == "__main__":
    sys.exit(0 if audit() else 1)
