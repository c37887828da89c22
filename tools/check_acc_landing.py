#!/usr/bin/env python3
"""Static check of the software-pipelined one-wave-per-row kernels (fft_wave.hip, pass_threads = 65): their look-ahead
operands are loaded into accumulation registers by inline assembly, which the compiler's s_waitcnt insertion does not
see.  The kernels are only correct if the compiler itself never READS such a register (a copy, a spill, a move to a
vector register): every legitimate read is one of the v_accvgpr_read_b32 of wtake_row(), inline assembly as well, placed
behind the explicit wait.  The assembler output marks inline assembly (";;#ASMSTART" ... ";;#ASMEND"), so the check is:
   * landing registers = accumulation registers written by a global_load inside an inline-assembly block;
   * no instruction OUTSIDE such blocks may name a landing register as a source (writes - the zero fill of a skipped
     request - are harmless), and the kernel may have no scratch traffic.
Usage: tools/check_acc_landing.py            (compiles fdes_amd/csrc/fft_wave.hip to assembly; exit code 1 on a finding)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fdes_amd", "csrc", "fft_wave.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "w.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S",
                           "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    asm = open(out).read()

def regs(tok):
    m = re.match(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"a(\d+)$", tok)
    return {int(m.group(1))} if m else set()

bad = 0
for m in re.finditer(r"^(_ZN4fdes12_GLOBAL__N_17k_wpassILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb1E\w*):", asm, re.M):
    name = "k_wpass<%s, %s, %s, %s, %s, pipe>" % m.groups()[1:]
    body = asm[m.end():asm.index("s_endpgm", m.end())].splitlines()
    landing, findings, in_asm = set(), [], False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_load"):
            landing |= regs(t.replace(",", " ").split()[1])
    if not landing:
        continue  # this pass has no look-ahead (its pipelined form falls back, pipe_fits())
    in_asm = False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm or not t or t.startswith((";", ".")):
            continue
        tok = t.replace(",", " ").split()
        if tok[0].startswith(("scratch_", "buffer_")):
            findings.append("scratch traffic: " + t)
            continue
        srcs = tok[2:] if len(tok) > 2 else []
        if tok[0].startswith(("global_store", "ds_write", "flat_store")):
            srcs = tok[1:]  # stores have no destination register
        for x in srcs:
            if regs(x) & landing:
                findings.append("compiler reads a landing register: " + t)
                break
    print(("FAIL " if findings else "ok   ") + name + f"  ({len(landing)} landing registers)")
    for f in findings[:5]:
        print("      ", f)
    bad += bool(findings)
sys.exit(1 if bad else 0)
