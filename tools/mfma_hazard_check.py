"""Static check of the asm-MFMA kernels (cdlrm_amd/csrc/gemm_wide.h: k_gemm3).

The MFMAs of k_gemm3 are asm statements (accumulators tied in place), so hipcc's hazard recognizer does not see them: a VALU /
memory instruction the COMPILER places behind such an MFMA (a register copy at a loop edge, an epilogue read) may read the
accumulator before the matrix pipe has written it.  This script scans the code of every k_gemm3 instantiation -- a listing
(`hipcc -save-temps`: *.s) or the disassembly of a code object / shared library (llvm-objdump -d) -- in program order and reports
every non-MFMA instruction that names a register an asm MFMA (VGPR accumulator, destination = SrcC) wrote fewer than WAIT wait
states earlier, and every MFMA that reads such a register as an A / B operand or as a DIFFERENT accumulator range.  Wait states: 1 per instruction, N+1 for `s_nop N`,
8 per v_mfma_f32_16x16x4_f32 (the pipe is busy 32 cycles = 8 issue slots; the ISA asks for 11 behind an 8-pass MFMA).
Fall-through order only (a branch target is also checked as the textual successor): conservative for the straight-line tiles.

    python3 tools/mfma_hazard_check.py cdlrm_amd/csrc/libcdlrm_hip.so        # exit code 1 on a finding
"""
import re
import subprocess
import sys

WAIT = 12
VALU_TO_MFMA = 3     # wait states between a VALU write of a register and an asm MFMA that reads it as its accumulator
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OFFLOAD = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def regs_of(tok):
    """set of VGPR numbers named by one operand token ('v12', 'v[4:7]'); AGPRs as 1000+n"""
    out = set()
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        out.add((1000 if m.group(1) == "a" else 0) + int(m.group(2)))
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return out


def split_ops(rest):
    return [t.strip() for t in re.split(r",\s*|\s+", rest) if t.strip()]


def check_function(name, lines):
    pending = {}        # reg -> wait states still required (written by an asm MFMA)
    vwritten = {}       # reg -> wait states since a VALU instruction wrote it (an asm MFMA reading it as SrcC needs VALU_TO_MFMA)
    findings = []
    for ln, text in lines:
        ins = text.split("//")[0].split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        parts = ins.split(None, 1)
        op = parts[0]
        ops = split_ops(parts[1]) if len(parts) > 1 else []
        touched = set()
        for t in ops:
            touched |= regs_of(t)
        cost = 1
        if op.startswith("v_mfma"):
            dst = regs_of(ops[0])
            srcc = regs_of(ops[3]) if len(ops) > 3 else set()
            ab = regs_of(ops[1]) | regs_of(ops[2])
            bad = {r for r in ab if pending.get(r, 0) > 0}
            if srcc != dst:
                bad |= {r for r in (srcc | dst) if pending.get(r, 0) > 0}
            if bad:
                findings.append((ln, ins, sorted(bad)))
            # the asm forms of gemm_wide.h: VGPR accumulator tied in place, or SrcC the inline constant 0 (g3_mfma_init)
            asm_form = all(r < 1000 for r in dst) and (srcc == dst or (len(ops) > 3 and ops[3] == "0"))
            if asm_form and srcc == dst:
                late = {r for r in dst if r in vwritten and vwritten[r] < VALU_TO_MFMA}
                if late:
                    findings.append((ln, ins + "    [a VALU write of the accumulator too close in front]", sorted(late)))
            cost = 8
            for r in list(vwritten):
                vwritten[r] += cost
                if vwritten[r] > 16:
                    del vwritten[r]
            for r in list(pending):
                pending[r] -= cost
                if pending[r] <= 0:
                    del pending[r]
            # only the asm form (VGPR accumulator tied in place) is unknown to the compiler; an MFMA it emitted itself
            # (the progressive epilogue's builtin: AGPR accumulators, dst != SrcC) is covered by its own hazard recognizer
            if asm_form:
                for r in dst:
                    pending[r] = WAIT
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending.clear()             # what follows in the text is not reached by falling through
            vwritten.clear()
            continue
        if op == "s_nop":
            cost = int(ops[0], 0) + 1
        else:
            bad = {r for r in touched if pending.get(r, 0) > 0}
            if bad:
                findings.append((ln, ins, sorted(bad)))
        for r in list(pending):
            pending[r] -= cost
            if pending[r] <= 0:
                del pending[r]
        for r in list(vwritten):
            vwritten[r] += cost
            if vwritten[r] > 16:
                del vwritten[r]
        if op.startswith("v_") and ops:
            for r in regs_of(ops[0]):
                if r < 1000:
                    vwritten[r] = 0
    return findings


def functions_from_listing(text):
    """{name: [(lineno, text)]} for .s listings and llvm-objdump output alike"""
    funcs, cur = {}, None
    for ln, line in enumerate(text.splitlines(), 1):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w*k_gemm3\w*)>?:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None:
            continue
        if re.match(r"^(?:[0-9a-f]+ <)?[_A-Za-z]\w*>?:\s*$", line) and "k_gemm3" not in line and not line.startswith(".L"):
            cur = None
            continue
        body = line
        m = re.match(r"^\s*(.*?)\s*//\s*[0-9A-Fa-f]+:.*$", line)      # objdump: "\tins  // addr: bytes"
        if m:
            body = m.group(1)
        funcs[cur].append((ln, body))
        if "s_endpgm" in body:
            cur = None
    return funcs


def disassemble(path):
    """listing text of every gfx950 code object in `path` (a .s listing, a code object, an object file or a shared library:
    the .hip_fatbin section holds one offload bundle per translation unit)"""
    if path.endswith(".s"):
        return open(path).read()
    import os
    import tempfile
    tmp = tempfile.mkdtemp()
    sec = os.path.join(tmp, "fatbin")
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, sec], check=True)
    blob = open(sec, "rb").read() if os.path.exists(sec) else b""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    if not starts:
        return subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", path], capture_output=True, text=True, check=True).stdout
    text = []
    for k, st in enumerate(starts):
        piece = os.path.join(tmp, f"bundle{k}")
        open(piece, "wb").write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        r = subprocess.run([OFFLOAD, "--list", "--type=o", f"--input={piece}"], capture_output=True, text=True)
        for t in [t for t in r.stdout.split() if "gfx950" in t]:
            out = os.path.join(tmp, f"dev{k}.co")
            subprocess.run([OFFLOAD, "--unbundle", "--type=o", f"--input={piece}", f"--targets={t}", f"--output={out}"], check=True)
            text.append(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", out], capture_output=True, text=True, check=True).stdout)
    return "\n".join(text)


def main(argv):
    total = 0
    nfun = 0
    for path in argv[1:]:
        funcs = functions_from_listing(disassemble(path))
        for name, lines in funcs.items():
            nfun += 1
            nm = sum(1 for _, t in lines if "v_mfma" in t)
            f = check_function(name, lines)
            print(f"{name}: {len(lines)} lines, {nm} MFMAs, {len(f)} finding(s)")
            for ln, ins, regs in f[:12]:
                print(f"   line {ln}: {ins}    <- registers {regs[:8]} still in the matrix pipe")
            total += len(f)
    if nfun == 0:
        print("no k_gemm3 kernel found")
        return 2
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
