"""Build-time check of trsm_strip8_kernel's hand-issued loads (kernels_trsm.hip): the loads for the next block are
issued by inline asm, so the compiler does not know their results arrive later.  That is only sound if no instruction
touches a destination register between the load and the point where the kernel has waited for it (the block end, first
use: v_xor / v_mov / ds_write).  This script compiles the file to ISA and verifies exactly that, for both instantiations.
usage: python scripts/check_hand_issued_loads.py [device-assembly.s]  (exit code 0 = ok).  With a path (what
cbo_with_oop_amd/csrc/Makefile does for the product AND the DIAG=1 object, on the assembly that becomes the object) it
checks that file; without, it compiles the source itself (CHECK_DIAG=1: with -DCBO_DIAG_KNOBS); also run by tests/test_abi.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cbo_with_oop_amd", "csrc", "kernels_trsm.hip")


def registers(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(asm_text):
    """Every instruction that names a destination register of a hand-issued load must be a hand-issued load itself or
    come after the kernel's pin (an empty inline-asm statement, which the source places behind the stage-top wait that
    retires the loads) within its basic block.  Those registers stay live from the loads at kernel start to the kernel's
    end, so any other mention is a copy or a reuse the loads would race with.  Code ahead of the first such load in the
    listing (the entry block) is exempt."""
    problems, checked = [], 0
    kernels = re.split(r"\n(?=_ZN3cbo18trsm_strip8_kernel\S*:)", asm_text)[1:]
    for body in kernels:
        name = body.split(":", 1)[0]
        raw = body.split("\n")
        end = next(i for i, l in enumerate(raw) if "s_endpgm" in l)
        raw = raw[:end + 1]
        code = [l.split(";")[0].rstrip() for l in raw]
        in_asm, asm_lines, pins = False, set(), set()      # inline asm is bracketed by ;;#ASMSTART / ;;#ASMEND
        start = None
        for i, l in enumerate(raw):
            if "#ASMSTART" in l:
                in_asm, start = True, i
            elif "#ASMEND" in l:
                in_asm = False
                if i == start + 1:
                    pins.add(i)                            # empty statement: a pin
            elif in_asm:
                asm_lines.add(i)
        loads = [i for i in sorted(asm_lines) if re.match(r"\s*global_load_dwordx[24] v", code[i])]
        if not loads:
            problems.append(f"{name}: no hand-issued loads found")
            continue
        dst = set()
        for i in loads:
            dst |= registers(code[i].split(",")[0])
        checked += len(loads)
        for i in range(loads[0] + 1, len(code)):
            t = code[i].strip()
            if i in loads or not t or t.startswith(".") or t.endswith(":") or not (registers(t) & dst):
                continue
            # back towards the start of the straight-line code this instruction is reached through
            ok = False
            for b in range(i - 1, -1, -1):
                tb = code[b].strip()
                if b in pins:
                    ok = True
                    break
                if tb.endswith(":") or tb.startswith("s_branch") or tb.startswith("s_setpc"):
                    break                                  # (a conditional branch is passed: the fall-through block has
                                                           #  no label, its only predecessor is the code above it)
            if not ok:
                problems.append(f"{name}: line {i}: '{t}' names {sorted(registers(t) & dst)}, a register of a hand-issued "
                                f"load, with no pin before it in its basic block")
    return checked, problems


def check_pair(asm_text):
    """trsm_pair_kernel: its hand-issued loads are issued in the diagonal phase of one pair and read back behind the first
    stage top of the next, across the loop's back edge, and their registers are free for other uses in between -- so this is
    a forward data-flow over the kernel's control-flow graph: a load puts its destination registers "in flight", the named
    pin of a register ("; ahead-pin vN" inside an inline-asm bracket) lands it, a load of the same registers re-arms them,
    and any OTHER instruction that names an in-flight register is a violation (a copy, a spill or a reuse the load would
    race with).  At a merge the in-flight sets are united."""
    problems, checked = [], 0
    for body in re.split(r"\n(?=_ZN3cbo16trsm_pair_kernel\S*:)", asm_text)[1:]:
        name = body.split(":", 1)[0]
        raw = body.split(".end_amdhsa_kernel")[0].split("\n") if ".end_amdhsa_kernel" in body else body.split("\n")
        last = max(i for i, l in enumerate(raw) if "s_endpgm" in l)
        raw = raw[:last + 1]
        code = [l.split(";")[0].strip() for l in raw]
        in_asm, asm_lines = False, set()
        pins = {}
        for i, l in enumerate(raw):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm:
                asm_lines.add(i)
                if "ahead-pin" in l:
                    pins[i] = registers(l.split("ahead-pin", 1)[1])
        loads = {i: registers(code[i].split(",")[0]) for i in asm_lines if re.match(r"global_load_dwordx[24] v", code[i])}
        if not loads or not pins:
            problems.append(f"{name}: no hand-issued loads / pins found")
            continue
        checked += len(loads)
        # basic blocks by label; successors
        labels = {code[i][:-1]: i for i in range(len(code)) if re.match(r"\.?\w+:$", code[i])}
        def succ(i):
            t = code[i]
            if t.startswith("s_endpgm"):
                return []
            if t.startswith("s_branch"):
                return [labels[t.split()[1]]]
            if t.startswith("s_cbranch"):
                return [labels[t.split()[1]], i + 1]
            return [i + 1] if i + 1 < len(code) else []
        state = {0: frozenset()}
        work = [0]
        seen_bad = set()
        while work:
            i = work.pop()
            cur = set(state[i])
            t = code[i]
            if i in loads:
                cur |= loads[i]
            elif i in pins:
                cur -= pins[i]
            elif re.match(r"s_waitcnt\b.*\bvmcnt\(0\)", t):
                cur = set()                                # everything has landed (the kernel's drain; the last pair's
                                                           # loads are never read back)
            elif t and not t.startswith(".") and not t.endswith(":") and i not in asm_lines or (i in asm_lines and t):
                hit = registers(t) & cur
                if hit and i not in loads and i not in pins and i not in seen_bad:
                    seen_bad.add(i)
                    problems.append(f"{name}: line {i}: '{t}' names {sorted(hit)} while a hand-issued load into it is in flight")
            for j in succ(i):
                new = frozenset(cur) | state.get(j, frozenset())
                if j not in state or new != state[j]:
                    state[j] = new
                    work.append(j)
    return checked, problems


def main():
    if len(sys.argv) > 1:
        # the Makefile's form: the device assembly of the very compilation that becomes the object (-save-temps=obj)
        text = open(sys.argv[1]).read()
        checked, problems = check(text)
        c2, p2 = check_pair(text)
        checked, problems = checked + c2, problems + p2
    else:
        flags = ["-DCBO_DIAG_KNOBS"] if os.environ.get("CHECK_DIAG") else []
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"] + flags +
                                  ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-S",
                                   "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
            checked, problems = check(text)
            c2, p2 = check_pair(text)
            checked, problems = checked + c2, problems + p2
    print(f"checked {checked} hand-issued load sites; {len(problems)} problem(s)")
    for p in problems:
        print("  ", p)
    return 1 if problems or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
