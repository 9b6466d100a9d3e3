"""Build-time check of trsm_strip8_kernel's hand-issued loads (kernels_trsm.hip): the loads for the next block are
issued by inline asm, so the compiler does not know their results arrive later.  That is only sound if no instruction
touches a destination register between the load and the point where the kernel has waited for it (the block end, first
use: v_xor / v_mov / ds_write).  This script compiles the file to ISA and verifies exactly that, for both instantiations.
usage: python scripts/check_hand_issued_loads.py  (exit code 0 = ok); also run by tests/test_abi.py"""
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
    problems, checked = [], 0
    kernels = re.split(r"\n(?=_ZN3cbo18trsm_strip8_kernel\S*:)", asm_text)[1:]
    for body in kernels:
        name = body.split(":", 1)[0]
        raw = body.split("\n")
        lines = [l.split(";")[0].rstrip() for l in raw]
        end = next(i for i, l in enumerate(lines) if "s_endpgm" in l)
        lines = lines[:end + 1]
        in_asm, hand = False, set()                       # inline asm is bracketed by ;;#ASMSTART / ;;#ASMEND
        for i, l in enumerate(raw[:end + 1]):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm:
                hand.add(i)
        # the hand-issued loads: "global_load_dwordx2/x4 vDST, v[ADDR], off" inside an inline-asm bracket
        for i, l in enumerate(lines):
            m = re.match(r"\s*global_load_dwordx([24]) (v\[\d+:\d+\]), (v\[\d+:\d+\]), off\s*$", l)
            if not m or i not in hand:
                continue
            dst = registers(m.group(2))
            # walk forward (straight-line order of the listing) to the first instruction that names a destination register
            for j in range(i + 1, len(lines)):
                t = lines[j].strip()
                if not t or t.startswith(".") or t.endswith(":"):
                    continue
                if t.startswith("global_load_dword") and registers(t.split(",")[0]) & dst:
                    checked += 1                          # untouched up to the next load site of the same register
                    break
                if registers(t) & dst:
                    checked += 1
                    # legitimate first uses: after the kernel's own stage-top waits, at the block end
                    window = [x for x in lines[i + 1:j] if "s_barrier" in x]
                    waited = any("vmcnt" in x for x in lines[i + 1:j])
                    if not waited or len(window) < 1:
                        problems.append(f"{name}: line {j}: '{t}' touches {sorted(registers(t) & dst)} loaded at line {i} "
                                        f"with no stage-top wait in between")
                    break
    return checked, problems


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-S", "--cuda-device-only",
                               SRC, "-o", out], stderr=subprocess.DEVNULL)
        checked, problems = check(open(out).read())
    print(f"checked {checked} hand-issued load sites; {len(problems)} problem(s)")
    for p in problems:
        print("  ", p)
    return 1 if problems or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
