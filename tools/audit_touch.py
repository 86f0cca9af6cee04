"""Audit of the compositing kernels' generated code (no GPU needed): the register hazards of their asm statements.

1. The touch-ahead loads (`global_load_dword vSINK, vOFF, s[base]` inside an asm statement) land in their
   destination register long after the compiler believes it written.  For every kernel and every such load this
   script follows the code from the load to the drain (`s_waitcnt vmcnt(0)` inside an asm statement) along every
   path and fails if any other instruction writes the sink register on the way, or if a path reaches the end of
   the kernel without a drain.
2. Every asm statement whose vector memory instruction takes a scalar base must open with `s_nop 4`: the base
   may come fresh from `v_readfirstlane`, and the compiler does not pad inside asm statements.

Usage: python tools/audit_touch.py [composite.s]   (without an argument it compiles csrc/composite.hip to assembly)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-gaussian-splatting_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-sanitize"]

TOUCH = re.compile(r"^\s*global_load_dword (v\d+), v\d+, s\[\d+:\d+\]")
SBASE = re.compile(r"^\s*global_(load|store)_dword\S* v\d+, v\d+, s\[\d+:\d+\]")
BRANCH = re.compile(r"^\s*s_(c?branch\S*)\s+(\.LBB\d+_\d+)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def assembly(path=None):
    if path:
        return open(path).read()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "composite.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, "-S", "--cuda-device-only",
                        os.path.join(CSRC, "composite.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(text):
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if m and i + 1 < len(lines):
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i:j]
            i = j
        i += 1


def writes(line, reg):
    """True when the instruction's destination (its first operand) covers vector register `reg`."""
    t = line.strip()
    if not t.startswith(("v_", "global_load", "buffer_load", "ds_read", "flat_load", "scratch_load")):
        return False
    if t.startswith(("v_cmp", "v_nop")):      # compares write vcc / scalar registers
        return False
    ops = t.split(None, 1)[1] if " " in t else ""
    dst = ops.split(",")[0].strip()
    n = int(reg[1:])
    m = re.match(r"^v(\d+)$", dst)
    if m:
        return int(m.group(1)) == n
    m = re.match(r"^v\[(\d+):(\d+)\]$", dst)
    if m:
        return int(m.group(1)) <= n <= int(m.group(2))
    return False


def swaps(line, reg):
    t = line.strip()
    return t.startswith("v_permlane") and "swap" in t and re.search(r"\b%s\b" % reg, t) is not None


def audit(text):
    problems, touches = [], 0
    for name, body in kernels(text):
        label_at = {m.group(1): i for i, l in enumerate(body) for m in [LABEL.match(l)] if m}
        in_asm = [False] * len(body)
        flag = False
        for i, l in enumerate(body):
            if "#ASMSTART" in l:
                flag = True
            in_asm[i] = flag
            if "#ASMEND" in l:
                flag = False
        # rule 2: scalar-base memory instructions inside asm statements open with s_nop 4
        for i, l in enumerate(body):
            if in_asm[i] and SBASE.match(l):
                j = i
                while j > 0 and "#ASMSTART" not in body[j]:
                    j -= 1
                if body[j + 1].strip() != "s_nop 4":
                    problems.append(f"{name}: line {i}: asm statement with a scalar-base memory instruction does not open with s_nop 4")
        # rule 1
        for i, l in enumerate(body):
            m = TOUCH.match(l)
            if not (m and in_asm[i]):
                continue
            touches += 1
            sink = m.group(1)
            seen, stack = set(), [i + 1]
            while stack:
                k = stack.pop()
                while k < len(body):
                    if k in seen:
                        break
                    seen.add(k)
                    t = body[k]
                    if in_asm[k] and t.strip().startswith("s_waitcnt vmcnt(0)"):
                        break                                   # drained on this path
                    if t.strip().startswith("s_endpgm"):
                        problems.append(f"{name}: a touch load into {sink} (line {i}) reaches the end of the kernel without a drain")
                        break
                    tm = TOUCH.match(t)
                    if (writes(t, sink) or swaps(t, sink)) and not (tm and in_asm[k] and tm.group(1) == sink):
                        problems.append(f"{name}: line {k}: `{t.strip()}` writes {sink} while a touch load (line {i}) may be in flight")
                        break
                    b = BRANCH.match(t)
                    if b:
                        if b.group(2) in label_at:
                            stack.append(label_at[b.group(2)])
                        if b.group(1) == "branch":
                            break
                    k += 1
    return touches, problems


if __name__ == "__main__":
    n, bad = audit(assembly(sys.argv[1] if len(sys.argv) > 1 else None))
    for p in bad:
        print("HAZARD", p)
    print(f"{n} touch loads audited, {len(bad)} problems")
    sys.exit(1 if bad or n == 0 else 0)
