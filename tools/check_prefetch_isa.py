#!/usr/bin/env python3
"""Static guard for the inline-asm software prefetch of the tile sweeps.

tree_up_micro_kernel and qlt_down_sp_kernel keep the NEXT chunk's cell records in flight with
`asm volatile("global_load_...")` and tie the destination registers to a later
`asm volatile("s_waitcnt vmcnt(0)" : "+v"(...))`. hipcc does not count inline-asm loads, so
between the two statements it believes the destination registers already hold their values: a
copy, spill or reuse of one of them there would read (or be overwritten by) data still in
flight -- silently wrong sums. This script compiles the kernels to gfx950 assembly and checks,
function by function in layout order, that NO compiler-generated instruction (anything outside
`;;#ASMSTART ... ;;#ASMEND`) names a VGPR that an inline-asm load has targeted and no inline-asm
wait has retired yet. It is run by __graft_entry__.build() and by tests/test_cpu_abi.py.

usage: check_prefetch_isa.py [listing.s]      (without an argument it compiles cedr_kernels.hip)
exit status 0 = clean; 1 = violations (printed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "compose_amd", "csrc", "cedr_kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", "-S",
         "--cuda-device-only"]

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def listing(extra_flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC] + FLAGS + list(extra_flags) + ["-o", out, SRC],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\.L\w+)")


def check(text):
    """Returns (violations, stats): stats[function] = (asm loads, asm waits).

    A forward dataflow over each function's control-flow graph (basic blocks from the listing's labels and
    s_branch / s_cbranch_* instructions): the set of VGPRs some inline-asm load may still be filling at a point is the
    union over all paths reaching it; an inline-asm `s_waitcnt vmcnt(0)` empties it. (Until round 4 this walked the
    listing in layout order, which missed a real hazard: a wait on a side path that happens to be laid out first
    made the main path's copies of in-flight registers look safe.)"""
    violations, stats = [], {}
    # ---- split into functions
    funcs, fn, body = [], None, []
    for ln, line in enumerate(text.splitlines(), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, body = m.group(1), []
            continue
        if fn is None:
            continue
        if line.strip().startswith(".Lfunc_end"):
            funcs.append((fn, body))
            fn = None
            continue
        body.append((ln, line))
    for fn, body in funcs:
        # ---- basic blocks: list of (kind, ln, text, regs) items; kind in {"load", "wait", "code"}
        blocks, labels, cur = [], {}, {"items": [], "succ": [], "fall": True}
        blocks.append(cur)
        in_asm, nload, nwait = False, 0, 0

        def new_block():
            nonlocal cur
            cur = {"items": [], "succ": [], "fall": True}
            blocks.append(cur)
        for ln, line in body:
            s = line.strip()
            m = re.match(r"^(\.L\w+):", s)
            if m and not in_asm:
                if cur["items"] or cur["succ"]:
                    new_block()
                labels[m.group(1)] = len(blocks) - 1
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            code = s.split(";")[0].strip()
            if in_asm:
                if re.match(r"(global|buffer|flat)_load", code) and " lds" not in code and "_lds_" not in code:
                    dest = code.split(None, 1)[1].split(",")[0]
                    cur["items"].append(("load", ln, s, vregs(dest)))
                    nload += 1
                elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
                    cur["items"].append(("wait", ln, s, set()))
                    nwait += 1
                continue
            cur["items"].append(("code", ln, s, vregs(code)))
            mb = BRANCH.match(code)
            if mb:
                cur["succ"].append(mb.group(2))
                if mb.group(1) == "s_branch":
                    cur["fall"] = False
                new_block()
            elif code.startswith("s_endpgm") or code.startswith("s_setpc") or code.startswith("s_trap"):
                cur["fall"] = False
                new_block()
        if not (nload or nwait):
            continue
        stats[fn] = (nload, nwait)
        succ = []
        for i, b in enumerate(blocks):
            out = [labels[t] for t in b["succ"] if t in labels]
            if b["fall"] and i + 1 < len(blocks):
                out.append(i + 1)
            succ.append(out)
        # ---- fixed point of IN sets
        IN = [set() for _ in blocks]
        work = list(range(len(blocks)))
        while work:
            i = work.pop()
            st = set(IN[i])
            for kind, ln, s, regs in blocks[i]["items"]:
                if kind == "load":
                    st |= regs
                elif kind == "wait":
                    st = set()
            for j in succ[i]:
                if not st <= IN[j]:
                    IN[j] |= st
                    work.append(j)
        # ---- report
        seen = set()
        for i, b in enumerate(blocks):
            st = set(IN[i])
            for kind, ln, s, regs in b["items"]:
                if kind == "load":
                    st |= regs
                elif kind == "wait":
                    st = set()
                elif st:
                    hit = regs & st
                    if hit and ln not in seen:
                        seen.add(ln)
                        violations.append((fn, ln, s, sorted(hit)))
    violations.sort(key=lambda v: v[1])
    return violations, stats


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else listing()
    violations, stats = check(text)
    for fn, (nl, nw) in sorted(stats.items()):
        print("%-100s asm loads %3d  asm waits %2d" % (fn[:100], nl, nw))
    if not stats:
        print("no inline-asm prefetch found: nothing was checked")
        return 1
    for fn, ln, s, hit in violations[:40]:
        print("VIOLATION %s line %d: `%s` touches v%s while an asm load into it is in flight" % (fn[:60], ln, s, hit))
    print("%d violation(s)" % len(violations))
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main())
