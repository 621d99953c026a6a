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


def check(text):
    """Returns (violations, stats): stats[function] = (asm loads, asm waits)."""
    violations, stats = [], {}
    fn, in_asm, inflight, nload, nwait = None, False, set(), 0, 0
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, in_asm, inflight, nload, nwait = m.group(1), False, set(), 0, 0
            continue
        if fn is None:
            continue
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            if nload or nwait:
                stats[fn] = (nload, nwait)
            if s.startswith(".Lfunc_end"):
                fn = None
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        code = s.split(";")[0]
        if in_asm:
            if re.match(r"(global|buffer|flat)_load", code):
                dest = code.split(None, 1)[1].split(",")[0]
                inflight |= vregs(dest)
                nload += 1
            elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
                inflight = set()
                nwait += 1
            continue
        if inflight:
            hit = vregs(code) & inflight
            if hit:
                violations.append((fn, ln, s, sorted(hit)))
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
