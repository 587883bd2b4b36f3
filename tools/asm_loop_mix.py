#!/usr/bin/env python
"""tools/asm_loop_mix.py — per kernel of a `hipcc -S --cuda-device-only` listing: instruction mix of every loop body that holds MFMAs
(matrix / other vector / scalar / LDS / VMEM instructions, waits, nops). fp32 MFMAs issue on the vector ALU's own pipe on gfx950
(tools/probes/probe_mfma_f32.hip: ~5 cycles of matrix time lost per VALU instruction), so "valu per mfma" in a k-loop is a direct cost.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icelebbasis_amd/csrc -S --cuda-device-only -o /tmp/x.s celebbasis_amd/csrc/cb_flash.hip
    python tools/asm_loop_mix.py /tmp/x.s [substring of the demangled kernel name]
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    kernels, cur, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            kernels[cur] = body
        elif cur and line.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            body.append(line.rstrip("\n"))
    names = demangle(list(kernels))
    for k, body in kernels.items():
        nm = names[k].replace("cb::", "")
        nm = re.sub(r"\(.*", "", nm).replace("void ", "")
        if flt not in nm:
            continue
        # basic blocks; a loop = label ... backward branch to that label
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        seen = False
        for a, b in loops:
            ops = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
            mf = sum(o.startswith("v_mfma") for o in ops)
            if not mf:
                continue
            va = sum(o.startswith("v_") and not o.startswith("v_mfma") for o in ops)
            acc = sum("accvgpr" in o for o in ops)
            sa = sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio")) for o in ops)
            lds = sum(o.startswith("ds_") for o in ops)
            vm = sum(o.startswith(("buffer_", "global_", "flat_")) for o in ops)
            wt = sum(o.startswith("s_waitcnt") for o in ops)
            nop = sum(o.startswith("s_nop") for o in ops)
            bar = sum(o.startswith("s_barrier") for o in ops)
            if not seen:
                print(nm)
                seen = True
            print(f"    loop @{a:5d}..{b:5d}: mfma {mf:4d}  valu {va:4d} ({va / mf:.2f}/mfma, accvgpr {acc})  salu {sa:3d}  lds {lds:3d}  vmem {vm:3d}  waitcnt {wt:3d}  nop {nop:2d}  barrier {bar}")


if __name__ == "__main__":
    main()
