#!/usr/bin/env python3
"""Build guard: the DMA-staged kernels count their own `s_waitcnt vmcnt(N)` around inline-asm loads and
global->LDS DMA.  A register spill is a scratch load / store -- a vector-memory operation the compiler adds to that same
in-order queue -- and silently breaks the count.  This reads hipcc's `-Rpass-analysis=kernel-resource-usage` remarks (the
Makefile saves them per object) and fails the build when one of those kernels has a non-zero ScratchSize.

    python tools/check_no_scratch.py build/conv_trunk.rpt build/conv_tail.rpt build/conv_fnet.rpt
"""
import re
import sys

GUARDED = ("trunk_conv_kernel", "trunk_convT_kernel", "trunk6_conv_kernel", "tail_kernel", "fnet_conv_kernel", "convT3x3_gather_kernel")


def main(paths):
    bad, seen = [], 0
    for path in paths:
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                continue
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and name and any(g in name for g in GUARDED):
                seen += 1
                if int(m.group(1)):
                    bad.append((path, name, int(m.group(1))))
    if not seen:
        print("check_no_scratch: no guarded kernel found in", paths, file=sys.stderr)
        return 1
    for path, name, n in bad:
        print(f"check_no_scratch: {name} ({path}) uses {n} bytes of scratch per lane: spills break its hand-counted vmcnt waits",
              file=sys.stderr)
    if not bad:
        print(f"check_no_scratch: {seen} guarded kernels, no scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
