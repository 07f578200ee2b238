#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for writes to a VGPR that an MFMA issued a few
instructions earlier still names as SrcA/SrcB: prints, per kernel, the histogram of distances (in instructions)
between the MFMA and the first later instruction that overwrites one of its A/B registers."""
import re
import sys
from collections import Counter

RANGE = re.compile(r"([va])\[(\d+):(\d+)\]|([va])(\d+)\b")


def regs(tok):
    m = RANGE.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(1):
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return {(m.group(4), int(m.group(5)))}


def main(path, window=12):
    name, ins = None, []
    out = {}
    for line in open(path):
        s = line.split(";")[0].strip()
        if s.endswith(":") and s.startswith("_Z"):
            name, ins = s[:-1], []
            out[name] = ins
        elif name and s and not s.startswith((".", ";")) and not s.endswith(":"):
            ins.append(s.split(";")[0].strip())
    for name, ins in out.items():
        if "conv_f16x3" not in name:
            continue
        hist, worst = Counter(), []
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            src = regs(ops[1]) | regs(ops[2])
            for d in range(1, window + 1):
                if i + d >= len(ins):
                    break
                u = ins[i + d]
                if u.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store")):
                    continue
                if u.startswith("v_mfma"):
                    dst = regs(u.split(None, 1)[1].split(",")[0])
                else:
                    dst = regs(u.split(None, 1)[1].split(",")[0]) if " " in u else set()
                if dst & src:
                    hist[d] += 1
                    if d <= 3:
                        worst.append((i, t, d, u))
                    break
        print(name[:90], dict(sorted(hist.items())))
        for w in worst[:6]:
            print("     @%d  %s\n        +%d  %s" % w)


if __name__ == "__main__":
    main(sys.argv[1])
