#!/usr/bin/env python3
"""Developer helper: per-kernel VGPR count and instruction histogram of the hottest loop from a hipcc -S listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/api.s multimodal_transformer_amd/csrc/api.hip
  python tools/isa_hist.py /tmp/api.s attn_bwd_dkv_kernelILi16ELb1E [--top 25]
The "hottest loop" is taken to be the innermost loop (by label range) with the most instructions."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    meta = [l for l in lines[end:end + 80] if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize", l)]
    print(lines[start][:100])
    for m in meta[:6]:
        print("  ", m.strip())
    # loops: the blocks the compiler's comments assign to one inner loop header (the header itself and "in Loop: Header=" blocks)
    lab = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    loops = []
    for i, l in lab:
        if "Inner Loop Header" not in l:
            continue
        name = l.split(":")[0][2:]                       # BBn_m
        member = [k for k, (j, t) in enumerate(lab) if j == i or ("Header=" + name + " ") in t]
        a = lab[member[0]][0]
        b = lab[member[-1] + 1][0] if member[-1] + 1 < len(lab) else len(body)
        loops.append((a, b))
    if not loops:
        print("no loop found"); return
    def ninstr(a, b):
        return sum(1 for l in body[a:b] if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";"))
    a, b = max(loops, key=lambda ab: ninstr(*ab))
    hist = collections.Counter(l.split()[0] for l in body[a:b] if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";"))
    valu = sum(c * (4 if k.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_mul_lo", "v_mul_hi_u32")) else 1)
               for k, c in hist.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print("loop lines %d..%d: %d instructions, VALU issue slots (transcendentals x4) = %d, MFMA = %d"
          % (a, b, sum(hist.values()), valu, sum(c for k, c in hist.items() if k.startswith("v_mfma"))))
    for k, c in hist.most_common(top):
        print("  %4d %s" % (c, k))


if __name__ == "__main__":
    main()
