#!/usr/bin/env python3
"""Developer aid: instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/fx.s effex_amd/csrc/fxcorr.hip
    python tools/isa_count.py /tmp/fx.s 'fx_fused4096_kernelILb0ELb0EE'
Prints totals for the whole kernel and for each loop body (label-to-backward-branch) with >= 200 instructions.
"""
import re
import sys
from collections import Counter


def classify(c):
    return {"valu": sum(v for k, v in c.items() if k.startswith("v_")),
            "salu": sum(v for k, v in c.items() if k.startswith("s_")),
            "ds": sum(v for k, v in c.items() if k.startswith("ds_")),
            "vmem": sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "flat_", "scratch_")))}


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.split(";")[0].strip() for l in lines[start + 1:end + 1]]
    body = [l for l in body if l and not l.startswith((";", "."))or re.match(r"^\.LBB\S+:", l)]
    ins = [(i, l) for i, l in enumerate(body) if not l.endswith(":")]
    c = Counter(l.split()[0] for _, l in ins)
    print("kernel:", len(ins), "instructions", classify(c), "barriers", c["s_barrier"], "waitcnt", c["s_waitcnt"],
          "nop", c["s_nop"], "scratch", sum(v for k, v in c.items() if k.startswith("scratch_")))
    labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    for i, l in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                seg = [x for x in body[labels[tgt]:i + 1] if not x.endswith(":")]
                if len(seg) >= 200:
                    cc = Counter(x.split()[0] for x in seg)
                    print("loop %s: %d instructions %s barriers %d" % (tgt, len(seg), classify(cc), cc["s_barrier"]))
                    print("   top:", cc.most_common(14))


if __name__ == "__main__":
    main()
