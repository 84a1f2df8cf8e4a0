#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_stats.py /tmp/orb.s k_fast_waveILi48 [--dump out.s]"""
import sys, re, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
cnt = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
        continue
    op = l.split()[0]
    k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cnt[k] += 1
print(dict(cnt), "total", sum(cnt.values()))
