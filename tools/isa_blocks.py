#!/usr/bin/env python3
"""Instruction histogram per basic block of one kernel in a `hipcc -S` listing.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -Iinclude -Ipolymath_amd/csrc polymath_amd/csrc/ntt.hip -o /tmp/ntt.s
  python tools/isa_blocks.py /tmp/ntt.s k_ntt_pass28INS_6BlsFrP [min_instructions]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*%s\S*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
blocks, cur = [], ["entry", collections.Counter(), 0]
blocks.append(cur)
for ln in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m:
        cur = [m.group(1), collections.Counter(), 0]
        blocks.append(cur)
        continue
    t = ln.strip()
    if not t or t[0] in ";.":
        continue
    cur[1][t.split()[0]] += 1
    cur[2] += 1
print("kernel total:", sum(b[2] for b in blocks))
for b in blocks:
    if b[2] >= floor:
        print(b[0], b[2], dict(b[1].most_common(14)))
