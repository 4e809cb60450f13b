#!/usr/bin/env python3
"""Where one rank's share of an N-GPU proof goes: turns a `rocprofv3 --kernel-trace --output-format csv` trace of
tools/shard_emulation.py (N rank threads on one GPU, serialised) into per-rank-proof kernel statistics.

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o emu -- python3 tools/shard_emulation.py --ranks 8 --steps 2
  python tools/trace_per_rank.py OUT/emu_kernel_trace.csv > profiles/rNN_kernel_stats_per_rank_proof_8ranks.csv

The trace carries the dispatching host thread of every kernel, and a rank IS a thread, so the last proof of every rank thread
(everything after its last k_witness_cyclic launch) is averaged over the ranks.  Columns: kernel, launches per rank-proof,
total ms per rank-proof, share of the kernel time.  The last rows give the kernel sum, the union of the kernel intervals
(concurrent streams counted once) and the active time (union plus the gaps shorter than 150 us, i.e. launch gaps but not
the waits for the other ranks)."""
import collections
import csv
import sys


def main(path):
    by_thread = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pm::", "")
        by_thread[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    total, count, ranks = collections.Counter(), collections.Counter(), 0
    unions, actives = [], []
    for evs in by_thread.values():
        evs.sort()
        wit = [i for i, e in enumerate(evs) if "k_witness_cyclic" in e[2]]
        if len(wit) < 2:            # the setup thread and the warm-up threads
            continue
        seg = evs[wit[-1]:]
        ranks += 1
        for s, e, n in seg:
            total[n] += e - s
            count[n] += 1
        union = active = 0
        cs, ce = seg[0][0], seg[0][1]
        ws = cs
        for s, e, _ in seg[1:]:
            if s > ce:
                union += ce - cs
                if s - ce > 150000:
                    active += ce - ws
                    ws = s
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        active += ce - ws
        unions.append(union)
        actives.append(active)
    if not ranks:
        raise SystemExit("no rank threads with two proofs in this trace")
    ksum = sum(total.values())
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches_per_rank_proof", "ms_per_rank_proof", "share_of_kernel_time"])
    for n, d in total.most_common():
        w.writerow([n, "%.1f" % (count[n] / ranks), "%.4f" % (d / ranks / 1e6), "%.4f" % (d / ksum)])
    w.writerow(["# kernel sum", "", "%.4f" % (ksum / ranks / 1e6), "1.0"])
    w.writerow(["# union of kernel intervals", "", "%.4f" % (sum(unions) / ranks / 1e6), ""])
    w.writerow(["# active time (gaps < 150 us kept)", "", "%.4f" % (sum(actives) / ranks / 1e6), ""])
    w.writerow(["# rank threads averaged", ranks, "", ""])


if __name__ == "__main__":
    main(sys.argv[1])
