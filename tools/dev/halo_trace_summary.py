#!/usr/bin/env python3
"""Summarise the block timeline printed by build/conv_ubench_trace (H3_TRACE build of tools/ubench/conv_ubench.cpp):
phase durations, start-time distribution, blocks per CU and how much of a CU's time has one block contracting
while another stages or stores."""
import collections
import statistics as st
import sys

cur = None
runs = {}
for ln in open(sys.argv[1]):
    if ln.startswith("TRACE"):
        cur = ln.strip()
        runs[cur] = []
    elif ln.startswith("B ") and cur:
        f = ln.split()
        runs[cur].append(dict(i=int(f[1]), xcc=int(f[3]), se=int(f[5]), cu=int(f[7]), simd=int(f[9]), wave=int(f[11]),
                              t=[float(x) for x in f[12:16]]))
for name, bl in runs.items():
    print("==", name)
    for k, lab in ((0, "stage"), (1, "contract"), (2, "epilogue")):
        d = [b["t"][k + 1] - b["t"][k] for b in bl]
        print("  %-9s mean %.2f  median %.2f  p10 %.2f  p90 %.2f us" % (lab, st.mean(d), st.median(d),
              sorted(d)[len(d) // 10], sorted(d)[9 * len(d) // 10]))
    d = [b["t"][3] - b["t"][0] for b in bl]
    print("  block life mean %.2f median %.2f" % (st.mean(d), st.median(d)))
    starts = sorted(b["t"][0] for b in bl)
    print("  starts: first 512 by %.2f us; block 513 at %.2f; last start %.2f" % (starts[min(511, len(starts) - 1)],
          starts[min(512, len(starts) - 1)], starts[-1]))
    hist = collections.Counter(int(s) for s in starts)
    print("  start histogram (us: blocks):", " ".join("%d:%d" % kv for kv in sorted(hist.items())))
    cus = collections.defaultdict(list)
    for b in bl:
        cus[(b["xcc"], b["se"], b["cu"])].append(b)
    per = collections.Counter(len(v) for v in cus.values())
    print("  distinct CUs %d; blocks per CU: %s" % (len(cus), dict(per)))
    # overlap: sample time at 0.05 us steps per CU
    both_c = one_c_one_m = both_m = only_one = 0
    for v in cus.values():
        t = 0.0
        end = max(b["t"][3] for b in v)
        while t < end:
            ph = []
            for b in v:
                if b["t"][0] <= t < b["t"][3]:
                    ph.append("c" if b["t"][1] <= t < b["t"][2] else "m")
            if len(ph) >= 2:
                if ph.count("c") >= 2:
                    both_c += 1
                elif ph.count("c") == 1:
                    one_c_one_m += 1
                else:
                    both_m += 1
            elif len(ph) == 1:
                only_one += 1
            t += 0.05
    tot = both_c + one_c_one_m + both_m + only_one
    print("  CU-time: 2 contracting %.0f%%, 1 contracting + 1 in memory phase %.0f%%, 2 in memory phases %.0f%%, "
          "single block resident %.0f%%" % (100 * both_c / tot, 100 * one_c_one_m / tot, 100 * both_m / tot,
                                            100 * only_one / tot))
    k = sorted(cus)[5]
    print("  example CU", k)
    for b in sorted(cus[k], key=lambda b: b["t"][0]):
        print("    block %4d simd %d wave %d: %s" % (b["i"], b["simd"], b["wave"], " ".join("%6.2f" % x for x in b["t"])))
