#!/usr/bin/env python3
"""Per-segment time line of k_seg_decode from a -DJPEGX_DECODE_STATS build (JPEGX_DECODE_STATS=<dump file>):
python microbench/decode_trace.py <dump> [nseg]"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint8)
up16 = lambda v: (v + 15) // 16 * 16
# the info words sit in front of the trace, whose place follows from them: look for a consistent set
def layout(nseg, cmax, nblocks):
    return up16(nseg * 4 + 64) + up16((nseg + 1) * 4) + up16((nblocks + 63) // 64 * 4) + up16(nseg * cmax * 16)
words = raw.view(np.uint32)
o = None
for i in range(0, words.size - 8, 4):
    n4, sg, cm, nb = (int(x) for x in words[i + 4:i + 8])
    if n4 and sg and sg % 256 == 0 and sg <= 4096 and cm in (64, 128, 256, 512, 1024, 2048) and nb and layout(n4, cm, nb) == i * 4:
        o = i * 4
        break
assert o is not None, "no trace header found: is this a -DJPEGX_DECODE_STATS build?"
head = words[o // 4:o // 4 + 16]
nseg, nblocks = int(head[4]), int(head[7])
print('segment bytes', int(head[5]), 'table capacity', int(head[6]))
o += 64
t = raw[o:o + nseg * 128].view(np.uint64).reshape(nseg, 16).astype(np.int64)
t0 = t[:, 0].min()
names = ["stage", "cand", "parse", "double", "exitwait", "list"]
start = (t[:, 0] - t0) / 100.0
end = (t[:, 6] - t0) / 100.0
ok = t[:, 6] > 0
print("segments", nseg, "traced to the end", int(ok.sum()), "kernel span %.1f us" % end[ok].max())
print("start times us: p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(np.percentile(start, [0, 25, 50, 75, 100])))
for k, n in enumerate(names):
    sel = ok & (t[:, k + 1] > 0)
    d = (t[:, k + 1] - t[:, k])[sel] / 100.0
    if not len(d):
        continue
    print("%-9s mean %7.2f us  p50 %7.2f  p90 %7.2f  max %7.2f" % (n, d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
life = (end - start)[ok]
print("lifetime  mean %.1f us; waves alive on average %.0f" % (life.mean(), life.sum() / end[ok].max()))
order = np.argsort(start)
print("start order vs index: first 16 started:", order[:16].tolist())
print("polls of the exit in front: mean %.1f max %d" % (t[:, 12].mean(), t[:, 12].max()))
for lo in range(0, nseg, max(1, nseg // 12)):
    print("  seg %5d start %7.1f end %7.1f" % (lo, start[lo], end[lo]))

print('candidates per segment mean %.1f max %d, blocks %.1f' % (t[:, 15].mean(), t[:, 15].max(), t[:, 14].mean()))

k = t[:, 13] != 0
print("segments whose exit was published early: %d of %d; longest run without: %d" % (int(k.sum()), nseg, max((len(r) for r in "".join("k" if v else "n" for v in k).split("k")), default=0)))
w = (t[:, 5] - t[:, 4]) / 100.0
slow = np.argsort(w)[-8:]
print("longest exit waits:", [(int(i), round(float(w[i]), 1), "early" if k[i] else "late", "front early" if i and k[i - 1] else "front late") for i in slow])
pub = np.where(k, t[:, 4], t[:, 6])      # when the exit became visible (about): after the doubling, or after the entry was known
lag = (pub[:-1] - t[1:, 4]) / 100.0     # how long after segment s+1 was ready for it
print("exit of the segment in front later than own readiness: mean %.2f us, p90 %.2f, max %.2f" % (np.maximum(lag, 0).mean(), np.percentile(np.maximum(lag, 0), 90), lag.max()))

nw = (nblocks + 63) // 64
d3 = t[:min(nw, nseg)]
t3 = d3[:, 7].min()
print("block decoder: %d waves traced, span %.1f us; start p50 %.1f p75 %.1f p100 %.1f" % (len(d3), (d3[:, 11].max() - t3) / 100.0, *[float(x) for x in np.percentile((d3[:, 7] - t3) / 100.0, [50, 75, 100])]))
for nm, a, b in (("where + tile", 7, 8), ("bytes to LDS", 8, 9), ("parse", 9, 10), ("rows out", 10, 11)):
    d = (d3[:, b] - d3[:, a]) / 100.0
    print("  %-13s mean %6.2f us p50 %6.2f p90 %6.2f max %6.2f" % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
life = (d3[:, 11] - d3[:, 7]) / 100.0
print("  lifetime mean %.1f us; waves alive on average %.0f" % (life.mean(), life.sum() / ((d3[:, 11].max() - t3) / 100.0)))
