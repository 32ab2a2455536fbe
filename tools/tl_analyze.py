"""Timeline of the persistent conv kernel (tools/ubench/conv_lab.hip built with -DLM_H3_TIMELINE): per wave, the shader-clock
intervals between marks.  Marks: 0 chunk barrier reached, 1 barrier passed, 2 main loop of the item done, 5 staging writes of a
row pass done, 6 its stores issued, 3 epilogue done, 4 item-switch barrier passed.    python tools/tl_analyze.py FILE [wg]"""
import sys
path = sys.argv[1]; wg_sel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = {}
for ln in open(path):
    head, _, rest = ln.partition(":")
    wg, wv = head.split(); wg = int(wg[2:]); wv = int(wv[4:])
    ev = [tuple(map(int, t.split(":"))) for t in rest.split()]
    waves[(wg, wv)] = ev
t0 = min(ev[0][1] for (wg, wv), ev in waves.items() if wg == wg_sel and ev)
print("absolute marks (cycles since the first mark of the workgroup), waves side by side; partner waves share a SIMD: (0,4) (1,5) (2,6) (3,7)")
n = max(len(ev) for (wg, wv), ev in waves.items() if wg == wg_sel)
for k in range(min(n, int(sys.argv[3]) if len(sys.argv) > 3 else 64)):
    row = []
    for wv in range(8):
        ev = waves[(wg_sel, wv)]
        row.append("%d:%7d" % (ev[k][0], ev[k][1] - t0) if k < len(ev) else " " * 9)
    print("%3d  " % k + "  ".join(row))
# per-wave sums by segment kind
names = {(1, 0): "run(chunk)", (4, 0): "run(chunk0)", (0, 1): "barrier", (1, 2): "tap8+", (2, 5): "epi valu+stage", (5, 6): "epi stores", (6, 5): "epi valu+stage",
         (6, 3): "epi pool/tail", (3, 4): "switch barrier", (2, 3): "epi(head)"}
print()
for wv in range(8):
    ev = waves[(wg_sel, wv)]
    tot = {}
    for a, b in zip(ev, ev[1:]):
        key = names.get((a[0], b[0]), "%d->%d" % (a[0], b[0]))
        tot[key] = tot.get(key, 0) + (b[1] - a[1])
    all_ = ev[-1][1] - ev[0][1]
    print("wave %d: total %7d | " % (wv, all_) + "  ".join("%s %5.1f%%" % (k, 100.0 * v / all_) for k, v in sorted(tot.items())))
