"""Epilogue phases from the in-kernel timelines (tools/ubench/conv_lab_tl): per layer and traced workgroup, the mean cycles of
item period | row 0: arithmetic+staging, stores | row 1: the same | pool | switch barrier | per-chunk (rest / chunks)."""
import glob, re, sys
for path in sorted(glob.glob(sys.argv[1] + '/tl_*.txt')):
    H, Ci, Co = map(int, re.search(r'H(\d+)_Ci(\d+)_Co(\d+)', path).groups())
    for ln in open(path):
        tag, rest = ln.split(':', 1)
        if not tag.endswith(('wave0', 'wave4')): continue
        ev = [tuple(map(int, t.split(':'))) for t in rest.split()]
        items, cur = [], None
        for k, t in ev:
            if k == 2: cur = {'t2': t, 'sub': []}
            elif k in (5, 6) and cur is not None: cur['sub'].append(t)
            elif k == 3 and cur is not None: cur['t3'] = t
            elif k == 4 and cur is not None: cur['t4'] = t; items.append(cur); cur = None
        full = [it for it in items if len(it['sub']) == 4 and 't3' in it]
        if len(full) < 2: continue
        n = len(full)
        ph = [sum(it['sub'][0] - it['t2'] for it in full) / n, sum(it['sub'][1] - it['sub'][0] for it in full) / n,
              sum(it['sub'][2] - it['sub'][1] for it in full) / n, sum(it['sub'][3] - it['sub'][2] for it in full) / n,
              sum(it['t3'] - it['sub'][3] for it in full) / n, sum(it['t4'] - it['t3'] for it in full) / n]
        per = (full[-1]['t4'] - full[0]['t4']) / (n - 1)
        epi = sum(ph)
        print("H%-3d Ci%-4d Co%-4d %-10s items %2d period %7.0f (ideal %7.0f = %2.0f%%) | epi %6.0f = row0 %5.0f +%5.0f  row1 %5.0f +%5.0f  pool %5.0f  barrier %5.0f | per chunk %5.0f"
              % (H, Ci, Co, tag, n, per, Ci // 16 * 6912, 100 * (Ci // 16 * 6912) / per, epi, *ph, (per - epi) / (Ci // 16)))
