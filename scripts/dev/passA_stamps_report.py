import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
names = ["entry", "after sync1", "main loop done", "sums done", "sync2", "Hf+sync3", "V/g done", "transform done", "sync4", "inverse done", "obs compute done", "obs loads arrived"]
order = [0, 1, 11, 10, 2, 3, 4, 5, 6, 7, 8, 9]
print("frames stamped", len(a), " entry spread (cycles): min 0 median %.0f max %.0f" % (np.median(a[:, 0] - t0), (a[:, 0] - t0).max()))
prev = None
for i in order:
    rel = a[:, i] - a[:, 0]
    print("%-20s median %8.0f   p90 %8.0f  (cycles since the workgroup's entry)%s" % (names[i], np.median(rel), np.percentile(rel, 90), "" if prev is None else "   +%.0f" % (np.median(rel) - prev)))
    prev = np.median(rel)
print("last stamp - first entry: %.0f cycles" % (a[:, 9].max() - t0))
