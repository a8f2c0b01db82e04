"""Per-workgroup times of xs_kernel from the XSP lines a timing build prints (make AIDS=1,
CUGRAPH_AMD_XS_PROFILE=1): total, time inside item boundaries, items, items of other slices."""
import collections
import statistics as st
import sys

rows = [[int(x) for x in l.split()[1:]] for l in open(sys.argv[1]) if l.startswith("XSP ")]
t0 = min(r[8] for r in rows)
ends = [(r[8] - t0 + r[2]) * 0.01 for r in rows]
print("start spread us", (max(r[8] for r in rows) - t0) * 0.01, "end min/mean/max us", min(ends), sum(ends) / len(ends), max(ends))
print(len(rows), "workgroups")
for name, i, sc in (("total", 2, 0.01), ("boundaries", 3, 0.01), ("items", 4, 1), ("stolen", 5, 1), ("barrier1", 6, 0.01), ("writeback", 7, 0.01)):  # 100 MHz ticks -> us
    c = [r[i] for r in rows]
    print(f"{name:10s} min {min(c)*sc:10.1f} mean {st.mean(c)*sc:10.1f} max {max(c)*sc:10.1f}")
by = collections.defaultdict(list)
for r in rows:
    by[r[1]].append(r[2] * 0.01)
for k in sorted(by):
    print("xcd", k, len(by[k]), "wgs, total us min/mean/max", round(min(by[k]), 1), round(st.mean(by[k]), 1), round(max(by[k]), 1))
