import sys, collections
rows = [l.split()[1:] for l in open(sys.argv[1]) if l.startswith("XSP ")]
rows = [[int(x) for x in r] for r in rows]
t0=min(r[8] for r in rows)
ends=[(r[8]-t0+r[2])*0.01 for r in rows]
print("start spread us", (max(r[8] for r in rows)-t0)*0.01, "end min/mean/max us", min(ends), sum(ends)/len(ends), max(ends))
print(len(rows), "workgroups")
import statistics as st
def col(i): return [r[i] for r in rows]
for name, i in (("total", 2), ("setup", 3), ("pass", 4), ("writeback", 5), ("units", 6), ("stolen", 7)):
    c = col(i)
    sc = 0.01 if i < 6 else 1  # 100 MHz ticks -> us
    print(f"{name:10s} min {min(c)*sc:10.1f} mean {st.mean(c)*sc:10.1f} max {max(c)*sc:10.1f}")
by = collections.defaultdict(list)
for r in rows: by[r[1]].append(r[2] * 0.01)
for k in sorted(by): print("xcd", k, len(by[k]), "wgs, total us min/mean/max", round(min(by[k]),1), round(st.mean(by[k]),1), round(max(by[k]),1))
