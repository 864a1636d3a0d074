"""Timeline of ONE Match from a rocprofv3 kernel trace (rocpd .db): every kernel of the last complete Match in launch order with its
start offset, duration and the gap to the next kernel.   python tools/match_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {ks} s on k.kernel_id = s.id order by k.start"))
idx = [i for i, r in enumerate(rows) if "k_gray_census" in r[2]]
seg = rows[idx[-2]:idx[-1]]
t0 = seg[0][0]
tot_k = tot_g = 0.0
for i, (s, e, n) in enumerate(seg):
    gap = (seg[i + 1][0] - e) / 1000.0 if i + 1 < len(seg) else 0.0
    tot_k += (e - s) / 1000.0
    tot_g += max(gap, 0.0)
    name = n.split("(")[0].replace("void ", "")[:60]
    print("%8.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, gap, name))
print("kernels %.1f us, gaps %.1f us, span %.1f us" % (tot_k, tot_g, (seg[-1][1] - t0) / 1000.0))
