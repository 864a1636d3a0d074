"""Per-kernel picture of the region-voting chain from a rocprofv3 kernel trace (rocpd .db):
  python tools/irv_trace_summary.py <results.db> [kernel-name-substring]
prints, for the LAST Match in the trace, duration and gap (end -> next start) statistics of the chain's kernels."""
import sys, sqlite3, numpy as np
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_irv_u"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {ks} s on k.kernel_id = s.id order by k.start"))
idx = [i for i, r in enumerate(rows) if "k_gray_census" in r[2]]
seg = rows[idx[-2]:idx[-1]] if len(idx) >= 2 else rows
ch = [(i, r) for i, r in enumerate(seg) if pat in r[2]]
dur = np.array([(r[1] - r[0]) / 1000.0 for _, r in ch])
gap = np.array([(seg[i + 1][0] - r[1]) / 1000.0 for i, r in ch if i + 1 < len(seg)])
print(f"{pat}: {len(ch)} kernels in the last Match; duration us: sum {dur.sum():.1f} mean {dur.mean():.2f} median {np.median(dur):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.1f}")
print(f"   gap to the next kernel us: sum {gap.sum():.1f} mean {gap.mean():.2f} median {np.median(gap):.2f}")
edges = [0, 3, 4, 5, 6, 8, 10, 15, 20, 30, 50, 100, 1e9]
h, _ = np.histogram(dur, edges)
print("   duration histogram:", ", ".join(f"<{int(e) if e < 1e9 else 'inf'}us: {n} ({dur[(dur >= lo) & (dur < e)].sum():.0f} us)" for lo, e, n in zip(edges[:-1], edges[1:], h) if n))
print("   first 60 durations:", " ".join(f"{d:.1f}" for d in dur[:60]))
