#!/usr/bin/env python3
"""Prints the top_kernels table of a rocprofv3 sqlite result (gpurun_out/prof/*.db)."""
import glob
import sqlite3
import sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*.db"))[-1]
db = sqlite3.connect(path)
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for n, c, t, a, p in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("| `%s` | %d | %.1f | %.1f | %.2f |" % (n.split("(")[0][:60], c, t, a, p))
