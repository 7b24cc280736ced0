#!/usr/bin/env python
"""print the top kernels of a rocprofv3 (rocpd sqlite) kernel trace: name, calls, avg us, % of GPU time"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/*/*.db')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
cur = sqlite3.connect(db).cursor()
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'%':>6s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit ?", (n,)):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:70]:70s} {calls:6d} {avg:10.1f} {pct:6.1f}")
