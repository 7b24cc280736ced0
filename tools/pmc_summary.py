#!/usr/bin/env python
"""per-kernel average of a PMC counter from a rocprofv3 (rocpd sqlite) --pmc run: kernel, dispatches, mean value"""
import glob, sqlite3, sys, collections
db = glob.glob(sys.argv[1] + '/*/*.db')[0]
con = sqlite3.connect(db); cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tables if t.startswith('rocpd_pmc_event')][0]
kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
pi = [t for t in tables if t.startswith('rocpd_info_pmc')][0]
cols = lambda t: [r[1] for r in cur.execute(f"pragma table_info({t})")]
q = f"""select s.kernel_name, i.name, count(*), avg(e.value) from {pmc} e
        join {pi} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
        group by s.kernel_name, i.name order by avg(e.value) desc"""
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("schema:", {t: cols(t) for t in (pmc, kd, pi)}); raise
for name, ctr, n, v in rows:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "at::native" in name or "rocclr" in name: continue
    print(f"{name[:60]:60s} {ctr:14s} n={n:4d} mean={v:14.1f}")
