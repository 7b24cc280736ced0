#!/usr/bin/env python
"""per kernel of a rocprofv3 --kernel-trace --pmc run (rocpd sqlite): mean duration next to the mean of every counter (summed
over the counter's instances) -- to correlate a kernel's run-to-run modes with a counter:  python tools/pmc_vs_time.py <dir> [substr]"""
import glob, sqlite3, sys, collections
db = glob.glob(sys.argv[1] + '/*/*.db')[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db); cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tables if t.startswith('rocpd_pmc_event')][0]
kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
pi = [t for t in tables if t.startswith('rocpd_info_pmc')][0]
dur = collections.defaultdict(list)
for name, s, e in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"):
    dur[name].append((e - s) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for name, c, ev, v in cur.execute(f"select s.kernel_name, i.name, e.event_id, e.value from {pmc} e join {pi} i on e.pmc_id = i.id "
                                  f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"):
    ctr[name][c] += v
    cnt[name].add(ev)
for name in sorted(dur, key=lambda n: -sum(dur[n])):
    if sub not in name or "at::native" in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    n = max(1, len(cnt[name]))
    print(f"{short:48s} us={sum(dur[name]) / len(dur[name]):8.1f} " + " ".join(f"{c}={v / n:.0f}" for c, v in sorted(ctr[name].items())))
