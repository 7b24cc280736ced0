import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/*/*.db')[0]
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if 'kernel_dispatch' in t and not t.startswith('rocpd_info')][0]
ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if 'sphere_trace' in r[3]]
a, b = marks[-3], marks[-2]
t0 = rows[a][0]
for st, en, q, name in rows[a:b]:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q}  {name[:100]}")
print("step span us:", (rows[b][0] - t0) / 1e3)
