"""kernel timeline of one captured iteration of a stage loop (tools/time_loops.py's scene):
     rocprofv3 --kernel-trace -d gpurun_out/ltl -- python tools/loop_timeline.py run BA ; python tools/loop_timeline.py show gpurun_out/ltl"""
import glob, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def run(which):
    import torch
    import time_loops as TL                      # (its module body times the loops once: acceptable for a profile run)


def show(path):
    db = glob.glob(path + '/*/*.db')[0]
    cur = sqlite3.connect(db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tables if 'kernel_dispatch' in t and not t.startswith('rocpd_info')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    marks = [i for i, r in enumerate(rows) if 'adam_pair' in r[3]]
    a, b = marks[-3] + 1, marks[-2] + 1          # one iteration of the LAST loop timed (BALoop captured)
    t0 = rows[a][0]
    agg = {}
    for st, en, q, name in rows[a:b]:
        name = name.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at::")
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q}  {name[:80]}")
        key = name[:40]
        agg[key] = agg.get(key, [0, 0.0]); agg[key][0] += 1; agg[key][1] += (en - st) / 1e3
    print("iteration span us:", (rows[b][0] - t0) / 1e3, " launches:", b - a)
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"   {t:8.1f} us  x{n:3d}  {k}")


if __name__ == "__main__":
    run(sys.argv[2] if len(sys.argv) > 2 else "BA") if sys.argv[1] == "run" else show(sys.argv[2])
