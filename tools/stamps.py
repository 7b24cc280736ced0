import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
t0 = min(r[1] for r in rows); tend = max(r[4] for r in rows)
print("blocks", len(rows), "span ticks", tend - t0, "(100 MHz wall clock: 10 ns/tick)")
by = collections.defaultdict(list)
for b, a, z, i, e, l, sm in rows: by[l].append((a - t0, z - a, i - z, e - i))
for l in sorted(by):
    v = by[l]; n = len(v)
    print("level %2d blocks %4d  start %6.0f..%6.0f  zero+bound %5.1f  items %6.1f  flush %5.1f" % (l, n, min(x[0] for x in v), max(x[0] for x in v), sum(x[1] for x in v)/n, sum(x[2] for x in v)/n, sum(x[3] for x in v)/n))
