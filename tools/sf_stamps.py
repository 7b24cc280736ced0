import sys
rows=[list(map(int,l.split())) for l in open(sys.argv[1]) if l.strip()]
t0=min(r[0] for r in rows)
names=["start->weights","sdf chain","normal(J)+FE","geo stage","geo chain+dec part","decoder+lds","tps+scan"]
n=len(rows)
print("blocks",n,"kernel span us", (max(r[7] for r in rows)-t0)/100.0)
print("start spread us", (max(r[0] for r in rows)-t0)/100.0)
for q in range(7):
    print("%-22s %.2f us" % (names[q], sum(r[q+1]-r[q] for r in rows)/n/100.0))
