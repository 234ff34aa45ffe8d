import csv
def load(f):
    d={}
    for r in list(csv.reader(open(f)))[2:]:
        d[r[0]]=(float(r[1]),float(r[2]),float(r[3]))
    return d
g=load("gpurun_out/g4_graph_window.csv"); e=load("gpurun_out/g4_eager_window.csv")
diff=[]
for k in set(g)|set(e):
    a=g.get(k,(0,0,0)); b=e.get(k,(0,0,0))
    diff.append((a[2]-b[2],k,a,b))
diff.sort(reverse=True)
print("kernels whose ms/step differ most (graph - eager):")
for d,k,a,b in diff[:12]: print("%+.3f ms  graph %s  eager %s  %s"%(d,a,b,k[:90]))
print("...")
for d,k,a,b in diff[-6:]: print("%+.3f ms  graph %s  eager %s  %s"%(d,a,b,k[:90]))
print("sum graph %.3f eager %.3f"%(sum(v[2] for v in g.values()), sum(v[2] for v in e.values())))
