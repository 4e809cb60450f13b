import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
rows=list(c.execute(q)); tot=sum(r[2] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 30]: print(f"{r[0][:60]:60s} {r[1]:6d} {r[2]:9.2f} avg {r[3]:8.3f} min {r[4]:8.3f} {100*r[2]/tot:5.1f}%")
