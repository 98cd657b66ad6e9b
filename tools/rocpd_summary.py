"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / average / share (what --stats prints)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f'# rocprofv3 --kernel-trace summary of {sys.argv[1]}  (total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches)')
print(f'{"kernel":100s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}')
for name, n, s, a, mn, mx in rows:
    short = re.sub(r'\(anonymous namespace\)::', '', name)
    short = re.sub(r'void ', '', short)
    print(f'{short[:100]:100s} {n:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}')
