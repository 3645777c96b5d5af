"""Which aten element-wise functors run in the steady-state window of a rocpd trace (full template names)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 300
t1 = db.execute("select max(end) from kernels").fetchone()[0]
rows = db.execute("select name, count(*), sum(duration)/1000.0 from kernels where start >= ? and name like '%elementwise%' "
                  "group by name order by 3 desc", (t1 - int(last_ms * 1e6),)).fetchall()
for name, calls, us in rows[:25]:
    m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda\w*|lambda[^,>]*)", name[40:])
    print(f"{calls:6d} {us:9.1f}us  {name[:400]}")
