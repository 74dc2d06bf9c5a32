import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if not t.startswith('rocpd_') or 'region' in t or 'memory' in t][:60])
for t in ("regions", "memory_copies", "memory_copy"):
    if t in tabs:
        cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
        print(t, cols)
if "regions" in tabs:
    for r in db.execute("select name, count(*) from regions group by name order by 2 desc limit 25"):
        print(r)
if "memory_copies" in tabs:
    cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
    key = "name" if "name" in cols else cols[0]
    for r in db.execute(f"select {key}, count(*), avg(size) from memory_copies group by {key} order by 2 desc limit 10"):
        print(r)
