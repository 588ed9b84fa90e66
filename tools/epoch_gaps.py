"""From a rocprofv3 results .db of tools/rows_epoch_time.py (TRACE=1): step pitch (k_presplit to k_presplit), what runs that is
not a step kernel, and the idle time of the main queue per step: python tools/epoch_gaps.py <file.db>"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,queue_id from kernels order by start"))
nm = lambda n: (re.search(r"(k_\w+)", n).group(1) if re.search(r"(k_\w+)", n) else n[:40])
names = [nm(r[0]) for r in rows]
heads = [i for i, n in enumerate(names) if n == "k_presplit"]
heads = heads[len(heads) // 3:]
mainq = rows[heads[0]][3]
for a, b in zip(heads[:-1], heads[1:]):
    t0 = rows[a][1]
    pitch = (rows[b][1] - t0) / 1e3
    busy, idle, last = 0.0, 0.0, t0
    extra = []
    for i in range(a, b):
        n, s, e, q = names[i], rows[i][1], rows[i][2], rows[i][3]
        if q == mainq:
            if s > last:
                idle += (s - last) / 1e3
            last = max(last, e)
        if not n.startswith("k_"):
            extra.append("%s@%.0f+%.1f q%d" % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, q))
    idle += max(0, rows[b][1] - last) / 1e3
    print("pitch %7.1f us  main-queue idle %5.1f  %s" % (pitch, idle, "; ".join(extra)))
