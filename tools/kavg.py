"""Average duration per kernel name from a rocprofv3 results .db: python tools/kavg.py <file.db> [substring ...]"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels order by start"))
d = collections.defaultdict(list)
for n, s, e in rows[len(rows) // 4:]:
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    d[(m.group(1) + (m.group(2) or ""))[:60] if m else n[:40]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) < 3 or any(a in k for a in sys.argv[2:]):
        print(f"{k:62s} n={len(v):4d} avg={sum(v)/len(v):7.1f} us")
