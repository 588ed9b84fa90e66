"""Print the launches of one step of the shuffled-epoch section of bench.py (the step that follows a k_gather_rows) from a
rocprofv3 --kernel-trace csv: python tools/epoch_timeline.py <dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.search(r"(k_\w+)", r["Kernel_Name"]).group(1) if re.search(r"(k_\w+)", r["Kernel_Name"]) else r["Kernel_Name"][:24] for r in rows]
g = [i for i, n in enumerate(names) if n == "k_gather_rows"]
heads = [i for i, n in enumerate(names) if n in ("k_presplit", "k_make_xbits") and i > 0]
# a step head in the last third of the gathers (the pipelined epoch)
k = g[len(g) * 5 // 6]
a = max(h for h in heads if h <= k)
b = min(h for h in heads if h > k)
b2 = min(h for h in heads if h > b)
t0 = int(rows[a]["Start_Timestamp"])
for i in range(a, b2):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')}  {names[i]}")
