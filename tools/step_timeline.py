"""Print one train step's kernel timeline from a rocprofv3 --kernel-trace csv (start offset, duration, stream) to see gaps
and overlaps: python tools/step_timeline.py <dir> [step index]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.search(r"(k_\w+)", r["Kernel_Name"]).group(1) if re.search(r"(k_\w+)", r["Kernel_Name"]) else r["Kernel_Name"][:20] for r in rows]
# a step starts at its head launch (k_make_xbits, or the k_presplit launch that does its work too) behind the previous
# step's reduction
starts = [i for i, n in enumerate(names) if n in ("k_make_xbits", "k_presplit") and i > 0 and names[i - 1] == "k_reduce"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
for i in range(a, b):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:7.1f}  q{q}  gap {gap/1e3:6.1f}  {names[i]}")
print(f"step {(int(rows[b]['Start_Timestamp']) - t0)/1e3:.1f} us")
