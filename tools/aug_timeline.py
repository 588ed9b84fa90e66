"""One augmenter forward's launches (start offset, duration, grid, kernel) from a rocprofv3 --kernel-trace csv of tools/aug_time.py:
python tools/aug_timeline.py <dir>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"].replace("mmvae::", "").replace("void ", "")) for r in rows]
idx = [i for i, nm in enumerate(names) if "lat" in nm]
a, b = idx[-2] + 1, idx[-1] + 1          # the launches of one forward lie between two latent kernels; shift to its first GEMM
per = b - a
first = [i for i in range(a, b) if "gemm" in names[i]]
start = idx[-2] - (per - 1 - (first[-1] - a)) if False else a
t0 = int(rows[start]["Start_Timestamp"])
tot = 0.0
for i in range(start, start + per):
    r = rows[i]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    g = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    print("%9.1f +%7.1f  grid %-16s wg %-4s %s" % (s, d, g, r.get("Workgroup_Size_X", "?"), names[i][:80]))
print("sum of durations %.1f us over %d launches" % (tot, per))
