"""Durations of the one-launch 64-channel block kernel in a rocprofv3 kernel trace, split by where the launch stands: inside a
forward pass (runs of five and two launches between other kernels) or in the back-to-back loop of bench.py's probe.
usage: block_context.py <trace dir> <out.csv>"""
import csv, glob, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
runs, cur = [], []
for r in rows:
    if "nb1d_block64s" in r["Kernel_Name"]:
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    elif cur:
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
by = {}
for r in runs:
    by.setdefault(len(r), []).extend(r)
with open(sys.argv[2], "w") as out:
    out.write("consecutive_launches,launches,mean_us,median_us,min_us,max_us\n")
    for n in sorted(by):
        x = by[n]
        out.write(f"{n},{len(x)},{statistics.mean(x):.2f},{statistics.median(x):.2f},{min(x):.2f},{max(x):.2f}\n")
print(open(sys.argv[2]).read())
