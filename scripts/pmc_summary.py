"""Per-kernel mean of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) -> profiles/*.csv and
profiles/traffic.json.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and
gfx950 tallies the 128-byte read requests of wide coalesced streams at 64 bytes."""
import collections, csv, glob, json, sys

def means(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

fetch, write, out = means(sys.argv[1]), means(sys.argv[2]), sys.argv[3]
rows, traffic = [], {}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    fs, n = fetch[k]
    ws = write.get(k, (0.0, 0))[0]
    hbm = (2 * fs + ws) * 1024
    rows.append((k, n, fs, ws, hbm))
    traffic[k] = hbm
with open(out + ".csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
    w.writerows(rows)
json.dump(traffic, open(out + ".json", "w"), indent=1)
