"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide coalesced reads, so the read side is doubled."""
import collections, csv, glob, json, sys

def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    f, w = fetch[k], write.get(k, [0.0])
    out[k[:90]] = dict(launches=len(f), fetch_size_raw_per_launch=sum(f) / len(f), write_size_raw_per_launch=sum(w) / len(w),
                       hbm_read_bytes_per_launch_corrected=2 * 1024 * sum(f) / len(f), hbm_write_bytes_per_launch=1024 * sum(w) / len(w))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out)} kernels -> {sys.argv[3]}")
