"""Per-launch averages of rocprofv3 PMC counters for kernels whose name contains a filter.
usage: python tools/pmc_kernel.py <filter> <dir> [<dir> ...]"""
import collections, csv, glob, sys
flt, dirs = sys.argv[1], sys.argv[2:]
tot, n = collections.defaultdict(float), collections.Counter()
for d in dirs:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if flt in r["Kernel_Name"]:
                k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"])
                tot[k] += float(r["Counter_Value"]); n[k] += 1
for (name, c), v in sorted(tot.items()):
    print(f"{name[:60]:60s} {c:28s} {v / n[(name, c)]:16.1f}  (n={n[(name, c)]})")
