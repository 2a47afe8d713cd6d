"""Per-step table from a rocprofv3 kernel_stats.csv of `bench.py --steps 1 --warmup 1` (= 100 denoising steps + 2 setups)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / steps / 1e6:.2f} ms; launches per step: {sum(int(r['Calls']) for r in rows) / steps:.0f}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{name[:70]:70s} calls/step {int(r['Calls']) / steps:7.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / steps / 1e6:6.2f} ({100 * float(r['TotalDurationNs']) / tot:4.1f}%)")
