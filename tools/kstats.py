"""print the top rows of a rocprofv3 kernel_stats.csv:  python tools/kstats.py FILE [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:72]
    print(f"{name:74s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['AverageNs']) / 1e3:8.1f} us "
          f"{float(r['Percentage']):6.2f}%")
