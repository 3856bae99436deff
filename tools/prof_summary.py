import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'].split('(')[0][-28:]:30s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}")
