"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (directory argument)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)      # (gpurun merges calls: take the newest run)
rows = list(csv.DictReader(open(f)))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print("total ms", sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / div)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print("%-72s calls %5s tot %8.2f ms avg %8.1f us %6s%%" % (r['Name'][:72], r['Calls'], float(r['TotalDurationNs']) / 1e6 / div, float(r['AverageNs']) / 1e3, r['Percentage'][:5]))
