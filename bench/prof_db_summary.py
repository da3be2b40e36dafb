"""Per-kernel totals of a rocprofv3 --kernel-trace run that wrote the rocpd database (``*_results.db``):
``python bench/prof_db_summary.py gpurun_out/prof/x_results.db <steps> [out.csv]`` (steps = warm-up + timed)."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "MsPerStep", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, mn, mx in rows:
            w.writerow([n, c, round(c / steps, 2), t, round(t / steps / 1e6, 4), round(100.0 * t / tot, 3), mn, mx])
for n, c, t, mn, mx in rows[:28]:
    print(f"{n[:105]:105s} {c / steps:6.1f}/step {t / steps / 1e6:8.2f} ms/step {100 * t / tot:5.1f}%")
print(f"kernel time per step: {tot / steps / 1e6:.2f} ms")
