"""Per-kernel average durations from a rocprofv3 --kernel-trace --stats run (rocpd sqlite): usage show_kernel_stats.py <dir> [substring]"""
import glob
import sqlite3
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
for dbp in sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)):
    db = sqlite3.connect(dbp)
    print(dbp)
    for name, calls, tot, avg in db.execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc"):
        if pat in name:
            print("  %-90s calls %5d avg %9.1f us" % (name[:90], calls, avg))
