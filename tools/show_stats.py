"""Prints rows of a rocprofv3 kernel_stats.csv whose kernel name contains a substring."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for row in csv.reader(open(f)):
    if len(sys.argv) < 3 or sys.argv[2] in row[0]:
        print(row[0][:60].ljust(60), row[1], row[3])
