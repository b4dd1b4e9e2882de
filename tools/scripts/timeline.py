"""Merge rocprofv3's kernel and memory-copy traces (csv) into one timeline:  python tools/scripts/timeline.py DIR [first_event [count]]
Prints start / end / duration in ms relative to the first printed event; kernels with their queue, copies with their direction."""
import csv
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
ev = []
for r in csv.DictReader(open(d + "/t_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %-42s grid %s" % (r["Queue_Id"], r["Kernel_Name"][:42], r["Grid_Size_X"])))
try:
    for r in csv.DictReader(open(d + "/t_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s" % r["Direction"].replace("MEMORY_COPY_", "")))
except FileNotFoundError:
    pass
ev.sort()
if first < 0:
    first = max(0, len(ev) + first)
base = ev[first][0]
print("%d events; from #%d" % (len(ev), first))
for i, (s, e, n) in enumerate(ev[first:first + count]):
    print("%5d %9.3f %9.3f %8.3f  %s" % (first + i, (s - base) / 1e6, (e - base) / 1e6, (e - s) / 1e6, n))
