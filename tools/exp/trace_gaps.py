import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 5e6: print("LONG kernel %.2f ms" % ((e - s) / 1e6), r["Kernel_Name"][:90])
    if prev_end is not None and s - prev_end > 10e6:
        print("GAP %.2f ms before" % ((s - prev_end) / 1e6), r["Kernel_Name"][:70], "| after", rows[i - 1]["Kernel_Name"][:70])
    prev_end = max(prev_end or 0, e)
print(len(rows), "kernels")
