import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find last lstm_bwd_flow2 launch and print the window from previous clip_adam to the next clip_adam
idx = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  q%s  grid %-8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"].replace("amdspeech::", "").replace("void ", "")[:90]))
