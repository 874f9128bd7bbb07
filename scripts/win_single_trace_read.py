import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "collect_kernel" in r["Kernel_Name"] or "window_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
SHAPES = [(mode, nq, th, ori) for mode in (0, 2) for nq in (10, 100, 1000) for th in (3.0, 7.0) for ori in (0, 1)]
per = len(rows) // len(SHAPES)
for i, s in enumerate(SHAPES):
    seg = rows[i * per:(i + 1) * per]
    d = collections.defaultdict(list)
    for r in seg:
        d["collect" if "collect_kernel" in r["Kernel_Name"] else "claims"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    gaps = [int(seg[k + 1]["Start_Timestamp"]) - int(seg[k]["End_Timestamp"]) for k in range(0, len(seg) - 1, 2)]
    med = lambda v: sorted(v)[len(v) // 2] / 1e3 if v else float("nan")
    print("mode %d nq %4d th %.0f ori %d: collect %.1f us, claims %.1f us, gap between them %.1f us" % (*s, med(d["collect"]), med(d["claims"]), med(gaps)))
