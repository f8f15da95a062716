import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files: continue
    acc = collections.defaultdict(list); dur = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "gemm_f32_mfma" not in k and "gemm_split_kernel" not in k: continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[row["Counter_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print(os.path.basename(d), {c: round(sorted(v)[len(v) // 2]) for c, v in acc.items()},
          "us", round(sorted(next(iter(dur.values())))[len(next(iter(dur.values()))) // 2], 1) if dur else None)
