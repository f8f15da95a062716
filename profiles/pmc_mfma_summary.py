"""Per kernel: matrix-pipe busy fraction and effective clock from separate rocprofv3 --pmc passes (profiles/pmc_mfma.sh).
  clock  = GRBM_GUI_ACTIVE / 8 / wall           (the counter is summed over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back)
  busy   = SQ_VALU_MFMA_BUSY_CYCLES / (n_simd * clock * wall), n_simd = 256 CUs x 4     (cycles the matrix pipe of a SIMD was busy)
  rate   = busy x clock x 1024 flop / cycle / SIMD x n_simd  = the bf16 MFMA rate the kernel ran at (32x32x16: 32 768 flop in 32 cycles)
usage: python3 profiles/pmc_mfma_summary.py <dir with pmc_<COUNTER>/ subdirectories>"""
import collections, csv, glob, os, sys
root = sys.argv[1]
N_SIMD = 1024
acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> values; "wall" from the trace
for cdir in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    counter = os.path.basename(cdir)[4:]
    wall = {}
    for f in glob.glob(cdir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            wall[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for f in glob.glob(cdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            acc[k][counter].append(float(r["Counter_Value"]))
            w = wall.get(r["Dispatch_Id"])
            if w is not None:
                acc[k]["wall_" + counter].append(w[1])
print(f"{'kernel':70s} {'n':>4s} {'wall us':>8s} {'clock GHz':>9s} {'MFMA busy':>9s} {'bf16 TF at clock':>16s} {'SQ busy':>8s}")
rows = []
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or not c["SQ_VALU_MFMA_BUSY_CYCLES"] or max(c["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
        continue
    mean = lambda v: sum(v) / max(len(v), 1)
    wall = mean(c.get("wall_GRBM_GUI_ACTIVE", c.get("wall_SQ_VALU_MFMA_BUSY_CYCLES", [0])))
    clock = mean(c.get("GRBM_GUI_ACTIVE", [0])) / 8 / max(wall, 1e-12)
    wall_m = mean(c.get("wall_SQ_VALU_MFMA_BUSY_CYCLES", [wall]))
    busy = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (N_SIMD * max(clock, 1) * wall_m)
    sqb = mean(c.get("SQ_BUSY_CYCLES", [0])) / max(clock * mean(c.get("wall_SQ_BUSY_CYCLES", [wall])), 1e-12)
    rows.append((mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), k, len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), wall_m, clock, busy, sqb))
for _, k, n, wall, clock, busy, sqb in sorted(rows, reverse=True):
    print(f"{k[:70]:70s} {n:4d} {wall * 1e6:8.1f} {clock / 1e9:9.3f} {busy:9.3f} {busy * clock * 1024 * N_SIMD / 1e12:16.0f} {sqb:8.2f}")
print("(counter passes serialise every dispatch and run at lower clocks than the un-profiled loop: MI355X_MICROARCH.md, DVFS give-back item 2;")
print(" SQ busy = SQ_BUSY_CYCLES / (clock x wall): summed over shader engines, so it reads > 1)")
