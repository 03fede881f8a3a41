"""Per-launch view of the LAST replay in a rocprofv3 kernel trace (kernel_trace.csv): the trace ends with N identical graph replays;
this prints, for each launch position of one replay, the kernel, its grid and its duration averaged over the last R replays, plus
the gap to the previous kernel's end.  usage: python tools/trace_last_pass.py <kernel_trace.csv> <launches_per_replay, 0 = detect> [replays]"""
import csv
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if n <= 0:                       # auto-detect the replay length: smallest period of the kernel-name sequence at the end
        names = [r["Kernel_Name"] for r in rows]
        for cand in range(1, len(names) // (reps + 1)):
            if all(names[-1 - j] == names[-1 - j - cand] for j in range(cand * reps)):
                n = cand
                break
        print(f"launches per replay: {n}")
    # the process may end with a few set-up / read-back kernels behind the last replay: drop up to 8 trailing dispatches until the
    # last `reps` windows of n launches repeat
    for skip in range(9):
        cand = rows[:len(rows) - skip] if skip else rows
        tail = cand[-n * reps:]
        if len(tail) == n * reps and all(tail[k * n + i]["Kernel_Name"] == tail[i]["Kernel_Name"] for k in range(reps) for i in range(n)):
            break
    tot = 0.0
    for i in range(n):
        rs = [tail[k * n + i] for k in range(reps)]
        names = {r["Kernel_Name"] for r in rs}
        if len(names) != 1:
            print(f"position {i}: replays disagree ({len(names)} kernels) -- wrong launches_per_replay?")
            return
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / reps / 1e3
        gap = 0.0
        if i:
            gap = sum(int(tail[k * n + i]["Start_Timestamp"]) - int(tail[k * n + i - 1]["End_Timestamp"]) for k in range(reps)) / reps / 1e3
        tot += dur
        r = rs[0]
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        print(f"{i:4d} {dur:8.2f} us  gap {gap:6.2f}  wgs {wg:6d} x{r['Grid_Size_Y']:>2s}  vgpr {r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s}  lds {r['LDS_Block_Size']:>6s}  {nm[:90]}")
    span = sum(int(tail[k * n + n - 1]["End_Timestamp"]) - int(tail[k * n]["Start_Timestamp"]) for k in range(reps)) / reps / 1e3
    print(f"sum of durations {tot:.1f} us, first start -> last end {span:.1f} us")


if __name__ == "__main__":
    main()
