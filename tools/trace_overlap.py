"""Read a rocprofv3 kernel-trace CSV and report, for the last graph replay, the wall span vs the summed kernel time
and the time during which >=2 kernels were in flight (evidence for / against branch overlap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ev = ev[-n:]
span = max(e[1] for e in ev) - ev[0][0]
tot = sum(e[1] - e[0] for e in ev)
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
depth = 0; last = pts[0][0]; over = 0; idle = 0
for t, d in pts:
    if depth >= 2: over += t - last
    if depth == 0: idle += t - last
    depth += d; last = t
print(f"kernels {len(ev)}  span {span/1e6:.3f} ms  sum {tot/1e6:.3f} ms  overlapped {over/1e6:.3f} ms  idle {idle/1e6:.3f} ms")
