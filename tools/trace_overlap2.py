"""concurrency inside the captured step, from a rocprofv3 --kernel-trace CSV: per replay window, sum of kernel durations vs the
union of their [start, end] intervals, and which kernels overlapped with which.  usage: python tools/trace_overlap2.py kernel_trace.csv"""
import csv, sys, re, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# last 20 steps: take the last 60% of kernels
rows = rows[int(len(rows) * 0.5):]
tot = sum(e - s for s, e, _, _ in rows)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  sum of durations {tot/1e6:.3f} ms  union {union/1e6:.3f} ms  span {span/1e6:.3f} ms  overlap {100*(tot-union)/tot:.1f} %  idle {100*(span-union)/span:.1f} %")
ov = collections.Counter()
active = []
for s, e, n, q in rows:
    active = [(ee, nn) for ee, nn in active if ee > s]
    for ee, nn in active:
        ov[(nn, n)] += min(ee, e) - s
    active.append((e, n))
for (a, b), t in ov.most_common(12):
    print(f"  {t/1e3:9.1f} us  {a}  ||  {b}")
queues = collections.Counter(q for _, _, _, q in rows)
print("queues/streams:", dict(queues))
