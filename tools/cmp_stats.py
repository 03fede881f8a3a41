"""compare two rocprofv3 kernel_stats.csv files of the captured step: per-kernel time per step (total / executions)
usage: python tools/cmp_stats.py a.csv b.csv [executions=27] [min_us=3]"""
import csv, sys, re
def load(p, n):
    d = {}
    for r in csv.DictReader(open(p)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"^void ", "", name)
        name = name.split("(")[0][:100]
        d[name] = (int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1000.0)
    return d
a, b = sys.argv[1], sys.argv[2]
n = float(sys.argv[3]) if len(sys.argv) > 3 else 27.0
mn = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
A, B = load(a, n), load(b, n)
keys = sorted(set(A) | set(B), key=lambda k: -max(A.get(k, (0, 0))[1], B.get(k, (0, 0))[1]))
ta = tb = 0.0
for k in keys:
    ca, ua = A.get(k, (0, 0.0)); cb, ub = B.get(k, (0, 0.0))
    ta += ua; tb += ub
    if abs(ua - ub) >= mn or (ca != cb):
        print(f"{ca:6.1f} {ua:8.1f}us | {cb:6.1f} {ub:8.1f}us | {ub-ua:+8.1f}  {k}")
print(f"total {ta:.1f} us  vs  {tb:.1f} us  ({tb-ta:+.1f});  launches {sum(v[0] for v in A.values()):.0f} vs {sum(v[0] for v in B.values()):.0f}")
