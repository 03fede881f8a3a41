"""summarise the two rocprofv3 --pmc passes of tools/pmc_step.sh into gpurun_out/<PMC_STEP_NAME> (copied to profiles/ by hand)"""
import collections
import csv
import glob
import json
import os
import sys

out_dir, cmd = sys.argv[1], sys.argv[2]
STEPS = 7          # bench.py --steps 3 --warmup 1: 2 eager warm-up steps + (capture runs nothing) + 1 warm-up + 3 timed + ... see below


def load(sub, counter):
    rows = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out_dir, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            rows[k][0] += float(r["Counter_Value"])
            rows[k][1] += 1
    return rows


fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
# executions of the step in the process = launches of the loss kernel (once per step)
steps = max(1, next((v[1] for k, v in fetch.items() if "dicece_fwd_kernel" in k), STEPS))
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0.0, 0])
    w, nw = write.get(k, [0.0, 0])
    n = max(nf, nw)
    if n == 0:
        continue
    kernels[k.replace("(anonymous namespace)::", "")[:160]] = {
        "launches_per_step": n / steps, "hbm_read_bytes_per_step": f * 1024 * 2 / steps, "hbm_write_bytes_per_step": w * 1024 / steps,
        "hbm_bytes_per_step": (f * 1024 * 2 + w * 1024) / steps}
res = {"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- {cmd} (tools/pmc_step.sh; separate passes)",
       "steps_in_process": steps, "correction": "read = FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16 B/lane streams), write = WRITE_SIZE x 1024",
       "kernels": kernels}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", os.environ.get("PMC_STEP_NAME", "r03_pmc_step_kernels.json"))
json.dump(res, open(dst, "w"), indent=1)
tot = sum(v["hbm_bytes_per_step"] for v in kernels.values())
print(f"{len(kernels)} kernels, {tot / 1e9:.2f} GB of HBM traffic per step -> {dst}")
