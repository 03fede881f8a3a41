# PMC counters for the dominant conv kernel (run on the GPU box; separate --pmc passes as MI355X_MICROARCH.md prescribes).
# usage: bash tools/pmc_conv3.sh <outdir> ; writes <outdir>/pmc_summary.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/pmc}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# KERNEL / FILTER select the kernel_bench mode and the kernel-name substring (default: the 16->16 @ 96^3 forward conv)
KERNEL=${KERNEL:-conv3_fwd}
FILTER=${FILTER:-conv3_fwd}
CIN=${CIN:-16}
COUT=${COUT:-16}
# KARGS: the kernel_bench arguments after the kernel name (default: the conv shape); e.g. KERNEL=gemm_bf16 KARGS="--m 432 --n 2304 --k 768"
KARGS=${KARGS:---cin $CIN --cout $COUT --size 96 --batch 2}
PREC=${PREC:-bf16}      # precision mode of the benchmarked kernel (bf16 | fp32 | bf16x3)
ARGS="$KERNEL $KARGS --prec $PREC --iters 3 --warmup 1"
python3 $R/tools/kernel_bench.py $KERNEL $KARGS --prec $PREC --iters 20 --graph > $O/time.json 2>/dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/kernel_bench.py $ARGS > $O/$T.log 2>&1 || echo "pmc $T failed"
done
python3 - <<PY
import csv, glob, json, collections
out = {"command": "tools/kernel_bench.py $ARGS", "kernel_filter": "$FILTER", "family": """${FAMILY:-}""", "shape": """${SHAPE:-}"""}
out["time"] = json.load(open("$O/time.json"))
for d in ["FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU"]:
    for f in glob.glob(f"$O/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$FILTER" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = sum(v) / len(v)
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a 16 B/lane coalesced stream (guide, section HBM)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2
    out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024
    out["traffic_bytes_per_launch"] = out["hbm_read_bytes_corrected"] + out["hbm_write_bytes"]
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
print(json.dumps(out)[:900])
PY
