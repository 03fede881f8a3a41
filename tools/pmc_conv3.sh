set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc1
mkdir -p $O
KB="python3 $R/tools/kernel_bench.py"
$KB conv3_fwd --cin 16 --cout 16 --iters 10 > $O/times.log 2>&1
$KB conv3_fwd --cin 32 --cout 16 --iters 10 >> $O/times.log 2>&1
$KB conv3_dgrad --cin 32 --cout 16 --iters 10 >> $O/times.log 2>&1
$KB conv3_wgrad --cin 32 --cout 16 --iters 10 >> $O/times.log 2>&1
$KB conv3_fwd --cin 16 --cout 16 --prec fp32 --iters 5 >> $O/times.log 2>&1
$KB gemm --m 432 --n 768 --k 3072 --iters 50 >> $O/times.log 2>&1
$KB gemm --m 432 --n 3072 --k 768 --iters 50 >> $O/times.log 2>&1
$KB gemm --m 13824 --n 3072 --k 768 --iters 10 >> $O/times.log 2>&1
$KB gemm --m 13824 --n 768 --k 3072 --iters 10 >> $O/times.log 2>&1
$KB encoder_fwd --batch 2 --iters 10 >> $O/times.log 2>&1
$KB encoder_fwd --batch 32 --iters 3 >> $O/times.log 2>&1
grep -E '^\{' $O/times.log | cut -c1-200
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/kernel_bench.py conv3_fwd --cin 16 --cout 16 --iters 3 --warmup 1 > $O/$T.log 2>&1 || echo "pmc $T failed"
done
ls $O
