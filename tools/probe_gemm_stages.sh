for st in 2 3 4; do echo "== 128x128 tile, stages $st"; UNETR_GEMM_STAGES=$st PROBE_CFGS=128 python tools/probe_gemm_big.py 2>&1 | grep -v amdgpu; done
