mkdir -p gpurun_out/r06
for p in "0 0" "6 0" "12 0" "18 0" "8 2" "16 2"; do set -- $p; echo "== GRIT_GEMM_STAGGER_US=$1 MODE=$2"; GRIT_GEMM_STAGGER_US=$1 GRIT_GEMM_STAGGER_MODE=$2 timeout 600 python tools/micro/bench_fused_variants.py 2>&1 | grep "M51200\|M204800"; done | tee gpurun_out/r06/fused_stagger.txt
