R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
B=tools/micro/bin
for st in 1 0; do
echo "== GRIT_GEMM_W4_STAGGER=$st"
(export GRIT_GEMM_W4_STAGGER=$st; timeout 120 $B/gemm_ps_bench 51200 2048 512; timeout 120 $B/gemm_ps_bench 51200 1536 512; timeout 120 $B/gemm_ps_bench 51200 512 512; timeout 120 $B/gemm_ps_bench 204800 1024 256; timeout 60 $B/gemm_ps_bench 12800 4096 1024) 2>&1 | grep -v "stream\|waves\|max err\|differ"
done > $O/gemm_ps_bench_7.txt 2>&1
cat $O/gemm_ps_bench_7.txt
