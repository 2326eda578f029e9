R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
B=tools/micro/bin
(timeout 120 $B/gemm_ps_bench 51200 2048 512; timeout 120 $B/gemm_ps_bench 51200 512 512; timeout 120 $B/gemm_ps_bench 204800 1024 256; timeout 120 $B/gemm_ps_bench 51200 512 2048) > $O/gemm_ps_bench_3.txt 2>&1
(echo "== stamps dbg 0"; timeout 120 $B/gemm_ps_bench_stamps 51200 2048 512; echo "== stamps dbg 1 (nothing leaves)"; GRIT_GEMM_PS_DBG=1 timeout 120 $B/gemm_ps_bench_stamps 51200 2048 512; echo "== stamps dbg 2 (no global stores)"; GRIT_GEMM_PS_DBG=2 timeout 120 $B/gemm_ps_bench_stamps 51200 2048 512) >> $O/gemm_ps_bench_3.txt 2>&1
grep -v "max err\|elements differ" $O/gemm_ps_bench_3.txt
