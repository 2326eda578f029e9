cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 2 4; do
 for epi in 1; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_gemm_v${v}_a -- python3 $R/tools/prof_gemm.py 51200 2048 512 $epi $v 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc_gemm_v${v}_b -- python3 $R/tools/prof_gemm.py 51200 2048 512 $epi $v 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_gemm_v${v}_c -- python3 $R/tools/prof_gemm.py 51200 2048 512 $epi $v 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_gemm_v${v}_d -- python3 $R/tools/prof_gemm.py 51200 2048 512 $epi $v 5 > /dev/null 2>&1
 done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_gemm -- python3 $R/tools/bench_gemm.py 1 4 > $R/gpurun_out/gemm2.log 2>&1
ls $R/gpurun_out/pmc_gemm_v1_a/*
