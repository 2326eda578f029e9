O=gpurun_out/r06; mkdir -p $O
rm -f $O/filled_smooth.jsonl
for s in 5 10 20 40; do GRIT_TEST_SMOOTH=$s GRIT_TEST_MEASURE=$PWD/$O/filled_smooth.jsonl timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -k filled > $O/filled_s$s.log 2>&1; tail -n 2 $O/filled_s$s.log; done
cat $O/filled_smooth.jsonl
