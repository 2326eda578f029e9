# engine clock / power while a matrix-bound kernel runs (rocm-smi samples beside tools/micro/bench_wgrad_tn.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
(rocm-smi --showclocks --showpower 2>&1 | head -30) > $O/clocks_idle.txt
SHAPES=12800x4096x1024,12800x4096x1024,12800x4096x1024,12800x4096x1024,12800x4096x1024,12800x4096x1024,51200x2048x512,51200x2048x512,51200x2048x512 timeout 300 python tools/micro/bench_wgrad_tn.py > $O/clocks_load_bench.txt 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do (rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk\|fclk" | head -8) ; sleep 1.5; done > $O/clocks_load.txt
wait $BP
echo "--- idle"; grep -i "sclk\|power\|mclk" $O/clocks_idle.txt | head -6
echo "--- load"; head -24 $O/clocks_load.txt
grep "^M" $O/clocks_load_bench.txt | tail -3 | cut -c1-160
