R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python tools/op_profile.py --steps 2 > $O/op_profile.txt 2>&1
grep -n "aten::add\|aten::copy_\|aten::contiguous\|aten::clone\|aten::fill_\|aten::zero_\|aten::mul\|aten::cat\|aten::sum" $O/op_profile.txt | head -70 | cut -c1-230
