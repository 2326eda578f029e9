R=$GRAFT_REPO_ROOT
cd $R
cp grit_amd/csrc/libgrit_hip.so /tmp/lib_keep.so
for pass in 1 2; do
  for lib in tools/micro/bin/libgrit_old_winattn.so tools/micro/bin/libgrit_new.so; do
    cp $lib grit_amd/csrc/libgrit_hip.so
    timeout 400 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $lib)', round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
cp /tmp/lib_keep.so grit_amd/csrc/libgrit_hip.so
