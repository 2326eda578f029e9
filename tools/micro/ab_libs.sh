#!/bin/bash
# A/B of library builds on the training step, alternating on one box:  ab_libs.sh tools/micro/bin/a.so tools/micro/bin/b.so ...
mkdir -p gpurun_out/r02
cp grit_amd/csrc/libgrit_hip.so /tmp/lib_keep.so
for pass in 1 2; do
  for lib in "$@"; do
    cp $lib grit_amd/csrc/libgrit_hip.so
    timeout 400 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 2>/dev/null | tail -1 \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $lib)', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'loss', round(d.get('final_loss',0),4))"
  done
done | tee gpurun_out/r02/ab_libs.txt
cp /tmp/lib_keep.so grit_amd/csrc/libgrit_hip.so
