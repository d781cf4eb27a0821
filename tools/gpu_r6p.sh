#!/bin/bash
out=gpurun_out/r6p; mkdir -p $out
P=$PWD/drloco_amd/csrc/libdrloco_hip_dpp1.so; V=$PWD/build_variants/libdiet2.so
run() { tag=$1; lib=$2; shift 2; DL_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],1))"; }
for i in 1 2; do
  run "new  straight        " $P; run "prev straight        " $V
  run "new  no-split        " $P --no-split; run "prev no-split        " $V --no-split
  run "new  loco3d          " $P --walker loco3d; run "prev loco3d          " $V --walker loco3d
  run "new  loco3d no-split " $P --walker loco3d --no-split; run "prev loco3d no-split " $V --walker loco3d --no-split
  run "new  policy          " $P --policy; run "prev policy          " $V --policy
done
python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "full suite rc=$?"; tail -8 $out/tests.log
