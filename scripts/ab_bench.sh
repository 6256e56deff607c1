#!/bin/bash
# A/B of library builds on ONE box: scripts/ab/lib*.so, each through the same bench command.
# usage: scripts/ab_bench.sh [bench.py flags...]   (results: gpurun_out/ab/<name>.json)
mkdir -p gpurun_out/ab
for rep in 1 2; do
for lib in scripts/ab/lib*.so; do
  name=$(basename $lib .so)
  SWIMMER_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-aux --steps 100 --warmup 5 "$@" > gpurun_out/ab/$name.$rep.json 2> gpurun_out/ab/$name.$rep.err || exit 1
  python - <<P
import json
j = json.load(open("gpurun_out/ab/$name.$rep.json"))
print("$name rep$rep: ms/iter %.4f kernel_ms %.4f (post %.4f) value %.4e" % (j["ms_per_step"], j["roofline"]["kernel_ms_timed_region"], j["roofline"]["kernel_ms_postpass"], j["value"]))
P
done
done
