#!/bin/bash
# A/B of library builds on ONE box through a probe script instead of bench.py:
# usage: scripts/ab_probe.sh scripts/<probe>.py   (every scripts/ab/lib*.so, two repetitions)
for rep in 1 2; do
for lib in scripts/ab/lib*.so; do
  echo "== $(basename $lib .so) rep$rep"
  SWIMMER_HIP_LIB=$PWD/$lib python "$@" 2>/dev/null
done
done
