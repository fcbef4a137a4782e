#!/bin/bash
# what do the HIP events around the GEMM launches cost?  same box, interleaved: default (dominant kernel only), every launch, none
for i in 1 2; do
  for B in 256 16; do
    for f in "" "--time-all-launches" "--no-kernel-timer"; do
      echo -n "B=$B ${f:-default}: "; python bench.py --no-cpu-baseline --batch $B $f 2>/dev/null | tail -n 1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2), 'ms')"
    done
  done
done
