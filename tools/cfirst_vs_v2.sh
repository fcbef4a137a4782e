#!/bin/bash
# channels-per-lane kernel against the position-per-lane kernel on the shapes around the selection threshold (one box)
# bash tools/cfirst_vs_v2.sh "KIND Cin Cout K stride dil L B" ...      (KIND conv | convT)
for args in "$@"; do saved=("$@")
  for c in auto 0 1; do
    if [ $c = auto ]; then unset KALLE_CONV_CFIRST; else export KALLE_CONV_CFIRST=$c; fi
    set -- $args                      # (a residual only where the output has the input's shape: stride-1 convs)
    echo -n "cfirst=$c  "; python tools/conv_one.py $args 20 $( [ "$1" = conv ] && [ "$5" = 1 ] && [ "$2" = "$3" ] && echo res ) 2>&1 | grep -v amdgpu
  done
  set -- "${saved[@]}"
done
