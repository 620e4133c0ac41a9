#!/bin/bash
# same-box kernel A/B of alternative builds: tools/ab_lib.sh "<tag> <tag> ..." "<prof_one cfg>" ...   ("base" = in-tree library)
tags=$1; shift
for cfg in "$@"; do
  for t in $tags; do
    if [ "$t" = base ]; then unset FALNET_LIB; else export FALNET_LIB=$PWD/fal_net_amd/libfalnet_hip_$t.so; fi
    echo -n "$t: "; python tools/prof_one.py $cfg 2>&1 | tail -1
  done
done
