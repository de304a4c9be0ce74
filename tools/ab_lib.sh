#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <reps> <cmd...>   -- runs <cmd> alternately with libfr_hip.so (A) and libfr_hip_B.so (B) in place
reps=$1; shift
L=face-recognition_amd
cp $L/libfr_hip.so /tmp/libA.so; cp $L/libfr_hip_B.so /tmp/libB.so
for r in $(seq $reps); do
  for v in A B; do
    cp /tmp/lib$v.so $L/libfr_hip.so
    echo -n "$v: "; "$@" || exit 1
  done
done
cp /tmp/libA.so $L/libfr_hip.so
