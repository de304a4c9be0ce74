#!/bin/bash
# builds face-recognition_amd/libfr_hip_B.so from the sources of a git revision (default HEAD): the "B" of tools/ab_lib.sh
rev=${1:-HEAD}
set -e
R=$(git rev-parse --show-toplevel)
T=$(mktemp -d)
git -C $R archive $rev face-recognition_amd/csrc include | tar -x -C $T
make -C $T/face-recognition_amd/csrc -j8 > /dev/null
cp $T/face-recognition_amd/libfr_hip.so $R/face-recognition_amd/libfr_hip_B.so
rm -rf $T
echo "libfr_hip_B.so = $rev"
