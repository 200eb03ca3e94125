#!/bin/bash
# Build the kernels of another commit (or of the working tree with extra -D switches) as a variant library for same-box
# A/B timing (tools/ab.sh, abi.py MLMCPI_LIB_VARIANT):
#   tools/build_variant.sh <commit|WORK> <name> [EXTRA flags]     -> mlmcpathintegral_amd/libmlmcpi_hip_<name>.so
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$REV" = WORK ]; then
  make -C $ROOT/mlmcpathintegral_amd/csrc variant VARIANT=$NAME EXTRA="$*" -j4 > /dev/null
else
  T=$(mktemp -d)
  git -C $ROOT archive $REV mlmcpathintegral_amd/csrc include | tar -x -C $T
  make -C $T/mlmcpathintegral_amd/csrc EXTRA="$*" -j4 > /dev/null
  cp $T/mlmcpathintegral_amd/libmlmcpi_hip.so $ROOT/mlmcpathintegral_amd/libmlmcpi_hip_$NAME.so
  rm -rf $T
fi
ls -la $ROOT/mlmcpathintegral_amd/libmlmcpi_hip_$NAME.so
