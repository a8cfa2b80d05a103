#!/bin/bash
# interleaved A/B of two library builds on the full step, same box: tools/ab.sh [other.so] [rounds] [extra args]
cd "$(dirname "$0")/.."
OTHER=${1:-hand-gesture-recognition_amd/libmtn_base.so}
N=${2:-3}
shift; shift
for i in $(seq $N); do
  python tools/ab_step.py "$OTHER" "$@" 2>&1 | tail -1
  python tools/ab_step.py "$@" 2>&1 | tail -1
done
