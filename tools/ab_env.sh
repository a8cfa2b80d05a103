#!/bin/bash
# interleaved A/B of the full training step under two environments, same box, same library:
#   tools/ab_env.sh "VAR=0" "VAR=1" [rounds] [extra ab_step.py args]      (each spec: space-separated VAR=value pairs)
cd "$(dirname "$0")/.."
A="$1"; B="$2"; N=${3:-3}
shift; shift; shift
for i in $(seq $N); do
  echo -n "[$A] "; env $A python tools/ab_step.py "$@" 2>&1 | tail -1
  echo -n "[$B] "; env $B python tools/ab_step.py "$@" 2>&1 | tail -1
done
