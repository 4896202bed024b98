#!/bin/bash
# In-situ tuning sweep inside one gpurun call: bench ms/step under different environment settings (tuning aids of the library).
# usage: [FLAGS="--pair"] tools/sweep_env.sh "<VAR=val ...>" ...     (an empty string = defaults)
for spec in "$@"; do
  for r in 1 2; do
    ms=$(env $spec python bench.py --no-cpu-baseline --no-roofline $FLAGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "[$spec $FLAGS] $ms"
  done
done
