#!/bin/bash
# In-situ tuning sweep inside one gpurun call: bench ms/step under different environment settings (tuning aids of the library).
# Needs the diagnostic build (make -C asy-vrnet_amd/csrc tuning; VRNET_HIP_LIB=.../libvrnet_hip_tuning.so): the product library
# reads no environment.  bench.py runs with --diagnostic (its line is marked so).
# usage: [FLAGS="--pair"] tools/sweep_env.sh "<VAR=val ...>" ...     (an empty string = defaults)
for spec in "$@"; do
  for r in 1 2; do
    ms=$(env $spec python bench.py --diagnostic --no-cpu-baseline --no-roofline $FLAGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "[$spec $FLAGS] $ms"
  done
done
