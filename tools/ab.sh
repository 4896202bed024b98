#!/bin/bash
# A/B timing inside ONE gpurun call (devices differ by up to 12 % and a call may land on any of them: only numbers from
# the same call compare).  usage: tools/ab.sh ROUNDS "<label>|<dir>|<bench flags>" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    IFS='|' read -r label dir flags <<< "$spec"
    ms=$(cd $dir && python bench.py --no-cpu-baseline --no-roofline $flags 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $r $label $ms"
  done
done
