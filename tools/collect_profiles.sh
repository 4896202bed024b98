#!/bin/bash
# Copies the evidence tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ with the round prefix.
# usage: tools/collect_profiles.sh r03
cd "$(dirname "$0")/.." || exit 1
R=${1:?round prefix, e.g. r03}
src=gpurun_out/refresh
for f in $src/*; do
  b=$(basename "$f")
  case "$b" in
    *.log|*.err|*_pair.json) continue ;;      # run logs; (--pair is not part of the refresh: a file of that name is left over)
  esac
  [ -s "$f" ] && cp "$f" "profiles/${R}_$b"
done
ls profiles | grep "^${R}_"
