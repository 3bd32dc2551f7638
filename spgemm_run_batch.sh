#!/bin/bash
# Batch driver for bmsparse_spgemm_float, same contract as the reference's spgemm_run_batch.sh:1-16:
# ./bmsparse_spgemm_float <folder> <matrix> <matrix> <segmented> <tc_version> <verbose>  >> spgemm_out.txt
# with the reference's defaults segmented=0 tc_version=5 verbose=0 (overridable from the environment).
folder=${folder:-/media/matrices/ssget/MM/todas}
list=${list:-lista9.txt}
segmented=${segmented:-0}
tc_version=${tc_version:-5}
verbose=${verbose:-0}
here="$(cd "$(dirname "$0")" && pwd)"

rm -f spgemm_out.txt
while read -r line; do
  [ -z "$line" ] && continue
  matrix="$(basename -- "${line%}")"
  echo "Working on $matrix"
  "$here/bmsparse_spgemm_float" "$folder" "$matrix" "$matrix" "$segmented" "$tc_version" "$verbose" >> "spgemm_out.txt" || echo "FAILED: $matrix" >&2
done < "$list"
