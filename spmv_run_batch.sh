#!/bin/bash
# Batch driver for bmsparse_spmv_float, same contract as the reference's spmv_run_batch.sh:1-14:
# for every matrix named in $list, run  ./bmsparse_spmv_float <folder> <matrix> <matrix> <batched>  and append
# stdout to spmv_out.txt.  folder / list / batched may be overridden from the environment.
folder=${folder:-/media/matrices/ssget/MM/todas}
list=${list:-lista9.txt}
batched=${batched:-0}
here="$(cd "$(dirname "$0")" && pwd)"

rm -f spmv_out.txt
while read -r line; do
  [ -z "$line" ] && continue
  matrix="$(basename -- "${line%}")"
  echo "Working on $matrix"
  "$here/bmsparse_spmv_float" "$folder" "$matrix" "$matrix" "$batched" >> "spmv_out.txt" || echo "FAILED: $matrix" >&2
done < "$list"
