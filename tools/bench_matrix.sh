#!/bin/bash
# experiment helper: SpMV bench over several synthetic matrix families (prints one summary line per run)
for m in "" "--spmv-matrix banded:1048576:8" "--spmv-matrix banded:1048576:8 --batched 1" "--spmv-matrix rmat:20:16" "--spmv-matrix cage:1000000" "$@"; do
  timeout -k 10 300 python bench.py --skip-spgemm --skip-cpu $m 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print(d['config']['workload'][22:60], '|', d['config']['variant'][:7], 'nnz', d['config']['nnz'], 'blocks', d['config']['blocks'], 'ms', d['ms_per_step'], 'warm', d['warm_ms_per_step'], 'effGB/s', d['value'], 'frac', d['roofline']['frac'])
"
done
