#!/usr/bin/env python3
"""sha256 of every kernel source (bmsparse-spgemm-spmv_amd/csrc/*): printed as JSON {repo-relative path: hex digest}.  The profile scripts
run it on the GPU box next to the rocprofv3 passes (-> <out>/source_hash.json) and the summaries carry it; bench.py quotes a counter
figure only while the digests of that kernel's sources still match the tree it runs from -- no git history needed on the box."""
import glob, hashlib, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes(root=ROOT):
    out = {}
    for f in sorted(glob.glob(os.path.join(root, "bmsparse-spgemm-spmv_amd", "csrc", "*"))):
        if os.path.isfile(f):
            out[os.path.relpath(f, root)] = hashlib.sha256(open(f, "rb").read()).hexdigest()
    return out


if __name__ == "__main__":
    print(json.dumps(source_hashes(), indent=0))
