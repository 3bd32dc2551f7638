#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.  Run in the build container (needs
/root/reference for the half.hpp-anchored checker built by `make -C oracle ref`).

  half_rounding.json  -- fp16 rounding vectors produced by the REFERENCE's include/half.hpp (compiled where it
                         lies by oracle/Makefile into oracle/_ref/half_check): double -> half bits, and
                         half*half -> half bits.  Pins oracle/bmsp_oracle.c:orc_f64_to_f16_bits and the V15
                         "product rounded to fp16" rule.
  ragusa16_known.json -- known answers for the reference's only shipped matrices data/real/{A,B}_matrix.mtx,
                         computed here by an independent pure-Python dict-of-keys calculation straight from
                         the format definition -- NOT by the oracle.
"""
import json
import os
import random
import struct
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def half_vectors():
    exe = os.path.join(REPO, "oracle", "_ref", "half_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    rnd = random.Random(12345)
    doubles = [0.0, -0.0, 1.0, -1.0, 2049.0, 2051.0, 2050.0, 65504.0, 65519.999, 65520.0, 65536.0, 1e5, -1e5,
               2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -26, 3 * 2.0 ** -25, 2.0 ** -14, 2.0 ** -14 * (1 - 2.0 ** -11),
               1e-8, 1e-7, 6.1e-5, 0.1, 0.2, 0.3, 1.0 / 3.0, 3.14159265358979, 1.0009765625, 1.00048828125, 1.000732421875,
               1023.5, 1024.5, 1025.5, 4097.0, 4098.0, 4099.0]
    for _ in range(400):
        e = rnd.uniform(-28, 17)
        doubles.append(rnd.choice([-1, 1]) * 2.0 ** e * rnd.uniform(1, 2))
    for _ in range(200):  # exact ties between two halves
        m = rnd.randrange(1024, 2048)
        e = rnd.randrange(-14, 15)
        doubles.append((m + 0.5) * 2.0 ** (e - 10))
    inp = "".join("%r\n" % d for d in doubles)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == len(doubles)
    muls = []
    for _ in range(600):
        a = rnd.randrange(0, 0x7c00) | (rnd.randrange(2) << 15)
        b = rnd.randrange(0, 0x7c00) | (rnd.randrange(2) << 15)
        muls.append((a, b))
    inp = "".join("mul %04x %04x\n" % ab for ab in muls)
    mout = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    assert len(mout) == len(muls)
    return {"source": "include/half.hpp (half_float 2.2.0) compiled by oracle/Makefile:ref",
            "f64_to_f16": [[struct.pack(">d", d).hex(), o] for d, o in zip(doubles, out)],
            "f16_mul": [["%04x" % a, "%04x" % b, o] for (a, b), o in zip(muls, mout)]}


def read_mtx(path):
    with open(path) as f:
        lines = [l for l in f if not l.startswith("%")]
    nr, nc, nz = map(int, lines[0].split())
    ent = {}
    for l in lines[1:1 + nz]:
        r, c, v = l.split()
        ent[(int(r) - 1, int(c) - 1)] = float(v)
    return nr, nc, ent


def blocks_of(ent, transposed=False):
    blk = {}
    for (r, c), v in ent.items():
        pos = (c % 8) * 8 + (r % 8) if transposed else (r % 8) * 8 + (c % 8)
        blk.setdefault((r // 8, c // 8), {})[pos] = v
    return blk


def ragusa():
    nr, nc, A = read_mtx(os.path.join(HERE, "mtx", "real", "A_matrix.mtx"))
    _, _, B = read_mtx(os.path.join(HERE, "mtx", "real", "B_matrix.mtx"))
    ab = blocks_of(A)
    keys = sorted(ab)
    bmps = [sum(1 << (63 - p) for p in ab[k]) for k in keys]
    y = [sum(v for (r, c), v in A.items() if r == i) for i in range(nr)]

    def product(X, Y):
        Cd = {}
        prods = 0
        for (i, k), a in X.items():
            for (k2, j), b in Y.items():
                if k2 == k:
                    Cd[(i, j)] = Cd.get((i, j), 0.0) + a * b
                    prods += 1
        xb, yb = blocks_of(X), blocks_of(Y)
        cand = sum(1 for (bi, bk) in xb for (bk2, bj) in yb if bk2 == bk)
        surv = 0
        cblocks = {}
        for (bi, bk), xa in xb.items():
            for (bk2, bj), ya in yb.items():
                if bk2 != bk:
                    continue
                bits = 0
                for pa in xa:
                    for pb in ya:
                        if pa % 8 == pb // 8:
                            bits |= 1 << (63 - ((pa // 8) * 8 + pb % 8))
                if bits:
                    surv += 1
                    cblocks[(bi, bj)] = cblocks.get((bi, bj), 0) | bits
        ckeys = sorted(cblocks)
        return {"candidate_tasks": cand, "surviving_tasks": surv, "c_blocks": len(ckeys),
                "c_keys": ["%016x" % ((i << 32) | j) for i, j in ckeys],
                "c_bmps": ["%016x" % cblocks[k] for k in ckeys],
                "c_nnz": sum(bin(cblocks[k]).count("1") for k in ckeys), "scalar_products": prods,
                "sum": sum(Cd.values()), "max": max(Cd.values()),
                "entries": sorted([i, j, v] for (i, j), v in Cd.items())}

    return {"source": "data/real/A_matrix.mtx, B_matrix.mtx (Pajek/Ragusa16)", "num_rows": nr, "num_cols": nc, "nnz": len(A),
            "a_keys": ["%016x" % ((i << 32) | j) for i, j in keys], "a_bmps": ["%016x" % b for b in bmps],
            "a_popcounts": [bin(b).count("1") for b in bmps], "y_ones": y, "AxB": product(A, B), "AxA": product(A, A)}


if __name__ == "__main__":
    with open(os.path.join(HERE, "half_rounding.json"), "w") as f:
        json.dump(half_vectors(), f, indent=0)
    with open(os.path.join(HERE, "ragusa16_known.json"), "w") as f:
        json.dump(ragusa(), f, indent=0)
    print("golden vectors written")
