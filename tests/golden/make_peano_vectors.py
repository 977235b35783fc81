"""Generates tests/golden/peano_keys.npz: Peano-Hilbert and Morton keys of seeded integer
triplets under the REFERENCE'S definition of the curve.

The reference cannot be compiled here (allvars.h needs GSL, which this image lacks, and
stand-ins are not allowed), so this script does the next best thing: it reads the two 48x8
state tables of peano.c:195-295 from the reference source AS DATA at generation time and
evaluates the table-driven state machine the file documents (peano.c:300-316: per bit plane,
pix = 4*xbit + 2*ybit + zbit; key = key<<3 | subpix3[rot][pix]; rot = rottable3[rot][pix]).
Only the resulting input/output vectors are committed; no reference text is.
Run in the build container:  python tests/golden/make_peano_vectors.py
"""
import os
import re

import numpy as np

REF = "/root/reference/peano.c"


def table(src, name):
    m = re.search(name + r"\[48\]\[8\] = \{(.*?)\};", src, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    return np.array([[int(v) for v in r.split(",")] for r in rows], dtype=np.int64)


def main():
    src = open(REF).read()
    rot, sub = table(src, "rottable3"), table(src, "subpix3")
    rng = np.random.default_rng(20261004)
    cases = []
    for bits in (1, 2, 3, 5, 10, 21):
        n = 8 if bits == 1 else 400
        xyz = rng.integers(0, 1 << bits, size=(n, 3))
        if bits <= 2:  # exhaustive for the smallest grids
            g = np.arange(1 << bits)
            xyz = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
        cases.append((bits, xyz))
    B, X, PH, MO = [], [], [], []
    for bits, xyz in cases:
        for x, y, z in xyz:
            r, key, mort = 0, 0, 0
            for b in range(bits - 1, -1, -1):
                pix = (((x >> b) & 1) << 2) | (((y >> b) & 1) << 1) | ((z >> b) & 1)
                key = (key << 3) | int(sub[r][pix])
                r = int(rot[r][pix])
                mort = (mort << 3) + (((z >> b) & 1) << 2) + (((y >> b) & 1) << 1) + ((x >> b) & 1)
            B.append(bits)
            X.append((x, y, z))
            PH.append(key)
            MO.append(mort)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peano_keys.npz")
    np.savez_compressed(out, bits=np.array(B, np.int32), xyz=np.array(X, np.int32),
                        peano=np.array(PH, np.uint64), morton=np.array(MO, np.uint64))
    print("wrote", out, len(B), "vectors")


if __name__ == "__main__":
    main()
