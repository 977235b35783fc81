"""Development aid: how the walks' time scales with the number of targets (one tree, active
targets = a contiguous piece of the Peano-Hilbert order, as a multi-GPU shard has them)."""
import sys

import numpy as np

from common import Problem, ShardSet, bindings

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = bindings()
pr = Problem(ng=ng, gas=True, periodic=1)
fp = pr.device()
pr.device_tree(fp)
fp.set_field(B.F_OLDACC, np.zeros(pr.n))
fp.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
fp.gravity(pr.g_grav(pr.theta), B.WALK_EWALD)
fp.gravity_finish(pr.G)
probe = B.ForcePath(0)
probe.set_counts(pr.n, 0)
probe.set_field(B.F_POS, pr.ic["pos"])
probe.dd_init(0, 1)
probe.dd_set_domain(pr.extent[0], pr.extent[1], pr.extent[2], pr.force_soft)
order = np.argsort(probe.dd_keys(), kind="stable")
probe.close()
for frac in (1, 2, 4, 8):
    m = pr.n // frac
    act = np.sort(order[:m]).astype(np.int32)
    fp.set_active(None if frac == 1 else act)
    res = {}
    for name, walk in (("newton", B.WALK_NEWTON), ("ewald", B.WALK_EWALD), ("pair", B.WALK_NEWTON_EWALD)):
        t = []
        for r in range(3):
            fp.gravity(pr.g_grav(0.0), walk)
            fp.sync()
            s = fp.stats()
            t.append((s["ms_grav"], s["ms_ewald"]))
        res[name] = t[-1]
    print("1/%d of the targets (%d): newton alone %.2f  ewald alone %.2f  pair %.2f / %.2f ms" %
          (frac, m, res["newton"][0], res["ewald"][1], res["pair"][0], res["pair"][1]), flush=True)
