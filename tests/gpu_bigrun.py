"""Manual large-size run (not collected by pytest): python tests/gpu_bigrun.py NG
One full step (tree, Newtonian + Ewald walks, density, hmax, hydro) at NG^3 + NG^3 particles with
phase timings and a sampled oracle comparison of the gravity result."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import O, Problem, bindings, relerr  # noqa: E402

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = bindings()
t0 = time.time()
pr = Problem(ng=ng, gas=True, periodic=1)
n, ngas = pr.n, pr.ngas
print("ICs: n=%d  %.1f s" % (n, time.time() - t0), flush=True)
fp = pr.device()
for rep in range(2):
    pr.device_tree(fp)
    fp.set_field(B.F_OLDACC, np.full(n, 2.0))
    fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
    fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
    fp.density(pr.g_dens())
    fp.update_hmax()
    fp.hydro(pr.g_hydro())
    st = fp.stats()
    print("step %d: " % rep + " ".join("%s %.2f" % (k[3:], st[k]) for k in
          ("ms_tree", "ms_grav", "ms_ewald", "ms_dens", "ms_hmax", "ms_hydro")) +
          "  nodes %d  int/particle %.0f" % (st["tree_nodes"], (st["grav_interactions"] + st["ewald_interactions"]) / n), flush=True)
acc = fp.get_field(B.F_GRAVACCEL)
cost = fp.get_field(B.F_GRAVCOST)
assert np.isfinite(acc).all()
nn = fp.get_field(B.F_NUMNGB)
print("numngb window ok:", bool(np.all(np.abs(nn - pr.des_ngb) <= pr.max_dev + 1e-9)), flush=True)
t0 = time.time()
T = pr.oracle_tree()
print("oracle tree: %d nodes  %.1f s" % (T.numnodes, time.time() - t0), flush=True)
rng = np.random.default_rng(3)
sample = np.sort(rng.choice(n, 512, replace=False)).astype(np.int32)
old = np.full(n, 2.0)
oacc, ocost = T.gravity(pr.o_grav(0.0), sample, old)
T.gravity_ewald_add(pr.o_grav(0.0), O.ewald_table(pr.box), sample, old, oacc, ocost)
print("cost equal:", bool(np.array_equal(cost[sample], ocost)), " acc relerr: %.2e" % relerr(acc[sample], oacc), flush=True)
