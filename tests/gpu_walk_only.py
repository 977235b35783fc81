"""Profiling target: a few relative-criterion gravity walks at 64^3+64^3 (device only)."""
import sys

import numpy as np

from common import Problem, bindings

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = bindings()
pr = Problem(ng=ng, gas=True, periodic=1)
fp = pr.device()
pr.device_tree(fp)
fp.set_field(B.F_OLDACC, np.zeros(pr.n))
fp.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
fp.gravity(pr.g_grav(pr.theta), B.WALK_EWALD)
fp.gravity_finish(pr.G)
for r in range(reps):
    fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
    fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
s = fp.stats()
print("grav %.3f ms ewald %.3f ms" % (s["ms_grav"], s["ms_ewald"]))
