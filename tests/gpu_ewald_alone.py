"""Development aid: duration of the Ewald walk and of the Newtonian walk alone at c2."""
import sys

import numpy as np

from common import Problem, bindings

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = bindings()
pr = Problem(ng=ng, gas=True, periodic=1)
fp = pr.device()
pr.device_tree(fp)
fp.set_field(B.F_OLDACC, np.zeros(pr.n))
fp.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
fp.gravity(pr.g_grav(pr.theta), B.WALK_EWALD)
fp.gravity_finish(pr.G)
for rep in range(3):
    pr.device_tree(fp)
    fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
    fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
    s = fp.stats()
print("newton %.3f ms  ewald %.3f ms  (ewald interactions %d)"
      % (s["ms_grav"], s["ms_ewald"], s.get("ewald_interactions", -1)), flush=True)
