"""Device-only: time of each phase when this GPU evaluates 1/N of the buckets (what one rank of
an N-GPU run does).  python tests/gpu_shardscan.py [ng]"""
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
fp.density(pr.g_dens())
for nr in (1, 2, 4, 8):
    fp.set_shard(0, nr)
    for rep in range(2):
        pr.device_tree(fp)
        fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
        fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
        fp.density(pr.g_dens())
        fp.update_hmax()
        fp.hydro(pr.g_hydro())
        s = fp.stats()
    tot = s["ms_tree"] + s["ms_grav"] + s["ms_ewald"] + s["ms_dens"] + s["ms_hmax"] + s["ms_hydro"]
    print("1/%d of the buckets: tree %.3f grav %.3f ewald %.3f dens %.3f hmax %.3f hydro %.3f sum %.3f ms"
          % (nr, s["ms_tree"], s["ms_grav"], s["ms_ewald"], s["ms_dens"], s["ms_hmax"],
             s["ms_hydro"], tot), flush=True)

# the same shares as ONE overlapped step (Newton+Ewald pair, SPH underneath): wall time per step
import time
for nr in (1, 2, 4, 8):
    fp.set_shard(0, nr)
    for rep in range(3):
        if rep == 1:
            fp.sync()
            t0 = time.perf_counter()
        pr.device_tree(fp)
        fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
        fp.density(pr.g_dens())
        fp.update_hmax()
        fp.hydro(pr.g_hydro())
    fp.sync()
    print("1/%d overlapped step: %.3f ms wall" % (nr, (time.perf_counter() - t0) / 2 * 1e3), flush=True)
