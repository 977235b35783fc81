"""Debug: distribution of visited elements per wavefront (GHIP_DEBUG_STEPS=1)."""
import os
import sys

import numpy as np

os.environ["GHIP_DEBUG_STEPS"] = "1"
from common import Problem, bindings  # noqa: E402

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = bindings()
pr = Problem(ng=ng, gas=True, periodic=1)
fp = pr.device()
pr.device_tree(fp)
os.environ.pop("GHIP_DEBUG_STEPS")
fp.set_field(B.F_OLDACC, np.zeros(pr.n))
fp.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
fp.gravity_finish(pr.G)
os.environ["GHIP_DEBUG_STEPS"] = "1"
for theta in (pr.theta, 0.0):
    fp.gravity(pr.g_grav(theta), B.WALK_NEWTON)
    st = fp.get_field(B.F_GRAVCOST).astype(np.float64)
    print("theta", theta, "steps/wave: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" %
          (st.mean(), np.median(st), np.percentile(st, 90), np.percentile(st, 99), st.max()))
