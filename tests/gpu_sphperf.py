"""Development aid: the SPH phases alone at converged smoothing lengths (one density pass, hydro),
device-event times.  python tests/gpu_sphperf.py [ng] [reps]"""
import sys

import numpy as np

from common import Problem, bindings


def main():
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    B = bindings()
    pr = Problem(ng=ng, gas=True, periodic=1)
    fp = pr.device()
    pr.device_tree(fp)
    fp.density(pr.g_dens())          # converge h
    fp.update_hmax()
    fp.hydro(pr.g_hydro())
    d, h = [], []
    for _ in range(reps):
        pr.device_tree(fp)
        fp.density(pr.g_dens())      # h is converged: one pass
        s = fp.stats()
        assert s["dens_iterations"] == 0
        d.append(s["ms_dens"])
        fp.update_hmax()
        fp.hydro(pr.g_hydro())
        s = fp.stats()
        h.append(s["ms_hydro"])
    print("ng=%d density (1 pass incl. finalize + selection) %.3f ms  hydro %.3f ms  "
          "(neighbours %d, pairs %d)" % (ng, float(np.median(d)), float(np.median(h)),
                                         s["dens_neighbours"], s["hydro_pairs"]))


if __name__ == "__main__":
    main()
