"""Device-only timing of each phase (no oracle): python tests/gpu_perf.py [ng] [reps]"""
import sys

import numpy as np

from common import Problem, bindings


def main():
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
    s = fp.stats()
    print("BH pass: grav %.3f ms (%d int, %d steps, eff %.3f) ewald %.3f ms (%d int, %d steps)" %
          (s["ms_grav"], s["grav_interactions"], s["grav_wave_steps"],
           s["grav_interactions"] / (64.0 * max(s["grav_wave_steps"], 1)), s["ms_ewald"],
           s["ewald_interactions"], s["ewald_wave_steps"]), flush=True)
    for r in range(reps):
        pr.device_tree(fp)
        fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
        fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
        fp.set_field(B.F_HSML, pr.hsml0)
        pr.device_tree(fp)
        fp.density(pr.g_dens())
        fp.update_hmax()
        fp.hydro(pr.g_hydro())
        s = fp.stats()
        tot = (s["ms_tree"] + s["ms_grav"] + s["ms_ewald"] + s["ms_dens"] + s["ms_hmax"] +
               s["ms_hydro"])
        print("rep %d: tree %.3f grav %.3f ewald %.3f dens %.3f (it %d) hmax %.3f hydro %.3f  "
              "sum %.3f ms -> %.3e part-steps/s" %
              (r, s["ms_tree"], s["ms_grav"], s["ms_ewald"], s["ms_dens"], s["dens_iterations"],
               s["ms_hmax"], s["ms_hydro"], tot, pr.n / (tot * 1e-3)), flush=True)
        print("   grav: %d int, %d steps, lane eff %.3f, %.2f GB/s alg | ewald: %d int %d steps eff %.3f" %
              (s["grav_interactions"], s["grav_wave_steps"],
               s["grav_interactions"] / (64.0 * max(s["grav_wave_steps"], 1)),
               32.0 * s["grav_interactions"] / (s["ms_grav"] * 1e-3) / 1e9,
               s["ewald_interactions"], s["ewald_wave_steps"],
               s["ewald_interactions"] / (64.0 * max(s["ewald_wave_steps"], 1))), flush=True)


if __name__ == "__main__":
    main()
