"""Exploratory end-to-end check on a GPU box: every phase against the oracle, timings printed.
Usage: python tests/gpu_probe.py [ng] [periodic]
"""
import sys
import time

import numpy as np

from common import O, Problem, bindings, relerr


def main():
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    periodic = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = bindings()
    pr = Problem(ng=ng, gas=True, periodic=periodic)
    n, ngas = pr.n, pr.ngas
    print("n", n, "ngas", ngas, "periodic", periodic, flush=True)
    fp = pr.device()
    t0 = time.time()
    pr.device_tree(fp)
    fp.sync()
    print("device tree build wall %.3f s" % (time.time() - t0), fp.stats()["tree_nodes"],
          fp.stats()["gastree_nodes"], flush=True)
    t0 = time.time()
    T = pr.oracle_tree()
    print("oracle tree build %.3f s, nodes %d" % (time.time() - t0, T.numnodes), flush=True)

    tg = np.arange(n, dtype=np.int32)
    # pass 1: Barnes-Hut
    fp.set_field(B.F_OLDACC, np.zeros(n))
    fp.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
    acc = fp.get_field(B.F_GRAVACCEL)
    cost = fp.get_field(B.F_GRAVCOST)
    t0 = time.time()
    oacc, ocost = T.gravity(pr.o_grav(pr.theta), tg, np.zeros(n))
    t_or = time.time() - t0
    st = fp.stats()
    print("BH : relerr %.3e cost equal %s  ia/part %.1f  gpu %.3f ms  oracle %.3f s" %
          (relerr(acc, oacc), bool((cost == ocost).all()), cost.mean(), st["ms_grav"], t_or),
          flush=True)
    if periodic:
        tab = O.ewald_table(pr.box)
        fp.ewald_init(pr.box)
        gtab = fp.ewald_table()
        print("ewald table max abs diff %.3e (max |tab| %.3e)" %
              (np.abs(gtab - tab).max(), np.abs(tab).max()), flush=True)
        fp.gravity(pr.g_grav(pr.theta), B.WALK_EWALD)
        acc = fp.get_field(B.F_GRAVACCEL)
        cost = fp.get_field(B.F_GRAVCOST)
        T.gravity_ewald_add(pr.o_grav(pr.theta), tab, tg, np.zeros(n), oacc, ocost)
        st = fp.stats()
        print("BH+ewald : relerr %.3e cost equal %s  gpu %.3f ms" %
              (relerr(acc, oacc), bool((cost == ocost).all()), st["ms_ewald"]), flush=True)
    fp.gravity_finish(pr.G)
    old = fp.get_field(B.F_OLDACC)
    oold = np.linalg.norm(oacc, axis=1)
    print("oldacc relerr %.3e" % relerr(old, oold), flush=True)
    # pass 2: relative criterion
    fp.gravity(pr.g_grav(0.0), B.WALK_NEWTON)
    acc = fp.get_field(B.F_GRAVACCEL)
    cost = fp.get_field(B.F_GRAVCOST)
    t0 = time.time()
    oacc2, ocost2 = T.gravity(pr.o_grav(0.0), tg, oold)
    t_or = time.time() - t0
    st = fp.stats()
    print("REL: relerr %.3e cost equal %s  ia/part %.1f  gpu %.3f ms  oracle %.3f s (%d thr)" %
          (relerr(acc, oacc2), bool((cost == ocost2).all()), cost.mean(), st["ms_grav"], t_or,
           O.num_threads()), flush=True)
    if periodic:
        fp.gravity(pr.g_grav(0.0), B.WALK_EWALD)
        acc = fp.get_field(B.F_GRAVACCEL)
        cost = fp.get_field(B.F_GRAVCOST)
        T.gravity_ewald_add(pr.o_grav(0.0), tab, tg, oold, oacc2, ocost2)
        st = fp.stats()
        print("REL+ewald : relerr %.3e cost equal %s gpu %.3f ms" %
              (relerr(acc, oacc2), bool((cost == ocost2).all()), st["ms_ewald"]), flush=True)

    # density
    t0 = time.time()
    fp.density(pr.g_dens())
    fp.sync()
    t_g = time.time() - t0
    st = fp.stats()
    act = np.arange(ngas, dtype=np.int32)
    t0 = time.time()
    od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                   pr.ti_begstep, pr.hsml0)
    t_or = time.time() - t0
    h = fp.get_field(B.F_HSML)[:ngas]
    print("density: iters gpu %d oracle %d | wall gpu %.3f s oracle %.3f s | ngb visits gpu %d oracle %d" %
          (st["dens_iterations"], od["iterations"], t_g, t_or, st["dens_neighbours"],
           od["ngb_visits"]), flush=True)
    for name, fid in (("hsml", None), ("numngb", B.F_NUMNGB), ("density", B.F_DENSITY),
                      ("dhsmlfac", B.F_DHSMLFAC), ("divvel", B.F_DIVVEL),
                      ("curlvel", B.F_CURLVEL), ("pressure", B.F_PRESSURE)):
        g = h if fid is None else fp.get_field(fid)
        o = od[name][:ngas]
        print("   %-9s maxabs rel %.3e" % (name, np.max(np.abs(g - o)) / np.max(np.abs(o))),
              flush=True)
    # hydro
    fp.update_hmax()
    fp.hydro(pr.g_hydro())
    st = fp.stats()
    T.update_hmax(act, od["hsml"], od["divvel"])
    t0 = time.time()
    oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                 od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
    t_or = time.time() - t0
    ha = fp.get_field(B.F_HYDROACCEL)
    de = fp.get_field(B.F_DTENTROPY)
    ms = fp.get_field(B.F_MAXSIGNALVEL)
    print("hydro: pairs gpu %d oracle %d | gpu %.3f ms (hmax %.3f ms) oracle %.3f s" %
          (st["hydro_pairs"], oh["npairs"], st["ms_hydro"], st["ms_hmax"], t_or), flush=True)
    print("   hydroaccel relerr(max-norm) %.3e  dtentropy %.3e  maxsignalvel %.3e" %
          (np.abs(ha - oh["hydroaccel"][:ngas]).max() / np.abs(oh["hydroaccel"]).max(),
           np.abs(de - oh["dtentropy"][:ngas]).max() / np.abs(oh["dtentropy"]).max(),
           relerr(ms, oh["maxsignalvel"][:ngas])), flush=True)
    print("stats", st, flush=True)


if __name__ == "__main__":
    main()
