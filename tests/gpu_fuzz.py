"""Development aid: seeded random sweeps of the parity checks -- particle numbers, distributions
(plane-wave clustered lattices, Plummer spheres, uniform noise), periodic / open boundaries, equal /
unequal softenings, full and partial active lists -- device against the oracle, counts exactly.
python tests/gpu_fuzz.py [nseeds] [first seed]"""
import sys
import time

import numpy as np

from common import O, Problem, bindings, ics, relerr

B = bindings()
TOL = 1e-11


def one(seed):
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 3)
    periodic = int(rng.integers(0, 2))
    unequal = bool(rng.integers(0, 2))
    if kind == 0:
        ic = ics.make_ics(int(rng.integers(3, 13)), gas=True, seed=int(seed), clustered=True,
                          rms_disp=float(rng.uniform(0.2, 2.0)))
    elif kind == 1:
        ic = ics.make_plummer(int(rng.integers(70, 4000)), seed=int(seed),
                              a=float(rng.uniform(0.01, 0.1)),
                              gas_fraction=float(rng.uniform(0.1, 0.6)))
    else:
        n = int(rng.integers(2, 1500))
        ngas = int(rng.integers(0, n))
        pos = rng.random((n, 3))
        typ = np.where(np.arange(n) < ngas, 0, rng.integers(1, 6, n)).astype(np.int32)
        ic = dict(pos=pos, vel=rng.standard_normal((n, 3)), mass=rng.uniform(0.5, 2.0, n) / n,
                  type=typ, ngas=ngas, boxsize=1.0, spacing=1.0 / max(2.0, n ** (1 / 3)),
                  id=np.arange(1, n + 1, dtype=np.uint32), u=np.full(ngas, 0.01))
    # a quarter of the problems sit between two domain decompositions of a black-hole build: some
    # records of the gas block have become sinks / stars (Type != 0), some gas has been swallowed
    # (Mass == 0) and the neighbour loops follow the BLACK_HOLES / DUST rules
    rule = 0
    if ic["ngas"] > 60 and rng.random() < 0.25:
        ngas0 = int(ic["ngas"])
        conv = rng.choice(ngas0, max(1, ngas0 // 40), replace=False)
        ic["type"] = ic["type"].copy()
        ic["type"][conv] = rng.choice([4, 5], len(conv))
        gone = rng.choice(ngas0, max(1, ngas0 // 30), replace=False)
        ic["mass"] = ic["mass"].copy()
        ic["mass"][gone] = 0.0
        rule = int(rng.choice([1, 3]))
    O.set_massless_gas_rule(rule)
    pr = Problem(ic=ic, periodic=periodic, unequal=unequal,
                 des_ngb=float(min(33.0, max(ic["ngas"] - 1, 1))))
    n, ng = pr.n, pr.ngas
    fp = pr.device()
    fp.set_massless_gas_rule(rule)
    adaptive = bool(ng > 0 and rng.random() < 0.3)           # ADAPTIVE_GRAVSOFT_FORGAS
    if adaptive:
        pr.hsml0[:ng] *= rng.uniform(0.05, 1.0, ng)
        fp.set_field(B.F_HSML, pr.hsml0)
        fp.set_adaptive_gravsoft(True)
    pr.device_tree(fp)
    T = pr.oracle_tree()
    if adaptive:
        T.adaptive_gravsoft()
    assert fp.stats()["tree_nodes"] == T.numnodes, "node count"
    # density for every gas particle first: the later partial lists see inactive neighbours with a
    # valid (computed) state, as in a run
    allgas = np.where(pr.ic["type"][:ng] == 0)[0].astype(np.int32)
    sph = ng >= 40
    if sph:
        fp.density(pr.g_dens())
        od = T.density(pr.o_dens(), allgas, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                       pr.ti_begstep, pr.hsml0)
        assert relerr(fp.get_field(B.F_HSML)[allgas], od["hsml"][allgas]) < TOL, "hsml"
        assert relerr(fp.get_field(B.F_DENSITY)[allgas], od["density"][allgas]) < TOL, "density"
        assert fp.stats()["dens_iterations"] == od["iterations"], "h iterations"
        fp.update_hmax()
        T.update_hmax(allgas, od["hsml"], od["divvel"])
    act = None
    if rng.random() < 0.5 and n > 4:
        act = np.sort(rng.choice(n, int(rng.integers(1, n)), replace=False)).astype(np.int32)
        fp.set_active(act)
    tg = np.arange(n, dtype=np.int32) if act is None else act
    old = 0.2 + rng.random(n)
    fp.set_field(B.F_OLDACC, old)
    theta = float(rng.choice([0.0, 0.4, 0.8]))
    mode = "newton"
    if periodic:
        mode = str(rng.choice(["two calls", "pair", "shortrange"]))
    if mode == "shortrange":                                 # TreePM short-range walk
        asmth = 1.25 * pr.box / float(rng.choice([8, 16, 32]))
        fp.gravity(pr.g_grav(theta, 4.5 * asmth, asmth), B.WALK_SHORTRANGE)
        oacc, ocost = T.gravity(pr.o_grav(theta, rcut=4.5 * asmth, asmth=asmth), tg, old,
                                kind="shortrange")
    else:
        if mode == "pair":
            fp.gravity(pr.g_grav(theta), B.WALK_NEWTON_EWALD)
        else:
            fp.gravity(pr.g_grav(theta), B.WALK_NEWTON)
        oacc, ocost = T.gravity(pr.o_grav(theta), tg, old)
        if mode == "two calls":
            fp.gravity(pr.g_grav(theta), B.WALK_EWALD)
        if periodic:
            T.gravity_ewald_add(pr.o_grav(theta), O.ewald_table(pr.box), tg, old, oacc, ocost)
    assert np.array_equal(fp.get_field(B.F_GRAVCOST)[tg], ocost), "gravity counts"
    scale = np.abs(oacc).max() + 1e-300
    assert np.abs(fp.get_field(B.F_GRAVACCEL)[tg] - oacc).max() < 1e-10 * scale, "gravity"
    gas = tg[tg < ng]
    gas = gas[pr.ic["type"][gas] == 0]
    if sph and len(gas):
        fp.hydro(pr.g_hydro())
        oh = T.hydro(pr.o_hydro(), gas, pr.velpred, od["hsml"], od["density"], od["pressure"],
                     od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
        ha = fp.get_field(B.F_HYDROACCEL)[gas]
        want = oh["hydroaccel"][gas]
        assert fp.stats()["hydro_pairs"] == oh["npairs"], "pairs"
        assert np.abs(ha - want).max() <= 1e-10 * (np.abs(want).max() + 1e-300), "hydro"
    substeps = 0
    if not adaptive and rng.random() < 0.4:
        substeps = substep_sequence(rng, pr, periodic)
    O.set_massless_gas_rule(0)
    return kind, n, ng, mode, unequal, adaptive, (None if act is None else len(act)), substeps, rule


def substep_sequence(rng, pr, periodic):
    """Sub-steps on the tree of the last full build (forcetree.c:1356-1520): random kicks of random
    subsets, drifts of everybody, gravity for random active lists on the kept tree -- counts against the
    oracle's drifted and kicked insertion tree."""
    n = pr.n
    c, ce, ln = pr.extent
    ext = (c - 0.1 * ln, ce.copy(), 1.2 * ln)               # room to drift inside the domain cube
    vel = pr.ic["vel"].copy()
    vscale = np.abs(vel).max() + 1e-300
    fp = pr.device()
    fp.set_dynamic_tree(True)
    fp.tree_build(ext[0], ext[1], ext[2], pr.force_soft)
    T = O.Tree(pr.ic["pos"].copy(), vel, pr.ic["mass"], pr.ic["type"], pr.force_soft, hsml=pr.hsml0, extent=ext)
    old = 0.2 + 3.0 * rng.random(n)
    fp.set_field(B.F_OLDACC, old)
    tab = O.ewald_table(pr.box) if periodic else None
    nsub = int(rng.integers(1, 4))
    for _ in range(nsub):
        k = int(rng.integers(0, n + 1))
        act = np.sort(rng.choice(n, k, replace=False)).astype(np.int32)
        dv = 0.2 * vscale * rng.standard_normal((k, 3))
        T.vel[act] += dv
        fp.set_field(B.F_VEL, T.vel)
        if k:
            T.kick_nodes(act, dv)
            fp.tree_kick_nodes(act, dv)
        dt = float(rng.uniform(0.0005, 0.01)) * ln / vscale
        T.pos += T.vel * dt
        T.drift_nodes(dt)
        fp.set_field(B.F_POS, T.pos)
        fp.tree_substep(dt)
        m = int(rng.integers(1, n + 1))
        tg = np.sort(rng.choice(n, m, replace=False)).astype(np.int32)
        theta = float(rng.choice([0.0, 0.6]))
        fp.set_active(tg)
        fp.gravity(pr.g_grav(theta), B.WALK_NEWTON_EWALD if periodic else B.WALK_NEWTON)
        oa, oc = T.gravity(pr.o_grav(theta), tg, old)
        if periodic:
            T.gravity_ewald_add(pr.o_grav(theta), tab, tg, old, oa, oc)
        assert np.array_equal(fp.get_field(B.F_GRAVCOST)[tg], oc), "gravity counts on the kept tree"
        assert np.abs(fp.get_field(B.F_GRAVACCEL)[tg] - oa).max() < 1e-10 * (np.abs(oa).max() + 1e-300), \
            "gravity on the kept tree"
        fp.set_active(None)
    fp.close()
    return nsub


if __name__ == "__main__":
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    bad = 0
    for seed in range(first, first + nseeds):
        try:
            info = one(seed)
            print("seed %d ok %s" % (seed, info), flush=True)
        except AssertionError as e:
            bad += 1
            O.set_massless_gas_rule(0)
            print("seed %d FAILED: %s" % (seed, e), flush=True)
    print("%d seeds, %d failures, %.1f s" % (nseeds, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)
