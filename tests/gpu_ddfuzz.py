"""Development aid: seeded random sweeps of the domain-decomposed path -- particle numbers,
distributions (clustered lattices, Plummer spheres, uniform noise), periodic / open boundaries, equal
/ unequal softenings, 2..8 logical shards (some of them nearly or entirely empty), Barnes-Hut or
relative criterion, pair or separate walks, SPH, a drift + migration -- shards against the oracle's
single tree, counts exactly.   python tests/gpu_ddfuzz.py [nseeds] [first seed]"""
import os
import sys
import time

import numpy as np

from common import O, Problem, ShardSet, bindings, ics, relerr

B = bindings()
TOL = 1e-11


def make(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 3))
    periodic = int(rng.integers(0, 2))
    unequal = bool(rng.integers(0, 2))
    if kind == 0:
        ic = ics.make_ics(int(rng.integers(4, 13)), gas=True, seed=int(seed), clustered=True,
                          rms_disp=float(rng.uniform(0.2, 2.0)))
    elif kind == 1:
        ic = ics.make_plummer(int(rng.integers(200, 5000)), seed=int(seed),
                              a=float(rng.uniform(0.01, 0.1)),
                              gas_fraction=float(rng.uniform(0.1, 0.6)))
    else:
        n = int(rng.integers(40, 3000))
        ngas = int(rng.integers(0, n))
        pos = rng.random((n, 3))
        typ = np.where(np.arange(n) < ngas, 0, rng.integers(1, 6, n)).astype(np.int32)
        ic = dict(pos=pos, vel=rng.standard_normal((n, 3)), mass=rng.uniform(0.5, 2.0, n) / n,
                  type=typ, ngas=ngas, boxsize=1.0, spacing=1.0 / max(2.0, n ** (1 / 3)),
                  id=np.arange(1, n + 1, dtype=np.uint32), u=np.full(ngas, 0.01))
    pr = Problem(ic=ic, periodic=periodic, unequal=unequal,
                 des_ngb=float(min(33.0, max(ic["ngas"] - 1, 1))))
    return rng, kind, pr


def one(seed):
    rng, kind, pr = make(seed)
    n, ng = pr.n, pr.ngas
    P = int(rng.integers(2, 9))
    work = None
    if rng.random() < 0.3:                      # lopsided cuts: some shards (nearly) empty
        work = rng.random(n) ** 8
    old = 0.2 + 3.0 * rng.random(n)
    # -DMULTIPLEDOMAINS: every third problem gives each shard several pieces of the curve
    domains = int(rng.choice([1, 1, 1, 2, 4])) if os.environ.get("DDFUZZ_DOMAINS", "1") == "1" else 1
    S = ShardSet(pr, P, work=work, fields={"oldacc": old}, domains=domains)
    try:
        sizes = [len(g) for g in S.gid]
        tg = np.arange(n, dtype=np.int32)
        theta = float(rng.choice([0.0, 0.5, 0.8]))
        T = pr.oracle_tree()
        mode = "newton"
        if pr.periodic:
            mode = str(rng.choice(["pair", "two calls"]))
        if mode == "pair":
            S.run.gravity(pr.g_grav(theta), B.WALK_NEWTON_EWALD)
        else:
            S.run.gravity(pr.g_grav(theta), B.WALK_NEWTON)
        oacc, ocost = T.gravity(pr.o_grav(theta), tg, old)
        if mode == "two calls":
            S.run.gravity(pr.g_grav(theta), B.WALK_EWALD)
        if pr.periodic:
            T.gravity_ewald_add(pr.o_grav(theta), O.ewald_table(pr.box), tg, old, oacc, ocost)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), ocost), "gravity counts"
        scale = np.abs(oacc).max() + 1e-300
        assert np.abs(S.get_field(B.F_GRAVACCEL) - oacc).max() < 1e-10 * scale, "gravity"
        sph = ng >= 40
        if sph:
            act = np.arange(ng, dtype=np.int32)
            S.run.density(pr.g_dens())
            od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                           pr.ti_begstep, pr.hsml0)
            assert relerr(S.get_field(B.F_HSML)[:ng], od["hsml"][:ng]) < 1e-9, "hsml"
            assert relerr(S.get_field(B.F_DENSITY), od["density"][:ng]) < 1e-9, "density"
            st = S.each(lambda fp: fp.stats())
            assert sum(s["dens_neighbours"] for s in st) == od["ngb_visits"], "neighbour visits"
            S.each(lambda fp: fp.update_hmax())
            T.update_hmax(act, od["hsml"], od["divvel"])
            S.run.hydro(pr.g_hydro())
            oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                         od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
            st = S.each(lambda fp: fp.stats())
            assert sum(s["hydro_pairs"] for s in st) == oh["npairs"], "pairs"
            want = oh["hydroaccel"][:ng]
            assert np.abs(S.get_field(B.F_HYDROACCEL) - want).max() <= 1e-9 * (np.abs(want).max() + 1e-300), "hydro"
        # a shake and a migration; gravity again on the re-sharded set
        moved = -1
        if rng.random() < 0.6:
            lo, ln = pr.extent[0], pr.extent[2]
            amp = float(rng.choice([0.002, 0.02, 0.1])) * ln
            newpos = pr.ic["pos"] + amp * rng.standard_normal((n, 3))
            if pr.periodic:
                newpos = np.mod(newpos, pr.box)
                newpos[newpos >= pr.box] = 0.0
            # keep everybody inside the domain cube the shards were set up with
            newpos = np.clip(newpos, lo + 1e-9 * ln, lo + ln * (1 - 1e-9))
            S.set_field(B.F_POS, newpos)
            before = S.owner.copy()
            S.migrate()
            moved = int((S.owner != before).sum())
            assert np.array_equal(np.sort(np.concatenate(S.gid)), np.arange(n)), "lost or doubled"
            assert np.array_equal(S.get_field(B.F_POS), newpos), "positions after migration"
            pr.ic["pos"] = newpos
            T2 = pr.oracle_tree()
            S.set_field(B.F_OLDACC, old)
            o2, c2 = T2.gravity(pr.o_grav(theta), tg, old)
            S.run.gravity(pr.g_grav(theta), B.WALK_NEWTON)
            assert np.array_equal(S.get_field(B.F_GRAVCOST), c2), "gravity counts after migration"
            assert np.abs(S.get_field(B.F_GRAVACCEL) - o2).max() < 1e-10 * (np.abs(o2).max() + 1e-300), \
                "gravity after migration"
        return kind, n, ng, P, min(sizes), mode, theta, sph, moved
    finally:
        S.close()


if __name__ == "__main__":
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t0 = time.time()
    bad = 0
    for seed in range(first, first + nseeds):
        try:
            info = one(seed)
            print("seed %d ok %s" % (seed, info), flush=True)
        except AssertionError as e:
            bad += 1
            print("seed %d FAILED: %s" % (seed, e), flush=True)
        except Exception as e:   # noqa: BLE001 -- library errors are findings too
            bad += 1
            print("seed %d ERROR: %s: %s" % (seed, type(e).__name__, str(e)[:300]), flush=True)
    print("%d seeds, %d failures, %.1f s" % (nseeds, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)
