"""'Next' row N4: the sink (black-hole) neighbour passes and the per-particle part of
cooling_and_starformation on the device, through the C-ABI, against the oracle's restatement
(oracle/gadget_oracle.c: orc_sink_density, orc_blackhole_evaluate, orc_blackhole_swallow,
orc_cooling_and_starformation -- pinned by all-pairs numpy in tests/test_oracle_pins.py).
Marks (SwallowID), counts and iteration counts exact; sums to summation order."""
import numpy as np
import pytest

from common import O, ShardSet, SinkProblem, bindings, relerr

pytestmark = pytest.mark.gpu


def _device(sp):
    B = bindings()
    pr = sp.pr
    fp = pr.device()
    fp.set_field(B.F_HSML, sp.hsml)
    pr.device_tree(fp)
    return B, fp


@pytest.mark.parametrize("periodic", [0, 1])
@pytest.mark.parametrize("dust_only,acc_density", [(1, 1), (0, 1), (0, 0)])
def test_sink_passes_parity(periodic, dust_only, acc_density):
    sp = SinkProblem(ng=10, periodic=periodic, nsink=8, ndust=300)
    pr = sp.pr
    n, ngas = pr.n, pr.ngas
    B, fp = _device(sp)
    gas_density = 0.5 + np.random.default_rng(3).random(ngas)
    fp.set_field(B.F_DENSITY, gas_density)
    over = dict(accretion_of_dust_only=dust_only, accretion_density=acc_density, CritDensity=1.0)
    op, gp = sp.params(O.BhParams, **over), sp.params(B.BhParams, **over)
    mass0 = pr.ic["mass"].copy()
    T = O.Tree(pr.ic["pos"], pr.ic["vel"], mass0, pr.ic["type"], pr.force_soft, hsml=sp.hsml,
               extent=pr.extent)
    # ---- density of the sinks ----
    od = O.sink_density(T, pr.o_dens(), 1.5, sp.sinks, pr.velpred, pr.entropy, sp.hsml)
    gd = fp.sink_density(pr.g_dens(), 1.5, sp.sinks, sp.hsml[sp.sinks])
    assert gd["iterations"] == od["iterations"] >= 1
    assert relerr(gd["hsml"], od["hsml"][sp.sinks]) < 1e-13
    assert np.abs(gd["numngb"] - od["numngb"]).max() < 1e-10
    for k in ("density", "entropy"):
        assert relerr(gd[k], od[k]) < 1e-12, k
    assert np.abs(gd["gasvel"] - od["gasvel"]).max() < 1e-12 * np.abs(od["gasvel"]).max()
    assert relerr(fp.get_field(B.F_HSML)[sp.sinks], od["hsml"][sp.sinks]) < 1e-13
    hs = od["hsml"]
    # ---- marking + feedback ----
    fp.sink_reset()
    osw, oinj = O.blackhole_evaluate(T, op, sp.sinks, sp.ids, hs, pr.timebin, sp.mdot,
                                     od["density"], gas_density, np.zeros(n, np.uint32),
                                     np.zeros(ngas))
    fp.blackhole_evaluate(gp, sp.sinks, sp.ids[sp.sinks], sp.mdot, gd["density"])
    gsw, ginj = fp.sink_marks()
    assert np.array_equal(gsw, osw) and (osw > 0).sum() > 10
    assert np.abs(ginj - oinj).max() <= 1e-12 * np.abs(oinj).max() and np.abs(oinj).max() > 0
    # ---- swallowing ----
    oo = O.blackhole_swallow(T, op, sp.sinks, sp.ids, hs, osw, sp.bh_mass)
    go = fp.blackhole_swallow(gp, sp.sinks, sp.ids[sp.sinks], sp.bh_mass[sp.sinks])
    assert np.array_equal(go["counts"], oo["counts"]) and oo["counts"].sum() == (osw > 0).sum()
    for k in ("acc_mass", "acc_bhmass", "acc_dustmass"):
        assert np.abs(go[k] - oo[k]).max() <= 1e-13 * max(np.abs(oo[k]).max(), 1e-300), k
    assert np.abs(go["acc_momentum"] - oo["acc_momentum"]).max() <= \
        1e-13 * max(np.abs(oo["acc_momentum"]).max(), 1e-300)
    assert np.array_equal(fp.get_field(B.F_MASS), T.mass)          # victims at zero, nobody else touched
    assert np.array_equal(go["bh_mass"], oo["bh_mass"][sp.sinks])
    fp.close()


def _run_sink_op(S, op, make):
    """one sink operation on all shards; make(r) -> (DdSinkArgs, arrays) of shard r"""
    built = [make(r) for r in range(S.P)]
    S.run.run(op, [b[0] for b in built])
    return [b[1] for b in built]


@pytest.mark.parametrize("nshards,periodic,dust_only,acc_density",
                         [(3, 1, 1, 1), (8, 0, 0, 1), (5, 1, 0, 0)])
def test_sink_passes_on_shards(nshards, periodic, dust_only, acc_density):
    """N4 x 8(e): the sinks of all shards are evaluated by every shard against its own particles
    (GHIP_DD_SINK_DENSITY / _BH_EVALUATE / _BH_SWALLOW); the results equal the oracle's single
    global pass -- marks, counts and h-iteration counts exactly."""
    sp = SinkProblem(ng=10, periodic=periodic, nsink=8, ndust=300)
    pr = sp.pr
    n, ngas = pr.n, pr.ngas
    B = bindings()
    S = ShardSet(pr, nshards, fields=dict(hsml=sp.hsml))
    try:
        where = S.locate(sp.sinks)
        assert sum(len(w[0]) for w in where) == len(sp.sinks)
        gas_density = 0.5 + np.random.default_rng(3).random(ngas)
        over = dict(accretion_of_dust_only=dust_only, accretion_density=acc_density, CritDensity=1.0)
        op, gp = sp.params(O.BhParams, **over), sp.params(B.BhParams, **over)
        T = O.Tree(pr.ic["pos"], pr.ic["vel"], pr.ic["mass"].copy(), pr.ic["type"], pr.force_soft,
                   hsml=sp.hsml, extent=pr.extent)
        od = O.sink_density(T, pr.o_dens(), 1.5, sp.sinks, pr.velpred, pr.entropy, sp.hsml)
        # the trees of this step
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
        S.run.density(pr.g_dens())
        gd = pr.g_dens()
        res = _run_sink_op(S, B.DD_SINK_DENSITY, lambda r: B.dd_sink_args(
            where[r][0], dens=gd, ngb_factor=1.5, hsml=sp.hsml[sp.sinks][where[r][1]]))
        ns = len(sp.sinks)
        got = {k: np.zeros((ns, 3) if k == "gasvel" else ns) for k in
               ("hsml", "numngb", "density", "entropy", "gasvel")}
        for (loc, pos), a in zip(where, res):
            for k in got:
                got[k][pos] = a[k]
        assert relerr(got["hsml"], od["hsml"][sp.sinks]) < 1e-13
        assert np.abs(got["numngb"] - od["numngb"]).max() < 1e-10
        for k in ("density", "entropy"):
            assert relerr(got[k], od[k]) < 1e-12, k
        assert np.abs(got["gasvel"] - od["gasvel"]).max() < 1e-12 * np.abs(od["gasvel"]).max()
        assert relerr(S.get_field(B.F_HSML)[sp.sinks], od["hsml"][sp.sinks]) < 1e-13
        hs = od["hsml"]
        # ---- marking + feedback ----
        S.set_field(B.F_DENSITY, gas_density)
        S.each(lambda fp: fp.sink_reset())
        osw, oinj = O.blackhole_evaluate(T, op, sp.sinks, sp.ids, hs, pr.timebin, sp.mdot,
                                         od["density"], gas_density, np.zeros(n, np.uint32),
                                         np.zeros(ngas))
        _run_sink_op(S, B.DD_BH_EVALUATE, lambda r: B.dd_sink_args(
            where[r][0], sink_ids=sp.ids[sp.sinks][where[r][1]], bh=gp, mdot=sp.mdot[where[r][1]],
            bh_density=got["density"][where[r][1]]))
        gsw, ginj = np.zeros(n, np.uint32), np.zeros(ngas)
        for r, fp in enumerate(S.fp):
            sw, inj = fp.sink_marks()
            gsw[S.gid[r]] = sw
            ginj[S.gid[r][:S.ngas[r]]] = inj
        assert np.array_equal(gsw, osw) and (osw > 0).sum() > 10
        assert np.abs(ginj - oinj).max() <= 1e-12 * np.abs(oinj).max() and np.abs(oinj).max() > 0
        # ---- swallowing ----
        oo = O.blackhole_swallow(T, op, sp.sinks, sp.ids, hs, osw, sp.bh_mass)
        res = _run_sink_op(S, B.DD_BH_SWALLOW, lambda r: B.dd_sink_args(
            where[r][0], sink_ids=sp.ids[sp.sinks][where[r][1]], bh=gp,
            sink_bh_mass=sp.bh_mass[sp.sinks][where[r][1]]))
        go = {k: np.zeros((ns, 3) if k == "acc_momentum" else ns) for k in
              ("bh_mass", "acc_mass", "acc_bhmass", "acc_dustmass", "acc_momentum")}
        counts = np.zeros(3, np.int64)
        for (loc, pos), a in zip(where, res):
            for k in go:
                go[k][pos] = a[k]
            counts += a["counts"]
        assert np.array_equal(counts, oo["counts"]) and oo["counts"].sum() == (osw > 0).sum()
        for k in ("acc_mass", "acc_bhmass", "acc_dustmass"):
            assert np.abs(go[k] - oo[k]).max() <= 1e-13 * max(np.abs(oo[k]).max(), 1e-300), k
        assert np.abs(go["acc_momentum"] - oo["acc_momentum"]).max() <= \
            1e-13 * max(np.abs(oo["acc_momentum"]).max(), 1e-300)
        assert np.array_equal(S.get_field(B.F_MASS), T.mass)      # victims at zero, nobody else touched
        assert np.array_equal(go["bh_mass"], oo["bh_mass"][sp.sinks])
    finally:
        S.close()


def test_cooling_and_starformation_parity():
    sp = SinkProblem(ng=10, periodic=1)
    pr = sp.pr
    ngas = pr.ngas
    B, fp = _device(sp)
    rng = np.random.default_rng(9)
    dens = 0.2 + 3.0 * rng.random(ngas)
    inj = np.where(rng.random(ngas) < 0.3, 1e-3 * rng.random(ngas), 0.0)
    inj[5] = 1.0e9                                                 # runs into the 5e9 K ceiling
    mass = pr.ic["mass"].copy()
    mass[7] = 0.0                                                  # a swallowed particle
    dte = pr.dtentropy.copy()
    dte[11] = -1.0e6                                               # runs into the -A/(2 dt) floor
    fp.set_field(B.F_MASS, mass)
    fp.set_field(B.F_DENSITY, dens)
    fp.set_field(B.F_DTENTROPY, dte)
    fp.set_sink_marks(injected=inj)
    act = np.sort(rng.choice(ngas, ngas // 2, replace=False)).astype(np.int32)
    fp.set_active(act)
    crit, minegy, u2t = 2.5, 1e-4, 3.0e4
    od, oi, of = O.cooling_and_starformation(act, ngas, pr.ic["type"], mass, pr.timebin, pr.timebase,
                                             crit, minegy, u2t, dens, pr.entropy, dte, inj)
    gf = fp.cooling_and_starformation(pr.timebase, crit, minegy, u2t)
    assert np.array_equal(gf, of) and 0 < of.sum() < len(act)
    # dA/dt = (A_new - A) / dt is a difference of nearly equal numbers: agreement is to rounding of
    # A / dt (pow() of the two sides differs in the last bit)
    dtmin = (1 << int(pr.timebin[:ngas].min())) * pr.timebase
    assert np.abs(fp.get_field(B.F_DTENTROPY) - od).max() < 1e-13 * np.abs(pr.entropy).max() / dtmin
    _, gi = fp.sink_marks()
    assert np.array_equal(gi, oi)
    rest = np.setdiff1d(np.arange(ngas), act)
    assert np.array_equal(fp.get_field(B.F_DTENTROPY)[rest], dte[rest])      # inactive untouched
    fp.close()
