"""Multi-GPU path: Peano-Hilbert domain decomposition with tree-node (locally essential
tree) and ghost exchange, run as several LOGICAL shards one after the other on one GPU -- buffers are
handed across in device memory by ghip_dd_exchange_local, everything else is the code the RCCL path
runs (ghip_dd_* in include/ghip.h).  Checked against the oracle's SINGLE global tree: the reference's
forces do not depend on the number of ranks (domain.c:20-22; SURVEY 8c measured identical
interactions per particle for NTask = 1, 2, 4), so interaction counts, neighbour counts and pair
counts must be exact and the sums agree to summation order."""
import numpy as np
import pytest

from common import O, Problem, ShardSet, bindings, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _gravity_two_passes(pr, S, walk_pair):
    """accel.c:61-68 at Ti_Current == 0: a Barnes-Hut pass gives OldAcc, the second pass uses the
    relative criterion (gravtree.c:396-397).  Returns the oracle's and the shards' results."""
    B = S.B
    n = pr.n
    tg = np.arange(n, dtype=np.int32)
    T = pr.oracle_tree()
    tab = O.ewald_table(pr.box) if pr.periodic else None
    res = []
    oldacc = np.zeros(n)
    for theta in (pr.theta, 0.0):
        oacc, ocost = T.gravity(pr.o_grav(theta), tg, oldacc)
        if pr.periodic:
            T.gravity_ewald_add(pr.o_grav(theta), tab, tg, oldacc, oacc, ocost)
        S.set_field(B.F_OLDACC, oldacc)
        if pr.periodic and walk_pair:
            S.run.gravity(pr.g_grav(theta), B.WALK_NEWTON_EWALD)
        else:
            S.run.gravity(pr.g_grav(theta), B.WALK_NEWTON)
            if pr.periodic:
                S.run.gravity(pr.g_grav(theta), B.WALK_EWALD)
        res.append((oacc, ocost, S.get_field(B.F_GRAVACCEL), S.get_field(B.F_GRAVCOST),
                    S.each(lambda fp: fp.dd_info())))
        S.each(lambda fp: fp.gravity_finish(pr.G))
        oldacc, _ = O.gravity_finish(oacc, pr.G)
        assert relerr(S.get_field(B.F_OLDACC), oldacc) < TOL
    return res


@pytest.mark.parametrize("nshards,periodic,pair", [(3, 1, True), (2, 0, False), (8, 1, False)])
def test_sharded_gravity_equals_the_single_global_tree(nshards, periodic, pair):
    pr = Problem(ng=12, gas=True, periodic=periodic)
    S = ShardSet(pr, nshards)
    try:
        assert sum(len(g) for g in S.gid) == pr.n and min(len(g) for g in S.gid) > 0
        for oacc, ocost, acc, cost, info in _gravity_two_passes(pr, S, pair):
            assert np.array_equal(cost, ocost), "interaction counts differ from the single tree"
            assert relerr(acc, oacc) < TOL
            # something crossed the links, and nobody received the whole remote tree as particles
            assert all(i["let_imported"] > 0 for i in info)
    finally:
        S.close()


@pytest.mark.parametrize("nshards,periodic", [(3, 1), (2, 0), (8, 1)])
def test_sharded_density_and_hydro_equal_the_single_rank_sums(nshards, periodic):
    B = bindings()
    pr = Problem(ng=12, gas=True, periodic=periodic)
    S = ShardSet(pr, nshards)
    try:
        S.each(lambda fp: fp.dd_set_ghost_margin(2.0))    # the first guess of h is poor (common.py)
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)  # the tree of this step
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        T = pr.oracle_tree()
        act = np.arange(pr.ngas, dtype=np.int32)
        od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                       pr.ti_begstep, pr.hsml0)
        T.update_hmax(act, od["hsml"], od["divvel"])
        oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                     od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
        ng = pr.ngas
        for fid, key in ((B.F_HSML, "hsml"), (B.F_NUMNGB, "numngb"), (B.F_DENSITY, "density"),
                         (B.F_DHSMLFAC, "dhsmlfac"), (B.F_DIVVEL, "divvel"),
                         (B.F_CURLVEL, "curlvel"), (B.F_PRESSURE, "pressure")):
            got = S.get_field(fid)[:ng]
            assert relerr(got, od[key][:ng]) < 1e-9, key
        st = S.each(lambda fp: fp.stats())
        assert sum(s["dens_neighbours"] for s in st) == od["ngb_visits"]
        assert sum(s["hydro_pairs"] for s in st) == oh["npairs"]
        ha = S.get_field(B.F_HYDROACCEL)
        assert np.abs(ha - oh["hydroaccel"][:ng]).max() < 1e-9 * np.abs(oh["hydroaccel"]).max()
        assert relerr(S.get_field(B.F_DTENTROPY), oh["dtentropy"][:ng]) < 1e-8
        info = S.each(lambda fp: fp.dd_info())
        assert all(i["ghosts_imported"] > 0 for i in info)
        assert sum(i["ghosts_imported"] for i in info) == sum(i["ghosts_sent"] for i in info)
    finally:
        S.close()


def test_config_c4_128cubed_as_eight_logical_shards():
    """c4 (128^3 DM + 128^3 gas, Peano-Hilbert domain decomposition over 8 GPUs): the whole workload
    as 8 logical shards run one after the other on one GPU, tree nodes and ghosts handed across in
    device memory.  Gravity (relative criterion, Newtonian + Ewald walks) and SPH against the
    oracle's single global tree on a sample of targets spread over all shards: interaction counts
    and neighbour counts exact.  Prints what crossed the links."""
    B = bindings()
    pr = Problem(ng=128, gas=True, periodic=1)
    n, ng = pr.n, pr.ngas
    assert n == 2 * 128 ** 3
    rng = np.random.default_rng(11)
    old = 0.5 + 3.0 * rng.random(n)                    # OldAcc spread over a factor 7
    S = ShardSet(pr, 8, fields={"oldacc": old})
    try:
        sizes = [len(g) for g in S.gid]
        assert sum(sizes) == n and max(sizes) < 1.3 * n / 8
        S.run.gravity(pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
        S.each(lambda fp: fp.dd_set_ghost_margin(1.6))
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        acc, cost = S.get_field(B.F_GRAVACCEL), S.get_field(B.F_GRAVCOST)
        st = S.each(lambda fp: fp.stats())
        info = S.each(lambda fp: fp.dd_info())
        assert int(cost.astype(np.int64).sum()) == sum(s["grav_interactions"] + s["ewald_interactions"]
                                                      for s in st)
        sample = np.sort(rng.choice(n, 1024, replace=False)).astype(np.int32)
        assert len(np.unique(S.owner[sample])) == 8
        T = pr.oracle_tree()
        oacc, ocost = T.gravity(pr.o_grav(0.0), sample, old)
        T.gravity_ewald_add(pr.o_grav(0.0), O.ewald_table(pr.box), sample, old, oacc, ocost)
        assert np.array_equal(cost[sample], ocost)
        assert relerr(acc[sample], oacc) < TOL
        nn = S.get_field(B.F_NUMNGB)
        assert np.all(np.abs(nn - pr.des_ngb) <= pr.max_dev + 1e-9)
        act = sample[sample < ng][:256]
        od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                       pr.ti_begstep, pr.hsml0)
        assert relerr(S.get_field(B.F_DENSITY)[act], od["density"][act]) < TOL
        assert relerr(S.get_field(B.F_HSML)[act], od["hsml"][act]) < TOL
        assert np.abs(nn[act] - od["numngb"][act]).max() < 1e-10
        ha = S.get_field(B.F_HYDROACCEL)
        mg = pr.ic["mass"][:ng]
        assert np.abs((mg[:, None] * ha).sum(axis=0)).max() < 1e-10 * np.abs(mg[:, None] * ha).sum()
        for r, i in enumerate(info):
            print("shard %d: %d particles, %d tree elements imported (%.1f MB sent), %d ghosts "
                  "imported (%.1f MB sent), merged tree %d elements" %
                  (r, sizes[r], i["let_imported"], i["bytes_gravity"] / 1e6, i["ghosts_imported"],
                   i["bytes_density"] / 1e6, i["grav_elements"]))
        assert all(0 < i["let_imported"] < 0.6 * n for i in info)
    finally:
        S.close()
