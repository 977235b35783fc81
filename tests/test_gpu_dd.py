"""Multi-GPU path: Peano-Hilbert domain decomposition with tree-node (locally essential
tree) and ghost exchange, run as several LOGICAL shards one after the other on one GPU -- buffers are
handed across in device memory by ghip_dd_exchange_local, everything else is the code the RCCL path
runs (ghip_dd_* in include/ghip.h).  Checked against the oracle's SINGLE global tree: the reference's
forces do not depend on the number of ranks (domain.c:20-22; SURVEY 8c measured identical
interactions per particle for NTask = 1, 2, 4), so interaction counts, neighbour counts and pair
counts must be exact and the sums agree to summation order."""
import numpy as np
import pytest

from common import O, Problem, ShardSet, SinkProblem, bindings, relerr, sampled_hydro_check

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
            if key in ("divvel", "curlvel"):
                # sums of terms of both signs that cancel: measured against the field's scale, not
                # element by element (an element near zero has no relative accuracy to lose)
                assert np.abs(got - od[key][:ng]).max() < TOL * np.abs(od[key][:ng]).max(), key
            else:
                assert relerr(got, od[key][:ng]) < TOL, key
        st = S.each(lambda fp: fp.stats())
        assert sum(s["dens_neighbours"] for s in st) == od["ngb_visits"]
        assert sum(s["hydro_pairs"] for s in st) == oh["npairs"]
        ha = S.get_field(B.F_HYDROACCEL)
        assert np.abs(ha - oh["hydroaccel"][:ng]).max() < TOL * np.abs(oh["hydroaccel"]).max()
        de = S.get_field(B.F_DTENTROPY)
        assert np.abs(de - oh["dtentropy"][:ng]).max() < TOL * np.abs(oh["dtentropy"][:ng]).max()
        info = S.each(lambda fp: fp.dd_info())
        assert all(i["ghosts_imported"] > 0 for i in info)
        assert sum(i["ghosts_imported"] for i in info) == sum(i["ghosts_sent"] for i in info)
    finally:
        S.close()


@pytest.mark.parametrize("nshards", [3, 8])
def test_smoothing_lengths_that_triple_in_one_call_reselect_their_ghosts(nshards):
    """The first density() of a run starts from a poor guess and grows h by 1.26 per pass
    (density.c:610-642; init.c:740-741): here every smoothing length starts at a quarter of the usual
    first guess and must grow two- to more than threefold within ONE call.  The reference re-exports a target at every
    iteration; the shards select their ghosts once per call with the default padding (1.3), notice
    -- all of them together, on the all-gathered growth factors -- that the search radii outgrew it,
    select again with the radius that was needed and repeat the call from the smoothing lengths it
    started with.  Result: the single global tree's, iteration and neighbour-visit counts of the
    repeated attempt included."""
    B = bindings()
    pr = Problem(ng=12, gas=True, periodic=1)
    ng = pr.ngas
    h_start = pr.hsml0.copy()
    h_start[:ng] = pr.hsml0[:ng] / 4.0
    S = ShardSet(pr, nshards)
    try:
        S.set_field(B.F_HSML, h_start)
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)  # the tree of this step
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        T = O.Tree(pr.ic["pos"], pr.ic["vel"], pr.ic["mass"], pr.ic["type"], pr.force_soft, hsml=h_start,
                   extent=pr.extent)
        act = np.arange(ng, dtype=np.int32)
        od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                       pr.ti_begstep, h_start)
        growth = od["hsml"][:ng] / h_start[:ng]
        assert growth.min() > 1.9 and growth.max() > 3.0                 # they double to more than triple
        T.update_hmax(act, od["hsml"], od["divvel"])
        oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                     od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
        for fid, key in ((B.F_HSML, "hsml"), (B.F_NUMNGB, "numngb"), (B.F_DENSITY, "density")):
            assert relerr(S.get_field(fid)[:ng], od[key][:ng]) < TOL, key
        st = S.each(lambda fp: fp.stats())
        assert sum(s["dens_neighbours"] for s in st) == od["ngb_visits"]
        assert max(s["dens_iterations"] for s in st) == od["iterations"]
        assert sum(s["hydro_pairs"] for s in st) == oh["npairs"]
        info = S.each(lambda fp: fp.dd_info())
        assert max(i["hsml_growth_e6"] for i in info) > 2.0e6      # beyond the default padding of 1.3
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
        # hydro_force of the sampled targets (from all shards) against the oracle's hydro_evaluate
        assert sampled_hydro_check(pr, T, S.get_field, act) > 20 * len(act)
        for r, i in enumerate(info):
            print("shard %d: %d particles, %d tree elements imported (%.1f MB sent), %d ghosts "
                  "imported (%.1f MB sent), merged tree %d elements" %
                  (r, sizes[r], i["let_imported"], i["bytes_gravity"] / 1e6, i["ghosts_imported"],
                   i["bytes_density"] / 1e6, i["grav_elements"]))
        assert all(0 < i["let_imported"] < 0.6 * n for i in info)
    finally:
        S.close()


def test_config_c5_256cubed_with_sinks_as_eight_logical_shards():
    """c5 (256^3 DM + 256^3 gas = 33.5 million particles with 300 sinks and 3000 dust grains, 8 GPUs):
    the whole workload as 8 logical shards on one GPU.  One force step -- Newtonian + Ewald walks under
    the relative criterion, density, hydro -- and the sink passes of blackhole.c on the shards, against
    the oracle's single global tree: gravity on a sample of targets from all shards (interaction counts
    exact), SPH through the neighbour window and momentum balance, the 300 sinks' h, marks and
    swallow counts against the oracle's global passes (marks and counts exact)."""
    B = bindings()
    sp = SinkProblem(ng=256, periodic=1, nsink=300, ndust=3000)
    pr = sp.pr
    n, ng = pr.n, pr.ngas
    assert n == 2 * 256 ** 3
    rng = np.random.default_rng(12)
    old = np.full(n, 2.0)
    S = ShardSet(pr, 8, fields={"oldacc": old})
    try:
        sizes = [len(g) for g in S.gid]
        assert sum(sizes) == n and max(sizes) < 1.3 * n / 8
        S.run.gravity(pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        acc, cost = S.get_field(B.F_GRAVACCEL), S.get_field(B.F_GRAVCOST)
        st = S.each(lambda fp: fp.stats())
        info = S.each(lambda fp: fp.dd_info())
        assert int(cost.astype(np.int64).sum()) == sum(s["grav_interactions"] + s["ewald_interactions"]
                                                      for s in st)
        nn = S.get_field(B.F_NUMNGB)
        assert np.all(np.abs(nn - pr.des_ngb) <= pr.max_dev + 1e-9)
        ha = S.get_field(B.F_HYDROACCEL)
        mg = pr.ic["mass"][:ng]
        assert np.abs((mg[:, None] * ha).sum(axis=0)).max() < 1e-10 * np.abs(mg[:, None] * ha).sum()
        sample = np.sort(rng.choice(n, 512, replace=False)).astype(np.int32)
        assert len(np.unique(S.owner[sample])) == 8
        T = pr.oracle_tree()
        oacc, ocost = T.gravity(pr.o_grav(0.0), sample, old)
        T.gravity_ewald_add(pr.o_grav(0.0), O.ewald_table(pr.box), sample, old, oacc, ocost)
        assert np.array_equal(cost[sample], ocost)
        assert relerr(acc[sample], oacc) < TOL
        # density() and hydro_force() of sampled gas targets from all shards against the oracle
        act = sample[sample < ng][:128]
        od0 = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin, pr.ti_begstep,
                        pr.hsml0)
        assert relerr(S.get_field(B.F_DENSITY)[act], od0["density"][act]) < TOL
        assert relerr(S.get_field(B.F_HSML)[act], od0["hsml"][act]) < TOL
        assert sampled_hydro_check(pr, T, S.get_field, act) > 20 * len(act)
        # ---- the sinks, spread over the shards ----
        where = S.locate(sp.sinks)
        assert sum(len(w[0]) for w in where) == 300 and sum(len(w[0]) > 0 for w in where) >= 4
        hs = S.get_field(B.F_HSML)
        hs[sp.sinks] = sp.hsml[sp.sinks]
        gas_density = S.get_field(B.F_DENSITY)
        par = dict(accretion_of_dust_only=0, CritDensity=float(np.median(gas_density)))
        T.hsml = hs
        od = O.sink_density(T, pr.o_dens(), 1.5, sp.sinks, pr.velpred, pr.entropy, hs)
        gd, gp = pr.g_dens(), sp.params(B.BhParams, **par)

        def run(op, make):
            built = [make(r) for r in range(8)]
            S.run.run(op, [b[0] for b in built])
            return [b[1] for b in built]

        res = run(B.DD_SINK_DENSITY, lambda r: B.dd_sink_args(
            where[r][0], dens=gd, ngb_factor=1.5, hsml=hs[sp.sinks][where[r][1]]))
        gh, grho = np.zeros(300), np.zeros(300)
        for (loc, pos), a in zip(where, res):
            gh[pos], grho[pos] = a["hsml"], a["density"]
        assert relerr(gh, od["hsml"][sp.sinks]) < 1e-13
        assert relerr(grho, od["density"]) < TOL
        S.each(lambda fp: fp.sink_reset())
        osw, oinj = O.blackhole_evaluate(T, sp.params(O.BhParams, **par), sp.sinks, sp.ids, od["hsml"],
                                         pr.timebin, sp.mdot, od["density"], gas_density,
                                         np.zeros(n, np.uint32), np.zeros(ng))
        run(B.DD_BH_EVALUATE, lambda r: B.dd_sink_args(
            where[r][0], sink_ids=sp.ids[sp.sinks][where[r][1]], bh=gp, mdot=sp.mdot[where[r][1]],
            bh_density=grho[where[r][1]]))
        gsw, ginj = np.zeros(n, np.uint32), np.zeros(ng)
        for r, fp in enumerate(S.fp):
            sw, inj = fp.sink_marks()
            gsw[S.gid[r]] = sw
            ginj[S.gid[r][:S.ngas[r]]] = inj
        assert np.array_equal(gsw, osw) and (osw > 0).sum() > 300
        assert np.abs(ginj - oinj).max() <= 1e-11 * np.abs(oinj).max()
        oo = O.blackhole_swallow(T, sp.params(O.BhParams, **par), sp.sinks, sp.ids, od["hsml"], osw,
                                 sp.bh_mass)
        res = run(B.DD_BH_SWALLOW, lambda r: B.dd_sink_args(
            where[r][0], sink_ids=sp.ids[sp.sinks][where[r][1]], bh=gp,
            sink_bh_mass=sp.bh_mass[sp.sinks][where[r][1]]))
        counts, gm = np.zeros(3, np.int64), np.zeros(300)
        for (loc, pos), a in zip(where, res):
            counts += a["counts"]
            gm[pos] = a["acc_mass"]
        assert np.array_equal(counts, oo["counts"])
        assert np.abs(gm - oo["acc_mass"]).max() <= 1e-13 * oo["acc_mass"].max()
        assert np.array_equal(S.get_field(B.F_MASS), T.mass)
        for r, i in enumerate(info):
            print("shard %d: %d particles (%d sinks), %d tree elements imported (%.1f MB sent), %d "
                  "ghosts imported (%.1f MB sent), merged tree %d elements" %
                  (r, sizes[r], len(where[r][0]), i["let_imported"], i["bytes_gravity"] / 1e6,
                   i["ghosts_imported"], i["bytes_density"] / 1e6, i["grav_elements"]))
    finally:
        S.close()


def test_a_particle_outside_the_domain_cube_is_refused_by_every_shard():
    """A non-periodic run whose particle has drifted out of the cube given to ghip_dd_set_domain (the
    reference recomputes the extent at every decomposition, domain.c:1972-2014): its integer
    coordinates are clamped to the cube, so the migration sends it to the owner of the boundary cell
    instead of to wherever a garbage key points -- nobody is lost -- and the next force computation is
    refused by ALL shards together (the status record behind the group tables), the shard that holds
    the particle saying why."""
    B = bindings()
    pr = Problem(ng=8, gas=True, periodic=0)
    n = pr.n
    S = ShardSet(pr, 3)
    try:
        pos = pr.ic["pos"].copy()
        far = int(np.argmax(pos[:, 0]))
        pos[far, 0] = pr.extent[0][0] + 1.2 * pr.extent[2]     # 20 % of the cube's length beyond its face
        S.set_field(B.F_POS, pos)
        S.migrate()
        assert np.array_equal(np.sort(np.concatenate(S.gid)), np.arange(n))   # nobody lost, nobody doubled
        with pytest.raises(B.GhipError) as e:
            S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
        assert "domain cube" in str(e.value) or "reported an error" in str(e.value)
    finally:
        S.close()


def test_migration_moves_every_field_with_its_particle():
    """GHIP_DD_MIGRATE (domain_exchange, domain.c:665-1060): after the particles have moved, every
    one lies in the key range of the shard that holds it, every resident field has followed its
    particle, gas stays in front -- and the forces of the re-sharded set equal the single tree's."""
    B = bindings()
    pr = Problem(ng=10, gas=True, periodic=1)
    n, ng = pr.n, pr.ngas
    S = ShardSet(pr, 4)
    try:
        rng = np.random.default_rng(21)
        tag = {B.F_VEL: pr.ic["vel"], B.F_MASS: pr.ic["mass"], B.F_OLDACC: rng.random(n),
               B.F_GRAVACCEL: rng.standard_normal((n, 3)), B.F_GRAVCOST: rng.integers(1, 900, n).astype(np.int32),
               B.F_TI_BEGSTEP: rng.integers(0, 64, n).astype(np.int32), B.F_TIMEBIN: pr.timebin,
               B.F_HSML: pr.hsml0, B.F_VELPRED: pr.velpred, B.F_ENTROPY: pr.entropy,
               B.F_DENSITY: 1 + rng.random(ng), B.F_HYDROACCEL: rng.standard_normal((ng, 3)),
               B.F_MAXSIGNALVEL: rng.random(ng)}
        for fid, arr in tag.items():
            S.set_field(fid, arr)
        # a sizeable shake: a quarter of the particles change shard
        newpos = np.mod(pr.ic["pos"] + 0.08 * rng.standard_normal((n, 3)), 1.0)
        newpos[newpos >= 1.0] = 0.0
        S.set_field(B.F_POS, newpos)
        # positions changed: a fresh extent, as domain_Decomposition takes one (domain.c:1972-2014) --
        # a particle outside the cube of the old one would be refused
        ext2 = O.domain_extent(newpos)
        S.each(lambda fp: fp.dd_set_domain(ext2[0], ext2[1], ext2[2], pr.force_soft))
        before = S.owner.copy()
        S.migrate()
        moved = int((S.owner != before).sum())
        info = S.each(lambda fp: fp.dd_info())
        assert moved > n // 10 and moved == sum(i["migrated_out"] for i in info)
        assert sum(i["migrated_out"] for i in info) == sum(i["migrated_in"] for i in info)
        allid = np.sort(np.concatenate(S.gid))
        assert np.array_equal(allid, np.arange(n))                       # nobody lost, nobody doubled
        for r, fp in enumerate(S.fp):
            k = fp.dd_keys()
            assert np.all((k >= S.splits[r]) & (k < S.splits[r + 1]))
            t = fp.get_field(B.F_TYPE)
            assert np.all(t[:S.ngas[r]] == 0) and np.all(t[S.ngas[r]:] != 0)
            assert np.all(S.gid[r][:S.ngas[r]] < ng) and np.all(S.gid[r][S.ngas[r]:] >= ng)
        assert np.array_equal(S.get_field(B.F_POS), newpos)
        for fid, arr in tag.items():
            assert np.array_equal(S.get_field(fid), arr), fid
        # and the path runs on the re-sharded set
        pr2 = Problem(ng=10, gas=True, periodic=1)
        pr2.ic["pos"] = newpos
        pr2.extent = ext2
        T = pr2.oracle_tree()
        tg = np.arange(n, dtype=np.int32)
        S.set_field(B.F_OLDACC, np.zeros(n))
        oacc, ocost = T.gravity(pr2.o_grav(pr.theta), tg, np.zeros(n))
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), ocost)
        assert relerr(S.get_field(B.F_GRAVACCEL), oacc) < TOL
    finally:
        S.close()


def test_rccl_path_single_rank():
    """The RCCL entry points on the one GPU of this box: a communicator of one rank (RCCL refuses
    two ranks on one device), the whole gravity + SPH sequence through ghip_dd_run -- i.e.
    ncclCommInitRank, ncclAllGather and the (empty) grouped send/receive are really called -- against
    the oracle."""
    B = bindings()
    pr = Problem(ng=8, gas=True, periodic=1)
    fp = pr.device()
    fp.dd_init(0, 1)
    fp.dd_set_domain(pr.extent[0], pr.extent[1], pr.extent[2], pr.force_soft)
    fp.dd_set_splits(np.array([0, 1 << 63], np.uint64))
    assert B.dd_rccl_library().endswith(("librccl.so.1", "librccl.so"))
    fp.dd_rccl_connect(B.dd_rccl_unique_id())
    fp.set_field(B.F_OLDACC, np.zeros(pr.n))
    fp.dd_run(B.DD_MIGRATE, None)
    fp.dd_run(B.DD_GRAVITY, pr.g_grav(pr.theta), B.WALK_NEWTON_EWALD)
    fp.dd_run(B.DD_DENSITY, pr.g_dens())
    fp.update_hmax()
    fp.dd_run(B.DD_HYDRO, pr.g_hydro())
    T = pr.oracle_tree()
    tg = np.arange(pr.n, dtype=np.int32)
    oacc, ocost = T.gravity(pr.o_grav(pr.theta), tg, np.zeros(pr.n))
    T.gravity_ewald_add(pr.o_grav(pr.theta), O.ewald_table(pr.box), tg, np.zeros(pr.n), oacc, ocost)
    assert np.array_equal(fp.get_field(B.F_GRAVCOST), ocost)
    assert relerr(fp.get_field(B.F_GRAVACCEL), oacc) < TOL
    act = np.arange(pr.ngas, dtype=np.int32)
    od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                   pr.ti_begstep, pr.hsml0)
    assert relerr(fp.get_field(B.F_DENSITY), od["density"][:pr.ngas]) < TOL
    fp.close()


@pytest.mark.parametrize("variant", ["unequal", "adaptive", "shortrange", "subset", "tiny", "plummer"])
def test_sharded_gravity_variants(variant):
    """The locally essential trees under the reference's other opening rules and edge cases: mixed
    softenings (forcetree.c:2108-2124) and ADAPTIVE_GRAVSOFT_FORGAS (:2125-2139), the short-range walk
    of a TreePM build, an active subset (sub-step: inactive particles are sources only), more shards
    than a tiny problem can fill (empty shards take part in every exchange), a strongly clustered
    non-periodic set.  Counts exact against the single tree every time."""
    B = bindings()
    kw = dict(ng=10, gas=True, periodic=1)
    nsh = 4
    if variant == "unequal":
        kw["unequal"] = True
    if variant == "tiny":
        kw = dict(ng=3, gas=True, periodic=1)          # 54 particles on 8 shards
        nsh = 8
    if variant == "plummer":
        from common import ics
        kw = dict(periodic=0, ic=ics.make_plummer(3000, gas_fraction=0.3))
        nsh = 5
    pr = Problem(**kw)
    n = pr.n
    rng = np.random.default_rng(17)
    old = 0.3 + 2.0 * rng.random(n)
    S = ShardSet(pr, nsh, fields={"oldacc": old})
    try:
        T = pr.oracle_tree()
        tg = np.arange(n, dtype=np.int32)
        walk, okind, rc = B.WALK_NEWTON, "newton", {}
        if variant == "adaptive":
            S.each(lambda fp: fp.set_adaptive_gravsoft(True))
            T.adaptive_gravsoft()
        if variant == "shortrange":
            asmth = 1.25 * pr.box / 16
            rc = dict(rcut=4.5 * asmth, asmth=asmth)
            walk, okind = B.WALK_SHORTRANGE, "shortrange"
        if variant == "subset":
            tg = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
            for r, fp in enumerate(S.fp):
                loc = np.nonzero(np.isin(S.gid[r], tg))[0].astype(np.int32)
                fp.set_active(loc)
        for theta in (pr.theta, 0.0):
            if rc:
                oacc, ocost = T.gravity(pr.o_grav(theta, **rc), tg, old, kind=okind)
                S.run.gravity(pr.g_grav(theta, rc["rcut"], rc["asmth"]), walk)
            else:
                oacc, ocost = T.gravity(pr.o_grav(theta), tg, old)
                S.run.gravity(pr.g_grav(theta), walk)
            assert np.array_equal(S.get_field(B.F_GRAVCOST)[tg], ocost), (variant, theta)
            assert relerr(S.get_field(B.F_GRAVACCEL)[tg], oacc) < TOL
        if variant == "tiny":
            assert min(len(g) for g in S.gid) <= 7
    finally:
        S.close()


@pytest.mark.parametrize("variant", ["subset", "tiny", "nonperiodic-clustered"])
def test_sharded_sph_variants(variant):
    """Ghost exchange on a sub-step (active subset: inactive gas particles are neighbours with their
    current state), with nearly empty shards, and on a clustered non-periodic set."""
    B = bindings()
    kw = dict(ng=10, gas=True, periodic=1)
    nsh = 4
    if variant == "tiny":
        kw, nsh = dict(ng=4, gas=True, periodic=1), 8
    if variant == "nonperiodic-clustered":
        from common import ics
        kw, nsh = dict(periodic=0, ic=ics.make_plummer(4000, gas_fraction=0.5)), 3
    pr = Problem(**kw)
    ng = pr.ngas
    S = ShardSet(pr, nsh)
    try:
        act = np.arange(ng, dtype=np.int32)
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
        if variant == "subset":
            # converge everybody first (full step), then re-evaluate a third of them
            S.run.density(pr.g_dens())
            hs = S.get_field(B.F_HSML)
            pr.hsml0 = hs.copy()
            act = np.sort(np.random.default_rng(2).choice(ng, ng // 3, replace=False)).astype(np.int32)
            for r, fp in enumerate(S.fp):
                fp.set_active(np.nonzero(np.isin(S.gid[r], act))[0].astype(np.int32))
            S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON)
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        T = pr.oracle_tree(hsml=pr.hsml0)
        full = np.arange(ng, dtype=np.int32)
        if variant == "subset":
            # the oracle's state of the inactive particles = the converged full step
            od0 = T.density(pr.o_dens(), full, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                            pr.ti_begstep, pr.hsml0)
            T.update_hmax(full, od0["hsml"], od0["divvel"])
            od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                           pr.ti_begstep, od0["hsml"])
            for k in ("hsml", "density", "pressure", "dhsmlfac", "divvel", "curlvel"):
                od0[k][act] = od[k][act]
            od = od0
        else:
            od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                           pr.ti_begstep, pr.hsml0)
        T.update_hmax(full, od["hsml"], od["divvel"])
        oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                     od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
        assert relerr(S.get_field(B.F_DENSITY)[act], od["density"][act]) < TOL
        assert relerr(S.get_field(B.F_HSML)[act], od["hsml"][act]) < TOL
        st = S.each(lambda fp: fp.stats())
        assert sum(s["hydro_pairs"] for s in st) == oh["npairs"]
        ha = S.get_field(B.F_HYDROACCEL)[act]
        assert np.abs(ha - oh["hydroaccel"][act]).max() < TOL * np.abs(oh["hydroaccel"][act]).max()
    finally:
        S.close()


@pytest.mark.parametrize("nshards,ng,pmgrid", [(3, 16, 32), (8, 32, 128)])
def test_treepm_on_shards(nshards, ng, pmgrid):
    """TreePM across shards (c3's force split on c4's decomposition): GHIP_DD_PM -- every shard
    deposits its particles, the meshes are all-gathered and added in rank order, every shard solves
    the identical mesh and interpolates for its own particles -- against the numpy restatement of
    pmforce_periodic; the short-range walk on the same shards against the oracle's (counts exact);
    and their sum against the exact periodic force (direct sum + Ewald) on a sample."""
    B = bindings()
    pr = Problem(ng=ng, gas=True, periodic=1)
    n = pr.n
    G = 43007.1
    S = ShardSet(pr, nshards)
    try:
        asmth = 1.25 * pr.box / pmgrid
        S.run.run(B.DD_PM, B.PmParams(pmgrid, pr.box, G, asmth))
        got = S.get_field(B.F_GRAVPM)
        want = O.pm_periodic(pr.ic["pos"], pr.ic["mass"], pr.box, G, pmgrid)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() < 1e-11 * scale
        info = S.each(lambda fp: fp.dd_info())
        assert all(fp.stats()["ms_pm"] > 0 for fp in S.fp)
        T = pr.oracle_tree()
        tg = np.arange(n, dtype=np.int32)
        old = np.zeros(n)
        oacc, ocost = T.gravity(pr.o_grav(0.3, rcut=4.5 * asmth, asmth=asmth), tg, old, kind="shortrange")
        S.run.gravity(pr.g_grav(0.3, 4.5 * asmth, asmth), B.WALK_SHORTRANGE)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), ocost)
        assert relerr(S.get_field(B.F_GRAVACCEL), oacc) < TOL
        if pmgrid == 128:
            total = G * S.get_field(B.F_GRAVACCEL) + got
            sample = np.sort(np.random.default_rng(1).choice(n, 128, replace=False)).astype(np.int32)
            d = G * O.gravity_direct(pr.ic["pos"], pr.ic["mass"], pr.ic["type"], pr.force_soft, sample,
                                     periodic=1, boxsize=pr.box, ewald_tab=O.ewald_table(pr.box))
            err = np.linalg.norm(total[sample] - d, axis=1) / np.linalg.norm(d, axis=1)
            assert np.median(err) < 0.01 and np.percentile(err, 95) < 0.05
        del info
    finally:
        S.close()


@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_ranks_share_one_gpu_through_the_host_transport(nranks):
    """The multi-process form of the domain-decomposed path, as the driver launches it
    (torch.distributed.run, one rank per process): several ranks on this box's one GPU, which RCCL refuses
    (duplicate device), so every exchange is staged through the host and gloo
    (ghip_dd_exchange_host) -- the same state machine, migration included.  Checks the bench
    contract's JSON line and that no particle was lost."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_TRANSPORT="host", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", str(nranks), "--steps", "3", "--warmup", "1", "--ng", "24"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == nranks and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["n_particles"] == 2 * 24 ** 3
    assert "domain decomposition" in out["config"]["parallelism"]
    x = out["exchange_per_step_rank0"]
    assert x["bytes_sent_tree_nodes"] > 0 and x["bytes_sent_ghosts"] > 0
    assert out["work_per_step_rank0"]["grav_interactions"] > 0


@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_ranks_exchange_through_the_rccl_entry_points(nranks):
    """The RCCL branch of the exchanges -- ncclAllGather of the counts, then ONE group of ncclSend /
    ncclRecv (ghip_dd_exchange, csrc/ghip_comm.hip) -- with more than one rank.  The real RCCL refuses
    several ranks on one device, and this box has one: the nine entry points the library binds are
    supplied by tests/mock_rccl (shared memory between the rank processes), selected with
    GHIP_RCCL_LIB.  Same bench, same checks as over the host transport, and the run must report that
    it went through the RCCL entry points -- and agree, interaction for interaction, with the run
    staged through the host."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock = os.path.join(root, "tests", "mock_rccl", "librccl_mock.so")
    assert os.path.exists(mock), "build tests/mock_rccl first (__graft_entry__.build())"
    res = {}
    for transport in ("rccl", "host"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, BENCH_TRANSPORT=transport, MASTER_ADDR="127.0.0.1", GHIP_RCCL_LIB=mock)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
               "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--ng", "20"]
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        res[transport] = json.loads(lines[0])
    out = res["rccl"]
    assert out["transport"] == "rccl" and out["rccl_library"].endswith("librccl_mock.so"), out.get("transport")
    assert res["host"]["transport"] == "host"
    assert out["n_gpus"] == nranks and out["config"]["n_particles"] == 2 * 20 ** 3
    x = out["exchange_per_step_rank0"]
    assert x["bytes_sent_tree_nodes"] > 0 and x["bytes_sent_ghosts"] > 0
    for k in ("grav_interactions", "ewald_interactions", "dens_neighbours", "hydro_pairs"):
        assert out["work_per_step_rank0"][k] == res["host"]["work_per_step_rank0"][k] > 0, k
    assert x == res["host"]["exchange_per_step_rank0"]


def test_hydro_release_variants_do_the_same_work():
    """When the hydro kernel is let go underneath a gravity pair is scheduling only: on the word the
    Ewald walk sets when its last workgroup starts (default, hipStreamWaitValue32), on the walk's end
    event (GHIP_HYDRO_TRIGGER=drain) or right away (GHIP_HYDRO_EARLY=1).  The switches are read once
    per process, so each variant is one short bench run; the steps' interaction, neighbour and pair
    counts must be identical and every run must finish."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = []
    for extra in ({}, {"GHIP_HYDRO_TRIGGER": "drain"}, {"GHIP_HYDRO_EARLY": "1"}):
        env = dict(os.environ, **extra)
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--ng", "24",
               "--no-dropin", "--no-cpu-baseline"]
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        w = json.loads(lines[0])["work_per_step_rank0"]
        work.append({k: w[k] for k in ("grav_interactions", "ewald_interactions", "dens_neighbours",
                                       "hydro_pairs")})
    assert work[0]["hydro_pairs"] > 0 and work[0]["ewald_interactions"] > 0
    assert work[1] == work[0] and work[2] == work[0]


@pytest.mark.gpu
@pytest.mark.parametrize("nshards,domains", [(3, 4), (4, 16)])
def test_shards_that_own_several_pieces_of_the_curve(nshards, domains):
    """-DMULTIPLEDOMAINS > 1 (the shipped Makefile sets 16): a rank owns several disjoint pieces of the
    Peano-Hilbert curve (domain.c:482-494, 1158-1215), given as segments with owners
    (ghip_dd_set_segments).  A tree cell is a shard's own only when it lies inside ONE of its pieces;
    the target groups of a shard span its pieces.  Everything else is the one-range path: gravity
    counts exact against the single global tree, density / hydro to rounding, and a migration after a
    shake sends every particle to the owner of the piece its new key falls into."""
    B = bindings()
    pr = Problem(ng=12, gas=True, periodic=1)
    n, ng = pr.n, pr.ngas
    S = ShardSet(pr, nshards, domains=domains)
    try:
        pieces, seg_owner = S.segments
        assert len(seg_owner) == nshards * domains and len(np.unique(seg_owner)) == nshards
        assert np.all(seg_owner[1:] != seg_owner[:-1])               # no two neighbours merge
        sizes = [len(g) for g in S.gid]
        assert sum(sizes) == n and min(sizes) > 0
        T = pr.oracle_tree()
        tg = np.arange(n, dtype=np.int32)
        tab = O.ewald_table(pr.box)
        S.run.gravity(pr.g_grav(pr.theta), B.WALK_NEWTON_EWALD)
        a0, c0 = T.gravity(pr.o_grav(pr.theta), tg, np.zeros(n))
        T.gravity_ewald_add(pr.o_grav(pr.theta), tab, tg, np.zeros(n), a0, c0)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), c0)
        assert relerr(S.get_field(B.F_GRAVACCEL), a0) < TOL
        old = np.linalg.norm(a0, axis=1)
        S.set_field(B.F_OLDACC, old)
        S.run.gravity(pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
        a1, c1 = T.gravity(pr.o_grav(0.0), tg, old)
        T.gravity_ewald_add(pr.o_grav(0.0), tab, tg, old, a1, c1)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), c1)
        assert relerr(S.get_field(B.F_GRAVACCEL), a1) < TOL
        info = S.each(lambda fp: fp.dd_info())
        assert all(i["let_imported"] > 0 for i in info)
        # SPH through the ghosts
        S.run.density(pr.g_dens())
        S.each(lambda fp: fp.update_hmax())
        S.run.hydro(pr.g_hydro())
        act = np.arange(ng, dtype=np.int32)
        od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                       pr.ti_begstep, pr.hsml0)
        assert relerr(S.get_field(B.F_HSML)[:ng], od["hsml"][:ng]) < TOL
        assert relerr(S.get_field(B.F_DENSITY), od["density"][:ng]) < TOL
        T.update_hmax(act, od["hsml"], od["divvel"])
        oh = T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
                     od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
        st = S.each(lambda fp: fp.stats())
        assert sum(s["hydro_pairs"] for s in st) == oh["npairs"]
        ha = S.get_field(B.F_HYDROACCEL)
        assert np.abs(ha - oh["hydroaccel"][:ng]).max() < TOL * np.abs(oh["hydroaccel"]).max()
        # migration: shake the particles inside the cube, everybody to the owner of its new piece
        rng = np.random.default_rng(3)
        lo, ln = pr.extent[0], pr.extent[2]
        newpos = np.clip(pr.ic["pos"] + 0.02 * pr.box * rng.standard_normal((n, 3)), lo + 1e-9 * ln,
                         lo + ln * (1 - 1e-9))
        S.set_field(B.F_POS, newpos)
        S.migrate()
        probe = B.ForcePath(0)
        probe.set_counts(n, 0)
        probe.set_field(B.F_POS, newpos)
        probe.dd_init(0, 1)
        probe.dd_set_domain(pr.extent[0], pr.extent[1], pr.extent[2], pr.force_soft)
        keys = probe.dd_keys()
        probe.close()
        want_owner = seg_owner[np.searchsorted(pieces[1:-1], keys, side="right")]
        assert np.array_equal(S.owner, want_owner)
        assert sum(len(g) for g in S.gid) == n and (want_owner != seg_owner[np.searchsorted(
            pieces[1:-1], S.keys, side="right")]).sum() > 10
        # and the forces on the re-sharded set are again the single tree's
        T2 = O.Tree(newpos, pr.ic["vel"], pr.ic["mass"], pr.ic["type"], pr.force_soft, hsml=pr.hsml0,
                    extent=pr.extent)
        S.run.gravity(pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
        a2, c2 = T2.gravity(pr.o_grav(0.0), tg, old)
        T2.gravity_ewald_add(pr.o_grav(0.0), tab, tg, old, a2, c2)
        assert np.array_equal(S.get_field(B.F_GRAVCOST), c2)
    finally:
        S.close()
