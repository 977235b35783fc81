"""ctypes front-end of the CPU oracle (oracle/gadget_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  PARITY UNPINNED BY UPSTREAM (see
gadget_oracle.h): the reference has no fixtures for this path and is unbuildable here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EN = 64
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class GravParams(C.Structure):
    _fields_ = [("ErrTolTheta", C.c_double), ("ErrTolForceAcc", C.c_double),
                ("BoxSize", C.c_double), ("periodic", C.c_int),
                ("unequal_softenings", C.c_int), ("rcut", C.c_double), ("asmth", C.c_double)]


class DensParams(C.Structure):
    _fields_ = [("DesNumNgb", C.c_double), ("MaxNumNgbDeviation", C.c_double),
                ("MinGasHsml", C.c_double), ("BoxSize", C.c_double), ("periodic", C.c_int),
                ("Ti_Current", C.c_int), ("Timebase_interval", C.c_double), ("maxiter", C.c_int)]


class HydroParams(C.Structure):
    _fields_ = [("ArtBulkViscConst", C.c_double), ("BoxSize", C.c_double), ("periodic", C.c_int),
                ("ComovingIntegrationOn", C.c_int), ("hubble_a2", C.c_double),
                ("fac_mu", C.c_double), ("fac_vsic_fix", C.c_double),
                ("Timebase_interval", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libgadget_oracle.so")
    src = os.path.join(_HERE, "gadget_oracle.c")
    hdr = os.path.join(_HERE, "gadget_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                                     os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "CC=gcc"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_morton_key.restype = C.c_ulonglong
        L.orc_peano_hilbert_key.restype = C.c_ulonglong
        L.orc_tree_build.restype = C.c_void_p
        L.orc_tree_build.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_double, C.c_int]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_numnodes.argtypes = [C.c_void_p]
        L.orc_tree_dump.argtypes = [C.c_void_p] * 10
        L.orc_tree_dump_ext.argtypes = [C.c_void_p] * 6
        L.orc_tree_dump_particles.argtypes = [C.c_void_p] * 3
        L.orc_update_hmax.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_massless_gas_rule.argtypes = [C.c_int]
        L.orc_tree_drift_nodes.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_tree_kick_nodes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_tree_dump_dynamic.argtypes = [C.c_void_p] * 5
        L.orc_gravity.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_gravity_shortrange.argtypes = L.orc_gravity.argtypes
        L.orc_gravity_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_gravity_ext_soft.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.orc_tree_adaptive_gravsoft.argtypes = [C.c_void_p]
        L.orc_gravity_direct_psoft.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p]
        L.orc_gravity_ewald.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_ewald_init.argtypes = [C.c_void_p, C.c_double]
        L.orc_ewald_force.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_gravity_direct.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p]
        L.orc_ngb_treefind_variable.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                C.c_double, C.c_void_p]
        L.orc_ngb_treefind_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                             C.c_int, C.c_double, C.c_void_p]
        L.orc_density_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                           C.c_void_p, C.c_void_p]
        L.orc_density.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 14
        L.orc_hydro.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 13
        L.orc_set_num_threads.argtypes = [C.c_int]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def morton_key(x, y, z, bits=21):
    return int(lib().orc_morton_key(int(x), int(y), int(z), int(bits)))


def peano_hilbert_key(x, y, z, bits=21):
    return int(lib().orc_peano_hilbert_key(int(x), int(y), int(z), int(bits)))


def domain_extent(pos):
    pos = _f64(pos)
    corner = np.zeros(3)
    center = np.zeros(3)
    ln = C.c_double()
    lib().orc_domain_extent.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib().orc_domain_extent(len(pos), _p(pos), _p(corner), _p(center), C.byref(ln))
    return corner, center, ln.value


_EWALD_CACHE = {}


def ewald_table(boxsize=1.0, cache_dir=None):
    """[3][65][65][65] correction-force table, scaled by 1/BoxSize^2 (forcetree.c:4402-4527).
    Cached on disk because the CPU evaluation takes ~10 s on 8 cores."""
    key = float(boxsize)
    if key in _EWALD_CACHE:
        return _EWALD_CACHE[key]
    cache_dir = cache_dir or os.path.join(_HERE, "_cache")
    fn = os.path.join(cache_dir, "ewald_oracle_box1.npy")
    if os.path.exists(fn):
        unit = np.load(fn)
    else:
        unit = np.zeros((3, EN + 1, EN + 1, EN + 1))
        lib().orc_ewald_init(_p(unit), 1.0)
        try:
            os.makedirs(cache_dir, exist_ok=True)
            np.save(fn, unit)
        except OSError:
            pass
    tab = np.ascontiguousarray(unit / (boxsize * boxsize))
    _EWALD_CACHE[key] = tab
    return tab


def ewald_force(i, j, k, x):
    x = _f64(x)
    f = np.zeros(3)
    lib().orc_ewald_force(i, j, k, _p(x), _p(f))
    return f


def set_massless_gas_rule(rule):
    """The SPH neighbour loops skip gas of mass 0: rule 1 = density() only (-DDUST without
    -DBLACK_HOLES, density.c:831-834), 3 = density() and hydro_force() (-DBLACK_HOLES,
    hydra.c:1235-1238), 0 = neither.  A process-wide switch of the oracle, like the compile-time flags."""
    lib().orc_set_massless_gas_rule(int(rule))


class Tree:
    """The reference-style insertion oct-tree over SoA particle arrays."""

    def __init__(self, pos, vel, mass, ptype, soft, hsml=None, divvel=None, extent=None,
                 toplevels=0):
        self.pos = _f64(pos)
        self.n = len(self.pos)
        self.vel = _f64(vel) if vel is not None else np.zeros_like(self.pos)
        self.mass = _f64(mass)
        self.type = _i32(ptype)
        self.soft = _f64(soft)
        self.hsml = None if hsml is None else _f64(hsml)
        self.divvel = None if divvel is None else _f64(divvel)
        if extent is None:
            extent = domain_extent(self.pos)
        self.corner, self.center, self.len = _f64(extent[0]), _f64(extent[1]), float(extent[2])
        self.h = lib().orc_tree_build(self.n, _p(self.pos), _p(self.vel), _p(self.mass),
                                      _p(self.type), _p(self.hsml), _p(self.divvel),
                                      _p(self.soft), _p(self.corner), _p(self.center),
                                      self.len, int(toplevels))
        if not self.h:
            raise MemoryError("orc_tree_build failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tree_free(self.h)
            self.h = None

    @property
    def numnodes(self):
        return lib().orc_tree_numnodes(self.h)

    def dump(self):
        k = self.numnodes
        out = dict(len=np.zeros(k), center=np.zeros((k, 3)), s=np.zeros((k, 3)), mass=np.zeros(k),
                   sibling=np.zeros(k, np.int32), nextnode=np.zeros(k, np.int32),
                   father=np.zeros(k, np.int32), multi=np.zeros(k, np.int32), hmax=np.zeros(k))
        lib().orc_tree_dump(self.h, *[_p(out[x]) for x in
                                      ("len", "center", "s", "mass", "sibling", "nextnode",
                                       "father", "multi", "hmax")])
        out.update(vs=np.zeros((k, 3)), vmax=np.zeros(k), divvmax=np.zeros(k), maxsoft=np.zeros(k),
                   mixedsoft=np.zeros(k, np.int32))
        lib().orc_tree_dump_ext(self.h, *[_p(out[x]) for x in
                                          ("vs", "vmax", "divvmax", "maxsoft", "mixedsoft")])
        pn = np.zeros(self.n, np.int32)
        pf = np.zeros(self.n, np.int32)
        lib().orc_tree_dump_particles(self.h, _p(pn), _p(pf))
        out["p_nextnode"] = pn
        out["p_father"] = pf
        return out

    def update_hmax(self, active, hsml, divvel=None):
        active = _i32(active)
        self._hs = _f64(hsml)
        self._dv = None if divvel is None else _f64(divvel)
        lib().orc_update_hmax(self.h, len(active), _p(active), _p(self._hs), _p(self._dv))

    # ---- the tree between two builds (forcetree.c:1356-1520) ----
    def drift_nodes(self, dt_drift, dt_drift_hmax=None):
        """force_drift_node for every node.  The particles move with the caller: self.pos / self.vel
        are the arrays the C tree reads (update them in place)."""
        lib().orc_tree_drift_nodes(self.h, float(dt_drift),
                                   float(dt_drift if dt_drift_hmax is None else dt_drift_hmax))

    def kick_nodes(self, idx, dv):
        """force_kick_node for the particles idx, whose self.vel already holds the new velocities."""
        idx = _i32(idx)
        dv = _f64(dv)
        assert dv.shape == (len(idx), 3)
        lib().orc_tree_kick_nodes(self.h, len(idx), _p(idx), _p(dv))

    def dump_dynamic(self, numnodes):
        out = {"s": np.zeros((numnodes, 3)), "len": np.zeros(numnodes), "vs": np.zeros((numnodes, 3)),
               "vmax": np.zeros(numnodes)}
        lib().orc_tree_dump_dynamic(self.h, _p(out["s"]), _p(out["len"]), _p(out["vs"]), _p(out["vmax"]))
        return out

    # ---- gravity ----
    def gravity(self, params, targets, oldacc, kind="newton", ewald_tab=None):
        targets = _i32(targets)
        oldacc = _f64(oldacc)
        acc = np.zeros((len(targets), 3))
        cost = np.zeros(len(targets), np.int32)
        fn = {"newton": lib().orc_gravity, "shortrange": lib().orc_gravity_shortrange}[kind]
        fn(self.h, C.byref(params), len(targets), _p(targets), _p(oldacc), _p(acc), _p(cost))
        return acc, cost

    def gravity_ewald_add(self, params, tab, targets, oldacc, acc, cost):
        targets = _i32(targets)
        oldacc = _f64(oldacc)
        lib().orc_gravity_ewald(self.h, C.byref(params), _p(tab), len(targets), _p(targets),
                                _p(oldacc), _p(acc), _p(cost))

    def gravity_ext(self, params, tpos, ttype, toldacc, tsoft=None):
        tpos = _f64(tpos)
        ttype = _i32(ttype)
        toldacc = _f64(toldacc)
        tsoft = None if tsoft is None else _f64(tsoft)
        acc = np.zeros((len(tpos), 3))
        cost = np.zeros(len(tpos), np.int32)
        lib().orc_gravity_ext_soft(self.h, C.byref(params), len(tpos), _p(tpos), _p(ttype),
                                   _p(tsoft), _p(toldacc), _p(acc), _p(cost))
        return acc, cost

    def adaptive_gravsoft(self):
        """ADAPTIVE_GRAVSOFT_FORGAS: gas softening = Hsml, NODE.maxsoft (forcetree.c:705-726)."""
        lib().orc_tree_adaptive_gravsoft(self.h)
        return self

    # ---- neighbours ----
    def ngb_variable(self, c, h, periodic, boxsize):
        c = _f64(c)
        buf = np.zeros(self.n, np.int32)
        k = lib().orc_ngb_treefind_variable(self.h, _p(c), float(h), int(periodic),
                                            float(boxsize), _p(buf))
        return buf[:k].copy()

    def ngb_pairs(self, c, h, hsml, periodic, boxsize):
        c = _f64(c)
        hsml = _f64(hsml)
        buf = np.zeros(self.n, np.int32)
        k = lib().orc_ngb_treefind_pairs(self.h, _p(c), float(h), _p(hsml), int(periodic),
                                         float(boxsize), _p(buf))
        return buf[:k].copy()

    # ---- SPH ----
    def density_evaluate(self, params, target, h, velpred):
        velpred = _f64(velpred)
        out = np.zeros(7)
        lib().orc_density_evaluate(self.h, C.byref(params), int(target), float(h), _p(velpred),
                                   _p(out))
        return out

    def density(self, params, active, velpred, entropy, dtentropy, timebin, ti_begstep, hsml):
        """Runs the h-iteration; returns dict of arrays sized n (gas entries meaningful)."""
        n = self.n
        active = _i32(active)
        velpred = _f64(velpred)
        entropy = _f64(entropy)
        dtentropy = _f64(dtentropy)
        timebin = _i32(timebin)
        ti_begstep = _i32(ti_begstep)
        out = dict(hsml=_f64(hsml).copy(), numngb=np.zeros(n), density=np.zeros(n),
                   dhsmlfac=np.zeros(n), divvel=np.zeros(n), curlvel=np.zeros(n),
                   pressure=np.zeros(n))
        visits = C.c_longlong(0)
        it = lib().orc_density(self.h, C.byref(params), len(active), _p(active), _p(velpred),
                               _p(entropy), _p(dtentropy), _p(timebin), _p(ti_begstep),
                               _p(out["hsml"]), _p(out["numngb"]), _p(out["density"]),
                               _p(out["dhsmlfac"]), _p(out["divvel"]), _p(out["curlvel"]),
                               _p(out["pressure"]), C.byref(visits))
        out["iterations"] = it
        out["ngb_visits"] = visits.value
        return out

    def hydro(self, params, active, velpred, hsml, density, pressure, dhsmlfac, divvel, curlvel,
              timebin):
        n = self.n
        active = _i32(active)
        arrs = [_f64(x) for x in (velpred, hsml, density, pressure, dhsmlfac, divvel, curlvel)]
        timebin = _i32(timebin)
        out = dict(hydroaccel=np.zeros((n, 3)), dtentropy=np.zeros(n), maxsignalvel=np.zeros(n))
        npairs = C.c_longlong(0)
        lib().orc_hydro(self.h, C.byref(params), len(active), _p(active), *[_p(a) for a in arrs],
                        _p(timebin), _p(out["hydroaccel"]), _p(out["dtentropy"]),
                        _p(out["maxsignalvel"]), C.byref(npairs))
        out["npairs"] = npairs.value
        return out


class BhParams(C.Structure):
    """orc_bh_params (gadget_oracle.h): the sink passes of the shipped flag bundle"""
    _fields_ = [("BoxSize", C.c_double), ("periodic", C.c_int), ("ascale", C.c_double),
                ("dt_fac", C.c_double), ("SMBHmass", C.c_double), ("InnerBoundary", C.c_double),
                ("SinkBoundary", C.c_double), ("SofteningBndry", C.c_double),
                ("CritDensity", C.c_double), ("FeedbackCoeff", C.c_double),
                ("UnitMass_in_g", C.c_double), ("dust", C.c_int),
                ("accretion_of_dust_only", C.c_int), ("accretion_density", C.c_int)]


def sink_density(tree, params, ngbfactor, sinks, velpred, entropy, hsml):
    """density() for Type-5 targets (density.c BLACK_HOLES branches).  Returns dict."""
    sinks = _i32(sinks)
    ns = len(sinks)
    vp = np.zeros((tree.n, 3))
    vp[:len(velpred)] = velpred
    en = np.zeros(tree.n)
    en[:len(entropy)] = entropy
    out = dict(hsml=_f64(hsml).copy(), numngb=np.zeros(ns), density=np.zeros(ns),
               entropy=np.zeros(ns), gasvel=np.zeros((ns, 3)))
    L = lib()
    L.orc_sink_density.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int] + [C.c_void_p] * 8
    it = L.orc_sink_density(tree.h, C.byref(params), float(ngbfactor), ns, _p(sinks), _p(vp), _p(en),
                            _p(out["hsml"]), _p(out["numngb"]), _p(out["density"]),
                            _p(out["entropy"]), _p(out["gasvel"]))
    out["iterations"] = it
    return out


def blackhole_evaluate(tree, params, sinks, ids, hsml, timebin, mdot, bh_density, gas_density,
                       swallowid, injected):
    """blackhole_evaluate (blackhole.c:794) over the sinks in list order; swallowid [n] (uint32)
    and injected [n] are updated and returned (copies)."""
    sinks = _i32(sinks)
    sw = np.ascontiguousarray(swallowid, np.uint32).copy()
    inj = np.zeros(tree.n)
    inj[:len(injected)] = injected
    gd = np.zeros(tree.n)
    gd[:len(gas_density)] = gas_density
    ids = np.ascontiguousarray(ids, np.uint32)
    L = lib()
    L.orc_blackhole_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
    L.orc_blackhole_evaluate.restype = None
    L.orc_blackhole_evaluate(tree.h, C.byref(params), len(sinks), _p(sinks), _p(ids), _p(_f64(hsml)),
                             _p(_i32(timebin)), _p(_f64(mdot)), _p(_f64(bh_density)), _p(gd), _p(sw),
                             _p(inj))
    return sw, inj[:len(injected)]


def blackhole_swallow(tree, params, sinks, ids, hsml, swallowid, particle_bh_mass):
    """blackhole_evaluate_swallow (blackhole.c:1201).  The tree's mass array IS modified (victims
    are set to zero), as the reference does.  Returns dict."""
    sinks = _i32(sinks)
    ns = len(sinks)
    ids = np.ascontiguousarray(ids, np.uint32)
    sw = np.ascontiguousarray(swallowid, np.uint32)
    pbh = _f64(particle_bh_mass).copy()
    out = dict(acc_mass=np.zeros(ns), acc_bhmass=np.zeros(ns), acc_dustmass=np.zeros(ns),
               acc_momentum=np.zeros((ns, 3)), counts=np.zeros(3, np.int64), bh_mass=pbh)
    L = lib()
    L.orc_blackhole_swallow.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 11
    L.orc_blackhole_swallow.restype = None
    L.orc_blackhole_swallow(tree.h, C.byref(params), ns, _p(sinks), _p(ids), _p(_f64(hsml)), _p(sw),
                            _p(tree.mass), _p(pbh), _p(out["acc_mass"]), _p(out["acc_bhmass"]),
                            _p(out["acc_dustmass"]), _p(out["acc_momentum"]), _p(out["counts"]))
    out["mass"] = tree.mass
    return out


def cooling_and_starformation(active, ngas, ptype, mass, timebin, timebase, crit_density, min_egy,
                              u_to_temp_fac, density, entropy, dtentropy, injected):
    """the deterministic per-particle part of cooling_and_starformation (sfr_eff.c:82-947) with the
    cooling function as identity.  Returns (dtentropy, injected, flag_sink)."""
    active = _i32(active)
    dte = _f64(dtentropy).copy()
    inj = _f64(injected).copy()
    flag = np.zeros(len(mass), np.int32)
    L = lib()
    L.orc_cooling_and_starformation.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_double, C.c_double, C.c_double,
                                                C.c_double] + [C.c_void_p] * 5
    L.orc_cooling_and_starformation.restype = None
    L.orc_cooling_and_starformation(len(active), _p(active), int(ngas), _p(_i32(ptype)), _p(_f64(mass)),
                                    _p(_i32(timebin)), float(timebase), float(crit_density),
                                    float(min_egy), float(u_to_temp_fac), _p(_f64(density)),
                                    _p(_f64(entropy)), _p(dte), _p(inj), _p(flag))
    return dte, inj, flag[:ngas]


def gravity_finish(acc, G, pos=None, gravpm=None, comoving_fac=0.0):
    """The post-pass of gravity_tree() over the active particles (gravtree.c:362-403), in the
    reference's order.  acc: G-less tree accelerations [n,3]; pos: P[].Pos (only with comoving_fac);
    gravpm: P[].GravPM (PMGRID builds, carries G).  Returns (OldAcc, GravAccel)."""
    a = np.array(acc, dtype=np.float64, copy=True)
    if comoving_fac != 0.0:
        # :362-373  comoving, !PERIODIC, !PMGRID: fac = 0.5 * Hubble^2 * Omega0 / G
        a = a + comoving_fac * np.asarray(pos, dtype=np.float64)
    b = a
    if gravpm is not None:
        # :379-383  PMGRID: ax = GravAccel[0] + GravPM[0] / All.G ...
        b = a + np.asarray(gravpm, dtype=np.float64) / G
    oldacc = np.sqrt(b[:, 0] * b[:, 0] + b[:, 1] * b[:, 1] + b[:, 2] * b[:, 2])   # :389
    return oldacc, a * G                                                         # :398-403


def gravity_direct(pos, mass, ptype, soft, targets, unequal=False, periodic=False, boxsize=1.0,
                   ewald_tab=None):
    pos = _f64(pos)
    mass = _f64(mass)
    ptype = _i32(ptype)
    soft = _f64(soft)
    targets = _i32(targets)
    acc = np.zeros((len(targets), 3))
    lib().orc_gravity_direct(len(pos), _p(pos), _p(mass), _p(ptype), _p(soft), int(unequal),
                             int(periodic), float(boxsize), _p(ewald_tab), len(targets),
                             _p(targets), _p(acc))
    return acc


def gravity_direct_psoft(pos, mass, psoft, targets, periodic=False, boxsize=1.0, ewald_tab=None):
    """Direct sum with one softening per particle, pairs softened with the larger one."""
    pos = _f64(pos)
    mass = _f64(mass)
    psoft = _f64(psoft)
    targets = _i32(targets)
    acc = np.zeros((len(targets), 3))
    lib().orc_gravity_direct_psoft(len(pos), _p(pos), _p(mass), _p(psoft), int(periodic),
                                   float(boxsize), _p(ewald_tab), len(targets), _p(targets), _p(acc))
    return acc


def drift(time1, timebase, pos, vel, ptype, ti_current, timebin, ti_begstep, gravaccel, velpred,
          hydroaccel, density, hsml, divvel, entropy, dtentropy, pressure, minhsml=0.0, wrap=False,
          boxsize=1.0, tables=None, log_time_begin=0.0, log_time_max=0.0, gravpm=None):
    """In-place drift of copies; returns dict of the updated arrays.  gravpm: PMGRID."""
    n, ngas = len(pos), len(velpred)
    out = dict(pos=_f64(pos).copy(), ti_current=_i32(ti_current).copy(),
               velpred=_f64(velpred).copy(), density=_f64(density).copy(), hsml=_f64(hsml).copy(),
               pressure=_f64(pressure).copy())
    tabs = None if tables is None else _f64(np.concatenate([np.ravel(t) for t in tables]))
    L = lib()
    L.orc_drift_pm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_double,
                               C.c_double, C.c_double, C.c_int, C.c_double] + [C.c_void_p] * 16
    pm = None if gravpm is None else _f64(gravpm)
    rc = L.orc_drift_pm(n, ngas, int(time1), float(timebase), _p(tabs), float(log_time_begin),
                     float(log_time_max), float(minhsml), int(wrap), float(boxsize),
                     _p(out["pos"]), _p(_f64(vel)), _p(_i32(ptype)), _p(out["ti_current"]),
                     _p(_i32(timebin)), _p(_i32(ti_begstep)), _p(_f64(gravaccel)), _p(pm),
                     _p(out["velpred"]), _p(_f64(hydroaccel)), _p(out["density"]), _p(out["hsml"]),
                     _p(_f64(divvel)), _p(_f64(entropy)), _p(_f64(dtentropy)), _p(out["pressure"]))
    out["rc"] = rc
    return out


class KickParams(C.Structure):
    """orc_kick_params (timestep.c parameters of the minimal flag set)"""
    _fields_ = [("Ti_Current", C.c_int), ("Timebase_interval", C.c_double),
                ("ComovingIntegrationOn", C.c_int), ("Time", C.c_double), ("hubble_a", C.c_double),
                ("ErrTolIntAccuracy", C.c_double), ("CourantFac", C.c_double),
                ("MaxSizeTimestep", C.c_double), ("MinSizeTimestep", C.c_double),
                ("dt_displacement", C.c_double), ("SofteningTable", C.c_double * 6),
                ("MinEgySpec", C.c_double), ("TimeBinActive", C.c_uint), ("tables", C.c_void_p),
                ("logTimeBegin", C.c_double), ("logTimeMax", C.c_double),
                ("AdaptiveGravsoftForGasHsml", C.c_int), ("pmgrid", C.c_int),
                ("dt_gravkickB", C.c_double), ("gravpm", C.c_void_p)]


def velocity_moments(vel, mass, ptype):
    """find_dt_displacement_constraint's per-type sums (timestep.c:1140-1156)."""
    L = lib()
    v2, mm = np.zeros(6), np.zeros(6)
    cnt = np.zeros(6, np.int64)
    L.orc_velocity_moments.argtypes = [C.c_int] + [C.c_void_p] * 6
    L.orc_velocity_moments(len(mass), _p(_f64(vel)), _p(_f64(mass)), _p(_i32(ptype)), _p(v2),
                           _p(mm), _p(cnt))
    return v2, mm, cnt


def dt_displacement(v2, minmass, count, comoving, hfac, max_size_timestep, max_rms_fac, omega0,
                    omega_baryon, hubble, G, starformation=0):
    L = lib()
    L.orc_dt_displacement.restype = C.c_double
    L.orc_dt_displacement.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + \
        [C.c_double] * 7 + [C.c_int]
    return L.orc_dt_displacement(_p(_f64(v2)), _p(_f64(minmass)),
                                 _p(np.ascontiguousarray(count, dtype=np.int64)), int(comoving),
                                 float(hfac), float(max_size_timestep), float(max_rms_fac),
                                 float(omega0), float(omega_baryon), float(hubble), float(G),
                                 int(starformation))


def advance_timesteps(params, ptype, vel, gravaccel, hydroaccel, velpred, entropy, dtentropy,
                      density, pressure, hsml, maxsignalvel, timebin, ti_begstep, active=None,
                      tables=None, gravpm=None):
    """advance_and_find_timesteps + get_timestep + do_the_kick on copies; returns the updated
    arrays, the bin counts and rc (0 or the reference's endrun code)."""
    n, ngas = len(ptype), len(entropy)
    out = dict(vel=_f64(vel).copy(), velpred=_f64(velpred).copy(), entropy=_f64(entropy).copy(),
               dtentropy=_f64(dtentropy).copy(), timebin=_i32(timebin).copy(),
               ti_begstep=_i32(ti_begstep).copy())
    tabs = None
    if tables is not None:
        tabs = _f64(np.concatenate([np.ravel(t) for t in tables]))
        params.tables = tabs.ctypes.data
    pm = None
    if gravpm is not None:          # PMGRID (params.pmgrid / dt_gravkickB set by the caller)
        pm = _f64(gravpm)
        params.gravpm = pm.ctypes.data
    cnt = np.zeros(32, np.int64)
    sph = np.zeros(32, np.int64)
    act = None if active is None else _i32(active)
    L = lib()
    L.orc_advance_timesteps.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 16
    rc = L.orc_advance_timesteps(n, ngas, C.addressof(params), 0 if act is None else len(act),
                                 _p(act), _p(_i32(ptype)), _p(out["vel"]), _p(_f64(gravaccel)),
                                 _p(_f64(hydroaccel)), _p(out["velpred"]), _p(out["entropy"]),
                                 _p(out["dtentropy"]), _p(_f64(density)), _p(_f64(pressure)),
                                 _p(_f64(hsml)), _p(_f64(maxsignalvel)), _p(out["timebin"]),
                                 _p(out["ti_begstep"]), _p(cnt), _p(sph))
    out.update(rc=rc, bincount=cnt, bincount_sph=sph)
    return out


def pm_kick(ti_current, timebase, dt_gravkick, dt_gravkickB, ptype, timebin, ti_begstep, vel,
            gravaccel, gravpm, hydroaccel, velpred, tables=None, log_time_begin=0.0,
            log_time_max=0.0):
    """The long-range kick ending a PM step (timestep.c:301-345) on copies; returns vel, velpred."""
    n, ngas = len(ptype), len(velpred)
    out = dict(vel=_f64(vel).copy(), velpred=_f64(velpred).copy())
    tabs = None if tables is None else _f64(np.concatenate([np.ravel(t) for t in tables]))
    L = lib()
    L.orc_pm_kick.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_double,
                              C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 8
    L.orc_pm_kick.restype = None
    L.orc_pm_kick(n, ngas, int(ti_current), float(timebase), _p(tabs), float(log_time_begin),
                  float(log_time_max), float(dt_gravkick), float(dt_gravkickB), _p(_i32(ptype)),
                  _p(_i32(timebin)), _p(_i32(ti_begstep)), _p(out["vel"]), _p(_f64(gravaccel)),
                  _p(_f64(gravpm)), _p(_f64(hydroaccel)), _p(out["velpred"]))
    return out


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


def pm_periodic(pos, mass, boxsize, G, pmgrid, asmth=None):
    """pmforce_periodic (pm_periodic.c:199-800) restated with numpy: CIC assignment (:226-330),
    unnormalised forward FFT, Green's function with CIC deconvolution (:430-486), unnormalised
    inverse FFT, 4-point differences (:489-560), CIC interpolation (:640-690).  Returns GravPM
    [n][3] (G applied, as the reference).  The slab_z clamp typo of :263-264 is not reproduced."""
    N = int(pmgrid)
    pos = _f64(pos)
    mass = _f64(mass)
    asmth = 1.25 * boxsize / N if asmth is None else asmth     # ASMTH * BoxSize / PMGRID (:83)
    to_slab = N / boxsize
    sp = to_slab * pos
    s = sp.astype(np.int64)                                     # (int) truncation
    d = sp - s
    s = np.minimum(s, N - 1)
    rho = np.zeros((N, N, N))
    corners = [(xx, yy, zz) for xx in (0, 1) for yy in (0, 1) for zz in (0, 1)]
    w = {}
    for (xx, yy, zz) in corners:
        w[(xx, yy, zz)] = ((d[:, 0] if xx else 1.0 - d[:, 0]) * (d[:, 1] if yy else 1.0 - d[:, 1]) *
                           (d[:, 2] if zz else 1.0 - d[:, 2]))
        g = ((s[:, 0] + xx) % N, (s[:, 1] + yy) % N, (s[:, 2] + zz) % N)
        np.add.at(rho, g, mass * w[(xx, yy, zz)])
    fk = np.fft.rfftn(rho)                                      # unnormalised like FFTW
    k1 = np.arange(N)
    k1 = np.where(k1 > N // 2, k1 - N, k1).astype(np.float64)
    kz1 = np.arange(N // 2 + 1, dtype=np.float64)
    kx, ky, kz = np.meshgrid(k1, k1, kz1, indexing="ij")
    k2 = kx * kx + ky * ky + kz * kz
    asmth2 = ((2 * np.pi) * asmth / boxsize) ** 2

    def sinc(k):
        a = (np.pi * k) / N
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(k != 0, np.sin(a) / a, 1.0)

    with np.errstate(invalid="ignore", divide="ignore"):
        smth = -np.exp(-k2 * asmth2) / k2
    ff = 1.0 / (sinc(kx) * sinc(ky) * sinc(kz))
    smth = smth * ff * ff * ff * ff
    smth[0, 0, 0] = 0.0
    phi = np.fft.irfftn(fk * smth, s=(N, N, N), axes=(0, 1, 2)) * float(N) ** 3  # unnormalised inverse
    fac = G / (np.pi * boxsize) * (1 / (2 * boxsize / N))
    out = np.zeros((len(pos), 3))
    for dim in range(3):
        f = fac * ((4.0 / 3) * (np.roll(phi, 1, dim) - np.roll(phi, -1, dim)) -
                   (1.0 / 6) * (np.roll(phi, 2, dim) - np.roll(phi, -2, dim)))
        acc = np.zeros(len(pos))
        for (xx, yy, zz) in corners:
            g = ((s[:, 0] + xx) % N, (s[:, 1] + yy) % N, (s[:, 2] + zz) % N)
            acc += f[g] * w[(xx, yy, zz)]
        out[:, dim] = acc
    return out
