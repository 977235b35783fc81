"""ctypes view of libgadget_force.so (include/gadget_force.h): the reference's own call surface
(gravity_tree(), density(), force_treeevaluate(), ...) on its own global AoS arrays.

Used by the parity tests so that they read like tests of the reference's functions, and by
INTEGRATION.md as the worked example of the binding.  No CPU fallback.
"""
import ctypes as C
import importlib
import os

import numpy as np

_pkg = importlib.import_module(__package__)

# struct particle_data (112 B) / sph_particle_data (184 B), minimal periodic flag set
P_DTYPE = np.dtype({
    "names": ["Pos", "Vel", "Mass", "ID", "GravAccel", "OldAcc", "GravCost", "Ti_begstep",
              "Ti_current", "Type", "TimeBin"],
    "formats": [("f8", 3), ("f8", 3), "f8", "u4", ("f8", 3), "f8", "f4", "i4", "i4", "i2", "i2"],
    "offsets": [0, 24, 48, 56, 64, 88, 96, 100, 104, 108, 110],
    "itemsize": 112})

SPH_DTYPE = np.dtype({
    "names": ["Entropy", "Pressure", "VelPred", "MaxSignalVel", "Density", "DtEntropy",
              "HydroAccel", "DhsmlDensityFactor", "DivVel", "Rot", "Hsml", "Left", "Right",
              "NumNgb", "Injected_BH_Energy"],
    "formats": ["f8", "f8", ("f8", 3), "f8", "f8", "f8", ("f8", 3), "f8", "f8", ("f8", 3), "f8",
                "f8", "f8", "f8", "f8"],
    "offsets": [0, 8, 16, 40, 48, 56, 64, 88, 96, 104, 128, 136, 144, 152, 160],
    "itemsize": 184})

GRAVDATA_IN = np.dtype({"names": ["Pos", "Type", "OldAcc", "NodeList"],
                        "formats": [("f8", 3), "i4", "f8", ("i4", 8)],
                        "offsets": [0, 24, 32, 40], "itemsize": 72})
GRAVDATA_OUT = np.dtype({"names": ["Acc", "Ninteractions"], "formats": [("f8", 3), "i4"],
                         "offsets": [0, 24], "itemsize": 32})


class AllStruct(C.Structure):
    _fields_ = [("MaxPart", C.c_int), ("G", C.c_double), ("ErrTolTheta", C.c_double),
                ("ErrTolForceAcc", C.c_double), ("TypeOfOpeningCriterion", C.c_int),
                ("BoxSize", C.c_double), ("DesNumNgb", C.c_double),
                ("MaxNumNgbDeviation", C.c_double), ("MinGasHsmlFractional", C.c_double),
                ("MinGasHsml", C.c_double), ("ArtBulkViscConst", C.c_double),
                ("Ti_Current", C.c_int), ("Timebase_interval", C.c_double), ("Time", C.c_double),
                ("ComovingIntegrationOn", C.c_int), ("Hubble", C.c_double),
                ("Omega0", C.c_double), ("OmegaLambda", C.c_double),
                ("SofteningGas", C.c_double), ("SofteningHalo", C.c_double),
                ("SofteningDisk", C.c_double), ("SofteningBulge", C.c_double),
                ("SofteningStars", C.c_double), ("SofteningBndry", C.c_double),
                ("SofteningGasMaxPhys", C.c_double), ("SofteningHaloMaxPhys", C.c_double),
                ("SofteningDiskMaxPhys", C.c_double), ("SofteningBulgeMaxPhys", C.c_double),
                ("SofteningStarsMaxPhys", C.c_double), ("SofteningBndryMaxPhys", C.c_double),
                ("SofteningTable", C.c_double * 6), ("ForceSoftening", C.c_double * 6),
                ("Rcut", C.c_double * 2), ("Asmth", C.c_double * 2),
                ("TotNumOfForces", C.c_longlong), ("BunchSize", C.c_int),
                ("BufferSize", C.c_double),
                ("ErrTolIntAccuracy", C.c_double), ("CourantFac", C.c_double),
                ("MaxSizeTimestep", C.c_double), ("MinSizeTimestep", C.c_double),
                ("MaxRMSDisplacementFac", C.c_double), ("OmegaBaryon", C.c_double),
                ("MinEgySpec", C.c_double), ("TypeOfTimestepCriterion", C.c_int),
                ("StarformationOn", C.c_int),
                ("BlackHoleNgbFactor", C.c_double), ("BlackHoleFeedbackFactor", C.c_double),
                ("SMBHmass", C.c_double), ("InnerBoundary", C.c_double),
                ("SinkBoundary", C.c_double), ("CritOverDensity", C.c_double),
                ("UnitLength_in_cm", C.c_double), ("UnitMass_in_g", C.c_double),
                ("UnitEnergy_in_cgs", C.c_double)]


class Config(C.Structure):
    _fields_ = [("periodic", C.c_int), ("pmgrid", C.c_int), ("unequal_softenings", C.c_int),
                ("device", C.c_int), ("black_holes", C.c_int), ("dust", C.c_int),
                ("accretion_of_dust_only", C.c_int), ("accretion_density", C.c_int),
                ("overlap_sph", C.c_int), ("dynamic_tree", C.c_int), ("pin_records", C.c_int)]


class BhLayout(C.Structure):
    """struct gadget_force_bh_layout: byte offsets of the members the sink passes touch (-1: absent)"""
    _fields_ = [(k, C.c_int) for k in (
        "p_id", "p_swallowid", "p_bh_mass", "p_bh_mdot", "p_bh_density", "p_bh_entropy",
        "p_bh_gasvel", "p_bh_accreted_mass", "p_bh_accreted_bhmass", "p_bh_accreted_dustmass",
        "p_bh_accreted_momentum", "p_dust_density", "p_dust_entropy", "p_dust_gasvel",
        "p_dust_mass", "s_injected_bh_energy")]


class TopNode(C.Structure):
    """struct topnode_data, allvars.h:437-447"""
    _fields_ = [("Size", C.c_ulonglong), ("StartKey", C.c_ulonglong), ("Count", C.c_longlong),
                ("GravCost", C.c_double), ("Daughter", C.c_int), ("Pstart", C.c_int),
                ("Blocks", C.c_int), ("Leaf", C.c_int)]


HOST_ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


ENDRUN_CB = C.CFUNCTYPE(None, C.c_int)

# GADGET_FORCE_ALL_MEMBERS of include/gadget_force.h: (name, numpy type) in table order
ALL_MEMBERS = [
    ("MaxPart", "i4"), ("G", "f8"), ("ErrTolTheta", "f8"), ("ErrTolForceAcc", "f8"),
    ("TypeOfOpeningCriterion", "i4"), ("BoxSize", "f8"), ("DesNumNgb", "f8"),
    ("MaxNumNgbDeviation", "f8"), ("MinGasHsmlFractional", "f8"), ("MinGasHsml", "f8"),
    ("ArtBulkViscConst", "f8"), ("Ti_Current", "i4"), ("Timebase_interval", "f8"), ("Time", "f8"),
    ("ComovingIntegrationOn", "i4"), ("Hubble", "f8"), ("Omega0", "f8"), ("OmegaLambda", "f8"),
    ("SofteningGas", "f8"), ("SofteningHalo", "f8"), ("SofteningDisk", "f8"),
    ("SofteningBulge", "f8"), ("SofteningStars", "f8"), ("SofteningBndry", "f8"),
    ("SofteningGasMaxPhys", "f8"), ("SofteningHaloMaxPhys", "f8"), ("SofteningDiskMaxPhys", "f8"),
    ("SofteningBulgeMaxPhys", "f8"), ("SofteningStarsMaxPhys", "f8"),
    ("SofteningBndryMaxPhys", "f8"), ("SofteningTable", ("f8", 6)), ("ForceSoftening", ("f8", 6)),
    ("Rcut", ("f8", 2)), ("Asmth", ("f8", 2)), ("TotNumOfForces", "i8"), ("BunchSize", "i4"),
    ("BufferSize", "f8"), ("ErrTolIntAccuracy", "f8"), ("CourantFac", "f8"),
    ("MaxSizeTimestep", "f8"), ("MinSizeTimestep", "f8"), ("MaxRMSDisplacementFac", "f8"),
    ("OmegaBaryon", "f8"), ("MinEgySpec", "f8"), ("TypeOfTimestepCriterion", "i4"),
    ("StarformationOn", "i4"), ("BlackHoleNgbFactor", "f8"), ("BlackHoleFeedbackFactor", "f8"),
    ("SMBHmass", "f8"), ("InnerBoundary", "f8"), ("SinkBoundary", "f8"), ("CritOverDensity", "f8"),
    ("UnitLength_in_cm", "f8"), ("UnitMass_in_g", "f8"), ("UnitEnergy_in_cgs", "f8")]

EXPORTS = ["gadget_force_bind_all", "gadget_force_all_layout_count",
           "gadget_force_init", "gadget_force_finalize", "gadget_force_last_error",
           "gadget_force_ctx", "gadget_force_layout", "gadget_force_set_endrun",
           "gadget_force_mark_dirty", "endrun", "set_softenings", "data_index_compare",
           "mysort_dataindex", "domain_findExtent",
           "gadget_force_set_drift_table", "force_treebuild", "force_kick_node", "force_finish_kick_nodes", "ewald_init", "gravity_tree", "density", "density_isactive",
           "force_update_hmax", "hydro_force", "force_treeevaluate",
           "force_treeevaluate_shortrange", "force_treeevaluate_ewald_correction",
           "density_evaluate", "hydro_evaluate", "ngb_treefind_variable", "ngb_treefind_pairs",
           "peano_hilbert_key", "morton_key", "hubble_function",
           "P", "SphP", "All", "NumPart", "N_gas", "FirstActiveParticle", "NextActiveParticle",
           "DomainTask", "TreeReconstructFlag", "DomainCorner", "DomainCenter", "DomainLen", "DomainFac",
           "Ngblist", "GravDataGet", "GravDataResult",
           "advance_and_find_timesteps", "find_dt_displacement_constraint", "get_timestep_bin",
           "gadget_force_set_kick_tables", "TimeBinCount", "TimeBinCountSph", "TimeBinActive",
           "FirstInTimeBin", "LastInTimeBin", "NextInTimeBin", "PrevInTimeBin", "Flag_FullStep",
           "Nodes_base", "Nodes", "Extnodes_base", "Extnodes", "Nextnode", "Father", "MaxNodes",
           "Numnodestree",
           "gadget_force_bind_records", "blackhole_evaluate", "blackhole_evaluate_swallow",
           "blackhole_accretion_neighbour_passes", "N_gas_swallowed", "N_BH_swallowed",
           "N_dust_swallowed", "TopNodes", "NTopnodes", "NTopleaves", "DomainStartList",
           "DomainEndList", "gadget_force_unique_id", "gadget_force_connect",
           "gadget_force_set_allgather", "ThisTask", "NTask", "gadget_force_flush"]

_LIB = None


def lib_path():
    if not os.path.exists(_pkg.LIBHOST):
        raise RuntimeError("libgadget_force.so is missing (%s): run __graft_entry__.build()"
                           % _pkg.LIBHOST)
    return _pkg.LIBHOST


def lib():
    global _LIB
    if _LIB is None:
        _pkg.lib_path()  # libghip.so must exist too
        L = C.CDLL(lib_path(), mode=C.RTLD_GLOBAL)
        L.gadget_force_init.argtypes = [C.POINTER(Config)]
        L.gadget_force_bind_all.argtypes = [C.c_void_p, C.c_void_p]
        L.gadget_force_bind_all.restype = None
        L.gadget_force_last_error.restype = C.c_char_p
        L.gadget_force_ctx.restype = C.c_void_p
        L.gadget_force_set_endrun.argtypes = [ENDRUN_CB]
        for f in ("force_treeevaluate", "force_treeevaluate_shortrange",
                  "force_treeevaluate_ewald_correction", "density_evaluate", "hydro_evaluate"):
            getattr(L, f).argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in ("ngb_treefind_variable", "ngb_treefind_pairs"):
            getattr(L, f).argtypes = [C.POINTER(C.c_double), C.c_double, C.c_int,
                                      C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
        L.force_treebuild.argtypes = [C.c_int, C.c_void_p]
        L.density_isactive.argtypes = [C.c_int]
        L.peano_hilbert_key.argtypes = [C.c_int] * 4
        L.peano_hilbert_key.restype = C.c_ulonglong
        L.morton_key.argtypes = [C.c_int] * 4
        L.morton_key.restype = C.c_ulonglong
        L.hubble_function.argtypes = [C.c_double]
        L.hubble_function.restype = C.c_double
        L.gadget_force_bind_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gadget_force_bind_records.restype = None
        for f in ("blackhole_evaluate", "blackhole_evaluate_swallow"):
            getattr(L, f).argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.blackhole_accretion_neighbour_passes.restype = None
        L.gadget_force_unique_id.argtypes = [C.c_void_p]
        L.gadget_force_connect.argtypes = [C.c_void_p]
        L.gadget_force_set_allgather.argtypes = [HOST_ALLGATHER_CB, C.c_void_p]
        L.gadget_force_set_allgather.restype = None
        _LIB = L
    return _LIB


class Host:
    """Owns the numpy arrays standing in for the reference's P[], SphP[], NextActiveParticle[]
    and points the library's globals at them."""

    def __init__(self, periodic=1, pmgrid=0, unequal=0, device=0, black_holes=0, dust=0,
                 overlap_sph=0, accretion_of_dust_only=0, accretion_density=0, rank=0, nranks=1,
                 pin_records=0, dynamic_tree=0):
        self.L = lib()
        self.endrun_codes = []
        self._cb = ENDRUN_CB(lambda code: self.endrun_codes.append(code))
        self.L.gadget_force_set_endrun(self._cb)
        self._seti("ThisTask", rank)
        self._seti("NTask", nranks)
        cfg = Config(periodic, pmgrid, unequal, device, black_holes, dust, accretion_of_dust_only,
                     accretion_density, overlap_sph, dynamic_tree, pin_records)
        rc = self.L.gadget_force_init(C.byref(cfg))
        if rc != 0:
            raise RuntimeError("gadget_force_init failed (%d): %s" %
                               (rc, self.L.gadget_force_last_error().decode()))
        self.All = AllStruct.in_dll(self.L, "All")

    def bind_all(self, host_all, offsets):
        """gadget_force_bind_all: `host_all` a numpy structured scalar/array holding the host's own
        struct, `offsets` {member: byte offset} (missing members: -1).  None unbinds."""
        if host_all is None:
            self.L.gadget_force_bind_all(None, None)
            self._bound = None
            return
        assert self.L.gadget_force_all_layout_count() == len(ALL_MEMBERS)
        tab = (C.c_int * len(ALL_MEMBERS))(*[int(offsets.get(name, -1)) for name, _ in ALL_MEMBERS])
        self._bound = (host_all, tab)            # keep both alive
        self.L.gadget_force_bind_all(C.c_void_p(host_all.ctypes.data), tab)

    def bind_records(self, P, SphP, lay, bh=None):
        """gadget_force_bind_records: the host's own record arrays (any numpy structured dtype) by
        byte offsets -- `lay` a bindings.Layout, `bh` a BhLayout or None"""
        self.P, self.SphP = P, SphP
        self._rec = (P, SphP, lay, bh)          # keep alive
        self.L.gadget_force_bind_records(
            C.c_void_p(P.ctypes.data), C.c_void_p(SphP.ctypes.data) if SphP is not None else None,
            C.cast(C.byref(lay), C.c_void_p), C.cast(C.byref(bh), C.c_void_p) if bh is not None else None)
        self._seti("NumPart", len(P))
        self._seti("N_gas", 0 if SphP is None else len(SphP))
        self.All.MaxPart = len(P)
        self.L.gadget_force_mark_dirty()
        self._seti("TreeReconstructFlag", 1)

    def set_allgather(self, allgather):
        """the host's all-gather for more than one rank: allgather(send: bytes) -> bytes of all ranks"""
        def cb(_user, send, nbytes, recv):
            try:
                out = allgather(C.string_at(send, nbytes))
                C.memmove(recv, out, len(out))
                return 0
            except Exception:   # noqa: BLE001 -- reported through the C return code
                import traceback
                traceback.print_exc()
                return 1
        self._agcb = HOST_ALLGATHER_CB(cb)
        self.L.gadget_force_set_allgather(self._agcb, None)

    def set_topnodes(self, start_keys, sizes, start_list, end_list, domain_task=None):
        """the host's domain decomposition as the drivers read it: top-leaves in key order and the
        leaf ranges of the ranks (TopNodes / DomainStartList / DomainEndList, allvars.h:424-449);
        domain_task: DomainTask[] per leaf (-DMULTIPLEDOMAINS > 1)"""
        nleaf = len(start_keys)
        arr = (TopNode * nleaf)()
        for i in range(nleaf):
            arr[i].Size, arr[i].StartKey = int(sizes[i]), int(start_keys[i])
            arr[i].Daughter, arr[i].Leaf = -1, i
        self._top = arr
        self._dstart = np.ascontiguousarray(start_list, np.int32)
        self._dend = np.ascontiguousarray(end_list, np.int32)
        C.c_void_p.in_dll(self.L, "TopNodes").value = C.addressof(arr)
        self._seti("NTopnodes", nleaf)
        self._seti("NTopleaves", nleaf)
        self._setp("DomainStartList", self._dstart)
        self._setp("DomainEndList", self._dend)
        self._dtask = None if domain_task is None else np.ascontiguousarray(domain_task, np.int32)
        self._setp("DomainTask", self._dtask)

    def close(self):
        self.L.gadget_force_bind_all(None, None)
        self.L.gadget_force_bind_records(None, None, None, None)
        self.L.gadget_force_set_allgather(C.cast(None, HOST_ALLGATHER_CB), None)
        self._seti("ThisTask", 0)
        self._seti("NTask", 1)
        C.c_void_p.in_dll(self.L, "DomainTask").value = None
        self.L.gadget_force_finalize()

    def _setp(self, name, arr):
        C.c_void_p.in_dll(self.L, name).value = arr.ctypes.data if arr is not None else None

    def _seti(self, name, v):
        C.c_int.in_dll(self.L, name).value = int(v)

    def _geti(self, name):
        return C.c_int.in_dll(self.L, name).value

    def set_particles(self, P, SphP):
        assert P.dtype == P_DTYPE and (SphP is None or SphP.dtype == SPH_DTYPE)
        self.P, self.SphP = P, SphP
        self._setp("P", P)
        self._setp("SphP", SphP)
        self._seti("NumPart", len(P))
        self._seti("N_gas", 0 if SphP is None else len(SphP))
        self.All.MaxPart = len(P)
        self.L.gadget_force_mark_dirty()
        self._seti("TreeReconstructFlag", 1)

    def set_active(self, idx=None):
        """Thread FirstActiveParticle/NextActiveParticle like run.c:300-320."""
        n = len(self.P)
        nxt = np.full(n, -1, np.int32)
        idx = np.arange(n, dtype=np.int32) if idx is None else np.asarray(idx, np.int32)
        if len(idx):
            nxt[idx[:-1]] = idx[1:]
            first = int(idx[0])
        else:
            first = -1
        self.next_active = nxt
        self._setp("NextActiveParticle", nxt)
        self._seti("FirstActiveParticle", first)

    def domain(self):
        self.L.domain_findExtent()
        d3 = C.c_double * 3
        return (np.array(d3.in_dll(self.L, "DomainCorner")),
                np.array(d3.in_dll(self.L, "DomainCenter")),
                C.c_double.in_dll(self.L, "DomainLen").value)

    def set_domain(self, corner, center, length):
        """DomainCorner / DomainCenter / DomainLen as a domain decomposition left them."""
        d3 = C.c_double * 3
        for name, v in (("DomainCorner", corner), ("DomainCenter", center)):
            a = d3.in_dll(self.L, name)
            for j in range(3):
                a[j] = float(v[j])
        C.c_double.in_dll(self.L, "DomainLen").value = float(length)

    def ngblist(self, count):
        ptr = C.POINTER(C.c_int).in_dll(self.L, "Ngblist")
        return np.array([ptr[i] for i in range(count)], np.int32)
