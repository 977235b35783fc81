"""ctypes bindings of libghip.so (include/ghip.h).  Thin: every method is one C call.

No CPU fallback: constructing ForcePath without the library or without a GPU raises.
"""
import ctypes as C
import importlib
import os

import numpy as np

_pkg = importlib.import_module(__package__)

# enum ghip_field
(F_POS, F_VEL, F_MASS, F_TYPE, F_OLDACC, F_HSML, F_TIMEBIN, F_TI_BEGSTEP, F_VELPRED, F_ENTROPY,
 F_DTENTROPY, F_GRAVACCEL, F_GRAVCOST, F_NUMNGB, F_DENSITY, F_DHSMLFAC, F_DIVVEL, F_CURLVEL,
 F_PRESSURE, F_HYDROACCEL, F_MAXSIGNALVEL, F_TI_CURRENT, F_GRAVPM, F_ID, F_COUNT) = range(25)

_FIELD_INFO = {  # gas-sized?, ncomp, is int
    F_POS: (0, 3, 0), F_VEL: (0, 3, 0), F_MASS: (0, 1, 0), F_TYPE: (0, 1, 1), F_OLDACC: (0, 1, 0),
    F_HSML: (0, 1, 0), F_TIMEBIN: (0, 1, 1), F_TI_BEGSTEP: (0, 1, 1), F_VELPRED: (1, 3, 0),
    F_ENTROPY: (1, 1, 0), F_DTENTROPY: (1, 1, 0), F_GRAVACCEL: (0, 3, 0), F_GRAVCOST: (0, 1, 1),
    F_NUMNGB: (1, 1, 0), F_DENSITY: (1, 1, 0), F_DHSMLFAC: (1, 1, 0), F_DIVVEL: (1, 1, 0),
    F_CURLVEL: (1, 1, 0), F_PRESSURE: (1, 1, 0), F_HYDROACCEL: (1, 3, 0), F_MAXSIGNALVEL: (1, 1, 0),
    F_TI_CURRENT: (0, 1, 1), F_GRAVPM: (0, 3, 0), F_ID: (0, 1, 1)}

WALK_NEWTON, WALK_SHORTRANGE, WALK_EWALD, WALK_NEWTON_EWALD = 0, 1, 2, 3
EN = 64

GHIP_ERRORS = {-90001: "GHIP_EHIP", -90002: "GHIP_EINVAL", -90003: "GHIP_ENOMEM",
               -90004: "GHIP_ENOCONV", -90005: "GHIP_ENODEVICE", -90006: "GHIP_ETIMESTEP",
               -90008: "GHIP_EDEVICE", -90009: "GHIP_ECOMM"}


class Layout(C.Structure):
    _fields_ = [(k, C.c_int) for k in (
        "p_stride", "p_pos", "p_vel", "p_mass", "p_gravaccel", "p_oldacc", "p_gravcost",
        "p_ti_begstep", "p_type", "p_timebin", "p_hsml", "p_numngb",
        "s_stride", "s_entropy", "s_pressure", "s_velpred", "s_maxsignalvel", "s_density",
        "s_dtentropy", "s_hydroaccel", "s_dhsmlfac", "s_divvel", "s_curlvel", "s_hsml",
        "s_numngb", "p_ti_current", "p_gravpm")]


class GravParams(C.Structure):
    _fields_ = [("ErrTolTheta", C.c_double), ("ErrTolForceAcc", C.c_double),
                ("ForceSoftening", C.c_double * 6), ("BoxSize", C.c_double),
                ("periodic", C.c_int), ("unequal_softenings", C.c_int),
                ("Rcut", C.c_double), ("Asmth", C.c_double)]


class DensParams(C.Structure):
    _fields_ = [("DesNumNgb", C.c_double), ("MaxNumNgbDeviation", C.c_double),
                ("MinGasHsml", C.c_double), ("BoxSize", C.c_double), ("periodic", C.c_int),
                ("Ti_Current", C.c_int), ("Timebase_interval", C.c_double), ("MaxIter", C.c_int)]


class HydroParams(C.Structure):
    _fields_ = [("ArtBulkViscConst", C.c_double), ("BoxSize", C.c_double), ("periodic", C.c_int),
                ("ComovingIntegrationOn", C.c_int), ("hubble_a2", C.c_double),
                ("fac_mu", C.c_double), ("fac_vsic_fix", C.c_double),
                ("Timebase_interval", C.c_double), ("raw_dtentropy", C.c_int)]


class DriftParams(C.Structure):
    _fields_ = [("time1", C.c_int), ("Timebase_interval", C.c_double),
                ("ComovingIntegrationOn", C.c_int), ("logTimeBegin", C.c_double),
                ("logTimeMax", C.c_double), ("DriftTable", C.c_void_p),
                ("GravKickTable", C.c_void_p), ("HydroKickTable", C.c_void_p),
                ("MinGasHsml", C.c_double), ("box_wrap", C.c_int), ("BoxSize", C.c_double),
                ("pmgrid", C.c_int)]


class KickParams(C.Structure):
    _fields_ = [("Ti_Current", C.c_int), ("Timebase_interval", C.c_double),
                ("ComovingIntegrationOn", C.c_int), ("Time", C.c_double), ("hubble_a", C.c_double),
                ("ErrTolIntAccuracy", C.c_double), ("CourantFac", C.c_double),
                ("MaxSizeTimestep", C.c_double), ("MinSizeTimestep", C.c_double),
                ("dt_displacement", C.c_double), ("SofteningTable", C.c_double * 6),
                ("MinEgySpec", C.c_double), ("TimeBinActive", C.c_uint),
                ("logTimeBegin", C.c_double), ("logTimeMax", C.c_double),
                ("GravKickTable", C.c_void_p), ("HydroKickTable", C.c_void_p),
                ("AdaptiveGravsoftForGasHsml", C.c_int), ("pmgrid", C.c_int),
                ("dt_gravkickB", C.c_double)]


class PmKickParams(C.Structure):
    """ghip_pmkick_params (the long-range kick ending a PM step, timestep.c:269-345)"""
    _fields_ = [("Ti_Current", C.c_int), ("Timebase_interval", C.c_double),
                ("ComovingIntegrationOn", C.c_int), ("logTimeBegin", C.c_double),
                ("logTimeMax", C.c_double), ("GravKickTable", C.c_void_p),
                ("HydroKickTable", C.c_void_p), ("dt_gravkick", C.c_double),
                ("dt_gravkickB", C.c_double)]


class BhParams(C.Structure):
    """ghip_bh_params: the sink passes of the shipped flag bundle (include/ghip.h, row N4)"""
    _fields_ = [("BoxSize", C.c_double), ("periodic", C.c_int), ("ascale", C.c_double),
                ("dt_fac", C.c_double), ("SMBHmass", C.c_double), ("InnerBoundary", C.c_double),
                ("SinkBoundary", C.c_double), ("SofteningBndry", C.c_double),
                ("CritDensity", C.c_double), ("FeedbackCoeff", C.c_double),
                ("UnitMass_in_g", C.c_double), ("dust", C.c_int),
                ("accretion_of_dust_only", C.c_int), ("accretion_density", C.c_int)]


class PmParams(C.Structure):
    _fields_ = [("pmgrid", C.c_int), ("BoxSize", C.c_double), ("G", C.c_double),
                ("Asmth", C.c_double)]


class NodeLayout(C.Structure):
    """ghip_node_layout: byte offsets of struct NODE / struct extNODE (allvars.h:1847-1916)"""
    _fields_ = [(k, C.c_int) for k in
                ("node_stride", "n_len", "n_center", "n_s", "n_mass", "n_bitflags", "n_sibling",
                 "n_nextnode", "n_father", "n_ti_current", "ext_stride", "e_dp", "e_vs", "e_vmax",
                 "e_divvmax", "e_hmax", "e_ti_lastkicked", "e_flag", "n_maxsoft")]


# struct NODE / struct extNODE of the minimal periodic flag set (88 / 80 bytes)
NODE_DTYPE = np.dtype({"names": ["len", "center", "s", "mass", "bitflags", "sibling", "nextnode",
                                 "father", "Ti_current"],
                       "formats": ["f8", ("f8", 3), ("f8", 3), "f8", "u4", "i4", "i4", "i4", "i4"],
                       "offsets": [0, 8, 32, 56, 64, 68, 72, 76, 80], "itemsize": 88})
# struct NODE with ADAPTIVE_GRAVSOFT_FORGAS: maxsoft sits before the union (allvars.h:1857-1860), 96 bytes
NODE_ADAPTIVE_DTYPE = np.dtype({"names": ["len", "center", "maxsoft", "s", "mass", "bitflags",
                                          "sibling", "nextnode", "father", "Ti_current"],
                                "formats": ["f8", ("f8", 3), "f8", ("f8", 3), "f8", "u4", "i4", "i4",
                                            "i4", "i4"],
                                "offsets": [0, 8, 32, 40, 64, 72, 76, 80, 84, 88], "itemsize": 96})
EXTNODE_DTYPE = np.dtype({"names": ["dp", "vs", "vmax", "divVmax", "hmax", "Ti_lastkicked", "Flag"],
                          "formats": [("f8", 3), ("f8", 3), "f8", "f8", "f8", "i4", "i4"],
                          "offsets": [0, 24, 48, 56, 64, 72, 76], "itemsize": 80})


class Stats(C.Structure):
    _fields_ = [("grav_interactions", C.c_longlong), ("grav_targets", C.c_longlong),
                ("ewald_interactions", C.c_longlong), ("dens_neighbours", C.c_longlong),
                ("dens_target_evals", C.c_longlong), ("dens_iterations", C.c_int),
                ("hydro_pairs", C.c_longlong), ("hydro_targets", C.c_longlong),
                ("tree_nodes", C.c_int), ("gastree_nodes", C.c_int),
                ("ms_tree", C.c_float), ("ms_grav", C.c_float), ("ms_ewald", C.c_float),
                ("ms_dens", C.c_float), ("ms_hmax", C.c_float), ("ms_hydro", C.c_float),
                ("grav_wave_steps", C.c_longlong), ("ewald_wave_steps", C.c_longlong),
                ("ms_kick", C.c_float), ("ms_pm", C.c_float)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RunStats(C.Structure):
    """ghip_run_stats: a run of steps without a host synchronisation per step"""
    _fields_ = ([(k, C.c_longlong) for k in (
        "steps", "steps_timed", "launches", "blocking_syncs", "grav_interactions",
        "ewald_interactions", "dens_neighbours", "hydro_pairs", "grav_wave_steps",
        "ewald_wave_steps", "dens_extra_iterations")] +
        [(k, C.c_double) for k in (
            "ms_tree", "ms_grav", "ms_ewald", "ms_dens", "ms_hmax", "ms_hydro", "ms_kick",
            "ms_steps_device", "ms_between_steps", "ms_first_to_last")])

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class GhipError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        RuntimeError.__init__(self, "%s (%d): %s" % (GHIP_ERRORS.get(code, "GHIP_E?"), code, msg))


_LIB = None

# int allgather(void *user, const void *send, size_t bytes, void *recv)
ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

EXPORTS = [
    "ghip_create", "ghip_destroy", "ghip_last_error", "ghip_version", "ghip_set_counts",
    "ghip_set_field", "ghip_get_field", "ghip_upload_aos", "ghip_download_aos", "ghip_set_active",
    "ghip_set_shard", "ghip_tree_build", "ghip_ewald_init", "ghip_ewald_get_table", "ghip_gravity",
    "ghip_gravity_ext", "ghip_gravity_finish", "ghip_gravity_finish_ex", "ghip_gravity_direct",
    "ghip_density",
    "ghip_update_hmax", "ghip_hydro", "ghip_density_evaluate", "ghip_ngb_treefind",
    "ghip_peano_hilbert_keys", "ghip_morton_keys", "ghip_get_stats", "ghip_tree_dump",
    "ghip_stream", "ghip_sync", "ghip_shard_pack", "ghip_shard_unpack", "ghip_shard_count",
    "ghip_drift", "ghip_gravity_finish_all", "ghip_advance_timesteps",
    "ghip_timestep_endrun_code", "ghip_velocity_moments", "ghip_download_aos_kick",
    "ghip_tree_export", "ghip_pm_periodic", "ghip_set_adaptive_gravsoft", "ghip_gravity_ext_soft",
    "ghip_gravity_vacuum_energy", "ghip_pm_kick",
    "ghip_dd_init", "ghip_dd_set_domain", "ghip_dd_set_splits", "ghip_dd_set_segments", "ghip_dd_keys", "ghip_dd_find_split",
    "ghip_set_dynamic_tree", "ghip_tree_substep", "ghip_tree_kick_nodes", "ghip_tree_kick_nodes_vmax",
    "ghip_tree_dump_dynamic",
    "ghip_gas_block_mixed", "ghip_set_massless_gas_rule", "ghip_set_hydro_release", "ghip_download_aos_async",
    "ghip_gravity_to_records", "ghip_pin_host", "ghip_unpin_host", "ghip_dd_set_ghost_margin", "ghip_dd_rccl_unique_id", "ghip_dd_rccl_connect",
    "ghip_dd_rccl_library", "ghip_dd_begin", "ghip_dd_step", "ghip_dd_exchange",
    "ghip_dd_exchange_local", "ghip_dd_exchange_host", "ghip_dd_run", "ghip_dd_get_info",
    "ghip_sink_density", "ghip_sink_reset", "ghip_blackhole_evaluate", "ghip_blackhole_swallow",
    "ghip_sink_get_marks", "ghip_sink_set_marks", "ghip_cooling_and_starformation",
    "ghip_set_async", "ghip_timebin_counts", "ghip_run_begin", "ghip_step_begin", "ghip_step_end",
    "ghip_run_end"]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_pkg.lib_path())
        vp = C.c_void_p
        L.ghip_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ghip_destroy.argtypes = [vp]
        L.ghip_destroy.restype = None
        L.ghip_last_error.argtypes = [vp]
        L.ghip_last_error.restype = C.c_char_p
        L.ghip_version.restype = C.c_char_p
        L.ghip_set_counts.argtypes = [vp, C.c_int, C.c_int]
        L.ghip_set_field.argtypes = [vp, C.c_int, vp]
        L.ghip_get_field.argtypes = [vp, C.c_int, vp]
        L.ghip_upload_aos.argtypes = [vp, vp, vp, C.POINTER(Layout), C.c_int, C.c_int]
        L.ghip_download_aos.argtypes = [vp, vp, vp, C.POINTER(Layout), C.c_int, C.c_int, C.c_int]
        L.ghip_set_active.argtypes = [vp, vp, C.c_int]
        L.ghip_set_shard.argtypes = [vp, C.c_int, C.c_int]
        L.ghip_tree_build.argtypes = [vp, vp, vp, C.c_double, vp]
        L.ghip_ewald_init.argtypes = [vp, C.c_double]
        L.ghip_ewald_get_table.argtypes = [vp, vp]
        L.ghip_gravity.argtypes = [vp, C.POINTER(GravParams), C.c_int]
        L.ghip_gravity_ext.argtypes = [vp, C.POINTER(GravParams), C.c_int, C.c_int, vp, vp, vp,
                                       vp, vp]
        L.ghip_gravity_ext_soft.argtypes = [vp, C.POINTER(GravParams), C.c_int, C.c_int, vp, vp, vp,
                                            vp, vp, vp]
        L.ghip_set_adaptive_gravsoft.argtypes = [vp, C.c_int]
        L.ghip_gravity_vacuum_energy.argtypes = [vp, C.c_double]
        L.ghip_pm_kick.argtypes = [vp, C.POINTER(PmKickParams)]
        L.ghip_gravity_finish.argtypes = [vp, C.c_double]
        L.ghip_gravity_finish_all.argtypes = [vp, C.c_double]
        L.ghip_gravity_finish_ex.argtypes = [vp, C.c_double, C.c_int, C.c_double, C.c_int]
        L.ghip_gravity_direct.argtypes = [vp, C.POINTER(GravParams)]
        L.ghip_density.argtypes = [vp, C.POINTER(DensParams)]
        L.ghip_update_hmax.argtypes = [vp]
        L.ghip_hydro.argtypes = [vp, C.POINTER(HydroParams)]
        L.ghip_density_evaluate.argtypes = [vp, C.POINTER(DensParams), C.c_int, C.c_double, vp]
        L.ghip_ngb_treefind.argtypes = [vp, vp, C.c_double, C.c_int, C.c_int, C.c_double, vp,
                                        C.c_int, C.POINTER(C.c_int)]
        L.ghip_peano_hilbert_keys.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp]
        L.ghip_morton_keys.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp]
        L.ghip_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.ghip_tree_dump.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp, vp]
        L.ghip_set_dynamic_tree.argtypes = [vp, C.c_int]
        L.ghip_tree_substep.argtypes = [vp, C.c_double]
        L.ghip_tree_kick_nodes.argtypes = [vp, C.c_int, vp, vp]
        L.ghip_tree_dump_dynamic.argtypes = [vp, C.POINTER(C.c_int), vp, vp, vp, vp]
        L.ghip_stream.argtypes = [vp]
        L.ghip_stream.restype = vp
        L.ghip_sync.argtypes = [vp]
        L.ghip_drift.argtypes = [vp, C.POINTER(DriftParams)]
        L.ghip_advance_timesteps.argtypes = [vp, C.POINTER(KickParams), vp, vp]
        L.ghip_timestep_endrun_code.argtypes = [vp]
        L.ghip_velocity_moments.argtypes = [vp, vp, vp, vp]
        L.ghip_download_aos_kick.argtypes = [vp, vp, vp, C.POINTER(Layout)]
        L.ghip_pm_periodic.argtypes = [vp, C.POINTER(PmParams)]
        L.ghip_tree_export.argtypes = [vp, C.POINTER(NodeLayout), C.c_int, C.c_int, C.c_int, vp, vp,
                                       vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.ghip_shard_count.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ghip_shard_pack.argtypes = [vp, C.c_int, vp]
        L.ghip_shard_unpack.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ghip_dd_init.argtypes = [vp, C.c_int, C.c_int]
        L.ghip_dd_set_domain.argtypes = [vp, vp, vp, C.c_double, vp]
        L.ghip_dd_set_splits.argtypes = [vp, vp]
        L.ghip_dd_set_segments.argtypes = [vp, C.c_int, vp, vp]
        L.ghip_dd_keys.argtypes = [vp, vp]
        L.ghip_dd_find_split.argtypes = [C.c_int, C.c_int, vp, vp, vp]
        L.ghip_dd_set_ghost_margin.argtypes = [vp, C.c_double]
        L.ghip_dd_rccl_unique_id.argtypes = [vp]
        L.ghip_dd_rccl_connect.argtypes = [vp, vp]
        L.ghip_dd_rccl_library.restype = C.c_char_p
        L.ghip_dd_begin.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ghip_dd_step.argtypes = [vp]
        L.ghip_dd_exchange.argtypes = [vp]
        L.ghip_dd_exchange_local.argtypes = [vp, C.c_int]
        L.ghip_dd_run.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ghip_dd_exchange_host.argtypes = [vp, ALLGATHER_CB, vp]
        L.ghip_dd_get_info.argtypes = [vp, vp]
        L.ghip_sink_density.argtypes = [vp, C.POINTER(DensParams), C.c_double, C.c_int, vp, vp, vp,
                                        vp, vp, vp, C.POINTER(C.c_int)]
        L.ghip_sink_reset.argtypes = [vp]
        L.ghip_blackhole_evaluate.argtypes = [vp, C.POINTER(BhParams), C.c_int, vp, vp, vp, vp]
        L.ghip_blackhole_swallow.argtypes = [vp, C.POINTER(BhParams), C.c_int, vp, vp, vp, vp, vp,
                                             vp, vp, vp]
        L.ghip_set_async.argtypes = [vp, C.c_int]
        L.ghip_timebin_counts.argtypes = [vp, vp, vp]
        L.ghip_run_begin.argtypes = [vp, C.c_int]
        L.ghip_step_begin.argtypes = [vp]
        L.ghip_step_end.argtypes = [vp]
        L.ghip_run_end.argtypes = [vp, C.POINTER(RunStats)]
        L.ghip_sink_get_marks.argtypes = [vp, vp, vp]
        L.ghip_sink_set_marks.argtypes = [vp, vp, vp]
        L.ghip_cooling_and_starformation.argtypes = [vp, C.c_double, C.c_double, C.c_double,
                                                     C.c_double, vp]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class ForcePath:
    """One device context (one per process / GPU)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.ghip_create(int(device), C.byref(h))
        if rc != 0:
            raise GhipError(rc, "ghip_create(device=%d) failed: no usable GPU; this package has "
                                "no CPU path" % device)
        self.h = h
        self.n = 0
        self.ngas = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.ghip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # walk constant of the overlapped Newton+Ewald pair (sharded.py uses it when present)
    WALK_PAIR = (WALK_NEWTON, WALK_EWALD, WALK_NEWTON_EWALD)

    def _chk(self, rc):
        if rc != 0:
            e = GhipError(rc, self.L.ghip_last_error(self.h).decode())
            if GHIP_ERRORS.get(rc) == "GHIP_ETIMESTEP":   # (also when a deferred kick reports late)
                e.endrun = self.L.ghip_timestep_endrun_code(self.h)
            raise e

    # ---- data ----
    def set_counts(self, n, ngas):
        self._chk(self.L.ghip_set_counts(self.h, int(n), int(ngas)))
        self.n, self.ngas = int(n), int(ngas)

    def _shape(self, field):
        gas, ncomp, isint = _FIELD_INFO[field]
        cnt = self.ngas if gas else self.n
        return ((cnt, 3) if ncomp == 3 else (cnt,)), (np.int32 if isint else np.float64)

    def set_field(self, field, arr):
        shape, dt = self._shape(field)
        a = np.ascontiguousarray(arr, dtype=dt)
        if a.shape != shape:
            raise ValueError("field %d: expected shape %s, got %s" % (field, shape, a.shape))
        self._chk(self.L.ghip_set_field(self.h, field, _ptr(a)))

    def get_field(self, field):
        shape, dt = self._shape(field)
        a = np.zeros(shape, dtype=dt)
        self._chk(self.L.ghip_get_field(self.h, field, _ptr(a)))
        return a

    def upload_aos(self, P, SphP, layout):
        n, ng = len(P), (0 if SphP is None else len(SphP))
        self._chk(self.L.ghip_upload_aos(self.h, _ptr(P), _ptr(SphP) if ng else None,
                                         C.byref(layout), n, ng))
        self.n, self.ngas = n, ng

    def download_aos(self, P, SphP, layout, gravity=True, density=True, hydro=True):
        self._chk(self.L.ghip_download_aos(self.h, _ptr(P), _ptr(SphP) if self.ngas else None,
                                           C.byref(layout), int(gravity), int(density),
                                           int(hydro)))

    def set_active(self, idx=None):
        if idx is None:
            self._chk(self.L.ghip_set_active(self.h, None, 0))
        else:
            a = np.ascontiguousarray(idx, dtype=np.int32)
            self._active_keep = a
            self._chk(self.L.ghip_set_active(self.h, _ptr(a), len(a)))

    def set_shard(self, rank, nranks):
        self._chk(self.L.ghip_set_shard(self.h, int(rank), int(nranks)))

    # ---- the path ----
    def tree_build(self, corner, center, dlen, force_softening):
        c0 = np.ascontiguousarray(corner, dtype=np.float64)
        c1 = np.ascontiguousarray(center, dtype=np.float64)
        sf = np.ascontiguousarray(force_softening, dtype=np.float64)
        self._chk(self.L.ghip_tree_build(self.h, _ptr(c0), _ptr(c1), float(dlen), _ptr(sf)))

    def ewald_init(self, boxsize):
        self._chk(self.L.ghip_ewald_init(self.h, float(boxsize)))

    def ewald_table(self):
        t = np.zeros((3, EN + 1, EN + 1, EN + 1))
        self._chk(self.L.ghip_ewald_get_table(self.h, _ptr(t)))
        return t

    def gravity(self, params, walk=WALK_NEWTON):
        self._chk(self.L.ghip_gravity(self.h, C.byref(params), int(walk)))

    def gravity_ext(self, params, pos, ptype, oldacc, walk=WALK_NEWTON, soft=None):
        """soft: gravdata_in.Soft (Hsml of gas targets), used under set_adaptive_gravsoft."""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        ptype = np.ascontiguousarray(ptype, dtype=np.int32)
        oldacc = np.ascontiguousarray(oldacc, dtype=np.float64)
        soft = None if soft is None else np.ascontiguousarray(soft, dtype=np.float64)
        nt = len(pos)
        acc = np.zeros((nt, 3))
        nint = np.zeros(nt, np.int32)
        self._chk(self.L.ghip_gravity_ext_soft(self.h, C.byref(params), int(walk), nt, _ptr(pos),
                                               _ptr(ptype), None if soft is None else _ptr(soft),
                                               _ptr(oldacc), _ptr(acc), _ptr(nint)))
        return acc, nint

    def set_adaptive_gravsoft(self, on=True):
        """ADAPTIVE_GRAVSOFT_FORGAS: gas softening = Hsml; call before tree_build."""
        self._chk(self.L.ghip_set_adaptive_gravsoft(self.h, int(bool(on))))

    def gravity_finish(self, G):
        self._chk(self.L.ghip_gravity_finish(self.h, float(G)))

    def gravity_finish_ex(self, G, pmgrid=False, comoving_fac=0.0, all_shards=False):
        self._chk(self.L.ghip_gravity_finish_ex(self.h, float(G), int(bool(pmgrid)),
                                                float(comoving_fac), int(bool(all_shards))))

    def gravity_finish_all(self, G):
        self._chk(self.L.ghip_gravity_finish_all(self.h, float(G)))

    def gravity_vacuum_energy(self, fac):
        """GravAccel += fac * Pos (gravtree.c:470-483), fac = OmegaLambda * Hubble^2."""
        self._chk(self.L.ghip_gravity_vacuum_energy(self.h, float(fac)))

    def gravity_direct(self, params):
        self._chk(self.L.ghip_gravity_direct(self.h, C.byref(params)))

    def drift(self, time1, timebase, min_gas_hsml=0.0, box_wrap=False, boxsize=1.0, tables=None,
              log_time_begin=0.0, log_time_max=0.0, pmgrid=False):
        p = DriftParams()
        p.pmgrid = int(bool(pmgrid))
        p.time1, p.Timebase_interval = int(time1), float(timebase)
        p.MinGasHsml, p.box_wrap, p.BoxSize = float(min_gas_hsml), int(box_wrap), float(boxsize)
        if tables is not None:
            self._tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
            p.ComovingIntegrationOn = 1
            p.logTimeBegin, p.logTimeMax = float(log_time_begin), float(log_time_max)
            p.DriftTable, p.GravKickTable, p.HydroKickTable = [t.ctypes.data for t in self._tabs]
        self._chk(self.L.ghip_drift(self.h, C.byref(p)))

    def pm_kick(self, ti_current, timebase, dt_gravkick, dt_gravkickB, kick_tables=None,
                log_time_begin=0.0, log_time_max=0.0):
        """ghip_pm_kick: Vel += GravPM * dt_gravkick for every particle, VelPred of gas rebuilt."""
        p = PmKickParams()
        p.Ti_Current, p.Timebase_interval = int(ti_current), float(timebase)
        p.dt_gravkick, p.dt_gravkickB = float(dt_gravkick), float(dt_gravkickB)
        if kick_tables is not None:
            self._pktabs = [np.ascontiguousarray(t, dtype=np.float64) for t in kick_tables]
            p.ComovingIntegrationOn = 1
            p.logTimeBegin, p.logTimeMax = float(log_time_begin), float(log_time_max)
            p.GravKickTable, p.HydroKickTable = [t.ctypes.data for t in self._pktabs]
        self._chk(self.L.ghip_pm_kick(self.h, C.byref(p)))

    def set_async(self, on=True):
        """ghip_set_async: drift / kick stop waiting for the device (errors surface at the next sync)"""
        self._chk(self.L.ghip_set_async(self.h, int(bool(on))))

    def run_begin(self, max_steps):
        self._chk(self.L.ghip_run_begin(self.h, int(max_steps)))

    def step_begin(self):
        self._chk(self.L.ghip_step_begin(self.h))

    def step_end(self):
        self._chk(self.L.ghip_step_end(self.h))

    def run_end(self):
        s = RunStats()
        self._chk(self.L.ghip_run_end(self.h, C.byref(s)))
        return s.asdict()

    def timebin_counts(self):
        cnt = (C.c_longlong * 32)()
        sph = (C.c_longlong * 32)()
        self._chk(self.L.ghip_timebin_counts(self.h, cnt, sph))
        return np.array(cnt[:], dtype=np.int64), np.array(sph[:], dtype=np.int64)

    def advance_timesteps(self, params, kick_tables=None, counts=True):
        """ghip_advance_timesteps; returns (TimeBinCount[32], TimeBinCountSph[32]) -- or None with
        counts=False (no recount; under set_async the call then does not wait).  A timestep failure
        raises GhipError with .endrun = the reference's endrun code."""
        if kick_tables is not None:
            self._ktabs = [np.ascontiguousarray(t, dtype=np.float64) for t in kick_tables]
            params.GravKickTable, params.HydroKickTable = [t.ctypes.data for t in self._ktabs]
        if not counts:
            self._chk(self.L.ghip_advance_timesteps(self.h, C.byref(params), None, None))
            return None
        cnt = (C.c_longlong * 32)()
        sph = (C.c_longlong * 32)()
        rc = self.L.ghip_advance_timesteps(self.h, C.byref(params), cnt, sph)
        if rc != 0:
            try:
                self._chk(rc)
            except GhipError as e:
                e.endrun = self.L.ghip_timestep_endrun_code(self.h)
                raise
        return np.array(cnt[:], dtype=np.int64), np.array(sph[:], dtype=np.int64)

    def tree_export(self, maxpart=None, ti_current=0, unequal=0, max_nodes=None, adaptive=False):
        """ghip_tree_export into numpy records of the minimal-flag-set NODE / extNODE layout
        (adaptive: the 96-byte NODE of an ADAPTIVE_GRAVSOFT_FORGAS build, with maxsoft);
        returns (Nodes, Extnodes, Nextnode, Father) with Nodes[k] = node maxpart + k."""
        maxpart = self.n if maxpart is None else int(maxpart)
        max_nodes = int(2.0 * maxpart + 16) if max_nodes is None else int(max_nodes)
        lay = NodeLayout()
        ndt = NODE_ADAPTIVE_DTYPE if adaptive else NODE_DTYPE
        lay.node_stride = ndt.itemsize
        lay.n_maxsoft = ndt.fields["maxsoft"][1] if adaptive else -1
        for name, key in (("len", "n_len"), ("center", "n_center"), ("s", "n_s"), ("mass", "n_mass"),
                          ("bitflags", "n_bitflags"), ("sibling", "n_sibling"),
                          ("nextnode", "n_nextnode"), ("father", "n_father"),
                          ("Ti_current", "n_ti_current")):
            setattr(lay, key, ndt.fields[name][1])
        lay.ext_stride = EXTNODE_DTYPE.itemsize
        for name, key in (("dp", "e_dp"), ("vs", "e_vs"), ("vmax", "e_vmax"),
                          ("divVmax", "e_divvmax"), ("hmax", "e_hmax"),
                          ("Ti_lastkicked", "e_ti_lastkicked"), ("Flag", "e_flag")):
            setattr(lay, key, EXTNODE_DTYPE.fields[name][1])
        nodes = np.zeros(max_nodes, ndt)
        ext = np.zeros(max_nodes, EXTNODE_DTYPE)
        nxt = np.full(maxpart, -1, np.int32)
        fat = np.full(maxpart, -1, np.int32)
        nn = C.c_int(0)
        self._chk(self.L.ghip_tree_export(self.h, C.byref(lay), maxpart, int(ti_current),
                                          int(unequal), _ptr(nodes), _ptr(ext), _ptr(nxt), _ptr(fat),
                                          max_nodes, C.byref(nn)))
        return nodes[:nn.value], ext[:nn.value], nxt, fat

    def pm_periodic(self, pmgrid, boxsize, G, asmth=None):
        """ghip_pm_periodic: fills F_GRAVPM; asmth defaults to ASMTH * BoxSize / PMGRID."""
        p = PmParams(int(pmgrid), float(boxsize), float(G),
                     float(1.25 * boxsize / pmgrid if asmth is None else asmth))
        self._chk(self.L.ghip_pm_periodic(self.h, C.byref(p)))

    def velocity_moments(self):
        v2 = (C.c_double * 6)()
        mm = (C.c_double * 6)()
        cnt = (C.c_longlong * 6)()
        self._chk(self.L.ghip_velocity_moments(self.h, v2, mm, cnt))
        return np.array(v2[:]), np.array(mm[:]), np.array(cnt[:], dtype=np.int64)

    def density(self, params):
        self._chk(self.L.ghip_density(self.h, C.byref(params)))

    def update_hmax(self):
        self._chk(self.L.ghip_update_hmax(self.h))

    def hydro(self, params):
        self._chk(self.L.ghip_hydro(self.h, C.byref(params)))

    def density_evaluate(self, params, target, h):
        out = np.zeros(7)
        self._chk(self.L.ghip_density_evaluate(self.h, C.byref(params), int(target), float(h),
                                               _ptr(out)))
        return out

    def ngb_treefind(self, center, hsml, pairs, periodic, boxsize, cap=None):
        c = np.ascontiguousarray(center, dtype=np.float64)
        cap = self.ngas if cap is None else cap
        buf = np.zeros(max(cap, 1), np.int32)
        nf = C.c_int(0)
        self._chk(self.L.ghip_ngb_treefind(self.h, _ptr(c), float(hsml), int(pairs),
                                           int(periodic), float(boxsize), _ptr(buf), int(cap),
                                           C.byref(nf)))
        return buf[:min(nf.value, cap)].copy(), nf.value

    def peano_hilbert_keys(self, x, y, z, bits=21):
        x, y, z = [np.ascontiguousarray(a, dtype=np.int32) for a in (x, y, z)]
        k = np.zeros(len(x), np.uint64)
        self._chk(self.L.ghip_peano_hilbert_keys(self.h, len(x), _ptr(x), _ptr(y), _ptr(z),
                                                 int(bits), _ptr(k)))
        return k

    def morton_keys(self, x, y, z, bits=21):
        x, y, z = [np.ascontiguousarray(a, dtype=np.int32) for a in (x, y, z)]
        k = np.zeros(len(x), np.uint64)
        self._chk(self.L.ghip_morton_keys(self.h, len(x), _ptr(x), _ptr(y), _ptr(z), int(bits),
                                          _ptr(k)))
        return k

    def stats(self):
        s = Stats()
        self._chk(self.L.ghip_get_stats(self.h, C.byref(s)))
        return s.asdict()

    def tree_dump(self, which=0):
        ne = C.c_int(0)
        self._chk(self.L.ghip_tree_dump(self.h, which, C.byref(ne), None, None, None, None, None))
        ne = ne.value
        npart = self.ngas if which else self.n
        out = dict(xm=np.zeros((ne, 4)), cl=np.zeros((ne, 4)), lk=np.zeros((ne, 4), np.int32),
                   aux=np.zeros(ne), perm=np.zeros(npart, np.int32))
        n2 = C.c_int(0)
        self._chk(self.L.ghip_tree_dump(self.h, which, C.byref(n2), _ptr(out["xm"]),
                                        _ptr(out["cl"]), _ptr(out["lk"]), _ptr(out["aux"]),
                                        _ptr(out["perm"])))
        return out

    # ---- sub-steps on the tree of the last full build (forcetree.c:1356-1651) ----
    def set_massless_gas_rule(self, rule=3):
        self._chk(self.L.ghip_set_massless_gas_rule(self.h, int(rule)))

    def set_dynamic_tree(self, on=True):
        self._chk(self.L.ghip_set_dynamic_tree(self.h, int(bool(on))))

    def tree_substep(self, dt_drift):
        self._chk(self.L.ghip_tree_substep(self.h, float(dt_drift)))

    def tree_kick_nodes(self, idx, dv):
        idx = np.ascontiguousarray(idx, np.int32)
        dv = np.ascontiguousarray(dv, np.float64)
        assert dv.shape == (len(idx), 3)
        self._chk(self.L.ghip_tree_kick_nodes(self.h, len(idx), _ptr(idx), _ptr(dv)))

    def tree_dump_dynamic(self):
        ne = C.c_int(0)
        self._chk(self.L.ghip_tree_dump_dynamic(self.h, C.byref(ne), None, None, None, None))
        ne = ne.value
        out = dict(xm=np.zeros((ne, 4)), cl=np.zeros((ne, 4)), ev=np.zeros((ne, 4)),
                   lk=np.zeros((ne, 4), np.int32))
        n2 = C.c_int(0)
        self._chk(self.L.ghip_tree_dump_dynamic(self.h, C.byref(n2), _ptr(out["xm"]), _ptr(out["cl"]),
                                                _ptr(out["ev"]), _ptr(out["lk"])))
        return out

    def sync(self):
        self._chk(self.L.ghip_sync(self.h))

    @property
    def stream(self):
        return self.L.ghip_stream(self.h)

    # ---- "next" row N4: sinks ----
    def sink_density(self, params, ngb_factor, sinks, hsml):
        sinks = np.ascontiguousarray(sinks, np.int32)
        ns = len(sinks)
        out = dict(hsml=np.ascontiguousarray(hsml, np.float64).copy(), numngb=np.zeros(ns),
                   density=np.zeros(ns), entropy=np.zeros(ns), gasvel=np.zeros((ns, 3)))
        it = C.c_int(0)
        self._chk(self.L.ghip_sink_density(self.h, C.byref(params), float(ngb_factor), ns, _ptr(sinks),
                                           _ptr(out["hsml"]), _ptr(out["numngb"]),
                                           _ptr(out["density"]), _ptr(out["entropy"]),
                                           _ptr(out["gasvel"]), C.byref(it)))
        out["iterations"] = it.value
        return out

    def sink_reset(self):
        self._chk(self.L.ghip_sink_reset(self.h))

    def blackhole_evaluate(self, params, sinks, sink_ids, mdot, bh_density):
        sinks = np.ascontiguousarray(sinks, np.int32)
        ids = np.ascontiguousarray(sink_ids, np.uint32)
        md = np.ascontiguousarray(mdot, np.float64)
        rho = np.ascontiguousarray(bh_density, np.float64)
        self._chk(self.L.ghip_blackhole_evaluate(self.h, C.byref(params), len(sinks), _ptr(sinks),
                                                 _ptr(ids), _ptr(md), _ptr(rho)))

    def blackhole_swallow(self, params, sinks, sink_ids, sink_bh_mass):
        sinks = np.ascontiguousarray(sinks, np.int32)
        ids = np.ascontiguousarray(sink_ids, np.uint32)
        ns = len(sinks)
        out = dict(bh_mass=np.ascontiguousarray(sink_bh_mass, np.float64).copy(),
                   acc_mass=np.zeros(ns), acc_bhmass=np.zeros(ns), acc_dustmass=np.zeros(ns),
                   acc_momentum=np.zeros((ns, 3)), counts=np.zeros(3, np.int64))
        self._chk(self.L.ghip_blackhole_swallow(self.h, C.byref(params), ns, _ptr(sinks), _ptr(ids),
                                                _ptr(out["bh_mass"]), _ptr(out["acc_mass"]),
                                                _ptr(out["acc_bhmass"]), _ptr(out["acc_dustmass"]),
                                                _ptr(out["acc_momentum"]), _ptr(out["counts"])))
        return out

    def sink_marks(self):
        sw = np.zeros(self.n, np.uint32)
        inj = np.zeros(self.ngas)
        self._chk(self.L.ghip_sink_get_marks(self.h, _ptr(sw), _ptr(inj)))
        return sw, inj

    def set_sink_marks(self, swallow_id=None, injected=None):
        sw = None if swallow_id is None else np.ascontiguousarray(swallow_id, np.uint32)
        inj = None if injected is None else np.ascontiguousarray(injected, np.float64)
        self._chk(self.L.ghip_sink_set_marks(self.h, _ptr(sw), _ptr(inj)))

    def cooling_and_starformation(self, timebase, crit_density, min_egy, u_to_temp_fac):
        flag = np.zeros(self.ngas, np.int32)
        self._chk(self.L.ghip_cooling_and_starformation(self.h, float(timebase), float(crit_density),
                                                        float(min_egy), float(u_to_temp_fac),
                                                        _ptr(flag)))
        return flag

    # ---- multi-GPU: domain decomposition with tree-node / ghost exchange (include/ghip.h) ----
    def dd_init(self, rank, nranks):
        self._chk(self.L.ghip_dd_init(self.h, int(rank), int(nranks)))

    def dd_set_domain(self, corner, center, dlen, force_softening):
        c0 = np.ascontiguousarray(corner, np.float64)
        c1 = np.ascontiguousarray(center, np.float64)
        so = np.ascontiguousarray(force_softening, np.float64)
        self._chk(self.L.ghip_dd_set_domain(self.h, _ptr(c0), _ptr(c1), float(dlen), _ptr(so)))

    def dd_set_splits(self, splits):
        sp = np.ascontiguousarray(splits, np.uint64)
        self._chk(self.L.ghip_dd_set_splits(self.h, _ptr(sp)))

    def dd_keys(self):
        out = np.zeros(self.n, np.uint64)
        self._chk(self.L.ghip_dd_keys(self.h, _ptr(out)))
        return out

    def dd_set_ghost_margin(self, margin):
        self._chk(self.L.ghip_dd_set_ghost_margin(self.h, float(margin)))

    def dd_begin(self, op, params, walk=0):
        self._dd_params = params          # keep the struct alive
        ptr = None if params is None else C.cast(C.byref(params), C.c_void_p)
        self._chk(self.L.ghip_dd_begin(self.h, int(op), ptr, int(walk)))

    def counts(self):
        """(numpart, ngas) of the context -- they change when particles migrate"""
        out = np.zeros(16, np.int64)
        self._chk(self.L.ghip_dd_get_info(self.h, _ptr(out)))
        self.n, self.ngas = int(out[11]), int(out[12])
        return self.n, self.ngas

    def dd_step(self):
        rc = self.L.ghip_dd_step(self.h)
        if rc < 0:
            self._chk(rc)
        return rc

    def dd_exchange(self):
        self._chk(self.L.ghip_dd_exchange(self.h))

    @staticmethod
    def allgather_callback(allgather):
        """`allgather(send: bytes) -> bytes of all ranks, rank-major` as the C callback of
        ghip_dd_exchange_host (keep the returned object alive while it is in use)."""
        def cb(_user, send, nbytes, recv):
            try:
                data = C.string_at(send, nbytes)
                out = allgather(data)
                C.memmove(recv, out, len(out))
                return 0
            except Exception:   # noqa: BLE001 -- reported through the C return code
                import traceback
                traceback.print_exc()
                return 1
        return ALLGATHER_CB(cb)

    def dd_exchange_host(self, fn):
        self._chk(self.L.ghip_dd_exchange_host(self.h, fn, None))

    def dd_set_segments(self, keys, owner):
        keys = np.ascontiguousarray(keys, np.uint64)
        owner = np.ascontiguousarray(owner, np.int32)
        assert len(keys) == len(owner) + 1
        self._chk(self.L.ghip_dd_set_segments(self.h, len(owner), _ptr(keys), _ptr(owner)))

    def dd_run_host(self, op, params, allgather, walk=0):
        """The operation with every exchange staged through the host and `allgather(send: bytes)
        -> bytes of all ranks, rank-major` (ghip_dd_exchange_host)."""
        fn = self.allgather_callback(allgather)
        self.dd_begin(op, params, walk)
        while True:
            rc = self.dd_step()
            if rc == 0:
                break
            self._chk(self.L.ghip_dd_exchange_host(self.h, fn, None))
        if op == DD_MIGRATE:
            self.counts()

    def dd_run(self, op, params, walk=0):
        ptr = None if params is None else C.cast(C.byref(params), C.c_void_p)
        self._chk(self.L.ghip_dd_run(self.h, int(op), ptr, int(walk)))
        if op == DD_MIGRATE:
            self.counts()

    def dd_rccl_connect(self, id128):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(self.L.ghip_dd_rccl_connect(self.h, C.cast(buf, C.c_void_p)))

    def dd_info(self):
        out = np.zeros(16, np.int64)
        self._chk(self.L.ghip_dd_get_info(self.h, _ptr(out)))
        keys = ("rank", "nranks", "let_imported", "let_sent", "ghosts_imported", "ghosts_sent",
                "bytes_gravity", "bytes_density", "hsml_growth_e6", "grav_elements", "gas_elements",
                "numpart", "ngas", "migrated_out", "migrated_in", "bytes_migrate")
        return dict(zip(keys, (int(v) for v in out)))

    # ---- multi-GPU shard exchange helpers (device pointers, e.g. torch tensors' data_ptr()) ----
    def shard_count(self, gas):
        per = C.c_int(0)
        mine = C.c_int(0)
        self._chk(self.L.ghip_shard_count(self.h, int(gas), C.byref(per), C.byref(mine)))
        return per.value, mine.value

    def shard_pack(self, group, dev_ptr):
        self._chk(self.L.ghip_shard_pack(self.h, int(group), C.c_void_p(dev_ptr)))

    def shard_unpack(self, group, dev_ptr, nranks):
        self._chk(self.L.ghip_shard_unpack(self.h, int(group), C.c_void_p(dev_ptr), int(nranks)))


# ---- multi-GPU module-level helpers ----
DD_MIGRATE, DD_GRAVITY, DD_DENSITY, DD_HYDRO = 1, 2, 3, 4
DD_SINK_DENSITY, DD_BH_EVALUATE, DD_BH_SWALLOW, DD_PM = 5, 6, 7, 8


class DdSinkArgs(C.Structure):
    """ghip_dd_sink_args (include/ghip.h): the sink passes on a multi-GPU shard"""
    _fields_ = [("dens", C.POINTER(DensParams)), ("ngb_factor", C.c_double),
                ("bh", C.POINTER(BhParams)), ("nsink", C.c_int), ("sink_idx", C.c_void_p),
                ("sink_id", C.c_void_p), ("hsml", C.c_void_p), ("numngb", C.c_void_p),
                ("bh_density", C.c_void_p), ("bh_entropy", C.c_void_p), ("bh_gasvel", C.c_void_p),
                ("bh_mdot", C.c_void_p), ("bh_density_in", C.c_void_p),
                ("sink_bh_mass", C.c_void_p), ("acc_mass", C.c_void_p), ("acc_bhmass", C.c_void_p),
                ("acc_dustmass", C.c_void_p), ("acc_momentum", C.c_void_p), ("counts", C.c_void_p)]


def dd_sink_args(sinks, sink_ids=None, dens=None, ngb_factor=1.0, bh=None, hsml=None, mdot=None,
                 bh_density=None, sink_bh_mass=None):
    """(args, arrays): a filled DdSinkArgs and the dict of numpy arrays it points into -- inputs
    copied, outputs allocated; keep `arrays` alive until the operation has finished."""
    ns = len(sinks)
    f64 = lambda v: None if v is None else np.ascontiguousarray(v, np.float64).copy()
    a = dict(sink_idx=np.ascontiguousarray(sinks, np.int32),
             sink_id=None if sink_ids is None else np.ascontiguousarray(sink_ids, np.uint32),
             hsml=f64(hsml), numngb=np.zeros(ns), density=np.zeros(ns), entropy=np.zeros(ns),
             gasvel=np.zeros((ns, 3)), mdot=f64(mdot), bh_density_in=f64(bh_density),
             bh_mass=f64(sink_bh_mass), acc_mass=np.zeros(ns), acc_bhmass=np.zeros(ns),
             acc_dustmass=np.zeros(ns), acc_momentum=np.zeros((ns, 3)), counts=np.zeros(3, np.int64),
             _dens=dens, _bh=bh)
    A = DdSinkArgs()
    A.dens = C.pointer(dens) if dens is not None else None
    A.ngb_factor = float(ngb_factor)
    A.bh = C.pointer(bh) if bh is not None else None
    A.nsink = ns
    vp = lambda v: None if v is None else v.ctypes.data
    A.sink_idx, A.sink_id, A.hsml = vp(a["sink_idx"]), vp(a["sink_id"]), vp(a["hsml"])
    A.numngb, A.bh_density, A.bh_entropy = vp(a["numngb"]), vp(a["density"]), vp(a["entropy"])
    A.bh_gasvel, A.bh_mdot, A.bh_density_in = vp(a["gasvel"]), vp(a["mdot"]), vp(a["bh_density_in"])
    A.sink_bh_mass, A.acc_mass, A.acc_bhmass = vp(a["bh_mass"]), vp(a["acc_mass"]), vp(a["acc_bhmass"])
    A.acc_dustmass, A.acc_momentum, A.counts = vp(a["acc_dustmass"]), vp(a["acc_momentum"]), vp(a["counts"])
    return A, a


def dd_rccl_unique_id():
    """128 bytes of ncclGetUniqueId (rank 0 calls it, the host broadcasts them)."""
    buf = (C.c_char * 128)()
    rc = lib().ghip_dd_rccl_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:
        raise GhipError(rc, "ghip_dd_rccl_unique_id failed (no librccl next to the HIP runtime?)")
    return bytes(buf)


def dd_rccl_library():
    return lib().ghip_dd_rccl_library().decode()


def dd_exchange_local(paths):
    """Run the pending exchange of several shards living in this process (ghip_dd_exchange_local)."""
    arr = (C.c_void_p * len(paths))(*[p.h for p in paths])
    rc = lib().ghip_dd_exchange_local(C.cast(arr, C.c_void_p), len(paths))
    if rc != 0:
        raise GhipError(rc, lib().ghip_last_error(paths[0].h).decode())


def dd_run_local(paths, op, params, walk=0):
    """One operation on all shards of this process: compute phases shard after shard, exchanges in
    between (what ghip_dd_run does over RCCL with one shard per process)."""
    for p, prm in zip(paths, params):
        p.dd_begin(op, prm, walk)
    while True:
        rcs = [p.dd_step() for p in paths]
        if any(r != rcs[0] for r in rcs):
            raise RuntimeError("shards out of step: %r" % (rcs,))
        if rcs[0] == 0:
            if op == DD_MIGRATE:
                for p in paths:
                    p.counts()
            return
        dd_exchange_local(paths)


def dd_find_split(ncpu, work):
    """domain_findSplit_work_balanced (domain.c:1075-1113) through the library's host function."""
    w = np.ascontiguousarray(work, np.float64)
    start = np.zeros(ncpu, np.int32)
    end = np.zeros(ncpu, np.int32)
    rc = lib().ghip_dd_find_split(int(ncpu), len(w), _ptr(w), _ptr(start), _ptr(end))
    if rc != 0:
        raise GhipError(rc, "ghip_dd_find_split: bad arguments")
    return start, end
