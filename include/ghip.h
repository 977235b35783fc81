/*
 * ghip.h -- C-ABI of libghip.so: the MI355X (gfx950) force path of GADGET-3 (Leicester fork).
 *
 * Plain C, no C++/torch types.  One ghip_ctx per process / GPU / MPI rank (the reference is
 * single-threaded per rank, SURVEY.md 8b "Threading").  Every entry point returns 0 on success
 * or a negative GHIP_E* code and never calls exit(); the host glue maps a failure to the
 * reference's endrun(code) convention (endrun.c:23-38) -- see INTEGRATION.md.
 *
 * What each group replaces in the reference:
 *   ghip_upload_aos / ghip_download_aos   the lazy reads/writes of P[] / SphP[] inside the walks
 *                                         (allvars.h:1131-1377, 1384-1639)
 *   ghip_tree_build                       force_treebuild()            forcetree.c:67-872
 *   ghip_gravity                          the active-list loop over force_treeevaluate*()
 *                                         gravtree.c:130-168 + forcetree.c:1797, 2330, 2873
 *   ghip_ewald_init                       ewald_init()                 forcetree.c:4402-4527
 *   ghip_density                          density() incl. h iteration  density.c:89-704, 711-1029
 *   ghip_update_hmax                      force_update_hmax()          forcetree.c:1661-1786
 *   ghip_hydro                            hydro_force()                hydra.c:145-813, 822-1995
 *   ghip_peano_hilbert_keys               peano_hilbert_key()          peano.c:300-316
 *   ghip_gravity_finish / _vacuum_energy  OldAcc, *G, Lambda term      gravtree.c:381-403, 470-483
 *   ghip_set_adaptive_gravsoft            -DADAPTIVE_GRAVSOFT_FORGAS   forcetree.c:705-726, 2038-2139
 *   ghip_drift                            drift_particle()             predict.c:129-259
 *   ghip_advance_timesteps / ghip_pm_kick get_timestep, do_the_kick, long-range kick
 *                                                                      timestep.c:29-605
 *   ghip_tree_export                      Nodes[] / Extnodes[] / Nextnode[] / Father[] as
 *                                         force_treebuild leaves them  allvars.h:1847-1916
 *   ghip_pm_periodic                      pmforce_periodic()           pm_periodic.c:199-800
 * The host-side mirror with the reference's own names (gravity_tree(), density(), hydro_force(),
 * force_treeevaluate(), ...) is include/gadget_force.h.
 */
#ifndef GHIP_H
#define GHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ghip_ctx ghip_ctx;

/* error codes */
#define GHIP_OK 0
#define GHIP_EHIP (-90001)       /* a HIP runtime call failed (message in ghip_last_error) */
#define GHIP_EINVAL (-90002)     /* bad argument / call order */
#define GHIP_ENOMEM (-90003)     /* device or host allocation failed */
#define GHIP_ENOCONV (-90004)    /* density h-iteration did not converge (reference: endrun(1155)) */
#define GHIP_ENODEVICE (-90005)  /* no usable gfx950 device */
#define GHIP_EDEVICE (-90008)    /* an internal invariant broke on the device (plan overflow, a pruned
                                  * node of an imported tree that a target had to open, ...); results
                                  * of the step must not be used */
#define GHIP_ECOMM (-90009)      /* an RCCL call of the multi-GPU exchange failed */
#define GHIP_ETIMESTEP (-90006)  /* timestep criterion failed (reference: endrun(888|818|112313));
                                  * the code is returned by ghip_timestep_endrun_code */

/* particle fields held on the device in the host's particle order.  3-vectors are [n][3]
 * doubles on the host side; ints are 32-bit. */
enum ghip_field
{
  GHIP_F_POS = 0,        /* P[].Pos           [n][3] f64 in  */
  GHIP_F_VEL,            /* P[].Vel           [n][3] f64 in  (node vs/vmax only) */
  GHIP_F_MASS,           /* P[].Mass          [n]    f64 in  */
  GHIP_F_TYPE,           /* P[].Type          [n]    i32 in  */
  GHIP_F_OLDACC,         /* P[].OldAcc        [n]    f64 in/out */
  GHIP_F_HSML,           /* PPP[].Hsml        [n]    f64 in/out (gas entries used) */
  GHIP_F_TIMEBIN,        /* P[].TimeBin       [n]    i32 in  */
  GHIP_F_TI_BEGSTEP,     /* P[].Ti_begstep    [n]    i32 in  */
  GHIP_F_VELPRED,        /* SphP[].VelPred    [ngas][3] f64 in */
  GHIP_F_ENTROPY,        /* SphP[].Entropy    [ngas] f64 in  */
  GHIP_F_DTENTROPY,      /* SphP[].e.DtEntropy [ngas] f64 in (pressure prediction) / out (hydro) */
  GHIP_F_GRAVACCEL,      /* P[].g.GravAccel   [n][3] f64 out */
  GHIP_F_GRAVCOST,       /* P[].GravCost      [n]    i32 out (ninteractions) */
  GHIP_F_NUMNGB,         /* PPP[].n.NumNgb    [ngas] f64 out */
  GHIP_F_DENSITY,        /* SphP[].d.Density  [ngas] f64 in/out */
  GHIP_F_DHSMLFAC,       /* SphP[].h.DhsmlDensityFactor [ngas] f64 in/out */
  GHIP_F_DIVVEL,         /* SphP[].v.DivVel   [ngas] f64 in/out */
  GHIP_F_CURLVEL,        /* SphP[].r.CurlVel  [ngas] f64 in/out */
  GHIP_F_PRESSURE,       /* SphP[].Pressure   [ngas] f64 in/out */
  GHIP_F_HYDROACCEL,     /* SphP[].a.HydroAccel [ngas][3] f64 out */
  GHIP_F_MAXSIGNALVEL,   /* SphP[].MaxSignalVel [ngas] f64 out */
  GHIP_F_TI_CURRENT,     /* P[].Ti_current    [n]    i32 in/out (ghip_drift) */
  GHIP_F_GRAVPM,         /* P[].GravPM        [n][3] f64 out (ghip_pm_periodic; PMGRID builds) */
  GHIP_F_ID,             /* P[].ID            [n]    i32 in  (carried along by the multi-GPU
                          *                                   migration; no kernel reads it) */
  GHIP_F_COUNT
};

/* Byte layout of the host's AoS records (struct particle_data / sph_particle_data are
 * compile-flag dependent, allvars.h:1131-1639; offsets come from a probe TU).
 * An offset of -1 means "field absent".  Hsml/NumNgb live in P when BLACK_HOLES||DUST, else in
 * SphP (the PPP macro, allvars.h:266-270): set exactly one of the two pairs. */
typedef struct
{
  int p_stride, p_pos, p_vel, p_mass, p_gravaccel, p_oldacc, p_gravcost /* f32 */;
  int p_ti_begstep /* i32 */, p_type /* i16 */, p_timebin /* i16 */;
  int p_hsml, p_numngb; /* when PPP == P, else -1 */
  int s_stride, s_entropy, s_pressure, s_velpred, s_maxsignalvel, s_density, s_dtentropy;
  int s_hydroaccel, s_dhsmlfac, s_divvel, s_curlvel;
  int s_hsml, s_numngb; /* when PPP == SphP, else -1 */
  int p_ti_current;     /* i32 */
  int p_gravpm;         /* P[].GravPM (PMGRID builds, allvars.h:1180-1183): uploaded with the records
                           and written back with the gravity results; -1 otherwise */
} ghip_layout;

typedef struct
{
  double ErrTolTheta;        /* All.ErrTolTheta: != 0 Barnes-Hut, == 0 relative criterion */
  double ErrTolForceAcc;     /* All.ErrTolForceAcc */
  double ForceSoftening[6];  /* All.ForceSoftening[] = 2.8 * SofteningTable[] (gravtree.c:881) */
  double BoxSize;            /* All.BoxSize */
  int periodic;              /* built with PERIODIC */
  int unequal_softenings;    /* built with UNEQUALSOFTENINGS */
  double Rcut, Asmth;        /* All.Rcut[0], All.Asmth[0] (PMGRID short-range walk only) */
} ghip_grav_params;

#define GHIP_WALK_NEWTON 0      /* force_treeevaluate                   forcetree.c:1797 */
#define GHIP_WALK_SHORTRANGE 1  /* force_treeevaluate_shortrange        forcetree.c:2330 */
#define GHIP_WALK_EWALD 2       /* force_treeevaluate_ewald_correction  forcetree.c:2873 (adds) */
#define GHIP_WALK_NEWTON_EWALD 3 /* both of gravity_tree's passes of a PERIODIC && !PMGRID build
                                  * (gravtree.c:130-168) in one call: same results as 0 then 2, but
                                  * the two walks share the device (one is bound by fp64 issue, the
                                  * other by the table gathers) */

typedef struct
{
  double DesNumNgb, MaxNumNgbDeviation, MinGasHsml;  /* All.* (density.c:559-646) */
  double BoxSize;
  int periodic;
  int Ti_Current;            /* All.Ti_Current */
  double Timebase_interval;  /* All.Timebase_interval */
  int MaxIter;               /* MAXITER (150) */
} ghip_dens_params;

typedef struct
{
  double ArtBulkViscConst;   /* All.ArtBulkViscConst */
  double BoxSize;
  int periodic;
  int ComovingIntegrationOn;
  double hubble_a2, fac_mu, fac_vsic_fix; /* hydra.c:192-208 (1 when not comoving) */
  double Timebase_interval;
  int raw_dtentropy;         /* 1: leave DtEntropy as hydro_evaluate's raw sum (hydra.c:1934), skip
                              * hydro_force's conversion to dA/dt (hydra.c:583) */
} ghip_hydro_params;

/* drift_particle() for every particle + optional do_box_wrapping() (predict.c:129-259, 282-310) */
typedef struct
{
  int time1;                 /* All.Ti_Current to drift to */
  double Timebase_interval;
  int ComovingIntegrationOn;
  double logTimeBegin, logTimeMax;                          /* driftfac.c:20 */
  const double *DriftTable, *GravKickTable, *HydroKickTable; /* host, 1000 entries each */
  double MinGasHsml;
  int box_wrap;              /* also apply do_box_wrapping() with BoxSize */
  double BoxSize;
  int pmgrid;                /* PMGRID: VelPred += (GravAccel + GravPM) * dt_gravkick (predict.c:181-184) */
} ghip_drift_params;

/* "next" row N1: timestep criterion + kick for the active particles
 * (advance_and_find_timesteps timestep.c:29-362, get_timestep :607-1123 with
 * TypeOfTimestepCriterion 0, do_the_kick :364-605; minimal flag set) */
typedef struct
{
  int Ti_Current;            /* All.Ti_Current */
  double Timebase_interval;
  int ComovingIntegrationOn;
  double Time;               /* All.Time */
  double hubble_a;           /* hubble_function(All.Time) (timestep.c:56); ignored when not comoving */
  double ErrTolIntAccuracy, CourantFac, MaxSizeTimestep, MinSizeTimestep;
  double dt_displacement;    /* find_dt_displacement_constraint (timestep.c:1125) */
  double SofteningTable[6];  /* All.SofteningTable (set_softenings, gravtree.c:839) */
  double MinEgySpec;
  unsigned int TimeBinActive; /* bit b set <=> TimeBinActive[b] (timestep.c:163) */
  double logTimeBegin, logTimeMax;                /* driftfac.c:20 */
  const double *GravKickTable, *HydroKickTable;   /* host, 1000 entries each; comoving only */
  int AdaptiveGravsoftForGasHsml; /* ADAPTIVE_GRAVSOFT_FORGAS + _HSML: the gravity criterion of a
                                     gas particle uses Hsml/2.8 as its softening (timestep.c:740-743) */
  int pmgrid;                /* PMGRID: GRAVPM is added to the acceleration of the gravity criterion
                                (timestep.c:648-652) and VelPred += GravPM * dt_gravkickB (:511-513) */
  double dt_gravkickB;       /* timestep.c:66-72, from All.PM_Ti_begstep / PM_Ti_endstep */
} ghip_kick_params;

/* the long-range kick that ends a PM step (timestep.c:269-345).  The integer-timeline bookkeeping
 * (new PM step, All.PM_Ti_begstep/endstep, :273-300) stays with the host, which passes the two kick
 * factors: dt_gravkick over [mid-point of the old PM step, mid-point of the new one] and the new
 * dt_gravkickB (:296-300).  In/out VEL, VELPRED; in GRAVPM, GRAVACCEL, HYDROACCEL, TIMEBIN,
 * TI_BEGSTEP, TYPE.  ALL particles are kicked, not the active list. */
typedef struct
{
  int Ti_Current;
  double Timebase_interval;
  int ComovingIntegrationOn;
  double logTimeBegin, logTimeMax;
  const double *GravKickTable, *HydroKickTable;   /* host, 1000 entries each; comoving only */
  double dt_gravkick, dt_gravkickB;
} ghip_pmkick_params;

/* "next" row N2: byte offsets of struct NODE / struct extNODE (allvars.h:1847-1916) as the host
 * was compiled; -1 = member absent / not wanted.  Vectors are 3 consecutive doubles. */
typedef struct
{
  int node_stride, n_len, n_center, n_s, n_mass, n_bitflags, n_sibling, n_nextnode, n_father,
    n_ti_current;
  int ext_stride, e_dp, e_vs, e_vmax, e_divvmax, e_hmax, e_ti_lastkicked, e_flag;
  int n_maxsoft;   /* NODE.maxsoft (allvars.h:1857-1860, ADAPTIVE_GRAVSOFT_FORGAS builds only) */
} ghip_node_layout;

/* "next" row N3: periodic particle-mesh long-range force (pmforce_periodic, pm_periodic.c:199) */
typedef struct
{
  int pmgrid;        /* PMGRID (even) */
  double BoxSize;
  double G;          /* All.G: GravPM comes out with G applied, as in the reference (:224) */
  double Asmth;      /* All.Asmth[0] = ASMTH * BoxSize / PMGRID (pm_periodic.c:83) */
} ghip_pm_params;

/* work counters of the last phase, counted exactly as the reference counts them
 * (SURVEY.md 8d): used for roofline.achieved */
typedef struct
{
  long long grav_interactions;   /* sum of ninteractions over targets (forcetree.c:2214) */
  long long grav_targets;
  long long ewald_interactions;  /* sum of cost (forcetree.c:3170) */
  long long dens_neighbours;     /* neighbours with r2 < h2, summed over h-iterations (density.c:856) */
  long long dens_target_evals;   /* target evaluations summed over h-iterations */
  int dens_iterations;
  long long hydro_pairs;         /* pairs passing hydra.c:1266-1269 with r > 0 */
  long long hydro_targets;
  int tree_nodes, gastree_nodes;
  /* device time of the last call of each phase, ms, measured with hipEvents on the ctx stream */
  float ms_tree, ms_grav, ms_ewald, ms_dens, ms_hmax, ms_hydro;
  /* walk efficiency: elements visited summed over wavefronts (64 targets share each visit) */
  long long grav_wave_steps, ewald_wave_steps;
  float ms_kick;                 /* k_advance_timesteps of the last ghip_advance_timesteps */
  float ms_pm;                   /* the last ghip_pm_periodic (deposit .. interpolation) */
} ghip_stats;

/* ---- lifetime ---- */
int ghip_create(int device, ghip_ctx **out);
void ghip_destroy(ghip_ctx *ctx);
const char *ghip_last_error(const ghip_ctx *ctx);
const char *ghip_version(void);

/* ---- particle data ---- */
/* declare particle counts (gas = indices [0,ngas), allvars.h:1384); (re)allocates device arrays */
int ghip_set_counts(ghip_ctx *ctx, int numpart, int ngas);
/* copy one field host->device / device->host, host arrays in the plain layouts listed above */
int ghip_set_field(ghip_ctx *ctx, int field, const void *host);
int ghip_get_field(ghip_ctx *ctx, int field, void *host);
/* whole-record path: H2D of the raw P[]/SphP[] blocks + device-side unpack, and the reverse
 * (results are packed into the device image of the records, then one D2H per block) */
int ghip_upload_aos(ghip_ctx *ctx, const void *P, const void *SphP, const ghip_layout *lay,
                    int numpart, int ngas);
/* The same in two calls, for a host that starts the gravity walks before the gas data are across:
 * ghip_upload_aos_particles moves the P[] block (all the gravity tree and its walks read, unless
 * ADAPTIVE_GRAVSOFT_FORGAS needs smoothing lengths that live in SphP[]: refused then), and
 * ghip_upload_aos_gas the SphP[] block -- it does not wait for a GHIP_WALK_NEWTON_EWALD pair in
 * flight and must precede the first SPH call of the step. */
int ghip_upload_aos_particles(ghip_ctx *ctx, const void *P, const ghip_layout *lay, int numpart, int ngas);
int ghip_upload_aos_gas(ghip_ctx *ctx, const void *SphP, const ghip_layout *lay);
int ghip_download_aos(ghip_ctx *ctx, void *P, void *SphP, const ghip_layout *lay,
                      int want_gravity, int want_density, int want_hydro);
/* ghip_download_aos without the wait at its end: the packing kernels and the copies are queued on the
 * library's stream; the arrays are complete after the next synchronising call (ghip_sync). */
int ghip_download_aos_async(ghip_ctx *ctx, void *P, void *SphP, const ghip_layout *lay,
                            int want_gravity, int want_density, int want_hydro);
/* gravity_tree()'s post-pass (ghip_gravity_finish_ex with these arguments) and the gravity fields of
 * the P[] block (ghip_download_aos(..., 1, 0, 0)) in one call that, with a GHIP_WALK_NEWTON_EWALD pair
 * in flight, is ordered after the pair ONLY -- not behind SPH kernels queued underneath it, whose tail
 * the copy then overlaps.  Returns with P[] complete. */
int ghip_gravity_to_records(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac, void *P,
                            const ghip_layout *lay);
/* After ghip_upload_aos / ghip_upload_aos_particles: 1 when a record of the gas block [0, ngas) has a
 * Type other than 0 (a particle converted since the last rearrange_particle_sequence(); the block is
 * gas by allvars.h:1384), else 0.  Lets a host skip its own pass over P[].Type. */
int ghip_gas_block_mixed(ghip_ctx *ctx);
/* Page-lock a host array the record copies go through (the reference allocates P[] / SphP[] once
 * for All.MaxPart, allocate.c:30-60): the copies then run at the link's rate instead of through
 * the runtime's staging buffers.  Optional; a range that cannot be locked is left as it is
 * (returns GHIP_OK, ghip_last_error tells).  ghip_unpin_host before the array is freed;
 * ghip_destroy releases what is left. */
int ghip_pin_host(ghip_ctx *ctx, void *ptr, size_t bytes);
int ghip_unpin_host(ghip_ctx *ctx, void *ptr);

/* ---- active list (FirstActiveParticle/NextActiveParticle, run.c:300-320) ---- */
/* host indices of the active particles; NULL or n == numpart with idx NULL: all active */
int ghip_set_active(ghip_ctx *ctx, const int *idx, int nactive);
/* restrict evaluation to shard `rank` of `nranks` equal-count slices of the space-filling-curve
 * ordered active list (multi-GPU data parallelism; results of other slices are left untouched) */
int ghip_set_shard(ghip_ctx *ctx, int rank, int nranks);

/* shard exchange (multi-GPU; replaces the MPI export rounds gravtree.c:175-339,
 * density.c:193-389, hydra.c:274-526).  group 0 = gravity (width 4), 1 = density (width 7),
 * 2 = hydro (width 5).  ghip_shard_count: padded slice length `per` and this rank's count.
 * ghip_shard_pack writes this rank's finished results as [width][per] doubles into DEVICE memory;
 * the caller all-gathers the equal-sized blocks (RCCL) into [nranks][width][per];
 * ghip_shard_unpack fills in the other ranks' slices. */
int ghip_shard_count(ghip_ctx *ctx, int gas, int *per, int *mine);
int ghip_shard_pack(ghip_ctx *ctx, int group, void *dev_buf);
int ghip_shard_unpack(ghip_ctx *ctx, int group, const void *dev_buf_all, int nranks);

/* ---- multi-GPU: Peano-Hilbert domain decomposition with tree-node / ghost exchange ----
 * One process (or, for tests, one context) per GPU; each holds the particles of ONE contiguous
 * Peano-Hilbert key range (domain.c:100 domain_Decomposition; ranges cut by cumulative work,
 * domain.c:378-384, 1075-1113).  Replaces the top-tree pseudo-particles of force_treebuild
 * (forcetree.c:384-450, 879-1075) and the export rounds of gravity_tree / density / hydro_force
 * (gravtree.c:175-339, density.c:193-389, hydra.c:274-526):
 *   gravity  every shard receives, once per call, the part of every other shard's tree that its
 *            targets can possibly open (a locally essential tree: particles + pruned nodes with
 *            their moments), builds ONE tree over its particles and the imports and walks it --
 *            per target the opening decisions, hence GravCost, equal the single-rank tree's;
 *   SPH      gas particles within the padded search radius of another shard's targets (or whose own
 *            smoothing sphere reaches them) are imported as ghosts before density(); their records
 *            are refreshed once between density() and hydro_force().
 * Results of a shard cover its own (active) particles only; nothing is replicated.
 *
 * An operation is a small state machine, so that the same code serves RCCL (one rank per process)
 * and several logical shards in one process (parity tests on one GPU):
 *   ghip_dd_begin(op)   then   while(ghip_dd_step() == 1) <exchange>
 * where <exchange> is ghip_dd_exchange (this rank's part of a collective over RCCL: ncclAllGather,
 * or ncclAllGather of counts + grouped ncclSend/ncclRecv) or ghip_dd_exchange_local (all shards of
 * one process at once, device-to-device copies).  ghip_dd_run = the whole loop over RCCL. */
#define GHIP_DD_MIGRATE 1    /* params: NULL.  Particles that drifted out of their shard's key range
                              * move to the shard that owns them, with every resident field
                              * (domain_exchange, domain.c:665-1060); numpart / ngas of the
                              * contexts change, gas stays in front.  Run after ghip_drift. */
#define GHIP_DD_GRAVITY 2    /* params: ghip_grav_params, walk: GHIP_WALK_*; tree build included */
#define GHIP_DD_DENSITY 3    /* params: ghip_dens_params; needs the gravity tree of this step */
#define GHIP_DD_HYDRO 4      /* params: ghip_hydro_params; after GHIP_DD_DENSITY + ghip_update_hmax */
int ghip_dd_init(ghip_ctx *ctx, int rank, int nranks);
/* the global domain cube (identical on all ranks: the all-reduced extent of domain_findExtent,
 * domain.c:1972-2014) and All.ForceSoftening -- what ghip_tree_build takes in a single-GPU run */
int ghip_dd_set_domain(ghip_ctx *ctx, const double DomainCorner[3], const double DomainCenter[3],
                       double DomainLen, const double ForceSoftening[6]);
/* nranks+1 Peano-Hilbert keys (21 bits per dimension): rank r owns [splits[r], splits[r+1]);
 * splits[0] = 0, splits[nranks] = 2^63.  Every resident particle must lie in its rank's range. */
int ghip_dd_set_splits(ghip_ctx *ctx, const unsigned long long *splits);
/* The general form, for decompositions in which a rank owns several pieces of the curve
 * (-DMULTIPLEDOMAINS > 1, domain.c:482-494, 1158-1215: DomainTask[] per top-leaf): nseg segments
 * [keys[s], keys[s+1]) in key order with owner[s]; keys[0] = 0, keys[nseg] = 2^63.  Neighbouring
 * segments of one owner are merged.  A tree cell is "wholly this rank's" when it lies inside ONE of
 * its pieces; everything else works as for one range per rank (the target groups of a rank then
 * span its pieces, which only makes their boxes less compact). */
int ghip_dd_set_segments(ghip_ctx *ctx, int nseg, const unsigned long long *keys, const int *owner);
/* Peano-Hilbert keys of the resident particles (host array of numpart entries) */
int ghip_dd_keys(ghip_ctx *ctx, unsigned long long *keys_host);
/* domain_findSplit_work_balanced (domain.c:1075-1113, equal speed factors): cut ndomain
 * curve-ordered pieces of work into ncpu contiguous ranges [start[i], end[i]].  Host arithmetic. */
int ghip_dd_find_split(int ncpu, int ndomain, const double *domainWork, int *start, int *end);
/* search radii are padded by this factor when ghosts are first selected in a density() (default
 * 1.3).  Only a performance knob: when the h iteration takes a smoothing length beyond the padded
 * radius on any shard, all shards restore their starting Hsml, select ghosts again with
 * 1.26 x the worst growth seen and repeat the iteration (a collective decision). */
int ghip_dd_set_ghost_margin(ghip_ctx *ctx, double margin);
/* RCCL: rank 0 creates the id (128 bytes) and the host broadcasts it (MPI_Bcast in the reference's
 * world); every rank then connects.  The library binds the librccl that sits next to the HIP
 * runtime the process uses (ghip_dd_rccl_library tells which). */
int ghip_dd_rccl_unique_id(void *id128);
int ghip_dd_rccl_connect(ghip_ctx *ctx, const void *id128);
const char *ghip_dd_rccl_library(void);
int ghip_dd_begin(ghip_ctx *ctx, int op, const void *params, int walk);
int ghip_dd_step(ghip_ctx *ctx);                 /* 1: exchange pending, 0: done, < 0: error */
int ghip_dd_exchange(ghip_ctx *ctx);             /* RCCL */
int ghip_dd_exchange_local(ghip_ctx **ctxs, int nranks);
/* the pending exchange staged through host memory and the CALLER's all-gather of equal-sized
 * blocks (recv = [nranks][bytes]; return 0) -- MPI_Allgather for a host whose ranks have no RCCL
 * between them; also the only way to rehearse several ranks on ONE GPU.  Slow by construction. */
int ghip_dd_exchange_host(ghip_ctx *ctx,
                          int (*allgather)(void *user, const void *send, size_t bytes, void *recv),
                          void *user);
int ghip_dd_run(ghip_ctx *ctx, int op, const void *params, int walk);
/* out[0..10]: rank, nranks, elements imported into the gravity tree, elements sent, ghosts
 * imported, ghosts sent, bytes sent by the last gravity / density operation, largest Hsml growth
 * of the last density (x 1e6), elements of the gravity / gas tree */
int ghip_dd_get_info(const ghip_ctx *ctx, long long out[16]);

/* ---- pre-condition of the path: drift the resident particles (replaces the lazy
 * drift_particle() calls inside the walks, forcetree.c:1911, ngb.c:57) ---- */
int ghip_drift(ghip_ctx *ctx, const ghip_drift_params *p);

/* ---- "next" row N1: timestep + kick on the resident fields.  Works on the active list of
 * ghip_set_active.  In/out: VEL, VELPRED, ENTROPY, DTENTROPY, TIMEBIN, TI_BEGSTEP; in: GRAVACCEL
 * (final, xG), HYDROACCEL, HSML, MAXSIGNALVEL, DENSITY, TYPE.  TimeBinCount/TimeBinCountSph
 * (32 entries each, may be NULL) receive the recounted bin populations (allvars.h:337-338).
 * Returns GHIP_ETIMESTEP where the reference calls endrun(); ghip_timestep_endrun_code gives the
 * reference's code (888, 818, 112313).  With both arrays NULL the bins are not recounted
 * (ghip_timebin_counts does it on demand) and, under ghip_set_async, the call does not wait. ---- */
int ghip_advance_timesteps(ghip_ctx *ctx, const ghip_kick_params *p, long long *TimeBinCount,
                           long long *TimeBinCountSph);
int ghip_timestep_endrun_code(const ghip_ctx *ctx);
/* the bin populations after the last ghip_advance_timesteps, recounted over all particles
 * (reconstruct_timebins, predict.c:14-127); synchronises */
int ghip_timebin_counts(ghip_ctx *ctx, long long *TimeBinCount, long long *TimeBinCountSph);
int ghip_pm_kick(ghip_ctx *ctx, const ghip_pmkick_params *p);
/* per-type sums of find_dt_displacement_constraint (timestep.c:1140-1156): sum of |v|^2, smallest
 * positive mass (1e30 if none), particle count -- 6 entries each */
int ghip_velocity_moments(ghip_ctx *ctx, double v2sum[6], double min_mass[6], long long count[6]);
/* pack the kick's results (Vel, TimeBin, Ti_begstep; VelPred, Entropy, DtEntropy) into the device
 * image of the records and copy the blocks to the host */
int ghip_download_aos_kick(ghip_ctx *ctx, void *P, void *SphP, const ghip_layout *lay);

/* ---- "next" row N2: export the device-built gravity tree in the reference's representation
 * (what force_treebuild + force_update_node_recursive leave behind, forcetree.c:67-872): record k
 * of Nodes_base / Extnodes_base is node MaxPart + k (the root is node MaxPart), Nextnode[] and
 * Father[] are filled for the particles [0, numpart).  Needs a built tree and the VEL, HSML,
 * DIVVEL, TYPE fields (for vs, vmax, hmax, divVmax).  Single-rank semantics: no TOPLEVEL /
 * pseudo-particle entries. ---- */
int ghip_tree_export(ghip_ctx *ctx, const ghip_node_layout *lay, int MaxPart, int Ti_Current,
                     int unequal_softenings, void *Nodes_base, void *Extnodes_base, int *Nextnode,
                     int *Father, int max_nodes, int *numnodes);

/* ---- "next" row N3: P[].GravPM = long-range force of all particles on a PMGRID^3 periodic mesh
 * (replaces long_range_force() -> pmforce_periodic(), longrange.c / pm_periodic.c:199-800; the
 * field is zeroed first as long_range_force does).  Pairs with GHIP_WALK_SHORTRANGE. ---- */
int ghip_pm_periodic(ghip_ctx *ctx, const ghip_pm_params *p);

/* ---- "next" row N4: sink (black-hole) neighbour passes and the per-particle part of
 * cooling_and_starformation, for the reference's shipped flag bundle (BLACK_HOLES, SWALLOWGAS,
 * ACCRETION_RADIUS, ACCRETION_DENSITY, ACCRETION_OF_DUST_ONLY, BH_MERGERS_WITHIN_H,
 * BH_THERMALFEEDBACK + TMP_FEEDBACK, DUST, COOLING, SFR, BH_FORM).  The sinks are given by their
 * particle indices; their per-sink state (ID, Mdot, BH_Density, BH_Mass ...) travels in small host
 * arrays, the victims' marks (P[].SwallowID, SphP[].i.Injected_BH_Energy) are resident.  Scalar
 * bookkeeping per sink (blackhole_accretion(), blackhole.c:133-300, 680-760), the conversion of a
 * flagged gas particle into a sink (sfr_eff.c:606-640, GSL stream) and DoCooling (cooling.c) stay
 * host code.  On a multi-GPU shard the neighbour passes run through GHIP_DD_SINK_DENSITY /
 * GHIP_DD_BH_EVALUATE / GHIP_DD_BH_SWALLOW below. ---- */
typedef struct
{
  double BoxSize;
  int periodic;
  double ascale;            /* All.Time when comoving, else 1 (blackhole.c:89-95) */
  double dt_fac;            /* All.Timebase_interval / hubble_a (:822) */
  double SMBHmass, InnerBoundary, SinkBoundary, SofteningBndry;   /* All.* */
  double CritDensity;       /* All.CritOverDensity * UnitLength_in_cm^3 / UnitMass_in_g (:1099) */
  double FeedbackCoeff;     /* All.BlackHoleFeedbackFactor * 6.67e-8 * pow(4.*3.1415/3.*5., 0.3333)
                               / All.UnitEnergy_in_cgs (:1138-1139) */
  double UnitMass_in_g;
  int dust;                   /* -DDUST */
  int accretion_of_dust_only; /* -DACCRETION_OF_DUST_ONLY */
  int accretion_density;      /* -DACCRETION_DENSITY */
} ghip_bh_params;
/* density() for Type-5 targets (density.c BLACK_HOLES branches): h iteration against the gas tree
 * for DesNumNgb * ngb_factor (All.BlackHoleNgbFactor) neighbours.  hsml [nsink] in/out (also
 * written to the resident HSML field); numngb, bh_density, bh_entropy [nsink], bh_gasvel
 * [nsink][3].  Needs the tree of this step. */
int ghip_sink_density(ghip_ctx *ctx, const ghip_dens_params *p, double ngb_factor, int nsink,
                      const int *sink_idx, double *hsml, double *numngb, double *bh_density,
                      double *bh_entropy, double *bh_gasvel, int *iterations);
/* P[].SwallowID = 0, Injected_BH_Energy = 0 for all particles (start of blackhole_accretion) */
int ghip_sink_reset(ghip_ctx *ctx);
/* blackhole_evaluate (blackhole.c:794-1190): marks the victims (SwallowID = ID of the sink; a
 * victim claimed by several sinks goes to the largest ID) and spreads the feedback energy.
 * Reads POS, VEL, MASS, HSML, TIMEBIN, TYPE and the gas DENSITY from the resident fields. */
int ghip_blackhole_evaluate(ghip_ctx *ctx, const ghip_bh_params *p, int nsink, const int *sink_idx,
                            const unsigned int *sink_id, const double *bh_mdot,
                            const double *bh_density);
/* blackhole_evaluate_swallow (blackhole.c:1201-1346): per sink the accreted mass, BH mass, dust
 * mass [nsink] and momentum [nsink][3]; the victims' resident MASS becomes 0 (a tree built before
 * is stale afterwards), sink_bh_mass [nsink] (P[].BH_Mass of the sinks, in/out) becomes 0 for a
 * swallowed sink; counts = gas / sinks / dust swallowed. */
int ghip_blackhole_swallow(ghip_ctx *ctx, const ghip_bh_params *p, int nsink, const int *sink_idx,
                           const unsigned int *sink_id, double *sink_bh_mass, double *acc_mass,
                           double *acc_bhmass, double *acc_dustmass, double *acc_momentum,
                           long long counts[3]);
/* the resident marks: SwallowID [numpart] (u32), Injected_BH_Energy [ngas]; NULL = skip */
int ghip_sink_get_marks(ghip_ctx *ctx, unsigned int *swallow_id, double *injected_energy);
int ghip_sink_set_marks(ghip_ctx *ctx, const unsigned int *swallow_id, const double *injected_energy);
/* cooling_and_starformation (sfr_eff.c:82-947), per active gas particle (ghip_set_active), with the
 * cooling function as identity: flag_sink_host [ngas] = 1 where the particle qualifies for
 * conversion into a sink (:226-229), else the isochoric update of DTENTROPY incl. the injected
 * black-hole energy (:486-531, 582-594).  Non-comoving. */
int ghip_cooling_and_starformation(ghip_ctx *ctx, double Timebase_interval,
                                   double CritPhysDensity_code, double MinEgySpec,
                                   double u_to_temp_fac, int *flag_sink_host);

/* The sink passes on a multi-GPU shard (GHIP_DD_SINK_DENSITY / _BH_EVALUATE / _BH_SWALLOW through
 * ghip_dd_begin / ghip_dd_run): the sinks of every shard are made known to all shards (a few hundred
 * 96-byte records), each shard evaluates ALL sinks against ITS OWN particles, and the partial sums
 * (density: 6 doubles per sink and h iteration; swallow: 9) are all-gathered and added in rank order
 * -- the reference's export of the few sink targets (blackhole.c:310-600) instead of an import of
 * their neighbourhoods.  A victim is local to exactly one shard, which sees every sink's claim, so
 * the marks equal the single-GPU ones.  Arrays of this struct describe the sinks resident on THIS
 * shard; members an operation does not use may be NULL. */
typedef struct
{
  const ghip_dens_params *dens;   /* SINK_DENSITY */
  double ngb_factor;              /* All.BlackHoleNgbFactor */
  const ghip_bh_params *bh;       /* BH_EVALUATE, BH_SWALLOW */
  int nsink;                      /* sinks resident on this shard */
  const int *sink_idx;            /* their local particle indices */
  const unsigned int *sink_id;    /* P[].ID */
  double *hsml;                   /* [nsink] SINK_DENSITY in/out */
  double *numngb, *bh_density, *bh_entropy, *bh_gasvel;   /* SINK_DENSITY out ([nsink], gasvel [nsink][3]) */
  const double *bh_mdot;          /* BH_EVALUATE in */
  const double *bh_density_in;    /* BH_EVALUATE in */
  double *sink_bh_mass;           /* BH_SWALLOW in/out */
  double *acc_mass, *acc_bhmass, *acc_dustmass, *acc_momentum;   /* BH_SWALLOW out */
  long long *counts;              /* BH_SWALLOW out [3]: gas / sinks / dust swallowed ON THIS SHARD */
} ghip_dd_sink_args;
#define GHIP_DD_SINK_DENSITY 5   /* needs GHIP_DD_DENSITY of this step (the shard's gas tree) */
#define GHIP_DD_BH_EVALUATE 6    /* needs GHIP_DD_GRAVITY of this step (the shard's gravity tree) */
#define GHIP_DD_BH_SWALLOW 7
/* pmforce_periodic on shards (params: ghip_pm_params): every shard deposits ITS particles on the full
 * PMGRID^3 mesh, the meshes are all-gathered and added in rank order (8 PMGRID^3 bytes per shard:
 * 16.8 MB at PMGRID = 128), then every shard transforms the identical mesh and interpolates the
 * force for its own particles -- the reference's slab exchange (pm_periodic.c:263-450) with the
 * transform replicated instead of distributed.  Pairs with GHIP_WALK_SHORTRANGE of GHIP_DD_GRAVITY. */
#define GHIP_DD_PM 8

/* ---- the path ---- */
int ghip_tree_build(ghip_ctx *ctx, const double DomainCorner[3], const double DomainCenter[3],
                    double DomainLen, const double ForceSoftening[6]);
/* ---- sub-steps on the tree of the last full build (the reference's dynamic tree update,
 * forcetree.c:1356-1651: force_drift_node, force_kick_node, force_finish_kick_nodes) ----
 * Without this, every ghip_tree_build builds the tree of the CURRENT positions -- a valid Barnes-Hut
 * tree, but not the one the reference walks on a sub-step (TreeReconstructFlag == 0): there the cells
 * are those of the last full build, a node's centre of mass has moved with its mass-weighted velocity
 * vs, its side length has grown by 2 vmax dt and the momentum of the kicked particles is folded into
 * vs.  With ghip_set_dynamic_tree(ctx, 1):
 *   ghip_tree_build        = a full build (force_treebuild); a copy of the gravity tree is kept with
 *                            vs and vmax per node
 *   ghip_advance_timesteps   records the velocity change of every particle it kicks and hands Mass dv
 *                            and max|Vel| to all ancestors (force_kick_node); a host that kicks itself
 *                            calls ghip_tree_kick_nodes(ctx, n, idx, dv[n][3]) after it has stored the
 *                            new velocities (ghip_set_field VEL)
 *   ghip_tree_substep(ctx, dt_drift)
 *                          = what gravity_tree() does with TreeReconstructFlag == 0: every node of the
 *                            kept tree goes from the previous sync point to this one (pending kicks
 *                            into vs, s += vs dt_drift, len += 2 vmax dt_drift; dt_drift = (Ti_Current
 *                            - Ti_previous) * Timebase_interval, or get_drift_factor in comoving runs)
 *                            and its particle elements take the resident POS / MASS (drift the
 *                            particles first: ghip_drift, without box wrapping -- the reference wraps
 *                            at domain decompositions only); the tree of the current positions is
 *                            rebuilt as well, for the target order and the gas tree (SPH neighbour sets
 *                            are geometric: they do not depend on the tree that finds them).
 *                            The gravity walks read the kept tree until the next ghip_tree_build.
 * All nodes move at once; the reference moves a node when a walk or a kick first meets it, which is
 * the same state up to the rounding of s += vs dt in one piece or several.  Not on a multi-GPU shard
 * and not with ADAPTIVE_GRAVSOFT_FORGAS. */
int ghip_set_dynamic_tree(ghip_ctx *ctx, int on);
int ghip_tree_substep(ghip_ctx *ctx, double dt_drift);
int ghip_tree_kick_nodes(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3);
/* the same when the resident VEL is not the kicked one yet: vmax[k] = max_j |P[idx[k]].Vel[j]| of the
 * new velocity travels with the kick (what force_kick_node computes itself, forcetree.c:1478-1480).
 * A particle listed twice hands up the sum of its kicks. */
int ghip_tree_kick_nodes_vmax(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3,
                              const double *vmax);
/* the kept tree in pre-order (tests): xm = (s or pos, mass), cl = (centre, len), ev = (vs, vmax),
 * lk = links; NULL = skip */
int ghip_tree_dump_dynamic(ghip_ctx *ctx, int *nelem, double *xm4, double *cl4, double *ev4, int *lk4);
/* ADAPTIVE_GRAVSOFT_FORGAS (the shipped Makefile bundle): on != 0 makes the gravitational softening
 * of a gas particle its Hsml instead of ForceSoftening[0], as a target (forcetree.c:1851-1856), as
 * a source (:2038-2058) and in the nodes' maxsoft (:705-726, 845-846), which then opens a node
 * whenever the target lies inside it (:2125-2139).  Implies UNEQUALSOFTENINGS.  The softenings are
 * captured by ghip_tree_build: call this before it (it invalidates a built tree). */
int ghip_set_adaptive_gravsoft(ghip_ctx *ctx, int on);
int ghip_ewald_init(ghip_ctx *ctx, double BoxSize);
/* host copy of the ewald table [3][65][65][65] already scaled by 1/BoxSize^2 */
int ghip_ewald_get_table(ghip_ctx *ctx, double *host);
/* walk kind: GHIP_WALK_*.  Results: GRAVACCEL (G-less, as gravtree.c:381-393 expects) and
 * GRAVCOST of the active particles.  The EWALD walk adds to both. */
int ghip_gravity(ghip_ctx *ctx, const ghip_grav_params *p, int walk);
/* external targets (the mode==1 gravdata_in record, allvars.h:1690-1703): coordinates, type,
 * OldAcc in; Acc[3], Ninteractions out; all host arrays */
int ghip_gravity_ext(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt,
                     const double *pos, const int *type, const double *oldacc, double *acc,
                     int *ninteractions);
/* the post-pass of gravity_tree() over the active particles, in the reference's order
 * (gravtree.c:362-403):
 *   comoving_fac != 0 : GravAccel += comoving_fac * Pos, with comoving_fac = 0.5 * Hubble^2 *
 *                       Omega0 / G -- the term of comoving runs built without PERIODIC and PMGRID
 *                       (:362-373); pass 0 otherwise
 *   OldAcc = |GravAccel|; pmgrid != 0: |GravAccel + GRAVPM / G| (:375-391, the relative opening
 *                       criterion of a TreePM run sees the total acceleration)
 *   GravAccel *= G      (:398-403)
 * all_shards != 0: every active target regardless of ghip_set_shard (replicated multi-GPU mode,
 * after the all-gather). */
int ghip_gravity_finish_ex(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac,
                           int all_shards);
/* = ghip_gravity_finish_ex(ctx, G, 0, 0, 0) / (ctx, G, 0, 0, 1): builds without PMGRID that are
 * periodic or not comoving */
int ghip_gravity_finish(ghip_ctx *ctx, double G);
int ghip_gravity_finish_all(ghip_ctx *ctx, double G);
/* GravAccel += fac * Pos for all active particles, fac = OmegaLambda * Hubble^2: the vacuum-energy
 * term of runs in physical coordinates (gravtree.c:470-483, !PERIODIC && !PMGRID &&
 * ComovingIntegrationOn == 0); call it after ghip_gravity_finish */
int ghip_gravity_vacuum_energy(ghip_ctx *ctx, double fac);
/* the same with gravdata_in.Soft (allvars.h:1695, gravtree.c:214-220): soft[a] = Hsml of a gas
 * target, used when ghip_set_adaptive_gravsoft is on; NULL = ghip_gravity_ext */
int ghip_gravity_ext_soft(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt,
                          const double *pos, const int *type, const double *soft,
                          const double *oldacc, double *acc, int *ninteractions);
/* softened direct summation over all particles for the active targets (accuracy oracle on
 * device, formula of forcetree.c:4273-4336); writes GRAVACCEL */
int ghip_gravity_direct(ghip_ctx *ctx, const ghip_grav_params *p);
int ghip_density(ghip_ctx *ctx, const ghip_dens_params *p);
int ghip_update_hmax(ghip_ctx *ctx);
int ghip_hydro(ghip_ctx *ctx, const ghip_hydro_params *p);
/* -DBLACK_HOLES / -DDUST builds: a gas particle of mass 0 (swallowed, waiting for the next
 * rearrange_particle_sequence) is skipped by the neighbour loop of density() (rule 1: -DDUST or
 * -DBLACK_HOLES, density.c:831-834) and also by hydro_force()'s (rule 3: -DBLACK_HOLES,
 * hydra.c:1235-1238); it stays a target.  0 (default): the builds without these flags sum it with
 * weight 0 but count it in NumNgb.  Independent of this switch a record of the gas block whose Type is
 * not 0 is never a target or a neighbour, as in the reference's loops. */
int ghip_set_massless_gas_rule(ghip_ctx *ctx, int rule);
/* When ghip_hydro is called underneath a GHIP_WALK_NEWTON_EWALD pair in flight, its kernel is held
 * back on the device until the Ewald walk drains (fastest step for resident data).  early != 0: it
 * starts at once -- the step's kernels take 0.2 ms longer at c2, but a host that downloads the SPH
 * results gets them across while the walks still run. */
int ghip_set_hydro_release(ghip_ctx *ctx, int early);

/* one fixed-h evaluation for a single target (density_evaluate mode 0, density.c:711):
 * out7 = rho, numngb, dhsmlrho, divv, rot[3] (raw sums, before finalisation) */
int ghip_density_evaluate(ghip_ctx *ctx, const ghip_dens_params *p, int target, double h,
                          double out7[7]);
/* neighbour lists for one search centre (ngb_treefind_variable / _pairs, ngb.c:169, 32):
 * writes up to cap host indices, returns the count in *nfound */
int ghip_ngb_treefind(ghip_ctx *ctx, const double center[3], double hsml, int pairs,
                      int periodic, double boxsize, int *ngblist, int cap, int *nfound);

/* ---- keys (peano.c:300-358), device evaluation of n integer triplets ---- */
int ghip_peano_hilbert_keys(ghip_ctx *ctx, int n, const int *x, const int *y, const int *z,
                            int bits, unsigned long long *keys);
int ghip_morton_keys(ghip_ctx *ctx, int n, const int *x, const int *y, const int *z, int bits,
                     unsigned long long *keys);

/* ---- a resident step loop that never waits for the device (run.c:40-155 with P / SphP in HBM) ----
 * ghip_set_async(ctx, 1): ghip_drift and ghip_advance_timesteps (called with NULL count arrays)
 * return without waiting; what they would have reported -- the reference's endrun(12) of a particle
 * ahead of the drift target, endrun(888|818|112313) of a failed timestep criterion -- is reported
 * with the same codes by the next entry point that synchronises (ghip_sync, ghip_get_field, the
 * h iteration of ghip_density, ...).  ghip_tree_build needs no switch: whenever the particle number
 * equals the previous build's it is enqueued without waiting for the node counts, and the counts
 * are verified before anything persistent is modified on their basis (a build that overflowed its
 * buffers is repeated and the gravity calls made since are replayed: same results, later).
 * With both, the host enqueues a step while the device still works on the previous one; the only
 * wait left in a step with gas is the h iteration's unconverged count, which is read underneath
 * the gravity walks. */
int ghip_set_async(ghip_ctx *ctx, int on);

/* statistics of a run of steps without a host synchronisation per step: the library keeps one set
 * of phase events per step and sums its device counters on the device.
 *   ghip_run_begin(ctx, max_steps);  { ghip_step_begin(ctx); <the calls of a step>; ghip_step_end(ctx); } ...
 *   ghip_run_end(ctx, &stats)        -- synchronises once */
typedef struct
{
  long long steps;             /* ghip_step_end calls */
  long long steps_timed;       /* of them, with events kept (the last max_steps) */
  long long launches;          /* kernel launches issued by the library (its own and rocPRIM's) */
  long long blocking_syncs;    /* host waits for the device inside the library */
  long long grav_interactions, ewald_interactions, dens_neighbours, hydro_pairs;
  long long grav_wave_steps, ewald_wave_steps, dens_extra_iterations;
  double ms_tree, ms_grav, ms_ewald, ms_dens, ms_hmax, ms_hydro, ms_kick;   /* summed device spans */
  double ms_steps_device;      /* sum over steps of (step begin mark -> step end mark), device clock */
  double ms_between_steps;     /* sum of (end mark of step k -> begin mark of step k+1): the device
                                * waiting for the host between steps */
  double ms_first_to_last;     /* begin mark of the first timed step -> end mark of the last */
} ghip_run_stats;
int ghip_run_begin(ghip_ctx *ctx, int max_steps);
int ghip_step_begin(ghip_ctx *ctx);
int ghip_step_end(ghip_ctx *ctx);
int ghip_run_end(ghip_ctx *ctx, ghip_run_stats *out);

/* ---- introspection ---- */
int ghip_get_stats(const ghip_ctx *ctx, ghip_stats *out);
/* tree dump for parity tests: the pre-order element list of the gravity tree (which = 0) or the
 * gas tree (which = 1): xm4 = (x,y,z,mass), cl4 = (centre, len), lk4 = (skip, particle index in
 * tree order or -(level+1), first particle, count), aux, and perm = tree order -> host index.
 * Pass NULL arrays to query *nelem only. */
int ghip_tree_dump(ghip_ctx *ctx, int which, int *nelem, double *xm4, double *cl4, int *lk4,
                   double *aux, int *perm);
/* device stream (hipStream_t) the kernels run on, for callers that time with HIP events */
void *ghip_stream(ghip_ctx *ctx);
int ghip_sync(ghip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
