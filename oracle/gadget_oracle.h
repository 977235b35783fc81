/*
 * gadget_oracle.h -- CPU restatement of the GADGET-3 (Leicester fork) per-step force path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the CPU baseline.  The product path (libghip.so) never links or calls it.
 *
 * PARITY UNPINNED BY UPSTREAM: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built in this image without writing stand-ins
 * for GSL (allvars.h:21-25 includes gsl headers in every TU; GSL is absent).  The oracle is
 * therefore a restatement written from the source text, each function citing the reference
 * file:line it follows, and pinned by independent checks only (direct summation, brute-force
 * O(N^2) neighbour sums, analytic cases) -- see tests/test_oracle_*.py.
 *
 * All arithmetic is fp64 (reference: DOUBLEPRECISION, no FLTROUNDOFFREDUCTION; allvars.h:154-202).
 * Particles are plain SoA arrays; gas particles are the indices [0, ngas) as in the reference
 * (SphP[] is index-aligned with P[0..N_gas), allvars.h:1384).
 */
#ifndef GADGET_ORACLE_H
#define GADGET_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned long long orc_key;

#define ORC_BITS_PER_DIMENSION 21 /* allvars.h:58 */
#define ORC_EN 64                 /* forcetree.c:51 */
#define ORC_NTAB 1000             /* forcetree.c:28 */

typedef struct orc_tree orc_tree;

/* ---- keys (peano.c:300-358) ---- */
orc_key orc_morton_key(int x, int y, int z, int bits);
orc_key orc_peano_hilbert_key(int x, int y, int z, int bits);

/* ---- domain extent (domain.c:1972-2014) ---- */
void orc_domain_extent(int n, const double *pos, double corner[3], double center[3], double *len);

/* ---- tree build (forcetree.c:125-872) ----
 * pos,vel: [n][3]; hsml/divvel may be NULL (treated as 0); soft = All.ForceSoftening[6].
 * toplevels: depth of a pre-created complete top-level grid (force_create_empty_nodes analogue;
 * 0 = root only).  Returns NULL on allocation failure. */
orc_tree *orc_tree_build(int n, const double *pos, const double *vel, const double *mass,
                         const int *type, const double *hsml, const double *divvel,
                         const double soft[6], const double corner[3], const double center[3],
                         double len, int toplevels);
/* switch the built tree to ADAPTIVE_GRAVSOFT_FORGAS (gas softening = Hsml; NODE.maxsoft,
 * forcetree.c:535-541, 705-726, 845-846): every later gravity walk over it uses those rules
 * (forcetree.c:1851-1856, 2038-2058, 2125-2139) */
void orc_tree_adaptive_gravsoft(orc_tree *t);
void orc_tree_free(orc_tree *t);
int orc_tree_numnodes(const orc_tree *t);
/* node dump for tests: arrays sized numnodes; index k is node (n + k) */
void orc_tree_dump(const orc_tree *t, double *len, double *center3, double *s3, double *mass,
                   int *sibling, int *nextnode, int *father, int *multi, double *hmax);
void orc_tree_dump_ext(const orc_tree *t, double *vs3, double *vmax, double *divvmax,
                       double *maxsoft, int *mixedsoft);
/* Nextnode[] and Father[] for particles (size n) */
void orc_tree_dump_particles(const orc_tree *t, int *nextnode, int *father);
/* force_update_hmax (forcetree.c:1661-1786): raise hmax/divVmax up the Father chain */
void orc_update_hmax(orc_tree *t, int nactive, const int *active, const double *hsml,
                     const double *divvel);
/* the tree between two builds (forcetree.c:1356-1520): force_drift_node for every node,
 * force_kick_node for the kicked particles */
void orc_set_massless_gas_rule(int on);   /* density.c:831-834, hydra.c:1235-1238 */
void orc_tree_drift_nodes(orc_tree *t, double dt_drift, double dt_drift_hmax);
void orc_tree_kick_nodes(orc_tree *t, int n, const int *idx, const double *dv3);
void orc_tree_dump_dynamic(const orc_tree *t, double *s3, double *len, double *vs3, double *vmax);

typedef struct
{
  double ErrTolTheta;      /* != 0: Barnes-Hut criterion, == 0: relative criterion */
  double ErrTolForceAcc;
  double BoxSize;
  int periodic;            /* PERIODIC */
  int unequal_softenings;  /* UNEQUALSOFTENINGS */
  double rcut;             /* shortrange only: All.Rcut[0] */
  double asmth;            /* shortrange only: All.Asmth[0] */
} orc_grav_params;

/* force_treeevaluate (forcetree.c:1797-2317), mode 0, over a target list (OpenMP over targets).
 * oldacc: [n] per particle.  acc: [nt][3] G-less, OVERWRITTEN; cost: [nt] ninteractions. */
void orc_gravity(const orc_tree *t, const orc_grav_params *p, int nt, const int *targets,
                 const double *oldacc, double *acc, int *cost);
/* same walk for external targets given by coordinates (the mode==1 record of gravdata_in,
 * allvars.h:1690-1703, walking the whole local tree) */
void orc_gravity_ext(const orc_tree *t, const orc_grav_params *p, int nt, const double *tpos,
                     const int *ttype, const double *toldacc, double *acc, int *cost);
/* the same with gravdata_in.Soft (allvars.h:1695): tsoft[a] = Hsml of a gas target, used under
 * ADAPTIVE_GRAVSOFT_FORGAS; may be NULL */
void orc_gravity_ext_soft(const orc_tree *t, const orc_grav_params *p, int nt, const double *tpos,
                          const int *ttype, const double *tsoft, const double *toldacc, double *acc,
                          int *cost);
/* force_treeevaluate_shortrange (forcetree.c:2330-2845) */
void orc_gravity_shortrange(const orc_tree *t, const orc_grav_params *p, int nt,
                            const int *targets, const double *oldacc, double *acc, int *cost);
/* ewald tables (forcetree.c:4402-4527, 4727-4778): tab = [3][EN+1][EN+1][EN+1], already
 * divided by BoxSize^2 */
void orc_ewald_init(double *tab, double boxsize);
void orc_ewald_force(int i, int j, int k, const double x[3], double force[3]);
/* force_treeevaluate_ewald_correction (forcetree.c:2873-3204): acc ADDED to, cost ADDED to */
void orc_gravity_ewald(const orc_tree *t, const orc_grav_params *p, const double *tab, int nt,
                       const int *targets, const double *oldacc, double *acc, int *cost);
/* softened direct summation (formula of forcetree.c:4273-4336), nearest image if periodic,
 * optional ewald table correction */
void orc_gravity_direct(int n, const double *pos, const double *mass, const int *type,
                        const double soft[6], int unequal, int periodic, double boxsize,
                        const double *ewald_tab, int nt, const int *targets, double *acc);
void orc_gravity_direct_psoft(int n, const double *pos, const double *mass, const double *psoft,
                              int periodic, double boxsize, const double *ewald_tab, int nt,
                              const int *targets, double *acc);

/* ---- neighbour search (ngb.c:169-297, 32-160): returns count, writes indices ---- */
int orc_ngb_treefind_variable(const orc_tree *t, const double c[3], double h, int periodic,
                              double boxsize, int *ngblist);
int orc_ngb_treefind_pairs(const orc_tree *t, const double c[3], double h, const double *hsml,
                           int periodic, double boxsize, int *ngblist);

/* ---- SPH density (density.c:89-704, 711-1029) ---- */
typedef struct
{
  double DesNumNgb, MaxNumNgbDeviation, MinGasHsml;
  double BoxSize;
  int periodic;
  int Ti_Current;
  double Timebase_interval;
  int maxiter;             /* MAXITER 150 (density.c) */
} orc_dens_params;

/* one evaluation at fixed h (density_evaluate mode 0): out7 = rho, numngb, dhsmlrho, divv, rot[3] */
void orc_density_evaluate(const orc_tree *t, const orc_dens_params *p, int target, double h,
                          const double *velpred, double out7[7]);
/* the full driver.  hsml in/out [n]; outputs sized [ngas] except numngb [n].
 * returns number of h-iterations, or -1 if not converged. */
int orc_density(const orc_tree *t, const orc_dens_params *p, int nactive, const int *active,
                const double *velpred, const double *entropy, const double *dtentropy_in,
                const int *timebin, const int *ti_begstep, double *hsml, double *numngb,
                double *density, double *dhsmlfac, double *divvel, double *curlvel,
                double *pressure, long long *nngb_visits);

/* ---- SPH hydro (hydra.c:145-813, 822-1995) ---- */
typedef struct
{
  double ArtBulkViscConst;
  double BoxSize;
  int periodic;
  int ComovingIntegrationOn;
  double hubble_a2, fac_mu, fac_vsic_fix; /* hydra.c:192-208; all 1 when not comoving */
  double Timebase_interval;
} orc_hydro_params;

/* outputs [ngas]: hydroaccel [ngas][3], dtentropy (already converted, hydra.c:583), maxsignalvel */
void orc_hydro(const orc_tree *t, const orc_hydro_params *p, int nactive, const int *active,
               const double *velpred, const double *hsml, const double *density,
               const double *pressure, const double *dhsmlfac, const double *divvel,
               const double *curlvel, const int *timebin, double *hydroaccel, double *dtentropy,
               double *maxsignalvel, long long *npairs);

/* ---- drift_particle + do_box_wrapping (predict.c:129-259, 282-310); tables: NULL or
 * [3][1000] = DriftTable, GravKickTable, HydroKickTable.  Returns 0, or 12 (endrun(12)). ---- */
int orc_drift(int n, int ngas, int time1, double timebase, const double *tables,
              double logTimeBegin, double logTimeMax, double minhsml, int wrap, double boxsize,
              double *pos, const double *vel, const int *type, int *ti_current, const int *timebin,
              const int *ti_begstep, const double *gravaccel, double *velpred,
              const double *hydroaccel, double *density, double *hsml, const double *divvel,
              const double *entropy, const double *dtentropy, double *pressure);

/* ---- "next" row N1: timestep criterion + kick (timestep.c:29-605, 607-1123, 1125-1246),
 * minimal flag set; tables as in orc_drift (NULL when not comoving) ---- */
typedef struct
{
  int Ti_Current;
  double Timebase_interval;
  int ComovingIntegrationOn;
  double Time, hubble_a;
  double ErrTolIntAccuracy, CourantFac, MaxSizeTimestep, MinSizeTimestep, dt_displacement;
  double SofteningTable[6];
  double MinEgySpec;
  unsigned int TimeBinActive; /* bit b: TimeBinActive[b] */
  const double *tables;       /* [3][1000] drift, gravkick, hydrokick */
  double logTimeBegin, logTimeMax;
  int AdaptiveGravsoftForGasHsml; /* ADAPTIVE_GRAVSOFT_FORGAS(_HSML): timestep.c:740-743 */
  int pmgrid;                     /* PMGRID: timestep.c:648-652, 511-513 */
  double dt_gravkickB;            /* timestep.c:66-72 */
  const double *gravpm;           /* [n][3], PMGRID only */
} orc_kick_params;

void orc_velocity_moments(int n, const double *vel, const double *mass, const int *type,
                          double v2[6], double minmass[6], long long count[6]);
double orc_dt_displacement(const double v2[6], const double minmass[6], const long long count[6],
                           int comoving, double hfac, double MaxSizeTimestep,
                           double MaxRMSDisplacementFac, double Omega0, double OmegaBaryon,
                           double Hubble, double G, int StarformationOn);
int orc_drift_pm(int n, int ngas, int time1, double timebase, const double *tables,
                 double logTimeBegin, double logTimeMax, double minhsml, int wrap, double boxsize,
                 double *pos, const double *vel, const int *type, int *ti_current,
                 const int *timebin, const int *ti_begstep, const double *gravaccel,
                 const double *gravpm, double *velpred, const double *hydroaccel, double *density,
                 double *hsml, const double *divvel, const double *entropy, const double *dtentropy,
                 double *pressure);
void orc_pm_kick(int n, int ngas, int ti_current, double timebase, const double *tables,
                 double logTimeBegin, double logTimeMax, double dt_gravkick, double dt_gravkickB,
                 const int *type, const int *timebin, const int *ti_begstep, double *vel,
                 const double *gravaccel, const double *gravpm, const double *hydroaccel,
                 double *velpred);
int orc_advance_timesteps(int n, int ngas, const orc_kick_params *p, int nactive, const int *active,
                          const int *type, double *vel, const double *gravaccel,
                          const double *hydroaccel, double *velpred, double *entropy,
                          double *dtentropy, const double *density, const double *pressure,
                          const double *hsml, const double *maxsignalvel, int *timebin,
                          int *ti_begstep, long long bincount[32], long long bincount_sph[32]);

/* ---- "next" row N4: sink (black-hole) neighbour passes + the per-particle part of
 * cooling_and_starformation, for the shipped flag bundle (gadget_oracle.c has the flag list) ---- */
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
} orc_bh_params;

int orc_sink_density(const orc_tree *t, const orc_dens_params *p, double ngbfactor, int nsink,
                     const int *sink, const double *velpred, const double *entropy, double *hsml,
                     double *numngb, double *bh_density, double *bh_entropy, double *bh_gasvel);
void orc_blackhole_evaluate(const orc_tree *t, const orc_bh_params *p, int nsink, const int *sink,
                            const unsigned int *id, const double *hsml, const int *timebin,
                            const double *bh_mdot, const double *bh_density,
                            const double *gas_density, unsigned int *swallowid,
                            double *injected_energy);
void orc_blackhole_swallow(const orc_tree *t, const orc_bh_params *p, int nsink, const int *sink,
                           const unsigned int *id, const double *hsml,
                           const unsigned int *swallowid, double *mass, double *particle_bh_mass,
                           double *acc_mass, double *acc_bhmass, double *acc_dustmass,
                           double *acc_momentum, long long counts[3]);
void orc_cooling_and_starformation(int nactive, const int *active, int ngas, const int *type,
                                   const double *mass, const int *timebin, double timebase,
                                   double CritPhysDensity_code, double MinEgySpec,
                                   double u_to_temp_fac, const double *density,
                                   const double *entropy, double *dtentropy,
                                   double *injected_energy, int *flag_sink);

int orc_num_threads(void);
void orc_set_num_threads(int nthreads);

#ifdef __cplusplus
}
#endif
#endif
