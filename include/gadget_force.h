/*
 * gadget_force.h -- host-side mirror of the reference's call surface for the force path.
 *
 * The reference host is plain C on global AoS arrays (allvars.h, proto.h, forcetree.h).  This
 * header restates exactly the part of that interface the force path touches, with the same
 * names, argument meaning and error behaviour, so that
 *   (a) accel.c (accel.c:27-313) can call gravity_tree() / density() / force_update_hmax() /
 *       hydro_force() unchanged, and
 *   (b) the parity tests read like tests of the reference's own functions.
 * Everything computes on the GPU through include/ghip.h; there is no CPU fallback.
 *
 * Struct layout: the MINIMAL periodic flag set of BASELINE.md (-DPERIODIC -DPEANOHILBERT
 * -DDOUBLEPRECISION ...: sizeof(particle_data) = 112, sizeof(sph_particle_data) = 184,
 * SURVEY.md 8b).  A host built with another -D set keeps its own allvars.h and passes its own
 * offsets through ghip_layout; nothing in libghip.so depends on this struct.
 *
 * Reference interfaces replaced (file:line):
 *   gravity_tree            gravtree.c:27        set_softenings        gravtree.c:835
 *   force_treebuild         forcetree.c:67       force_update_hmax     forcetree.c:1661
 *   force_treeevaluate      forcetree.c:1797     ..._shortrange        forcetree.c:2330
 *   ..._ewald_correction    forcetree.c:2873     ewald_init            forcetree.c:4402
 *   density                 density.c:89         density_evaluate      density.c:711
 *   density_isactive        density.c:1035       hydro_force           hydra.c:145
 *   hydro_evaluate          hydra.c:822          ngb_treefind_variable ngb.c:169
 *   ngb_treefind_pairs      ngb.c:32             peano_hilbert_key     peano.c:300
 *   morton_key              peano.c:320          domain_findExtent     domain.c:1972
 *   endrun                  endrun.c:23
 */
#ifndef GADGET_FORCE_H
#define GADGET_FORCE_H

#include "ghip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef double MyFloat;       /* DOUBLEPRECISION, allvars.h:154-165 */
typedef double MyDouble;
typedef double MyLongDouble;  /* no FLTROUNDOFFREDUCTION, allvars.h:191-202 */
typedef unsigned long long peanokey; /* allvars.h:55 */

#define BITS_PER_DIMENSION 21 /* allvars.h:58 */
#define NODELISTLENGTH 8
#define GAMMA (7. / 5.)       /* allvars.h:64 */
#define GAMMA_MINUS1 (GAMMA - 1)

/* allvars.h:1131-1377, minimal flag set: 112 bytes */
struct particle_data
{
  MyDouble Pos[3];
  MyDouble Vel[3];
  MyDouble Mass;
  unsigned int ID;
  union
  {
    MyFloat GravAccel[3];
    MyLongDouble dGravAccel[3];
  } g;
  MyFloat OldAcc;
  float GravCost;
  int Ti_begstep;
  int Ti_current;
  short int Type;
  short int TimeBin;
};

/* allvars.h:1384-1639, minimal flag set (+BH_THERMALFEEDBACK's `i`): 184 bytes */
struct sph_particle_data
{
  MyDouble Entropy;
  MyFloat Pressure;
  MyDouble VelPred[3];
  MyFloat MaxSignalVel;
  union
  {
    MyFloat Density;
    MyLongDouble dDensity;
  } d;
  union
  {
    MyFloat DtEntropy;
    MyLongDouble dDtEntropy;
  } e;
  union
  {
    MyFloat HydroAccel[3];
    MyLongDouble dHydroAccel[3];
  } a;
  union
  {
    MyFloat DhsmlDensityFactor;
    MyLongDouble dDhsmlDensityFactor;
  } h;
  union
  {
    MyFloat DivVel;
    MyLongDouble dDivVel;
  } v;
  union
  {
    MyFloat CurlVel;
    MyFloat Rot[3];
    MyLongDouble dRot[3];
  } r;
  MyFloat Hsml;   /* PPP == SphP when !BLACK_HOLES && !DUST, allvars.h:266-270 */
  MyFloat Left, Right;
  union
  {
    MyFloat NumNgb;
    MyLongDouble dNumNgb;
  } n;
  union
  {
    MyFloat Injected_BH_Energy;
    MyLongDouble dInjected_BH_Energy;
  } i;
  MyFloat pad_[2];
};

#define PPP SphP /* allvars.h:266-270 */

/* the members of `All` (struct global_data_all_processes, allvars.h:530-1123) the path uses */
struct global_data_all_processes
{
  int MaxPart;
  double G;
  double ErrTolTheta, ErrTolForceAcc;
  int TypeOfOpeningCriterion;
  double BoxSize;
  double DesNumNgb, MaxNumNgbDeviation;
  double MinGasHsmlFractional, MinGasHsml;
  double ArtBulkViscConst;
  int Ti_Current;
  double Timebase_interval;
  double Time;
  int ComovingIntegrationOn;
  double Hubble, Omega0, OmegaLambda; /* hubble_function(), darkenergy.c:389 (no DARKENERGY) */
  double SofteningGas, SofteningHalo, SofteningDisk, SofteningBulge, SofteningStars, SofteningBndry;
  double SofteningGasMaxPhys, SofteningHaloMaxPhys, SofteningDiskMaxPhys, SofteningBulgeMaxPhys,
    SofteningStarsMaxPhys, SofteningBndryMaxPhys;
  double SofteningTable[6], ForceSoftening[6];
  double Rcut[2], Asmth[2];           /* PMGRID */
  long long TotNumOfForces;
  int BunchSize;
  double BufferSize;
  /* "next" row N1: timestep criterion + kick (timestep.c) */
  double ErrTolIntAccuracy, CourantFac, MaxSizeTimestep, MinSizeTimestep, MaxRMSDisplacementFac;
  double OmegaBaryon, MinEgySpec;
  int TypeOfTimestepCriterion, StarformationOn;
  /* "next" row N4: the sink passes (blackhole.c, density.c BLACK_HOLES / DUST branches) */
  double BlackHoleNgbFactor, BlackHoleFeedbackFactor;
  double SMBHmass, InnerBoundary, SinkBoundary, CritOverDensity;
  double UnitLength_in_cm, UnitMass_in_g, UnitEnergy_in_cgs;
};

/* A host whose `All` is the reference's full struct (allvars.h:530-1123, ~300 members whose
 * layout depends on the -D flags) cannot share the subset above.  Like P[]/SphP[] (ghip_layout) it
 * hands over BYTE OFFSETS instead: gadget_force_bind_all(&All_of_the_host, &offsets) makes every
 * entry point of this library read the members below from the host's struct and write back the ones
 * the path changes (ErrTolTheta gravtree.c:396-397, TotNumOfForces :783, SofteningTable /
 * ForceSoftening / MinGasHsml :835-884).  One int per member, in the order of GADGET_FORCE_ALL_MEMBERS;
 * -1 = the host's build has no such member (the library keeps its own value, 0 unless set).
 * A probe TU fills the table with offsetof(struct global_data_all_processes, member), INTEGRATION.md. */
#define GADGET_FORCE_ALL_MEMBERS(X)                                                               \
  X(MaxPart) X(G) X(ErrTolTheta) X(ErrTolForceAcc) X(TypeOfOpeningCriterion) X(BoxSize)          \
  X(DesNumNgb) X(MaxNumNgbDeviation) X(MinGasHsmlFractional) X(MinGasHsml) X(ArtBulkViscConst)   \
  X(Ti_Current) X(Timebase_interval) X(Time) X(ComovingIntegrationOn) X(Hubble) X(Omega0)        \
  X(OmegaLambda) X(SofteningGas) X(SofteningHalo) X(SofteningDisk) X(SofteningBulge)              \
  X(SofteningStars) X(SofteningBndry) X(SofteningGasMaxPhys) X(SofteningHaloMaxPhys)             \
  X(SofteningDiskMaxPhys) X(SofteningBulgeMaxPhys) X(SofteningStarsMaxPhys)                       \
  X(SofteningBndryMaxPhys) X(SofteningTable) X(ForceSoftening) X(Rcut) X(Asmth)                   \
  X(TotNumOfForces) X(BunchSize) X(BufferSize) X(ErrTolIntAccuracy) X(CourantFac)                 \
  X(MaxSizeTimestep) X(MinSizeTimestep) X(MaxRMSDisplacementFac) X(OmegaBaryon) X(MinEgySpec)     \
  X(TypeOfTimestepCriterion) X(StarformationOn) X(BlackHoleNgbFactor) X(BlackHoleFeedbackFactor)  \
  X(SMBHmass) X(InnerBoundary) X(SinkBoundary) X(CritOverDensity) X(UnitLength_in_cm)             \
  X(UnitMass_in_g) X(UnitEnergy_in_cgs)
struct gadget_force_all_layout
{
#define GADGET_FORCE_X(m) int m;
  GADGET_FORCE_ALL_MEMBERS(GADGET_FORCE_X)
#undef GADGET_FORCE_X
};
/* host_All == NULL unbinds (the library's own `All` is the state again).  The member types must be
 * the reference's (int / double / long long, MyFloat == double: DOUBLEPRECISION builds). */
void gadget_force_bind_all(void *host_All, const struct gadget_force_all_layout *offsets);
int gadget_force_all_layout_count(void);   /* number of ints in the table (ABI check for bindings) */

/* A host built with another -D set keeps its own struct particle_data / sph_particle_data -- the
 * shipped bundle's are 536 / 264 bytes, with Hsml and NumNgb in P (the PPP macro) and the
 * BLACK_HOLES / DUST unions -- and binds its arrays by BYTE OFFSETS: every driver of this library then
 * reads and writes the host's records through the tables (P / SphP below stay unused).  `lay` as for
 * ghip_upload_aos; `bh` (may be NULL) names the members the sink passes touch, -1 = absent:
 *   P[].ID, SwallowID (unsigned int), BH_Mass, BH_Mdot, b1.BH_Density, b2.BH_Entropy,
 *   b3.BH_SurroundingGasVel[3], b4.BH_accreted_Mass, b5.BH_accreted_BHMass, b5.BH_accreted_DustMass,
 *   b6.BH_accreted_momentum[3], d1.DUST_Density, d2.DUST_Entropy, d3.DUST_SurroundingGasVel[3],
 *   Dust_Mass (allvars.h:1206-1330), SphP[].i.Injected_BH_Energy (allvars.h:1580-1584).
 * A probe TU fills both tables with offsetof() under the host's flags. */
struct gadget_force_bh_layout
{
  int p_id, p_swallowid, p_bh_mass, p_bh_mdot, p_bh_density, p_bh_entropy, p_bh_gasvel;
  int p_bh_accreted_mass, p_bh_accreted_bhmass, p_bh_accreted_dustmass, p_bh_accreted_momentum;
  int p_dust_density, p_dust_entropy, p_dust_gasvel, p_dust_mass;
  int s_injected_bh_energy;
};
void gadget_force_bind_records(void *host_P, void *host_SphP, const ghip_layout *lay,
                               const struct gadget_force_bh_layout *bh);   /* host_P == NULL unbinds */

/* allvars.h:1673-1684: export bookkeeping record other translation units sort with the two
 * functions below (blackhole.c:366, dust.c:114, density.c:186 ...) */
struct data_index
{
  int Task;
  int Index;
  int IndexGet;
};

/* gravdata_in / gravdata_out, allvars.h:1690-1716 (mode == 1 records) */
struct gravdata_in
{
  MyFloat Pos[3];
  int Type;
  MyFloat OldAcc;
  int NodeList[NODELISTLENGTH];
};
struct gravdata_out
{
  MyLongDouble Acc[3];
  int Ninteractions;
};

/* struct NODE / struct extNODE (allvars.h:1847-1916) of the minimal flag set: 88 / 80 bytes */
struct NODE
{
  MyFloat len;
  MyFloat center[3];
  union
  {
    int suns[8];
    struct
    {
      MyFloat s[3];
      MyFloat mass;
      unsigned int bitflags;
      int sibling;
      int nextnode;
      int father;
    } d;
  } u;
  int Ti_current;
};
struct extNODE
{
  MyLongDouble dp[3];
  MyFloat vs[3];
  MyFloat vmax;
  MyFloat divVmax;
  MyFloat hmax;
  int Ti_lastkicked;
  int Flag;
};

/* compile-time switches of the reference that this library takes at run time */
struct gadget_force_config
{
  int periodic;            /* -DPERIODIC */
  int pmgrid;              /* -DPMGRID=n (0: off): gravity_tree uses the short-range walk */
  int unequal_softenings;  /* -DUNEQUALSOFTENINGS */
  int device;              /* GPU ordinal of this rank */
  int black_holes;         /* -DBLACK_HOLES (without NO_BH_ACCRETION): Type 5 is a density target
                              (density.c:1035-1041) */
  int dust;                /* -DDUST: Type 2 is a density target (density.c:1043-1046) */
  int accretion_of_dust_only;   /* -DACCRETION_OF_DUST_ONLY (blackhole.c:1003) */
  int accretion_density;        /* -DACCRETION_DENSITY (blackhole.c:1094) */
  int overlap_sph;         /* 1: on a step with gas, gravity_tree() returns with its walks still in
                              flight and density() / force_update_hmax() / hydro_force() run underneath
                              them on the device; the gravity results (GravAccel, OldAcc, GravCost) and
                              density()'s (one download of the SphP[] block per step instead of two)
                              reach the records when hydro_force() returns -- or gadget_force_flush().  Valid
                              for accel.c's sequence (accel.c:61-106: nothing reads P[].g.GravAccel
                              between the four calls); 0 (default): every driver returns with its own
                              results in P[] / SphP[] */
  int dynamic_tree;        /* All.DoDynamicUpdate: 1 = a gravity_tree() with TreeReconstructFlag == 0 walks
                              the tree of the last force_treebuild(), its nodes drifted and kicked as
                              force_drift_node / force_kick_node / force_finish_kick_nodes do
                              (forcetree.c:1356-1651; ghip_tree_substep), instead of a tree of the
                              current positions.  force_kick_node() and force_finish_kick_nodes() below
                              are then the reference's (timestep.c:261, 588); the library's own
                              advance_and_find_timesteps() hands its kicks to the nodes itself.
                              Comoving runs need gadget_force_set_drift_table(); 0 (default): every
                              gravity_tree() rebuilds */
  int pin_records;         /* 1: page-lock the first NumPart / N_gas records of P[] / SphP[] when they
                              are first uploaded (again when the arrays move or grow), so the record
                              copies run at the link's rate; released by gadget_force_finalize().  The
                              reference allocates both arrays once for All.MaxPart (allocate.c:30-60) */
};

/* ---- globals with the reference's names (allvars.c) ---- */
extern struct particle_data *P;
extern struct sph_particle_data *SphP;
extern struct global_data_all_processes All;
extern int NumPart, N_gas;
extern int FirstActiveParticle, *NextActiveParticle;
extern int TreeReconstructFlag;
extern double DomainCorner[3], DomainCenter[3], DomainLen, DomainFac;
extern int *Ngblist;
extern struct gravdata_in *GravDataGet;
extern struct gravdata_out *GravDataResult;
extern int ThisTask, NTask;
extern double CPU_Step_Treewalk, CPU_Step_Treebuild, CPU_Step_Density, CPU_Step_Hydro,
  CPU_Step_Hmaxupdate; /* the CPU_Step[] buckets the path fills (allvars.h:205-238) */

/* the host's tree arrays (allvars.h:1880-1916, forcetree.c:4560-4600 force_treeallocate).  When
 * Nodes_base is set, force_treebuild() also exports the device-built tree into them ("next" row
 * N2): Nodes = Nodes_base - All.MaxPart as in the reference, Numnodestree = number of nodes. */
extern struct NODE *Nodes_base, *Nodes;
extern struct extNODE *Extnodes_base, *Extnodes;
extern int *Nextnode, *Father;
extern int MaxNodes, Numnodestree;

/* time bins (allvars.h:337-346) and the step flag of run.c; TIMEBINS = 29 */
#define TIMEBINS 29
#define TIMEBASE (1 << TIMEBINS)
extern int TimeBinCount[TIMEBINS], TimeBinCountSph[TIMEBINS], TimeBinActive[TIMEBINS];
extern int FirstInTimeBin[TIMEBINS], LastInTimeBin[TIMEBINS];
extern int *NextInTimeBin, *PrevInTimeBin; /* sized MaxPart by the host */
extern int Flag_FullStep;

/* ---- library management (no counterpart in the reference) ---- */
int gadget_force_init(const struct gadget_force_config *cfg);  /* 0 or a GHIP_E* code */
void gadget_force_finalize(void);
const char *gadget_force_last_error(void);
ghip_ctx *gadget_force_ctx(void);
void gadget_force_layout(ghip_layout *lay);   /* offsets of the structs above */
/* endrun(code) handler: default prints "task %d: endrun called with an error level of %d" and
 * abort()s like MPI_Abort would; tests install a recording handler. */
void gadget_force_set_endrun(void (*handler)(int code));
/* the comoving kick-factor tables of driftfac.c (GravKickTable / HydroKickTable, 1000 entries each,
 * built by the host's init_drift_table) and their log(a) range; needed when
 * All.ComovingIntegrationOn is set */
/* the host's DriftTable[DRIFT_TABLE_LENGTH] (driftfac.c:26-60): cfg.dynamic_tree in comoving runs takes
 * force_drift_node's dt_drift from it (get_drift_factor, driftfac.c:123-163; logTimeBegin / logTimeMax
 * as given to gadget_force_set_kick_tables); without it a comoving sub-step rebuilds */
void gadget_force_set_drift_table(const double *drifttable);
void gadget_force_set_kick_tables(const double *gravkick, const double *hydrokick,
                                  double logTimeBegin, double logTimeMax);
/* tell the glue that the host changed P/SphP outside the four drivers */
void gadget_force_mark_dirty(void);
/* overlap_sph: complete a gravity_tree() whose results are still on the device (no-op otherwise) */
void gadget_force_flush(void);

/* ---- the reference's call surface ---- */
void endrun(int ierr);
void set_softenings(void);
/* gravtree.c:892-907 (order by Task, then Index) and :909-963 (a stable sort of data_index records
 * with that interface; they live in gravtree.c, which this library replaces) */
int data_index_compare(const void *a, const void *b);
void mysort_dataindex(void *b, size_t n, size_t s, int (*cmp)(const void *, const void *));
void domain_findExtent(void);
int force_treebuild(int npart, void *mp);
/* forcetree.c:1455, 1522 (cfg.dynamic_tree): the momentum a kicked particle hands to its ancestors;
 * collected per call and applied to the kept tree by force_finish_kick_nodes() */
void force_kick_node(int i, MyFloat *dv);
void force_finish_kick_nodes(void);
void ewald_init(void);
void gravity_tree(void);
void density(void);
int density_isactive(int n);
void force_update_hmax(void);
void hydro_force(void);
/* "next" row N1 (timestep.c:29, 1125, 1226): the particle loop runs on the device */
void advance_and_find_timesteps(void);
void find_dt_displacement_constraint(double hfac);
int get_timestep_bin(int ti_step);

int force_treeevaluate(int target, int mode, int *nexport, int *nsend_local);
int force_treeevaluate_shortrange(int target, int mode, int *nexport, int *nsend_local);
int force_treeevaluate_ewald_correction(int target, int mode, int *nexport, int *nsend_local);
int density_evaluate(int target, int mode, int *nexport, int *nsend_local);
int hydro_evaluate(int target, int mode, int *nexport, int *nsend_local);
int ngb_treefind_variable(MyDouble searchcenter[3], MyFloat hsml, int target, int *startnode,
                          int mode, int *nexport, int *nsend_local);
int ngb_treefind_pairs(MyDouble searchcenter[3], MyFloat hsml, int target, int *startnode,
                       int mode, int *nexport, int *nsend_local);

/* "next" row N4, BLACK_HOLES builds with bound records (gadget_force_bind_records with a bh table):
 * the two per-sink neighbour passes with the reference's signatures (proto.h:110-111; mode 0 = a
 * local sink, index into P[]) -- a batch of one on the device -- and the neighbour-pass core of
 * blackhole_accretion() (blackhole.c:70): for ALL active sinks, SwallowID / Injected_BH_Energy reset
 * (:300-305 via the marks), blackhole_evaluate (:346-353), blackhole_evaluate_swallow (:545-551), the
 * accreted sums added to the sinks' records (:648-660), the victims' Mass zeroed, and the swallow
 * counts (N_gas_swallowed, N_BH_swallowed, N_dust_swallowed).  The per-sink scalar bookkeeping
 * around it (Mdot, Eddington limit, logs: blackhole.c:133-290, 690-800) stays the host's. */
int blackhole_evaluate(int target, int mode, int *nexport, int *nsend_local);
int blackhole_evaluate_swallow(int target, int mode, int *nexport, int *nsend_local);
void blackhole_accretion_neighbour_passes(void);
extern int N_gas_swallowed, N_BH_swallowed, N_dust_swallowed;

/* ---- more than one rank (NTask > 1): the drivers above become collectives ----
 * The host has decomposed the domain (domain_Decomposition, domain.c:100) and every rank holds the
 * particles of its key range in P[0, NumPart).  With NTask > 1 gravity_tree() / density() /
 * force_update_hmax() / hydro_force() take the ranks' key ranges from the host's own top-tree
 *   splits[r] = TopNodes[<the leaf DomainStartList[r]>].StartKey          (domain.c:1075-1113, 1495)
 * and run the domain-decomposed device path (ghip_dd_*: locally essential trees and SPH ghosts in
 * place of the export loops gravtree.c:175-339, density.c:193-389, hydra.c:274-526 and of the
 * pseudo-particle exchange forcetree.c:879-1016, 1722-1747).  MULTIPLEDOMAINS must be 1.
 * Exchanges go over RCCL (gadget_force_connect: every rank passes the 128-byte id that rank 0
 * obtained from gadget_force_unique_id and broadcast over the host's MPI), or -- a host without
 * RCCL, several ranks on one GPU -- through the host's own all-gather (gadget_force_set_allgather:
 * `allgather(user, send, bytes, recv)` = MPI_Allgather(send, bytes, MPI_BYTE, recv, bytes, MPI_BYTE)). */
struct topnode_data   /* allvars.h:437-447 */
{
  peanokey Size;
  peanokey StartKey;
  long long Count;
  MyFloat GravCost;
  int Daughter;
  int Pstart;
  int Blocks;
  int Leaf;
};
extern struct topnode_data *TopNodes;
extern int NTopnodes, NTopleaves;
extern int *DomainStartList, *DomainEndList;
extern int *DomainTask;   /* allvars.h:429: the rank of every top-leaf.  When set (-DMULTIPLEDOMAINS > 1: a rank
                             owns several pieces of the curve, domain.c:1158-1215) the drivers take the
                             ownership of the curve from it, leaf by leaf, instead of from DomainStartList */
int gadget_force_unique_id(void *id128);
int gadget_force_connect(const void *id128);
void gadget_force_set_allgather(int (*allgather)(void *user, const void *send, size_t bytes, void *recv),
                                void *user);

peanokey peano_hilbert_key(int x, int y, int z, int bits);
peanokey morton_key(int x, int y, int z, int bits);
double hubble_function(double a);

#ifdef __cplusplus
}
#endif
#endif
