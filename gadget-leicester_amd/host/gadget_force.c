/*
 * gadget_force.c -- host C glue: the reference's force-path call surface on top of libghip.so.
 *
 * Mirrors, function by function, what accel.c (accel.c:27-313) and init.c:791 call:
 * gravity_tree() (gravtree.c:27-828), density() (density.c:89-704), force_update_hmax()
 * (forcetree.c:1661-1786), hydro_force() (hydra.c:145-813) plus the per-target *_evaluate and
 * ngb_treefind_* entry points (forcetree.h:31-35, 107-111; proto.h:205, 229-230).
 *
 * The batched drivers replace the reference's `for(i = FirstActiveParticle; i >= 0;
 * i = NextActiveParticle[i])` loops (gravtree.c:130, density.c:174, hydra.c:257) by one device
 * launch over the whole active list.  The per-target entry points run a batch of one on the GPU.
 * Nothing here computes forces on the CPU.
 */
#include "gadget_force.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define GHIP_DD_MAXRANKS_HOST 64   /* GHIP_MAXRANKS of the device library */

/* ---- globals (allvars.c) ---- */
struct particle_data *P = NULL;
struct sph_particle_data *SphP = NULL;
struct global_data_all_processes All;
int NumPart = 0, N_gas = 0;
int FirstActiveParticle = -1, *NextActiveParticle = NULL;
int TreeReconstructFlag = 1;
struct NODE *Nodes_base, *Nodes;
struct extNODE *Extnodes_base, *Extnodes;
int *Nextnode, *Father;
int MaxNodes, Numnodestree;
int TimeBinCount[TIMEBINS], TimeBinCountSph[TIMEBINS], TimeBinActive[TIMEBINS];
int FirstInTimeBin[TIMEBINS], LastInTimeBin[TIMEBINS];
int *NextInTimeBin, *PrevInTimeBin;
int Flag_FullStep;
static double dt_displacement = 0; /* timestep.c:17 */
static const double *KickTabGrav, *KickTabHydro, *DriftTab;
static double KickLogBegin, KickLogMax;
double DomainCorner[3], DomainCenter[3], DomainLen = 0, DomainFac = 0;
int *Ngblist = NULL;
struct gravdata_in *GravDataGet = NULL;
struct gravdata_out *GravDataResult = NULL;
int ThisTask = 0, NTask = 1;
double CPU_Step_Treewalk = 0, CPU_Step_Treebuild = 0, CPU_Step_Density = 0, CPU_Step_Hydro = 0,
  CPU_Step_Hmaxupdate = 0;

struct topnode_data *TopNodes = NULL;
int NTopnodes = 0, NTopleaves = 0;
int *DomainStartList = NULL, *DomainEndList = NULL;
int *DomainTask = NULL;
int N_gas_swallowed = 0, N_BH_swallowed = 0, N_dust_swallowed = 0;

static ghip_ctx *Ctx = NULL;
static struct gadget_force_config Cfg;

/* ---- the particle records: this header's structs, or the host's own through offsets tables ---- */
static ghip_layout Lay;                      /* byte offsets in use */
static struct gadget_force_bh_layout BhLay;  /* ... of the members the sink passes touch (-1: absent) */
static char *RecP = NULL, *RecS = NULL;      /* bound records (gadget_force_bind_records), else P / SphP */

/* (the offsets of this header's own structs from the moment the library is loaded: some entry points
 * -- domain_findExtent, density_isactive -- are legitimately called before gadget_force_init) */
__attribute__((constructor)) static void lay_defaults(void)
{
  gadget_force_layout(&Lay);
  memset(&BhLay, 0xff, sizeof(BhLay));
  BhLay.p_id = (int) offsetof(struct particle_data, ID);
  BhLay.s_injected_bh_energy = (int) offsetof(struct sph_particle_data, i);
}

static inline char *prec(int i)
{
  return (RecP ? RecP : (char *) P) + (size_t) i * (size_t) Lay.p_stride;
}
static inline char *srec(int i)
{
  return (RecP ? RecS : (char *) SphP) + (size_t) i * (size_t) Lay.s_stride;
}
#define PF64(i, off) ((double *) (prec(i) + (off)))
#define SF64(i, off) ((double *) (srec(i) + (off)))
static inline int p_type(int i) { return (int) *(short *) (prec(i) + Lay.p_type); }
static inline int p_timebin(int i) { return (int) *(short *) (prec(i) + Lay.p_timebin); }
static inline double *ppp_hsml(int i)   /* the PPP macro, allvars.h:266-270 */
{
  return Lay.p_hsml >= 0 ? PF64(i, Lay.p_hsml) : SF64(i, Lay.s_hsml);
}
static inline double *ppp_numngb(int i)
{
  return Lay.p_numngb >= 0 ? PF64(i, Lay.p_numngb) : SF64(i, Lay.s_numngb);
}
static void *records_p(void) { return RecP ? (void *) RecP : (void *) P; }
static void *records_s(void) { return RecP ? (void *) RecS : (void *) SphP; }

/* ---- more than one rank ---- */
static void gravity_tree_ranks(void);
static void density_ranks(void);
static void hydro_force_ranks(void);
static int (*AllgatherFn)(void *, const void *, size_t, void *) = NULL;
static void *AllgatherUser = NULL;
static int DdReady = 0;          /* ghip_dd_init done for (ThisTask, NTask) */
static int RcclConnected = 0;
static void (*EndrunHandler)(int) = NULL;
static int DeviceFresh = 0;      /* device copy of P/SphP matches the host arrays */
static int TreeOnDevice = 0;
static int Phase = 0;            /* 1 after gravity_tree(), 2 after density(): accel.c:61-106 order */
static int GravPending = 0;      /* overlap_sph: walks in flight, post-pass + download still to do */
static int GasPending = 0;       /* overlap_sph: P[] is on the device, SphP[] still to follow */
static int GravPendingActive = 0;
/* cfg.dynamic_tree: the tree force_treebuild() left on the device is kept across sub-steps */
static int KeptTree = 0, KeptN = -1, KeptTi = 0;
static int *KickIdx = NULL;
static double *KickDv = NULL, *KickVmax = NULL;
static int KickN = 0, KickCap = 0;
/* cfg.pin_records: the record arrays page-locked so far */
static void *PinP, *PinS;
static size_t PinPBytes, PinSBytes;
static int DensPending = 0;      /* overlap_sph: density()'s results are still on the device */
static int *ActiveBuf = NULL;
static int ActiveCap = 0;
static int NgblistCap = 0;
static char ErrBuf[1200];

static double wallclock(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* GADGET_FORCE_TRACE=1: wall-clock stamps of the drivers' stages on stderr (development aid) */
static int TraceOn = -1;
static double TraceT0;
static void trace(const char *what)
{
  if(TraceOn < 0)
    {
      const char *e = getenv("GADGET_FORCE_TRACE");
      TraceOn = e && atoi(e) > 0;
    }
  if(!TraceOn)
    return;
  double t = wallclock();
  if(!what)
    TraceT0 = t;
  else
    fprintf(stderr, "[gadget_force] %-28s +%8.3f ms\n", what, 1e3 * (t - TraceT0));
}

/* endrun.c:23-38 */
void endrun(int ierr)
{
  if(EndrunHandler)
    {
      EndrunHandler(ierr);
      return;
    }
  if(ierr)
    {
      printf("task %d: endrun called with an error level of %d\n\n\n", ThisTask, ierr);
      fflush(stdout);
      abort();
    }
  exit(0);
}

void gadget_force_set_endrun(void (*handler)(int code))
{
  EndrunHandler = handler;
}

const char *gadget_force_last_error(void)
{
  return ErrBuf;
}

ghip_ctx *gadget_force_ctx(void)
{
  return Ctx;
}

void gadget_force_mark_dirty(void)
{
  DeviceFresh = 0;
}

/* a failing device call ends the run the way the reference ends it: endrun(code).  Codes 9xxxx
 * are reserved for this library. */
static int chk(int rc, const char *where)
{
  if(rc == GHIP_OK)
    return 0;
  snprintf(ErrBuf, sizeof(ErrBuf), "%s: %d %s", where, rc, Ctx ? ghip_last_error(Ctx) : "");
  fprintf(stderr, "gadget_force: %s\n", ErrBuf);
  endrun(rc == GHIP_ENOCONV ? 1155 : -rc);
  return rc;
}

int gadget_force_init(const struct gadget_force_config *cfg)
{
  if(!cfg)
    return GHIP_EINVAL;
  if(Ctx)
    gadget_force_finalize();
  Cfg = *cfg;
  ErrBuf[0] = 0;
  int rc = ghip_create(cfg->device, &Ctx);
  if(rc != GHIP_OK)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "ghip_create(device %d) failed: %d (no GPU: this library "
               "has no CPU path)", cfg->device, rc);
      Ctx = NULL;
      return rc;
    }
  DeviceFresh = 0;
  TreeOnDevice = 0;
  DdReady = 0;
  RcclConnected = 0;
  if(!RecP)
    lay_defaults();
  /* density.c:831-834, hydra.c:1235-1238 */
  ghip_set_massless_gas_rule(Ctx, cfg->black_holes ? 3 : (cfg->dust ? 1 : 0));
  KeptTree = 0;
  KickN = 0;
  if(cfg->dynamic_tree && ghip_set_dynamic_tree(Ctx, 1) != GHIP_OK)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "ghip_set_dynamic_tree: %s", ghip_last_error(Ctx));
      return GHIP_EINVAL;
    }
  /* (measured, not kept as the default: releasing the hydro kernel at once so that the SphP[] block
   * crosses the link under the walks -- the kernel does not get its registers before the Ewald walk
   * drains anyway; GADGET_FORCE_HYDRO_EARLY=1 reproduces it) */
  if(cfg->overlap_sph && getenv("GADGET_FORCE_HYDRO_EARLY"))
    ghip_set_hydro_release(Ctx, atoi(getenv("GADGET_FORCE_HYDRO_EARLY")));
  return GHIP_OK;
}

void gadget_force_bind_records(void *host_P, void *host_SphP, const ghip_layout *lay,
                               const struct gadget_force_bh_layout *bh)
{
  if(host_P && lay)
    {
      RecP = (char *) host_P;
      RecS = (char *) host_SphP;
      Lay = *lay;
      if(bh)
        BhLay = *bh;
      else
        memset(&BhLay, 0xff, sizeof(BhLay));
    }
  else
    {
      RecP = RecS = NULL;
      lay_defaults();
    }
  DeviceFresh = 0;
  TreeOnDevice = 0;
}

void gadget_force_finalize(void)
{
  if(Ctx)
    {
      if(PinP)
        ghip_unpin_host(Ctx, PinP);
      if(PinS)
        ghip_unpin_host(Ctx, PinS);
      ghip_destroy(Ctx);
    }
  PinP = PinS = NULL;
  PinPBytes = PinSBytes = 0;
  free(KickIdx);
  free(KickDv);
  free(KickVmax);
  KickIdx = NULL;
  KickDv = KickVmax = NULL;
  KickN = KickCap = 0;
  KeptTree = 0;
  Ctx = NULL;
  free(ActiveBuf);
  ActiveBuf = NULL;
  ActiveCap = 0;
  free(Ngblist);
  Ngblist = NULL;
  NgblistCap = 0;
  DeviceFresh = 0;
  TreeOnDevice = 0;
  Phase = 0;
  GravPending = 0;
  GasPending = 0;
  DensPending = 0;
  /* the host's arrays are the host's: forget them, a later init must set them again */
  Nodes_base = Nodes = NULL;
  Extnodes_base = Extnodes = NULL;
  Nextnode = Father = NULL;
  MaxNodes = Numnodestree = 0;
  NextInTimeBin = PrevInTimeBin = NULL;
  KickTabGrav = KickTabHydro = DriftTab = NULL;
  dt_displacement = 0;
}

void gadget_force_layout(ghip_layout *lay)
{
  memset(lay, 0xff, sizeof(*lay)); /* every offset -1 */
  lay->p_stride = (int) sizeof(struct particle_data);
  lay->p_pos = (int) offsetof(struct particle_data, Pos);
  lay->p_vel = (int) offsetof(struct particle_data, Vel);
  lay->p_mass = (int) offsetof(struct particle_data, Mass);
  lay->p_gravaccel = (int) offsetof(struct particle_data, g);
  lay->p_oldacc = (int) offsetof(struct particle_data, OldAcc);
  lay->p_gravcost = (int) offsetof(struct particle_data, GravCost);
  lay->p_ti_begstep = (int) offsetof(struct particle_data, Ti_begstep);
  lay->p_type = (int) offsetof(struct particle_data, Type);
  lay->p_timebin = (int) offsetof(struct particle_data, TimeBin);
  lay->p_hsml = -1;
  lay->p_numngb = -1;
  lay->s_stride = (int) sizeof(struct sph_particle_data);
  lay->s_entropy = (int) offsetof(struct sph_particle_data, Entropy);
  lay->s_pressure = (int) offsetof(struct sph_particle_data, Pressure);
  lay->s_velpred = (int) offsetof(struct sph_particle_data, VelPred);
  lay->s_maxsignalvel = (int) offsetof(struct sph_particle_data, MaxSignalVel);
  lay->s_density = (int) offsetof(struct sph_particle_data, d);
  lay->s_dtentropy = (int) offsetof(struct sph_particle_data, e);
  lay->s_hydroaccel = (int) offsetof(struct sph_particle_data, a);
  lay->s_dhsmlfac = (int) offsetof(struct sph_particle_data, h);
  lay->s_divvel = (int) offsetof(struct sph_particle_data, v);
  lay->s_curlvel = (int) offsetof(struct sph_particle_data, r);
  lay->s_hsml = (int) offsetof(struct sph_particle_data, Hsml);
  lay->s_numngb = (int) offsetof(struct sph_particle_data, n);
  lay->p_ti_current = (int) offsetof(struct particle_data, Ti_current);
}

/* ---- `All` of a host with the reference's full struct: members by byte offset ---------------- */
static char *HostAll = NULL;
static struct gadget_force_all_layout HostAllLay;

struct all_member
{
  size_t own_off, size;
  int writeback;
};
#define GF_MEMBER(m) { offsetof(struct global_data_all_processes, m), sizeof(All.m), 0 },
static struct all_member AllMembers[] = { GADGET_FORCE_ALL_MEMBERS(GF_MEMBER) };
#undef GF_MEMBER
#define N_ALL_MEMBERS ((int) (sizeof(AllMembers) / sizeof(AllMembers[0])))

int gadget_force_all_layout_count(void)
{
  return N_ALL_MEMBERS;
}

static void all_mark_writeback(size_t own_off)
{
  for(int k = 0; k < N_ALL_MEMBERS; k++)
    if(AllMembers[k].own_off == own_off)
      AllMembers[k].writeback = 1;
}

void gadget_force_bind_all(void *host_All, const struct gadget_force_all_layout *offsets)
{
  HostAll = (host_All && offsets) ? (char *) host_All : NULL;
  if(HostAll)
    HostAllLay = *offsets;
  /* what the path itself changes (gravtree.c:396-397, 783, 835-884) */
  all_mark_writeback(offsetof(struct global_data_all_processes, ErrTolTheta));
  all_mark_writeback(offsetof(struct global_data_all_processes, TotNumOfForces));
  all_mark_writeback(offsetof(struct global_data_all_processes, SofteningTable));
  all_mark_writeback(offsetof(struct global_data_all_processes, ForceSoftening));
  all_mark_writeback(offsetof(struct global_data_all_processes, MinGasHsml));
}

/* host -> library: at the top of every entry point */
static void all_pull(void)
{
  if(!HostAll)
    return;
  const int *off = (const int *) &HostAllLay;
  for(int k = 0; k < N_ALL_MEMBERS; k++)
    if(off[k] >= 0)
      memcpy((char *) &All + AllMembers[k].own_off, HostAll + off[k], AllMembers[k].size);
}

/* library -> host: after the path changed one of its members */
static void all_push(void)
{
  if(!HostAll)
    return;
  const int *off = (const int *) &HostAllLay;
  for(int k = 0; k < N_ALL_MEMBERS; k++)
    if(off[k] >= 0 && AllMembers[k].writeback)
      memcpy(HostAll + off[k], (char *) &All + AllMembers[k].own_off, AllMembers[k].size);
}

/* gravtree.c:835-884 */
void set_softenings(void)
{
  all_pull();
  const double soft[6] = { All.SofteningGas, All.SofteningHalo, All.SofteningDisk,
    All.SofteningBulge, All.SofteningStars, All.SofteningBndry };
  const double maxphys[6] = { All.SofteningGasMaxPhys, All.SofteningHaloMaxPhys,
    All.SofteningDiskMaxPhys, All.SofteningBulgeMaxPhys, All.SofteningStarsMaxPhys,
    All.SofteningBndryMaxPhys };
  for(int i = 0; i < 6; i++)
    {
      if(All.ComovingIntegrationOn && soft[i] * All.Time > maxphys[i])
        All.SofteningTable[i] = maxphys[i] / All.Time;
      else
        All.SofteningTable[i] = soft[i];
    }
  for(int i = 0; i < 6; i++)
    All.ForceSoftening[i] = 2.8 * All.SofteningTable[i];
  All.MinGasHsml = All.MinGasHsmlFractional * All.ForceSoftening[0];
  all_push();
}

/* gravtree.c:892-907 */
int data_index_compare(const void *a, const void *b)
{
  const struct data_index *x = (const struct data_index *) a, *y = (const struct data_index *) b;
  if(x->Task != y->Task)
    return x->Task < y->Task ? -1 : +1;
  if(x->Index != y->Index)
    return x->Index < y->Index ? -1 : +1;
  return 0;
}

/* gravtree.c:909-963: stable sort of data_index records.  Bottom-up merge between the array and
 * one scratch copy (the reference recurses top-down; the result of a stable sort is unique). */
void mysort_dataindex(void *b, size_t n, size_t s, int (*cmp)(const void *, const void *))
{
  if(n < 2 || s != sizeof(struct data_index))
    {
      if(n >= 2)
        qsort(b, n, s, cmp);
      return;
    }
  struct data_index *src = (struct data_index *) b;
  struct data_index *tmp = (struct data_index *) malloc(n * sizeof(struct data_index));
  if(!tmp)
    {
      endrun(90004);
      return;
    }
  struct data_index *dst = tmp;
  for(size_t w = 1; w < n; w *= 2)
    {
      for(size_t lo = 0; lo < n; lo += 2 * w)
        {
          size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
          size_t i = lo, j = mid, k = lo;
          while(i < mid && j < hi)
            dst[k++] = (cmp(&src[j], &src[i]) < 0) ? src[j++] : src[i++];
          while(i < mid)
            dst[k++] = src[i++];
          while(j < hi)
            dst[k++] = src[j++];
        }
      struct data_index *t = src;
      src = dst;
      dst = t;
    }
  if(src != (struct data_index *) b)
    memcpy(b, src, n * sizeof(struct data_index));
  free(tmp);
}

/* darkenergy.c:389-409 without DARKENERGY */
double hubble_function(double a)
{
  all_pull();
  double hubble_a = All.Omega0 / (a * a * a) + (1 - All.Omega0 - All.OmegaLambda) / (a * a) +
    All.OmegaLambda;
  return All.Hubble * sqrt(hubble_a);
}

/* domain.c:1972-2014 (single rank) */
void domain_findExtent(void)
{
  double xmin[3] = { 1e300, 1e300, 1e300 }, xmax[3] = { -1e300, -1e300, -1e300 };
  for(int i = 0; i < NumPart; i++)
    {
      const double *pos = PF64(i, Lay.p_pos);
      for(int j = 0; j < 3; j++)
        {
          if(xmin[j] > pos[j])
            xmin[j] = pos[j];
          if(xmax[j] < pos[j])
            xmax[j] = pos[j];
        }
    }
  /* domain.c:1996-1997: more than one rank, the extent of all of them (through the host's all-gather) */
  if(NTask > 1 && AllgatherFn)
    {
      double mine[6] = { xmin[0], xmin[1], xmin[2], xmax[0], xmax[1], xmax[2] };
      double *all = (double *) malloc((size_t) NTask * sizeof(mine));
      if(all && AllgatherFn(AllgatherUser, mine, sizeof(mine), all) == 0)
        for(int r = 0; r < NTask; r++)
          for(int j = 0; j < 3; j++)
            {
              if(all[6 * r + j] < xmin[j])
                xmin[j] = all[6 * r + j];
              if(all[6 * r + 3 + j] > xmax[j])
                xmax[j] = all[6 * r + 3 + j];
            }
      free(all);
    }
  double len = 0;
  for(int j = 0; j < 3; j++)
    if(xmax[j] - xmin[j] > len)
      len = xmax[j] - xmin[j];
  len *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      DomainCenter[j] = 0.5 * (xmin[j] + xmax[j]);
      DomainCorner[j] = 0.5 * (xmin[j] + xmax[j]) - 0.5 * len;
    }
  DomainLen = len;
  DomainFac = 1.0 / len * (((peanokey) 1) << (BITS_PER_DIMENSION));
}

/* peano.c:320-333 */
peanokey morton_key(int x, int y, int z, int bits)
{
  peanokey m = 0;
  for(int b = bits - 1; b >= 0; b--)
    {
      m <<= 3;
      m += (peanokey) ((((z >> b) & 1) << 2) + (((y >> b) & 1) << 1) + ((x >> b) & 1));
    }
  return m;
}

/* peano.c:300-316: the reference's curve generated from its definition (base octant order +
 * per-octant cube symmetry) instead of its 48-state tables; see ghip_tree.hip */
peanokey peano_hilbert_key(int x, int y, int z, int bits)
{
  static const unsigned char base[8] = { 0, 7, 1, 6, 3, 4, 2, 5 };
  static const unsigned char cperm[8][3] = { {0, 2, 1}, {0, 2, 1}, {2, 1, 0}, {2, 1, 0},
  {0, 1, 2}, {0, 1, 2}, {2, 1, 0}, {2, 1, 0} };
  static const unsigned char cflip[8][3] = { {0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {1, 0, 1},
  {1, 1, 0}, {1, 1, 0}, {0, 0, 0}, {1, 0, 1} };
  int perm[3] = { 0, 1, 2 }, flip[3] = { 0, 0, 0 };
  peanokey key = 0;
  for(int b = bits - 1; b >= 0; b--)
    {
      int v[3] = { (x >> b) & 1, (y >> b) & 1, (z >> b) & 1 };
      int w[3];
      for(int i = 0; i < 3; i++)
        w[i] = v[perm[i]] ^ flip[i];
      int local = w[0] * 4 + w[1] * 2 + w[2];
      key = (key << 3) | base[local];
      int np[3], nf[3];
      for(int i = 0; i < 3; i++)
        {
          np[i] = perm[cperm[local][i]];
          nf[i] = flip[cperm[local][i]] ^ cflip[local][i];
        }
      memcpy(perm, np, sizeof(perm));
      memcpy(flip, nf, sizeof(flip));
    }
  return key;
}

/* ------------------------------------------------------------------------------------------ */
static int need_ctx(const char *who)
{
  all_pull();   /* every driver starts here: the host's `All` is the state (gadget_force_bind_all) */
  if(Ctx)
    return 0;
  snprintf(ErrBuf, sizeof(ErrBuf), "%s: gadget_force_init() has not succeeded (no GPU context)",
           who);
  fprintf(stderr, "gadget_force: %s\n", ErrBuf);
  endrun(90005);
  return -1;
}

static void pin_one(void **have, size_t *have_bytes, void *ptr, size_t bytes)
{
  if(*have == ptr && bytes <= *have_bytes)
    return;
  if(*have)
    ghip_unpin_host(Ctx, *have);
  *have = NULL;
  *have_bytes = 0;
  if(!ptr || bytes == 0)
    return;
  if(ghip_pin_host(Ctx, ptr, bytes) == 0)
    {
      *have = ptr;
      *have_bytes = bytes;
    }
}

static void pin_records(void)
{
  if(!Cfg.pin_records || !Ctx)
    return;
  /* a little room for a NumPart that creeps up between decompositions, so that not every step
   * locks again -- never beyond the array the host allocated (All.MaxPart records, allocate.c:38) */
  size_t np = (size_t) NumPart, ns = (size_t) N_gas;
  if(PinP != (void *) records_p() || np * (size_t) Lay.p_stride > PinPBytes)
    {
      np += np / 16;
      if(np > (size_t) All.MaxPart)
        np = (size_t) (All.MaxPart > NumPart ? All.MaxPart : NumPart);
    }
  pin_one(&PinP, &PinPBytes, records_p(), np * (size_t) Lay.p_stride);
  pin_one(&PinS, &PinSBytes, records_s(), ns * (size_t) Lay.s_stride);
}

/* split != 0 (gravity_tree with overlap_sph): only the P[] block now, the SphP[] block once the walks
 * are in flight (upload_gas_if_pending) */
static int upload_particles(int split)
{
  pin_records();
  if(split && N_gas > 0 && Lay.p_hsml < 0)
    {
      if(chk(ghip_upload_aos_particles(Ctx, records_p(), &Lay, NumPart, N_gas), "ghip_upload_aos_particles"))
        return -1;
      GasPending = 1;
    }
  else
    {
      if(chk(ghip_upload_aos(Ctx, records_p(), records_s(), &Lay, NumPart, N_gas), "ghip_upload_aos"))
        return -1;
      GasPending = 0;
    }
  DeviceFresh = 1;
  TreeOnDevice = 0;
  return 0;
}

static int upload_gas_if_pending(void)
{
  if(!GasPending)
    return 0;
  GasPending = 0;
  return chk(ghip_upload_aos_gas(Ctx, records_s(), &Lay), "ghip_upload_aos_gas") ? -1 : 0;
}

/* Everybody active, in index order?  (a full step: FirstActiveParticle = 0, NextActiveParticle[i] =
 * i + 1, run.c:300-320.)  A pass without the list's load-to-load dependency. */
static int active_list_is_everybody(void)
{
  if(NumPart <= 0 || FirstActiveParticle != 0 || !NextActiveParticle)
    return 0;
  int bad = 0;
  const int *next = NextActiveParticle;
  for(int i = 0; i < NumPart - 1; i++)
    bad |= next[i] ^ (i + 1);
  return bad == 0 && next[NumPart - 1] == -1;
}

/* the active list as the reference threads it (run.c:300-320); gas_only: the SPH drivers' filter
 * P[i].Type == 0 (density.c:1049, hydra.c:184) on the gas block */
static int collect_active(int gas_only)
{
  /* a full step needs no list: everybody for gravity; the whole gas block for SPH -- unless the
   * upload found a converted particle in it (ghip_gas_block_mixed), then the list decides */
  if(active_list_is_everybody())
    {
      if(!gas_only)
        return NumPart;
      if(DeviceFresh && ghip_gas_block_mixed(Ctx) == 0)
        return N_gas;
    }
  int n = 0;
  for(int i = FirstActiveParticle; i >= 0; i = NextActiveParticle[i])
    {
      if(gas_only && !(i < N_gas && p_type(i) == 0))
        continue;
      if(n >= ActiveCap)
        {
          int nc = ActiveCap ? 2 * ActiveCap : 1024;
          int *nb = (int *) realloc(ActiveBuf, (size_t) nc * sizeof(int));
          if(!nb)
            {
              endrun(90003);
              return -1;
            }
          ActiveBuf = nb;
          ActiveCap = nc;
        }
      ActiveBuf[n++] = i;
    }
  return n;
}

static void fill_grav_params(ghip_grav_params *g)
{
  memset(g, 0, sizeof(*g));
  g->ErrTolTheta = All.ErrTolTheta;
  g->ErrTolForceAcc = All.ErrTolForceAcc;
  for(int i = 0; i < 6; i++)
    g->ForceSoftening[i] = All.ForceSoftening[i];
  g->BoxSize = All.BoxSize;
  g->periodic = Cfg.periodic;
  g->unequal_softenings = Cfg.unequal_softenings;
  g->Rcut = All.Rcut[0];
  g->Asmth = All.Asmth[0];
}

static void fill_dens_params(ghip_dens_params *d)
{
  memset(d, 0, sizeof(*d));
  d->DesNumNgb = All.DesNumNgb;
  d->MaxNumNgbDeviation = All.MaxNumNgbDeviation;
  d->MinGasHsml = All.MinGasHsml;
  d->BoxSize = All.BoxSize;
  d->periodic = Cfg.periodic;
  d->Ti_Current = All.Ti_Current;
  d->Timebase_interval = All.Timebase_interval;
  d->MaxIter = 150; /* MAXITER */
}

static void fill_hydro_params(ghip_hydro_params *h, int raw)
{
  memset(h, 0, sizeof(*h));
  h->ArtBulkViscConst = All.ArtBulkViscConst;
  h->BoxSize = All.BoxSize;
  h->periodic = Cfg.periodic;
  h->ComovingIntegrationOn = All.ComovingIntegrationOn;
  if(All.ComovingIntegrationOn)
    {
      /* hydra.c:192-208 */
      double hubble_a = hubble_function(All.Time);
      h->hubble_a2 = All.Time * All.Time * hubble_a;
      h->fac_mu = pow(All.Time, 3 * (GAMMA - 1) / 2) / All.Time;
      h->fac_vsic_fix = hubble_a * pow(All.Time, 3 * GAMMA_MINUS1);
    }
  else
    h->hubble_a2 = h->fac_mu = h->fac_vsic_fix = 1.0;
  h->Timebase_interval = All.Timebase_interval;
  h->raw_dtentropy = raw;
}

/* "next" row N2: the host's own walks (potential.c, black-hole / dust neighbour loops) get the
 * device-built tree in their arrays, when the host has set them */
static int export_tree_to_host(void)
{
  if(!(Nodes_base && Extnodes_base && Nextnode && Father))
    return 0;
  ghip_node_layout nl;
  memset(&nl, 0xff, sizeof(nl));
  nl.node_stride = (int) sizeof(struct NODE);
  nl.n_len = (int) offsetof(struct NODE, len);
  nl.n_center = (int) offsetof(struct NODE, center);
  nl.n_s = (int) offsetof(struct NODE, u.d.s);
  nl.n_mass = (int) offsetof(struct NODE, u.d.mass);
  nl.n_bitflags = (int) offsetof(struct NODE, u.d.bitflags);
  nl.n_sibling = (int) offsetof(struct NODE, u.d.sibling);
  nl.n_nextnode = (int) offsetof(struct NODE, u.d.nextnode);
  nl.n_father = (int) offsetof(struct NODE, u.d.father);
  nl.n_ti_current = (int) offsetof(struct NODE, Ti_current);
  nl.ext_stride = (int) sizeof(struct extNODE);
  nl.e_dp = (int) offsetof(struct extNODE, dp);
  nl.e_vs = (int) offsetof(struct extNODE, vs);
  nl.e_vmax = (int) offsetof(struct extNODE, vmax);
  nl.e_divvmax = (int) offsetof(struct extNODE, divVmax);
  nl.e_hmax = (int) offsetof(struct extNODE, hmax);
  nl.e_ti_lastkicked = (int) offsetof(struct extNODE, Ti_lastkicked);
  nl.e_flag = (int) offsetof(struct extNODE, Flag);
  int nn = 0;
  if(chk(ghip_tree_export(Ctx, &nl, All.MaxPart, All.Ti_Current, Cfg.unequal_softenings, Nodes_base,
                          Extnodes_base, Nextnode, Father, MaxNodes, &nn),
         "ghip_tree_export"))
    return -1;
  Nodes = Nodes_base - All.MaxPart;       /* forcetree.c:4585-4590 */
  Extnodes = Extnodes_base - All.MaxPart;
  return 0;
}

/* forcetree.c:67-103: (re)build the tree over the current particles */
int force_treebuild(int npart, void *mp)
{
  (void) mp;
  if(need_ctx("force_treebuild"))
    return -1;
  double t0 = wallclock();
  if(!DeviceFresh)
    if(upload_particles(0))
      return -1;
  if(DomainLen <= 0)
    domain_findExtent();
  if(chk(ghip_tree_build(Ctx, DomainCorner, DomainCenter, DomainLen, All.ForceSoftening),
         "ghip_tree_build"))
    return -1;
  ghip_sync(Ctx);
  TreeOnDevice = 1;
  KeptTree = Cfg.dynamic_tree;   /* (ghip_tree_build kept a copy with vs / vmax per node) */
  KeptN = NumPart;
  KeptTi = All.Ti_Current;
  KickN = 0;
  ghip_stats st;
  ghip_get_stats(Ctx, &st);
  (void) npart;
  Numnodestree = st.tree_nodes;
  if(export_tree_to_host())
    return -1;
  CPU_Step_Treebuild += wallclock() - t0;
  return Numnodestree;
}

/* forcetree.c:4402-4527 */
void ewald_init(void)
{
  if(need_ctx("ewald_init"))
    return;
  chk(ghip_ewald_init(Ctx, All.BoxSize), "ghip_ewald_init");
}

static double drift_factor(int time0, int time1);

static int ensure_tree_split(int split)
{
  if(!DeviceFresh)
    {
      if(upload_particles(split))
        return -1;
    }
  if(TreeReconstructFlag || !TreeOnDevice)
    {
      /* gravtree.c:60-70.  cfg.dynamic_tree and no reconstruction asked for: the tree of the last
       * force_treebuild() goes to the current time (force_drift_node for every node, the kicks
       * recorded since folded in) and the walks read it */
      if(Cfg.dynamic_tree && !TreeReconstructFlag && KeptTree && KeptN == NumPart && NTask == 1 &&
         (!All.ComovingIntegrationOn || (DriftTab && KickLogMax > KickLogBegin)))
        {
          /* forcetree.c:1403-1425 */
          const double dt_drift = All.ComovingIntegrationOn ? drift_factor(KeptTi, All.Ti_Current)
                                                            : (All.Ti_Current - KeptTi) * All.Timebase_interval;
          if(chk(ghip_tree_substep(Ctx, dt_drift), "ghip_tree_substep"))
            return -1;
          KeptTi = All.Ti_Current;
          TreeOnDevice = 1;
          return 0;
        }
      if(force_treebuild(NumPart, NULL) < 0)
        return -1;
      TreeReconstructFlag = 0;
    }
  return 0;
}

/* forcetree.c:1455-1520: collected here, applied by force_finish_kick_nodes() */
void force_kick_node(int i, MyFloat *dv)
{
  if(!Cfg.dynamic_tree || !KeptTree || i < 0 || i >= NumPart)
    return;
  if(KickN >= KickCap)
    {
      const int nc = KickCap ? 2 * KickCap : 4096;
      int *ni = (int *) realloc(KickIdx, (size_t) nc * sizeof(int));
      if(ni)
        KickIdx = ni;
      double *nd = (double *) realloc(KickDv, (size_t) nc * 3 * sizeof(double));
      if(nd)
        KickDv = nd;
      double *nv = (double *) realloc(KickVmax, (size_t) nc * sizeof(double));
      if(nv)
        KickVmax = nv;
      if(!ni || !nd || !nv)
        {
          endrun(90003);
          return;
        }
      KickCap = nc;
    }
  KickIdx[KickN] = i;
  double vmax = 0;
  for(int j = 0; j < 3; j++)
    {
      KickDv[3 * (size_t) KickN + j] = dv[j];
      double v = fabs(PF64(i, Lay.p_vel)[j]);   /* :1478-1480: P[i].Vel already holds the new velocity */
      if(v > vmax)
        vmax = v;
    }
  KickVmax[KickN] = vmax;
  KickN++;
}

/* forcetree.c:1522-1651 (single rank: nothing to exchange) */
void force_finish_kick_nodes(void)
{
  if(!Cfg.dynamic_tree || !KeptTree || KickN == 0 || !Ctx)
    {
      KickN = 0;
      return;
    }
  chk(ghip_tree_kick_nodes_vmax(Ctx, KickN, KickIdx, KickDv, KickVmax), "ghip_tree_kick_nodes_vmax");
  KickN = 0;
}

static int ensure_tree(void)
{
  if(ensure_tree_split(0))
    return -1;
  return upload_gas_if_pending();   /* (a caller other than gravity_tree needs the gas data now) */
}

/* the tail of gravity_tree(): post-pass on the device, results into P[] (download != 0: by a
 * download of their own; 0: the caller downloads them together with its own results) */
static int gravity_complete(int download)
{
  GravPending = 0;
  /* gravtree.c:362-403: the comoving term of non-periodic builds without a PM mesh, then
   * OldAcc = |GravAccel (+ GravPM / G)|, then the multiplication by G */
  double comoving_fac = 0;
  if(!Cfg.periodic && !Cfg.pmgrid && All.ComovingIntegrationOn)
    comoving_fac = 0.5 * All.Hubble * All.Hubble * All.Omega0 / All.G;
  /* download == 2 (hydro_force with the walks in flight): post-pass and P[] block on the pair's own
   * stream, next to the tail of the hydro kernel */
  if(download == 2)
    {
      if(chk(ghip_gravity_to_records(Ctx, All.G, Cfg.pmgrid, comoving_fac, records_p(), &Lay),
             "ghip_gravity_to_records"))
        return -1;
    }
  else if(chk(ghip_gravity_finish_ex(Ctx, All.G, Cfg.pmgrid, comoving_fac, 0), "ghip_gravity_finish_ex"))
    return -1;
  if(All.TypeOfOpeningCriterion == 1)
    {
      All.ErrTolTheta = 0; /* gravtree.c:396-397 */
      all_push();
    }
  /* gravtree.c:470-483: vacuum energy in physical coordinates */
  if(!Cfg.periodic && !Cfg.pmgrid && All.ComovingIntegrationOn == 0)
    if(chk(ghip_gravity_vacuum_energy(Ctx, All.OmegaLambda * All.Hubble * All.Hubble),
           "ghip_gravity_vacuum_energy"))
      return -1;
  if(download == 1)
    {
      if(chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, 1, 0, 0), "ghip_download_aos"))
        return -1;
    }
  All.TotNumOfForces += GravPendingActive;
  all_push();
  return 0;
}

void gadget_force_flush(void)
{
  if(DensPending && Ctx)
    {
      DensPending = 0;
      chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, 0, 1, 0), "ghip_download_aos");
    }
  if(GravPending && Ctx)
    {
      all_pull();
      gravity_complete(1);
    }
}

/* gravtree.c:27-828 */
void gravity_tree(void)
{
  if(need_ctx("gravity_tree"))
    return;
  if(NTask > 1)
    {
      gravity_tree_ranks();
      return;
    }
  double t0 = wallclock();
  trace(NULL);
  if(GravPending && gravity_complete(1))   /* accel.c:63-64: the second pass of step 0 needs OldAcc */
    return;
  /* gravtree.c:55-56 */
  if(All.ComovingIntegrationOn)
    set_softenings();
  /* positions were drifted since the last force computation: refresh the device copy.  The
   * reference reuses a drifted tree on sub-steps (forcetree.c:1356-1452); this path rebuilds. */
  DeviceFresh = 0;
  TreeOnDevice = 0;
  int nact = collect_active(0);
  if(nact < 0)
    return;
  trace("gravity_tree: active list");
  /* overlap_sph: everything the walks read is in P[]; SphP[] follows once they are in flight */
  const int defer = Cfg.overlap_sph && Cfg.periodic && !Cfg.pmgrid && N_gas > 0 && nact == NumPart;
  if(ensure_tree_split(defer))
    return;
  trace("gravity_tree: upload + tree");
  if(chk(ghip_set_active(Ctx, nact == NumPart ? NULL : ActiveBuf, nact == NumPart ? 0 : nact),
         "ghip_set_active"))
    return;
  ghip_grav_params g;
  fill_grav_params(&g);
  /* gravtree.c:96-100, 130-168: PMGRID -> short-range walk; PERIODIC && !PMGRID -> a second,
   * Ewald-correction walk */
  int walk = Cfg.pmgrid ? GHIP_WALK_SHORTRANGE : GHIP_WALK_NEWTON;
  if(Cfg.periodic && !Cfg.pmgrid)
    walk = GHIP_WALK_NEWTON_EWALD; /* both passes in one call: the two walks share the device */
  if(chk(ghip_gravity(Ctx, &g, walk), "ghip_gravity"))
    return;
  trace("gravity_tree: walks launched");
  GravPendingActive = nact;
  /* overlap_sph: with gas to work on and everybody active (no list changes until hydro_force), the
   * walks stay in flight and the SPH phases run underneath them; the post-pass and the download
   * follow in hydro_force() / gadget_force_flush() */
  if(Cfg.overlap_sph && walk == GHIP_WALK_NEWTON_EWALD && N_gas > 0 && nact == NumPart)
    {
      if(upload_gas_if_pending())
        return;
      GravPending = 1;
      Phase = 1;
      CPU_Step_Treewalk += wallclock() - t0;
      return;
    }
  if(upload_gas_if_pending())
    return;
  if(gravity_complete(1))
    return;
  Phase = 1;
  CPU_Step_Treewalk += wallclock() - t0;
}

/* density.c:1030-1052: gas always; sinks under BLACK_HOLES, dust grains under DUST */
int density_isactive(int n)
{
  if(p_timebin(n) < 0)
    return 0;
  if(Cfg.black_holes && p_type(n) == 5)
    return 1;
  if(Cfg.dust && p_type(n) == 2)
    return 1;
  if(p_type(n) == 0)
    return 1;
  return 0;
}

/* density() for the active Type-5 (BLACK_HOLES) and Type-2 (DUST) targets -- density.c:125, 176, 393
 * loop over density_isactive(): the h iteration against the gas tree for DesNumNgb *
 * BlackHoleNgbFactor (sinks, density.c:549-550) or DesNumNgb (dust grains, :555-556) neighbours
 * without the Newton step (:613, 629), and the kernel-weighted density, entropy and gas velocity
 * around the target (:358-377, 522-545).  Their smoothing lengths live in P[] (PPP == P in such a
 * build, allvars.h:266-270): with the minimal records of this header there is nowhere to put them,
 * and the call is refused loudly rather than leaving PPP[].Hsml stale. */
static int dd_collective(int op, const void *params, int walk, const char *what);

static int density_of_sinks(const ghip_dens_params *d)
{
  if(!Cfg.black_holes && !Cfg.dust)
    return 0;
  for(int pass = 0; pass < 2; pass++)
    {
      const int type = pass == 0 ? 5 : 2;
      if((type == 5 && !Cfg.black_holes) || (type == 2 && !Cfg.dust))
        continue;
      int n = 0;
      for(int i = FirstActiveParticle; i >= 0; i = NextActiveParticle[i])
        if(p_type(i) == type && density_isactive(i))
          n++;
      /* on ranks the pass is a collective (every rank evaluates ALL ranks' sinks against its own gas,
       * ghip_dd_sink_args): entered by everybody, with or without sinks of its own */
      if(n == 0 && NTask == 1)
        continue;
      if(Lay.p_hsml < 0 && NTask > 1 && n == 0)
        {
          /* (a rank without such targets cannot know whether another one has them: with records that
           * cannot hold the result nobody may have them -- the ranks that do stop below, this one
           * must not wait in a collective for them) */
          continue;
        }
      if(Lay.p_hsml < 0)
        {
          snprintf(ErrBuf, sizeof(ErrBuf),
                   "density: %d active particles of type %d are density targets in this build "
                   "(BLACK_HOLES / DUST), but the bound records keep Hsml in SphP[]: bind the host's "
                   "records with gadget_force_bind_records()", n, type);
          fprintf(stderr, "gadget_force: %s\n", ErrBuf);
          endrun(90007);
          return -1;
        }
      int *idx = (int *) malloc((size_t) (n + 1) * sizeof(int));
      double *buf = (double *) malloc((size_t) (n + 1) * 7 * sizeof(double));
      unsigned int *ids = (unsigned int *) malloc((size_t) (n + 1) * sizeof(unsigned int));
      if(!idx || !buf || !ids)
        {
          free(idx);
          free(buf);
          free(ids);
          endrun(90003);
          return -1;
        }
      const size_t m = (size_t) n + 1;
      double *hs = buf, *nn = buf + m, *rho = buf + 2 * m, *ent = buf + 3 * m, *vel = buf + 4 * m;
      int k = 0;
      for(int i = FirstActiveParticle; i >= 0; i = NextActiveParticle[i])
        if(p_type(i) == type && density_isactive(i))
          {
            idx[k] = i;
            hs[k] = *ppp_hsml(i);
            ids[k] = BhLay.p_id >= 0 ? *(unsigned int *) (prec(i) + BhLay.p_id) : (unsigned int) i;
            k++;
          }
      int iter = 0;
      int rc;
      if(NTask > 1)
        {
          ghip_dd_sink_args a;
          memset(&a, 0, sizeof(a));
          a.dens = d;
          a.ngb_factor = type == 5 ? All.BlackHoleNgbFactor : 1.0;
          a.nsink = n;
          a.sink_idx = idx;
          a.sink_id = ids;
          a.hsml = hs;
          a.numngb = nn;
          a.bh_density = rho;
          a.bh_entropy = ent;
          a.bh_gasvel = vel;
          rc = dd_collective(GHIP_DD_SINK_DENSITY, &a, 0, "density of sinks (ranks)") ? GHIP_EDEVICE : GHIP_OK;
          if(rc != GHIP_OK)
            {
              free(idx);
              free(buf);
              free(ids);
              return -1;
            }
        }
      else
        rc = ghip_sink_density(Ctx, d, type == 5 ? All.BlackHoleNgbFactor : 1.0, n, idx, hs, nn, rho, ent,
                               vel, &iter);
      if(rc == GHIP_OK)
        {
          const int o_rho = type == 5 ? BhLay.p_bh_density : BhLay.p_dust_density;
          const int o_ent = type == 5 ? BhLay.p_bh_entropy : BhLay.p_dust_entropy;
          const int o_vel = type == 5 ? BhLay.p_bh_gasvel : BhLay.p_dust_gasvel;
          for(k = 0; k < n; k++)
            {
              const int i = idx[k];
              *ppp_hsml(i) = hs[k];
              *ppp_numngb(i) = nn[k];
              if(o_rho >= 0)
                *PF64(i, o_rho) = rho[k];
              if(o_ent >= 0)
                *PF64(i, o_ent) = ent[k];
              if(o_vel >= 0)
                for(int c = 0; c < 3; c++)
                  PF64(i, o_vel)[c] = vel[3 * (size_t) k + c];
            }
        }
      free(idx);
      free(buf);
      free(ids);
      if(chk(rc, "ghip_sink_density"))
        return -1;
    }
  return 0;
}

/* every gas particle is an active target: "everybody" selects the same gas targets as the list */
static int gravity_was_full_and_gas_is(int nact_gas)
{
  return N_gas > 0 && nact_gas == N_gas;
}

/* density.c:89-704 */
void density(void)
{
  if(need_ctx("density"))
    return;
  if(NTask > 1)
    {
      density_ranks();
      return;
    }
  double t0 = wallclock();
  /* directly after gravity_tree() (accel.c:61-84) the device copy is current; a stand-alone
   * call (init.c:791) re-reads P/SphP */
  trace(NULL);
  if(Phase != 1)
    DeviceFresh = 0;
  if(ensure_tree())
    return;
  trace("density: gas upload");
  int nact = collect_active(1);
  if(nact < 0)
    return;
  trace("density: active list");
  /* an empty list must not mean "all": give the device a real (possibly empty) list.  Everybody
   * active stays "everybody" (no list: a gravity pair in flight is not disturbed). */
  if(gravity_was_full_and_gas_is(nact))
    {
      if(chk(ghip_set_active(Ctx, NULL, 0), "ghip_set_active"))
        return;
    }
  else if(chk(ghip_set_active(Ctx, ActiveBuf ? ActiveBuf : &nact, nact), "ghip_set_active"))
    return;
  ghip_dens_params d;
  fill_dens_params(&d);
  if(chk(ghip_density(Ctx, &d), "ghip_density"))
    return;
  trace("density: ghip_density");
  /* overlap_sph with the walks in flight: nothing reads the density results between density() and
   * hydro_force() (accel.c:84-106: force_update_hmax is ours), so they travel with the hydro results --
   * one download of the SphP[] block per step instead of two */
  if(Cfg.overlap_sph && GravPending)
    DensPending = 1;
  else if(chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, 0, 1, 0), "ghip_download_aos"))
    return;
  if(density_of_sinks(&d))
    return;
  Phase = 2;
  CPU_Step_Density += wallclock() - t0;
}

/* forcetree.c:1661-1786 */
void force_update_hmax(void)
{
  if(need_ctx("force_update_hmax"))
    return;
  double t0 = wallclock();
  if(chk(ghip_update_hmax(Ctx), "ghip_update_hmax"))
    return;
  /* the reference refreshes Extnodes[].hmax / divVmax in place (forcetree.c:1661-1786): a host that
   * holds the exported tree gets it again with the new smoothing lengths and divergences */
  if(TreeOnDevice && NTask == 1)
    export_tree_to_host();
  CPU_Step_Hmaxupdate += wallclock() - t0;
}

/* hydra.c:145-813 */
void hydro_force(void)
{
  if(need_ctx("hydro_force"))
    return;
  if(NTask > 1)
    {
      hydro_force_ranks();
      return;
    }
  double t0 = wallclock();
  trace(NULL);
  if(Phase != 2)
    DeviceFresh = 0;
  if(ensure_tree())
    return;
  int nact = collect_active(1);
  if(nact < 0)
    return;
  trace("hydro_force: active list");
  if(gravity_was_full_and_gas_is(nact))
    {
      if(chk(ghip_set_active(Ctx, NULL, 0), "ghip_set_active"))
        return;
    }
  else if(chk(ghip_set_active(Ctx, ActiveBuf ? ActiveBuf : &nact, nact), "ghip_set_active"))
    return;
  ghip_hydro_params h;
  fill_hydro_params(&h, 0);
  if(chk(ghip_hydro(Ctx, &h), "ghip_hydro"))
    return;
  trace("hydro_force: ghip_hydro");
  /* overlap_sph: the gravity walks have been running underneath.  The SphP[] block is queued behind
   * the hydro kernel on the main stream; the walks' post-pass and the P[] block go on the pair's own
   * stream, so that the P[] block crosses the link while hydro's tail still runs.  (Records with Hsml
   * in P[] carry density results in the P[] block too: one download of everything then.) */
  const int with_gravity = GravPending;
  const int with_density = DensPending;
  DensPending = 0;
  if(with_gravity && Lay.p_hsml < 0 && !(Cfg.periodic == 0 && Cfg.pmgrid == 0))
    {
      if(chk(ghip_download_aos_async(Ctx, records_p(), records_s(), &Lay, 0, with_density, 1),
             "ghip_download_aos_async"))
        return;
      if(gravity_complete(2))
        return;
      trace("hydro_force: P download");
      if(chk(ghip_sync(Ctx), "ghip_sync"))
        return;
      trace("hydro_force: SphP download");
    }
  else
    {
      if(with_gravity && gravity_complete(0))
        return;
      trace("hydro_force: walks joined");
      if(chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, with_gravity, with_density, 1),
             "ghip_download_aos"))
        return;
      trace("hydro_force: download");
    }
  Phase = 0;
  CPU_Step_Hydro += wallclock() - t0;
}

/* ------------------------------------------------------------------------------------------
 * "next" row N1: timestep criterion + kick
 * ---------------------------------------------------------------------------------------- */
/* driftfac.c:123-163 get_drift_factor on the host's DriftTable (cfg.dynamic_tree in comoving runs:
 * force_drift_node's dt_drift, forcetree.c:1403-1412).  DRIFT_TABLE_LENGTH = 1000 (driftfac.c:12). */
void gadget_force_set_drift_table(const double *drifttable)
{
  DriftTab = drifttable;
}

static double drift_factor(int time0, int time1)
{
  const int len = 1000;
  double u[2], df[2];
  const int t[2] = { time0, time1 };
  for(int k = 0; k < 2; k++)
    {
      double a = KickLogBegin + t[k] * All.Timebase_interval;
      u[k] = (a - KickLogBegin) / (KickLogMax - KickLogBegin) * len;
      int i = (int) u[k];
      if(i >= len)
        i = len - 1;
      if(i <= 1)
        df[k] = u[k] * DriftTab[0];
      else
        df[k] = DriftTab[i - 1] + (DriftTab[i] - DriftTab[i - 1]) * (u[k] - i);
    }
  return df[1] - df[0];
}

void gadget_force_set_kick_tables(const double *gravkick, const double *hydrokick,
                                  double logTimeBegin, double logTimeMax)
{
  KickTabGrav = gravkick;
  KickTabHydro = hydrokick;
  KickLogBegin = logTimeBegin;
  KickLogMax = logTimeMax;
}

/* timestep.c:1226-1246 */
int get_timestep_bin(int ti_step)
{
  int bin = -1;
  if(ti_step == 0)
    return 0;
  if(ti_step == 1)
    {
      printf("time-step of integer size 1 not allowed\n");
      endrun(112313);
    }
  while(ti_step)
    {
      bin++;
      ti_step >>= 1;
    }
  return bin;
}

/* timestep.c:1125-1224: the per-type sums come from the device (ghip_velocity_moments replaces the
 * particle loop and the MPI_Allreduce of a single-rank run), the rest is the reference's host math */
void find_dt_displacement_constraint(double hfac)
{
  dt_displacement = All.MaxSizeTimestep;
  if(!All.ComovingIntegrationOn)
    return;
  if(need_ctx("find_dt_displacement_constraint"))
    return;
  if(!DeviceFresh && upload_particles(0))
    return;
  double v_sum[6], min_mass[6];
  long long count_sum[6];
  if(chk(ghip_velocity_moments(Ctx, v_sum, min_mass, count_sum), "ghip_velocity_moments"))
    return;
  for(int type = 0; type < 6; type++)
    if(count_sum[type] > 0)
      {
        double dmean, dt;
        if(type == 0 || (type == 4 && All.StarformationOn))
          dmean = pow(min_mass[type] / (All.OmegaBaryon * 3 * All.Hubble * All.Hubble / (8 * M_PI * All.G)),
                      1.0 / 3);
        else
          dmean = pow(min_mass[type] / ((All.Omega0 - All.OmegaBaryon) * 3 * All.Hubble * All.Hubble /
                                        (8 * M_PI * All.G)),
                      1.0 / 3);
        dt = All.MaxRMSDisplacementFac * hfac * dmean / sqrt(v_sum[type] / count_sum[type]);
        if(dt < dt_displacement)
          dt_displacement = dt;
      }
}

/* rebuild the linked lists of the time bins from P[].TimeBin, in index order
 * (reconstruct_timebins(), timestep.c / domain.c) */
static void rebuild_timebin_lists(void)
{
  for(int b = 0; b < TIMEBINS; b++)
    {
      TimeBinCount[b] = TimeBinCountSph[b] = 0;
      FirstInTimeBin[b] = LastInTimeBin[b] = -1;
    }
  for(int i = 0; i < NumPart; i++)
    {
      int bin = p_timebin(i);
      if(NextInTimeBin && PrevInTimeBin)
        {
          if(TimeBinCount[bin] > 0)
            {
              PrevInTimeBin[i] = LastInTimeBin[bin];
              NextInTimeBin[i] = -1;
              NextInTimeBin[LastInTimeBin[bin]] = i;
              LastInTimeBin[bin] = i;
            }
          else
            {
              FirstInTimeBin[bin] = LastInTimeBin[bin] = i;
              PrevInTimeBin[i] = NextInTimeBin[i] = -1;
            }
        }
      TimeBinCount[bin]++;
      if(p_type(i) == 0)
        TimeBinCountSph[bin]++;
    }
}

/* timestep.c:29-362 for the minimal flag set (no PMGRID long-range kick, no MAKEGLASS); the
 * particle loop :142-260 with get_timestep and do_the_kick runs on the device */
void advance_and_find_timesteps(void)
{
  if(need_ctx("advance_and_find_timesteps"))
    return;
  gadget_force_flush();   /* (overlap_sph without a hydro_force() call: the kick needs G * GravAccel) */
  if(All.TypeOfTimestepCriterion != 0)
    {
      endrun(888); /* timestep.c:726 */
      return;
    }
  double hubble_a = 1, atime = 1;
  if(All.ComovingIntegrationOn)
    {
      hubble_a = hubble_function(All.Time);
      atime = All.Time;
      if(!KickTabGrav || !KickTabHydro)
        {
          snprintf(ErrBuf, sizeof(ErrBuf),
                   "advance_and_find_timesteps: comoving kicks need gadget_force_set_kick_tables()");
          fprintf(stderr, "gadget_force: %s\n", ErrBuf);
          endrun(90002);
          return;
        }
    }
  /* the accelerations of this step are on the device (Phase 0 after hydro_force); velocities and
   * entropies are the host's: make the device image current */
  if(!DeviceFresh && upload_particles(0))
    return;
  if(Flag_FullStep || dt_displacement == 0)
    find_dt_displacement_constraint(hubble_a * atime * atime);
  int nact = collect_active(0);
  if(nact < 0)
    return;
  if(chk(ghip_set_active(Ctx, nact == NumPart ? NULL : ActiveBuf, nact == NumPart ? 0 : nact),
         "ghip_set_active"))
    return;
  ghip_kick_params k;
  memset(&k, 0, sizeof(k));
  k.Ti_Current = All.Ti_Current;
  k.Timebase_interval = All.Timebase_interval;
  k.ComovingIntegrationOn = All.ComovingIntegrationOn;
  k.Time = All.Time;
  k.hubble_a = hubble_a;
  k.ErrTolIntAccuracy = All.ErrTolIntAccuracy;
  k.CourantFac = All.CourantFac;
  k.MaxSizeTimestep = All.MaxSizeTimestep;
  k.MinSizeTimestep = All.MinSizeTimestep;
  k.dt_displacement = dt_displacement;
  for(int t = 0; t < 6; t++)
    k.SofteningTable[t] = All.SofteningTable[t];
  k.MinEgySpec = All.MinEgySpec;
  for(int b = 0; b < TIMEBINS; b++)
    if(TimeBinActive[b])
      k.TimeBinActive |= 1u << b;
  k.logTimeBegin = KickLogBegin;
  k.logTimeMax = KickLogMax;
  k.GravKickTable = KickTabGrav;
  k.HydroKickTable = KickTabHydro;
  long long cnt[32], sph[32];
  int rc = ghip_advance_timesteps(Ctx, &k, cnt, sph);
  if(rc == GHIP_ETIMESTEP)
    {
      endrun(ghip_timestep_endrun_code(Ctx)); /* the reference's own code: 888, 818, 112313 */
      return;
    }
  if(chk(rc, "ghip_advance_timesteps"))
    return;
  if(chk(ghip_download_aos_kick(Ctx, records_p(), records_s(), &Lay), "ghip_download_aos_kick"))
    return;
  rebuild_timebin_lists();
}

/* ------------------------------------------------------------------------------------------
 * per-target surface: a batch of one on the device
 * ---------------------------------------------------------------------------------------- */
static int treeevaluate_one(int target, int mode, int *nexport, int walk)
{
  if(need_ctx("force_treeevaluate"))
    return -1;
  if(ensure_tree())
    return -1;
  ghip_grav_params g;
  fill_grav_params(&g);
  if(mode == 0)
    {
      if(target < 0 || target >= NumPart)
        {
          endrun(90002);
          return -1;
        }
      /* results of the device's previous state for this target are needed for the "+=" of the
       * Ewald walk (forcetree.c:3190-3193) */
      static double *acc3 = NULL;
      static int *cost = NULL;
      static int cap = 0;
      if(cap < NumPart)
        {
          acc3 = (double *) realloc(acc3, (size_t) NumPart * 3 * sizeof(double));
          cost = (int *) realloc(cost, (size_t) NumPart * sizeof(int));
          cap = NumPart;
        }
      if(walk == GHIP_WALK_EWALD)
        {
          /* seed the device result with the host's current partial sum */
          if(chk(ghip_get_field(Ctx, GHIP_F_GRAVACCEL, acc3), "ghip_get_field") ||
             chk(ghip_get_field(Ctx, GHIP_F_GRAVCOST, cost), "ghip_get_field"))
            return -1;
          for(int k = 0; k < 3; k++)
            acc3[3 * (size_t) target + k] = PF64(target, Lay.p_gravaccel)[k];
          cost[target] = (int) *(float *) (prec(target) + Lay.p_gravcost);
          if(chk(ghip_set_field(Ctx, GHIP_F_GRAVACCEL, acc3), "ghip_set_field") ||
             chk(ghip_set_field(Ctx, GHIP_F_GRAVCOST, cost), "ghip_set_field"))
            return -1;
        }
      if(chk(ghip_set_active(Ctx, &target, 1), "ghip_set_active"))
        return -1;
      if(chk(ghip_gravity(Ctx, &g, walk), "ghip_gravity"))
        return -1;
      if(chk(ghip_get_field(Ctx, GHIP_F_GRAVACCEL, acc3), "ghip_get_field") ||
         chk(ghip_get_field(Ctx, GHIP_F_GRAVCOST, cost), "ghip_get_field"))
        return -1;
      /* forcetree.c:2277-2293 */
      int before = (int) *(float *) (prec(target) + Lay.p_gravcost);
      for(int k = 0; k < 3; k++)
        PF64(target, Lay.p_gravaccel)[k] = acc3[3 * (size_t) target + k];
      *(float *) (prec(target) + Lay.p_gravcost) = (float) cost[target];
      return (walk == GHIP_WALK_EWALD) ? cost[target] - before : cost[target];
    }
  /* mode 1: an imported target, GravDataGet[target] -> GravDataResult[target]
   * (forcetree.c:1862-1880, 2296-2314).  Single rank: the whole local tree is walked. */
  if(!GravDataGet || !GravDataResult)
    {
      endrun(90002);
      return -1;
    }
  double pos[3] = { GravDataGet[target].Pos[0], GravDataGet[target].Pos[1],
    GravDataGet[target].Pos[2] };
  int type = Cfg.unequal_softenings ? GravDataGet[target].Type : p_type(0); /* forcetree.c:1868-1872 */
  double oldacc = GravDataGet[target].OldAcc, acc[3];
  int nint = 0;
  if(chk(ghip_gravity_ext(Ctx, &g, walk, 1, pos, &type, &oldacc, acc, &nint), "ghip_gravity_ext"))
    return -1;
  for(int k = 0; k < 3; k++)
    GravDataResult[target].Acc[k] = acc[k];
  GravDataResult[target].Ninteractions = nint;
  if(nexport)
    *nexport = 1; /* nodesinlist */
  return nint;
}

int force_treeevaluate(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nsend_local;
  return treeevaluate_one(target, mode, nexport, GHIP_WALK_NEWTON);
}

int force_treeevaluate_shortrange(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nsend_local;
  return treeevaluate_one(target, mode, nexport, GHIP_WALK_SHORTRANGE);
}

int force_treeevaluate_ewald_correction(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nsend_local;
  int c = treeevaluate_one(target, mode, nexport, GHIP_WALK_EWALD);
  return c < 0 ? c : 0;
}

/* density.c:711-1029, mode 0: raw sums into the d-unions (density.c:942-989) */
int density_evaluate(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nexport;
  (void) nsend_local;
  if(need_ctx("density_evaluate"))
    return -1;
  if(mode != 0 || target < 0 || target >= N_gas)
    {
      endrun(90002);
      return -1;
    }
  if(ensure_tree())
    return -1;
  ghip_dens_params d;
  fill_dens_params(&d);
  double out7[7];
  if(chk(ghip_density_evaluate(Ctx, &d, target, *ppp_hsml(target), out7), "ghip_density_evaluate"))
    return -1;
  *SF64(target, Lay.s_density) = out7[0];
  *ppp_numngb(target) = out7[1];
  *SF64(target, Lay.s_dhsmlfac) = out7[2];
  *SF64(target, Lay.s_divvel) = out7[3];
  SF64(target, Lay.s_curlvel)[0] = out7[4];   /* r.dRot[3] shares the union with CurlVel */
  SF64(target, Lay.s_curlvel)[1] = out7[5];
  SF64(target, Lay.s_curlvel)[2] = out7[6];
  return 0;
}

/* hydra.c:822-1995, mode 0: raw sums (hydra.c:1931-1940) */
int hydro_evaluate(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nexport;
  (void) nsend_local;
  if(need_ctx("hydro_evaluate"))
    return -1;
  if(mode != 0 || target < 0 || target >= N_gas)
    {
      endrun(90002);
      return -1;
    }
  if(ensure_tree())
    return -1;
  if(chk(ghip_set_active(Ctx, &target, 1), "ghip_set_active"))
    return -1;
  ghip_hydro_params h;
  fill_hydro_params(&h, 1);
  if(chk(ghip_hydro(Ctx, &h), "ghip_hydro"))
    return -1;
  static double *buf = NULL;
  static int cap = 0;
  if(cap < N_gas)
    {
      buf = (double *) realloc(buf, (size_t) N_gas * 3 * sizeof(double));
      cap = N_gas;
    }
  if(chk(ghip_get_field(Ctx, GHIP_F_HYDROACCEL, buf), "ghip_get_field"))
    return -1;
  for(int k = 0; k < 3; k++)
    SF64(target, Lay.s_hydroaccel)[k] = buf[3 * (size_t) target + k];
  if(chk(ghip_get_field(Ctx, GHIP_F_DTENTROPY, buf), "ghip_get_field"))
    return -1;
  *SF64(target, Lay.s_dtentropy) = buf[target];
  if(chk(ghip_get_field(Ctx, GHIP_F_MAXSIGNALVEL, buf), "ghip_get_field"))
    return -1;
  *SF64(target, Lay.s_maxsignalvel) = buf[target];
  return 0;
}

static int ngb_find(MyDouble c[3], MyFloat hsml, int *startnode, int mode, int pairs)
{
  if(need_ctx("ngb_treefind"))
    return -1;
  if(mode != 0)
    {
      endrun(23131); /* ngb.c:92: pseudo-particle handling is not available in mode 1 */
      return -1;
    }
  if(ensure_tree())
    return -1;
  if(NgblistCap < NumPart)
    {
      /* density.c:143 / hydra.c:230: Ngblist holds up to NumPart indices */
      Ngblist = (int *) realloc(Ngblist, (size_t) (NumPart > 0 ? NumPart : 1) * sizeof(int));
      NgblistCap = NumPart;
    }
  int nfound = 0;
  if(chk(ghip_ngb_treefind(Ctx, c, hsml, pairs, Cfg.periodic, All.BoxSize, Ngblist, NgblistCap,
                           &nfound), "ghip_ngb_treefind"))
    return -1;
  if(startnode)
    *startnode = -1;
  return nfound;
}

/* ngb.c:169-297 */
int ngb_treefind_variable(MyDouble searchcenter[3], MyFloat hsml, int target, int *startnode,
                          int mode, int *nexport, int *nsend_local)
{
  (void) target;
  (void) nexport;
  (void) nsend_local;
  return ngb_find(searchcenter, hsml, startnode, mode, 0);
}

/* ngb.c:32-160 */
int ngb_treefind_pairs(MyDouble searchcenter[3], MyFloat hsml, int target, int *startnode,
                       int mode, int *nexport, int *nsend_local)
{
  (void) target;
  (void) nexport;
  (void) nsend_local;
  return ngb_find(searchcenter, hsml, startnode, mode, 1);
}

/* ------------------------------------------------------------------------------------------
 * "next" row N4: the black-hole neighbour passes (blackhole.c) on bound records
 * ---------------------------------------------------------------------------------------- */
static void fill_bh_params(ghip_bh_params *b)
{
  memset(b, 0, sizeof(*b));
  double hubble_a = 1, ascale = 1;
  if(All.ComovingIntegrationOn)   /* blackhole.c:89-95 */
    {
      ascale = All.Time;
      hubble_a = hubble_function(All.Time);
    }
  b->BoxSize = All.BoxSize;
  b->periodic = Cfg.periodic;
  b->ascale = ascale;
  b->dt_fac = All.Timebase_interval / hubble_a;   /* blackhole.c:822 */
  b->SMBHmass = All.SMBHmass;
  b->InnerBoundary = All.InnerBoundary;
  b->SinkBoundary = All.SinkBoundary;
  b->SofteningBndry = All.SofteningBndry;
  /* blackhole.c:1099 */
  b->CritDensity = All.CritOverDensity * All.UnitLength_in_cm * All.UnitLength_in_cm * All.UnitLength_in_cm /
                   All.UnitMass_in_g;
  /* blackhole.c:1138-1139 */
  b->FeedbackCoeff = All.BlackHoleFeedbackFactor * 6.67e-8 * pow(4. * 3.1415 / 3. * 5., 0.3333) /
                     All.UnitEnergy_in_cgs;
  b->UnitMass_in_g = All.UnitMass_in_g;
  b->dust = Cfg.dust;
  b->accretion_of_dust_only = Cfg.accretion_of_dust_only;
  b->accretion_density = Cfg.accretion_density;
}

static int bh_ready(const char *who)
{
  if(need_ctx(who))
    return -1;
  if(!Cfg.black_holes || BhLay.p_id < 0 || BhLay.p_swallowid < 0 || BhLay.p_bh_mass < 0 ||
     BhLay.p_bh_mdot < 0 || BhLay.p_bh_density < 0 || Lay.p_hsml < 0)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "%s: needs a BLACK_HOLES configuration and records bound with "
               "their black-hole members (gadget_force_bind_records)", who);
      fprintf(stderr, "gadget_force: %s\n", ErrBuf);
      endrun(90002);
      return -1;
    }
  if(NTask > 1)
    {
      /* the passes are collectives on the trees gravity_tree() and density() of THIS step left on the
       * device (blackhole_accretion follows them in compute_accelerations / run.c) */
      if(!DeviceFresh || !DdReady)
        {
          snprintf(ErrBuf, sizeof(ErrBuf), "%s on %d ranks must follow gravity_tree() and density() of the "
                   "same step", who, NTask);
          fprintf(stderr, "gadget_force: %s\n", ErrBuf);
          endrun(90002);
          return -1;
        }
      return 0;
    }
  return ensure_tree();
}

/* device marks <-> records: P[].SwallowID, SphP[].i.Injected_BH_Energy */
static int bh_marks_to_device(void)
{
  unsigned int *sw = (unsigned int *) malloc((size_t) (NumPart > 0 ? NumPart : 1) * sizeof(unsigned int));
  double *inj = (double *) malloc((size_t) (N_gas > 0 ? N_gas : 1) * sizeof(double));
  if(!sw || !inj)
    {
      free(sw);
      free(inj);
      endrun(90003);
      return -1;
    }
  for(int i = 0; i < NumPart; i++)
    sw[i] = *(unsigned int *) (prec(i) + BhLay.p_swallowid);
  for(int i = 0; i < N_gas; i++)
    inj[i] = BhLay.s_injected_bh_energy >= 0 ? *SF64(i, BhLay.s_injected_bh_energy) : 0.0;
  int rc = ghip_sink_set_marks(Ctx, sw, inj);
  free(sw);
  free(inj);
  return chk(rc, "ghip_sink_set_marks") ? -1 : 0;
}

static int bh_marks_to_records(int with_mass)
{
  unsigned int *sw = (unsigned int *) malloc((size_t) (NumPart > 0 ? NumPart : 1) * sizeof(unsigned int));
  double *inj = (double *) malloc((size_t) (N_gas > 0 ? N_gas : 1) * sizeof(double));
  double *mass = with_mass ? (double *) malloc((size_t) (NumPart > 0 ? NumPart : 1) * sizeof(double)) : NULL;
  int rc = (sw && inj && (mass || !with_mass)) ? ghip_sink_get_marks(Ctx, sw, inj) : GHIP_ENOMEM;
  if(rc == GHIP_OK && with_mass)
    rc = ghip_get_field(Ctx, GHIP_F_MASS, mass);
  if(rc == GHIP_OK)
    {
      for(int i = 0; i < NumPart; i++)
        {
          *(unsigned int *) (prec(i) + BhLay.p_swallowid) = sw[i];
          if(with_mass)
            *PF64(i, Lay.p_mass) = mass[i];   /* blackhole.c:1290, 1312, 1330: a victim's mass becomes 0 */
        }
      if(BhLay.s_injected_bh_energy >= 0)
        for(int i = 0; i < N_gas; i++)
          *SF64(i, BhLay.s_injected_bh_energy) = inj[i];
    }
  free(sw);
  free(inj);
  free(mass);
  return chk(rc, "ghip_sink_get_marks") ? -1 : 0;
}

/* blackhole_evaluate / _swallow of the sinks idx[0, n): results into the records */
static int bh_evaluate_batch(int n, const int *idx)
{
  unsigned int *id = (unsigned int *) malloc((size_t) (n + 1) * sizeof(unsigned int));
  double *md = (double *) malloc((size_t) (n + 1) * 2 * sizeof(double));
  if(!id || !md)
    {
      free(id);
      free(md);
      endrun(90003);
      return -1;
    }
  double *rho = md + n;
  for(int k = 0; k < n; k++)
    {
      id[k] = *(unsigned int *) (prec(idx[k]) + BhLay.p_id);
      md[k] = *PF64(idx[k], BhLay.p_bh_mdot);
      rho[k] = *PF64(idx[k], BhLay.p_bh_density);
    }
  ghip_bh_params b;
  fill_bh_params(&b);
  int rc;
  if(NTask > 1)
    {
      /* every rank's sinks against every rank's particles (blackhole.c:310-470's export, turned round) */
      ghip_dd_sink_args a;
      memset(&a, 0, sizeof(a));
      a.bh = &b;
      a.nsink = n;
      a.sink_idx = idx;
      a.sink_id = id;
      a.bh_mdot = md;
      a.bh_density_in = rho;
      rc = dd_collective(GHIP_DD_BH_EVALUATE, &a, 0, "blackhole_evaluate (ranks)") ? GHIP_EDEVICE : GHIP_OK;
      free(id);
      free(md);
      return rc == GHIP_OK ? 0 : -1;
    }
  rc = ghip_blackhole_evaluate(Ctx, &b, n, idx, id, md, rho);
  free(id);
  free(md);
  return chk(rc, "ghip_blackhole_evaluate") ? -1 : 0;
}

static int bh_swallow_batch(int n, const int *idx)
{
  unsigned int *id = (unsigned int *) malloc((size_t) (n + 1) * sizeof(unsigned int));
  double *buf = (double *) malloc((size_t) (n + 1) * 7 * sizeof(double));
  if(!id || !buf)
    {
      free(id);
      free(buf);
      endrun(90003);
      return -1;
    }
  const size_t m = (size_t) n + 1;
  double *bhm = buf, *am = buf + m, *ab = buf + 2 * m, *ad = buf + 3 * m, *mom = buf + 4 * m;
  for(int k = 0; k < n; k++)
    {
      id[k] = *(unsigned int *) (prec(idx[k]) + BhLay.p_id);
      bhm[k] = *PF64(idx[k], BhLay.p_bh_mass);
    }
  ghip_bh_params b;
  fill_bh_params(&b);
  long long counts[3] = { 0, 0, 0 };
  int rc;
  if(NTask > 1)
    {
      ghip_dd_sink_args a;
      memset(&a, 0, sizeof(a));
      a.bh = &b;
      a.nsink = n;
      a.sink_idx = idx;
      a.sink_id = id;
      a.sink_bh_mass = bhm;
      a.acc_mass = am;
      a.acc_bhmass = ab;
      a.acc_dustmass = ad;
      a.acc_momentum = mom;
      a.counts = counts;   /* victims swallowed ON THIS RANK, as the reference counts them (:1296-1309) */
      rc = dd_collective(GHIP_DD_BH_SWALLOW, &a, 0, "blackhole_evaluate_swallow (ranks)") ? GHIP_EDEVICE : GHIP_OK;
      if(rc != GHIP_OK)
        {
          free(id);
          free(buf);
          return -1;
        }
    }
  else
    rc = ghip_blackhole_swallow(Ctx, &b, n, idx, id, bhm, am, ab, ad, mom, counts);
  if(rc == GHIP_OK)
    {
      /* blackhole.c:1326-1333 (mode 0): the sums are ASSIGNED, in this order -- in the reference's
       * struct BH_accreted_BHMass and BH_accreted_DustMass share the union b5 (allvars.h:1261-1267),
       * so with both offsets equal the dust mass is what the record holds afterwards, as there */
      for(int k = 0; k < n; k++)
        {
          const int i = idx[k];
          *PF64(i, BhLay.p_bh_mass) = bhm[k];   /* (0 for a sink that was itself swallowed, :1312) */
          if(BhLay.p_bh_accreted_mass >= 0)
            *PF64(i, BhLay.p_bh_accreted_mass) = am[k];
          if(BhLay.p_bh_accreted_bhmass >= 0)
            *PF64(i, BhLay.p_bh_accreted_bhmass) = ab[k];
          if(BhLay.p_bh_accreted_dustmass >= 0 && Cfg.dust)
            *PF64(i, BhLay.p_bh_accreted_dustmass) = ad[k];
          if(BhLay.p_bh_accreted_momentum >= 0)
            for(int c = 0; c < 3; c++)
              PF64(i, BhLay.p_bh_accreted_momentum)[c] = mom[3 * (size_t) k + c];
        }
      N_gas_swallowed += (int) counts[0];
      N_BH_swallowed += (int) counts[1];
      N_dust_swallowed += (int) counts[2];
    }
  free(id);
  free(buf);
  return chk(rc, "ghip_blackhole_swallow") ? -1 : 0;
}

/* blackhole.c:794-1190, mode 0 */
int blackhole_evaluate(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nexport;
  (void) nsend_local;
  if(bh_ready("blackhole_evaluate"))
    return -1;
  if(mode != 0 || target < 0 || target >= NumPart || NTask > 1)
    {
      /* (on ranks the pass is a collective over ALL ranks' sinks: blackhole_accretion_neighbour_passes) */
      endrun(90002);
      return -1;
    }
  if(bh_marks_to_device() || bh_evaluate_batch(1, &target) || bh_marks_to_records(0))
    return -1;
  return 0;
}

/* blackhole.c:1201-1346, mode 0 */
int blackhole_evaluate_swallow(int target, int mode, int *nexport, int *nsend_local)
{
  (void) nexport;
  (void) nsend_local;
  if(bh_ready("blackhole_evaluate_swallow"))
    return -1;
  if(mode != 0 || target < 0 || target >= NumPart || NTask > 1)
    {
      /* (on ranks the pass is a collective over ALL ranks' sinks: blackhole_accretion_neighbour_passes) */
      endrun(90002);
      return -1;
    }
  if(bh_marks_to_device() || bh_swallow_batch(1, &target) || bh_marks_to_records(1))
    return -1;
  TreeOnDevice = 0;   /* masses changed: a tree built before is stale */
  return 0;
}

/* the neighbour-pass core of blackhole_accretion(), blackhole.c:294-660 */
void blackhole_accretion_neighbour_passes(void)
{
  if(bh_ready("blackhole_accretion"))
    return;
  N_gas_swallowed = N_BH_swallowed = N_dust_swallowed = 0;   /* blackhole.c:306 */
  int n = 0;
  for(int i = FirstActiveParticle; i >= 0; i = NextActiveParticle[i])
    if(p_type(i) == 5)
      n++;
  /* the marks start from the records (the reference resets SwallowID at the start of a step,
   * run.c / blackhole.c; Injected_BH_Energy accumulates until cooling consumes it) */
  if(bh_marks_to_device())
    return;
  if(n > 0 || NTask > 1)   /* (ranks: a collective, entered with or without sinks of one's own) */
    {
      int *idx = (int *) malloc((size_t) (n + 1) * sizeof(int));
      if(!idx)
        {
          endrun(90003);
          return;
        }
      int k = 0;
      for(int i = FirstActiveParticle; i >= 0; i = NextActiveParticle[i])
        if(p_type(i) == 5)
          idx[k++] = i;
      int bad = bh_evaluate_batch(n, idx) || bh_swallow_batch(n, idx);
      free(idx);
      if(bad)
        return;
    }
  if(bh_marks_to_records(1))
    return;
  TreeOnDevice = 0;
}

/* ------------------------------------------------------------------------------------------
 * more than one rank: the drivers as collectives over the domain-decomposed device path
 * ---------------------------------------------------------------------------------------- */
int gadget_force_unique_id(void *id128)
{
  return ghip_dd_rccl_unique_id(id128);
}

int gadget_force_connect(const void *id128)
{
  if(need_ctx("gadget_force_connect"))
    return -1;
  if(!DdReady)
    {
      if(chk(ghip_dd_init(Ctx, ThisTask, NTask), "ghip_dd_init"))
        return -1;
      DdReady = 1;
    }
  int rc = ghip_dd_rccl_connect(Ctx, id128);
  if(rc == GHIP_OK)
    RcclConnected = 1;
  else
    snprintf(ErrBuf, sizeof(ErrBuf), "ghip_dd_rccl_connect: %d %s", rc, ghip_last_error(Ctx));
  return rc;
}

void gadget_force_set_allgather(int (*allgather)(void *user, const void *send, size_t bytes, void *recv),
                                void *user)
{
  AllgatherFn = allgather;
  AllgatherUser = user;
}

/* the ranks' key ranges out of the host's top-tree, the global cube, the softenings: what
 * domain_Decomposition() left behind (domain.c:100-393) */
static int dd_prepare(void)
{
  if(NTask > GHIP_DD_MAXRANKS_HOST)
    {
      endrun(90002);
      return -1;
    }
  if(!DdReady)
    {
      if(chk(ghip_dd_init(Ctx, ThisTask, NTask), "ghip_dd_init"))
        return -1;
      DdReady = 1;
    }
  if(!RcclConnected && !AllgatherFn)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "NTask = %d: call gadget_force_connect() (RCCL) or "
               "gadget_force_set_allgather() first", NTask);
      fprintf(stderr, "gadget_force: %s\n", ErrBuf);
      endrun(90002);
      return -1;
    }
  if(DomainLen <= 0)
    domain_findExtent();
  if(chk(ghip_dd_set_domain(Ctx, DomainCorner, DomainCenter, DomainLen, All.ForceSoftening),
         "ghip_dd_set_domain"))
    return -1;
  if(TopNodes && DomainTask && NTopnodes > 0 && NTopleaves > 0)
    {
      /* -DMULTIPLEDOMAINS > 1: the curve leaf by leaf (top-leaves are numbered along the curve,
       * domain.c:1495-1509) with the rank that owns each (DomainTask[], domain.c:1208-1215) */
      unsigned long long *keys = (unsigned long long *) malloc((size_t) (NTopleaves + 1) * sizeof(unsigned long long));
      if(!keys)
        {
          endrun(90003);
          return -1;
        }
      for(int l = 0; l <= NTopleaves; l++)
        keys[l] = ~0ULL;
      for(int i = 0; i < NTopnodes; i++)
        if(TopNodes[i].Daughter == -1 && TopNodes[i].Leaf >= 0 && TopNodes[i].Leaf < NTopleaves)
          keys[TopNodes[i].Leaf] = TopNodes[i].StartKey;
      int bad = 0;
      for(int l = 0; l < NTopleaves; l++)
        if(keys[l] == ~0ULL || (l > 0 && keys[l] < keys[l - 1]) || DomainTask[l] < 0 || DomainTask[l] >= NTask)
          bad = 1;
      keys[0] = 0;
      keys[NTopleaves] = 1ULL << (3 * BITS_PER_DIMENSION);
      int rc = bad ? GHIP_EINVAL : ghip_dd_set_segments(Ctx, NTopleaves, keys, DomainTask);
      free(keys);
      if(bad)
        {
          snprintf(ErrBuf, sizeof(ErrBuf), "TopNodes / DomainTask do not describe %d top-leaves along the curve "
                   "owned by ranks 0..%d", NTopleaves, NTask - 1);
          fprintf(stderr, "gadget_force: %s\n", ErrBuf);
          endrun(90002);
          return -1;
        }
      return chk(rc, "ghip_dd_set_segments") ? -1 : 0;
    }
  if(!TopNodes || !DomainStartList || NTopnodes <= 0)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "NTask = %d: TopNodes / DomainStartList of the host's domain "
               "decomposition are not set", NTask);
      fprintf(stderr, "gadget_force: %s\n", ErrBuf);
      endrun(90002);
      return -1;
    }
  unsigned long long splits[GHIP_DD_MAXRANKS_HOST + 1];
  for(int r = 0; r < NTask; r++)
    {
      splits[r] = ~0ULL;
      for(int i = 0; i < NTopnodes; i++)   /* the top-leaf DomainStartList[r] (domain.c:1495-1509) */
        if(TopNodes[i].Daughter == -1 && TopNodes[i].Leaf == DomainStartList[r])
          {
            splits[r] = TopNodes[i].StartKey;
            break;
          }
      if(splits[r] == ~0ULL)
        {
          snprintf(ErrBuf, sizeof(ErrBuf), "no top-leaf %d (DomainStartList[%d]) in TopNodes",
                   DomainStartList[r], r);
          fprintf(stderr, "gadget_force: %s\n", ErrBuf);
          endrun(90002);
          return -1;
        }
    }
  splits[0] = 0;
  splits[NTask] = 1ULL << (3 * BITS_PER_DIMENSION);   /* PEANOCELLS */
  return chk(ghip_dd_set_splits(Ctx, splits), "ghip_dd_set_splits") ? -1 : 0;
}

/* one collective operation of the state machine: over RCCL, or staged through the host's all-gather */
static int dd_collective(int op, const void *params, int walk, const char *what)
{
  if(RcclConnected && !AllgatherFn)
    return chk(ghip_dd_run(Ctx, op, params, walk), what) ? -1 : 0;
  if(chk(ghip_dd_begin(Ctx, op, params, walk), what))
    return -1;
  for(;;)
    {
      int r = ghip_dd_step(Ctx);
      if(r == 0)
        return 0;
      if(r < 0)
        return chk(r, what) ? -1 : 0;
      if(chk(ghip_dd_exchange_host(Ctx, AllgatherFn, AllgatherUser), what))
        return -1;
    }
}

static int set_active_list(int nact, int everybody)
{
  if(everybody)
    return chk(ghip_set_active(Ctx, NULL, 0), "ghip_set_active") ? -1 : 0;
  return chk(ghip_set_active(Ctx, ActiveBuf ? ActiveBuf : &nact, nact), "ghip_set_active") ? -1 : 0;
}

/* gravtree.c:27-828 on NTask ranks: the local walk + the export rounds :175-339 become one pass of the
 * domain-decomposed device path (this rank's tree, the others' locally essential trees, one merged
 * tree, the walks); every target meets the interaction set of the reference's single global tree */
static void gravity_tree_ranks(void)
{
  double t0 = wallclock();
  if(All.ComovingIntegrationOn)
    set_softenings();
  DeviceFresh = 0;
  TreeOnDevice = 0;
  GravPending = 0;
  int nact = collect_active(0);
  if(nact < 0)
    return;
  if(upload_particles(0))
    return;
  if(dd_prepare())
    return;
  if(set_active_list(nact, nact == NumPart))
    return;
  ghip_grav_params g;
  fill_grav_params(&g);
  int walk = Cfg.pmgrid ? GHIP_WALK_SHORTRANGE : GHIP_WALK_NEWTON;
  if(Cfg.periodic && !Cfg.pmgrid)
    walk = GHIP_WALK_NEWTON_EWALD;
  if(dd_collective(GHIP_DD_GRAVITY, &g, walk, "gravity_tree (ranks)"))
    return;
  TreeOnDevice = 1;
  TreeReconstructFlag = 0;
  GravPendingActive = nact;
  if(gravity_complete(1))
    return;
  Phase = 1;
  CPU_Step_Treewalk += wallclock() - t0;
}

/* density.c:89-704 on NTask ranks: the export rounds :193-389 become an import of ghost gas
 * particles; the h iteration then runs without another exchange */
static void density_ranks(void)
{
  double t0 = wallclock();
  if(Phase != 1)
    {
      DeviceFresh = 0;
      if(upload_particles(0) || dd_prepare())
        return;
    }
  int nact = collect_active(1);
  if(nact < 0)
    return;
  if(set_active_list(nact, active_list_is_everybody()))
    return;
  ghip_dens_params d;
  fill_dens_params(&d);
  if(dd_collective(GHIP_DD_DENSITY, &d, 0, "density (ranks)"))
    return;
  if(chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, 0, 1, 0), "ghip_download_aos"))
    return;
  /* Type-5 / Type-2 targets: every rank's sinks against every rank's gas (GHIP_DD_SINK_DENSITY) */
  if(density_of_sinks(&d))
    return;
  Phase = 2;
  CPU_Step_Density += wallclock() - t0;
}

/* hydra.c:145-813 on NTask ranks: the ghosts carry their owners' density results (refreshed at the end
 * of density()), so the pass itself is local */
static void hydro_force_ranks(void)
{
  double t0 = wallclock();
  if(Phase != 2)
    {
      snprintf(ErrBuf, sizeof(ErrBuf), "hydro_force on %d ranks must follow density() and "
               "force_update_hmax() of the same step (accel.c:84-106)", NTask);
      fprintf(stderr, "gadget_force: %s\n", ErrBuf);
      endrun(90002);
      return;
    }
  ghip_hydro_params h;
  fill_hydro_params(&h, 0);
  if(dd_collective(GHIP_DD_HYDRO, &h, 0, "hydro_force (ranks)"))
    return;
  if(chk(ghip_download_aos(Ctx, records_p(), records_s(), &Lay, 0, 0, 1), "ghip_download_aos"))
    return;
  Phase = 0;
  CPU_Step_Hydro += wallclock() - t0;
}
