// ghip_internal.h -- shared declarations of the gfx950 force-path library (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ghip.h"
#include "ghip_count.h"

#define GHIP_BITS 21       // BITS_PER_DIMENSION, allvars.h:58
#define GHIP_EN 64         // EN, forcetree.c:51
// A second copy of the Ewald table is brick-tiled: one 128-byte line holds the values of 2 (first
// index) x 2 (second index) rows x 4 consecutive values of the last index, and consecutive bricks
// along the last index OVERLAP by one value (3 cells per brick), so that the pair (k, k+1) of a cell
// never straddles two lines.  Value (i, j, v) sits in brick (i>>1, j>>1, v/3) at
// ((i&1)*2 + (j&1))*4 + v%3 -- and, for v%3 == 0 and v > 0, also in slot 3 of brick v/3 - 1.
#define GHIP_EW_NB ((GHIP_EN >> 1) + 1)   // bricks along the first and the second index (values 0..EN)
#define GHIP_EW_NKB (GHIP_EN / 3 + 1)     // bricks along the last index: 3 cells each
#define GHIP_EW_DOUBLES (GHIP_EW_NB * GHIP_EW_NB * GHIP_EW_NKB * 16)
static inline __host__ __device__ unsigned int ghip_ew_offset(int i, int j, int kb, int t)
{
  return (unsigned int) ((((i >> 1) * GHIP_EW_NB + (j >> 1)) * GHIP_EW_NKB + kb) * 16 +
                         ((i & 1) * 2 + (j & 1)) * 4 + t);
}
#define GHIP_NTAB 1000     // NTAB, forcetree.c:28
#define GHIP_WAVE 64
#define GHIP_BLOCK 256
#define GHIP_MAXRANKS 64      // shards of a multi-GPU run (u64 rank masks)

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;  // bytes
};

// Sizes of a tree as the DEVICE knows them: written by k_tree_info at the end of the counting stage
// of a build, into device memory (the build's later kernels and every hot-path consumer read them
// there) and into a pinned host mirror (the host learns them without a copy at its next
// verification point, ghip_tree_verify).  The host therefore enqueues a whole build, and the walks
// behind it, without waiting for the node count.
struct TreeSizes
{
  int nelem;      // n + nnodes; 0 while the tree is unusable (bad != 0): every consumer then does nothing
  int nnodes;
  int maxlevel;
  int bad;        // 1: more nodes than the element buffers hold; 2: key runs too long for the 32-bit sort
  int n;          // sources
  int gen;        // build generation (host counter): tells a fresh mirror from a stale one
  int longrun;
  int pad;
};

struct TreeDev
{
  int n = 0;        // particles in this tree
  int nnodes = 0;   // (host copies: exact after a synchronous build or after ghip_tree_verify)
  int nelem = 0;    // n + nnodes
  int maxlevel = GHIP_BITS;  // deepest node level
  // sort
  DevBuf key, skey, idx, perm, iperm;  // u64[n], u64[n], i32[n], i32[n] (sorted->host index), i32[nhost]
  DevBuf cpl, cnt, nb;                 // i32[n]: common prefix levels, node counts, exclusive scan
  DevBuf phkey, phorder;               // u64[n], i32[n]: tree-order indices in Peano-Hilbert order
  // pre-order element list
  DevBuf xm, cl, lk, aux;              // double4[nelem], double4[nelem], int4[nelem], f64[nelem]
  // walk segments (ghip_walk.h): start[ns+1], nanc[ns], anc[ns][GHIP_MAXANC]
  DevBuf seg_start, seg_nanc, seg_anc;
  int ns = 1;            // segments of the fine table
  int seg_ns[3] = {1, 1, 1}, seg_soff[3] = {0, 0, 0}, seg_noff[3] = {0, 0, 0};   // fine / mid / coarse
  DevBuf mq, mq2;                      // WalkHot[nelem], WalkCold[nelem]: walk records (gravity tree)
  DevBuf slvl;                         // i32[n]: level of an imported pruned node in sorted order, 0: particle
  DevBuf father, arrived;              // moment pass: per-level lists of the chunk-crossing nodes, their counts
  DevBuf dsz;                          // TreeSizes on the device
  TreeSizes *hsz = nullptr;            // its pinned host mirror
  int cap_nodes = 0;                   // nodes the element buffers are sized for
  int last_n = -1, last_nnodes = 0;    // the last verified build: sources, nodes
  bool built = false;
};

// ---------------------------------------------------------------------------------------------
// multi-GPU domain decomposition (ghip_dd.hip, ghip_comm.hip)
// ---------------------------------------------------------------------------------------------
// Target groups of a shard, two levels: DD_NSUPER super-groups of DD_NSUB groups of consecutive
// targets along the space-filling curve.  A group is (bounding box, the least OldAcc of its
// targets, the largest search radius); other shards test their tree nodes / gas particles against
// these boxes to decide what this shard can possibly need (ghip_dd.hip).
#define DD_NSUPER 64   // = lanes of a wavefront: one test of a node against all super-groups of a rank
#define DD_NSUB 16
#define DD_NGROUPS (DD_NSUPER * DD_NSUB)
#define DD_TABLE (DD_NSUPER + DD_NGROUPS)   // records per shard: super-groups first
// ... and one status record behind them (cx != 0: this shard met an error while it prepared the
// table -- a particle outside its key range, a device invariant): every shard reads every status after
// the all-gather and all of them fail TOGETHER, instead of one returning while its peers wait for it
// inside the next collective
#define DD_STRIDE (DD_TABLE + 1)
struct __attribute__((aligned(64))) DDGroup
{
  double cx, cy, cz;   // box centre
  double ex, ey, ez;   // half extents; ex < 0: empty group
  double amin;         // gravity: least OldAcc of the targets; SPH: unused
  double rmax;         // SPH: largest search radius (Hsml * margin) of the targets; gravity: unused
};

// one element of a locally essential tree as it crosses a link (64 B)
struct __attribute__((aligned(64))) LetRec
{
  double x, y, z, m;          // particle: position, mass; pruned node: centre of mass, mass
  unsigned long long key;     // Morton key (node: the cell's prefix, zero padded)
  double aux;                 // particle: its softening; node: max softening below, negative = mixed
  int level;                  // 0: particle; L > 0: node of level L that no target of the receiver opens
  int pad;
  double spare;
};

// a ghost gas particle as it crosses a link: the two 64-byte records of the SPH kernels
struct __attribute__((aligned(64))) GhostRec
{
  double p[8];   // x, y, z, m, vx, vy, vz, h
  double q[8];   // pressure, density, dhsmlfac, divvel, curlvel, timestep, 0, 0
};

// a pending exchange of the state machine (ghip_dd_step): executed over RCCL by ghip_dd_exchange
// or, for several shards living in one process, by ghip_dd_exchange_local
struct DDXchg
{
  int kind = 0;                 // 0 none, 1 all-gather of fixed-size blocks, 2 all-to-all-v of records
  const void *send = nullptr;   // device
  size_t bytes = 0;             // kind 1: bytes per rank; kind 2: bytes per record
  DevBuf *recv = nullptr;       // kind 1: [nranks][bytes]; kind 2: records, source-rank major
  int scount[GHIP_MAXRANKS], soff[GHIP_MAXRANKS];   // kind 2, in records (soff into `send`)
  int rcount[GHIP_MAXRANKS], roff[GHIP_MAXRANKS];   // filled by the exchange
  int rtotal = 0;
};

// state of the h iteration of a set of sinks (ghip_sink.hip)
struct SinkIter
{
  std::vector<double> h, left, right, numngb, rho, entropy, gasvel;
  std::vector<int> todo;
};

struct DDState
{
  bool on = false;
  int rank = 0, nranks = 1;
  unsigned long long splits[GHIP_MAXRANKS + 1];   // Peano-Hilbert key range of rank r: [splits[r], splits[r+1])
  // The general form (MULTIPLEDOMAINS > 1: a rank owns several pieces of the curve, domain.c:482-494,
  // 1158-1215): nseg segments [segkey[s], segkey[s+1]) with their owners, adjacent segments of one
  // owner merged; ownlo/ownhi = this rank's own pieces.  Device copies for the three kernels that ask
  // "whose is this key" (range check, shared-cell test of the LET selection, migration).
  int nseg = 0, nown = 0;
  DevBuf segkey, segowner, ownlo, ownhi;
  void *nccl = nullptr;                           // ncclComm_t (ghip_comm.hip)
  // state machine
  int op = 0, phase = 0, walk = 0;
  ghip_grav_params gp;
  ghip_dens_params dp;
  ghip_hydro_params hp;
  DDXchg x;
  DevBuf xstage;                  // device staging of the count all-gather
  // target-group tables: own, and everybody's after the all-gather
  DevBuf grp_own, grp_all;        // DDGroup[DD_TABLE], DDGroup[nranks][DD_TABLE]
  // gravity: locally essential trees
  DevBuf reach, sendm;            // u64[nelem]: ranks whose targets reach / are sent this element
  DevBuf selcnt;                  // i32[nranks][nblocks] block counts / offsets of the packing
  DevBuf let_list;                // i32: per destination, the elements of this tree it is sent
  DevBuf let_send, let_recv;      // LetRec[]
  int let_sent = 0, gh_sent = 0;
  double gh_growth = 0;
  DevBuf gas_tgt;                 // i32: the local gas targets in curve order (gravity-tree indices)
  DevBuf gas_src;                 // i32: all local gas particles in gravity-tree order
  int gt_nimp = 0;                // imported elements the next gravity-tree build includes
  DevBuf src_x, src_y, src_z, src_m, src_aux, src_key, src_lvl;   // sources = local particles + imports
  // SPH: ghost gas particles
  DevBuf gh_mask;                 // u64[ngas]: ranks this gas particle is a ghost on
  DevBuf gh_list;                 // i32: per destination, the local gas-tree indices sent (fixed for the step)
  int gh_scount[GHIP_MAXRANKS], gh_soff[GHIP_MAXRANKS];
  DevBuf gh_send, gh_recv;        // GhostRec[]
  int nghost = 0;
  double gh_margin = 1.3;         // search radii are padded by this factor when ghosts are selected
  double gh_margin_cur = 1.3;     // ... in the density call in progress: grows when an h outgrew it (see density_step)
  int gh_retries = 0;             // re-selections of the density call in progress
  DevBuf status_own, status_all;  // f64[2] per shard: {largest Hsml growth, error flag} of a density call
  int local_err = 0;              // what this shard itself met while it prepared its group table
  std::string local_msg;
  int dens_rc = 0;                // ... and in the h iteration of the density call in progress
  std::string dens_msg;
  DevBuf gh_ratio;                // f64[2]: largest Hsml growth seen by the density iterations
  DevBuf gsx, gsy, gsz, gsm, gsh; // gas sources = local gas + ghosts (tree build input)
  DevBuf h0;                      // f64[ngas]: smoothing lengths the ghost selection was made with
  // migration (domain_exchange): destination masks, send lists, records, shadow field arrays
  DevBuf mig_mask, mig_list, mig_send, mig_recv, mig_scan;
  DevBuf fshadow[GHIP_F_COUNT];
  int mig_out = 0, mig_in = 0;
  // sinks on shards (ghip_sink.hip): every shard's sinks are made known to all, each shard
  // evaluates all of them against its own particles, the partial sums are all-gathered
  ghip_dd_sink_args sink;         // arguments of the operation in progress
  DevBuf sk_send, sk_all;         // SinkRec[] of this shard / of all shards (rank order)
  DevBuf sk_part, sk_parts;       // partial sums of this shard [k][nsink_total] / of all shards
  DevBuf sk_work;                 // per pass: h[total] | slots[total]; swallow: BH masses [n] | victims [n]
  int sk_total = 0, sk_off = 0, sk_iter = 0, sk_ncur = 0;
  SinkIter sk_it;                 // h iteration state of ALL sinks (identical on all ranks)
  // traffic of the last operation (bytes this rank sent over links, excluding its own block)
  long long bytes_sent[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  ghip_pm_params pm;              // GHIP_DD_PM
  DevBuf pm_all;                  // the density meshes of all shards
};

// the pending exchange of a state-machine step (ghip_dd.hip, ghip_sink.hip)
static inline void ghip_dd_set_allgather(DDState &D, const void *send, size_t bytes, DevBuf *recv)
{
  D.x.kind = 1;
  D.x.send = send;
  D.x.bytes = bytes;
  D.x.recv = recv;
}

static inline void ghip_dd_set_alltoallv(DDState &D, const void *send, size_t recbytes,
                                         const int *scount, const int *soff, DevBuf *recv)
{
  D.x.kind = 2;
  D.x.send = send;
  D.x.bytes = recbytes;
  D.x.recv = recv;
  for(int r = 0; r < D.nranks; r++)
    {
      D.x.scount[r] = scount[r];
      D.x.soff[r] = soff[r];
    }
}

struct ghip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  int n = 0, ngas = 0;
  // host-order fields
  DevBuf f[GHIP_F_COUNT];
  // staging for host<->device transfers / AoS images
  DevBuf stage, aosP, aosS;
  void *pinned = nullptr;
  size_t pinned_cap = 0;

  // domain
  double corner[3] = {0, 0, 0}, center[3] = {0, 0, 0}, dlen = 0;
  double soft[6] = {0, 0, 0, 0, 0, 0};

  TreeDev gt;  // gravity tree: all particles
  TreeDev st;  // gas tree
  // sorted gravity-source side arrays (targets read these): x,y,z,soft,oldacc  [n]
  DevBuf sx, sy, sz, ssoft, soldacc;
  // gas records in gas-tree order
  DevBuf gp;  // 8 doubles: x,y,z,m,vx,vy,vz,h
  DevBuf gq;  // 8 doubles: pressure, density, dhsmlfac, divvel, curlvel, timestep, 0, 0
  // density work arrays (gas-tree order)
  DevBuf dleft, dright, drho, dnumngb, ddhsml, ddivv, drot, dflags, dtgt_a, dtgt_b, dhcur;
  DevBuf hpart;  // hydro partial sums

  // active lists
  DevBuf act_host_idx;  // i32[nactive] host indices (uploaded)
  int nactive = -1;     // -1: all
  bool gas_mixed = false;        // a record of the gas block [0, ngas) has Type != 0 (ghip_check_gas_types)
  int skip_massless = 0;         // a gas particle of mass 0 is nobody's neighbour: 1 in density() (-DDUST or
                                 // -DBLACK_HOLES), 3 in density() and hydro_force() (-DBLACK_HOLES)
  bool massless_marked = false;  // the gas records carry the mark of the massless rule
  bool gas_types_unknown = false;  // ... not known since a migration: checked before the next gas tree
  bool lists_dirty = true;       // gravity target list
  bool gas_list_dirty = true;    // gas target list (made once the deferred gas tree exists)
  DevBuf tg_grav, tg_gas;  // sorted-order target lists
  int nt_grav = 0, nt_gas = 0;
  int shard_rank = 0, shard_n = 1;

  // walk outputs in target order
  DevBuf tax, tay, taz, tcost;
  // second set (slot 1) for the Ewald walk of an overlapped Newton+Ewald pair
  DevBuf tax2, tay2, taz2, tcost2, plan_nsub2, plan_woff2, plan_wave2, cubtmp2;
  hipStream_t stream2 = nullptr;   // the pair's Ewald walk runs here
  hipStream_t stream3 = nullptr;   // ... and its Newtonian walk here, so the main stream stays free
  bool adaptive_gravsoft = false;   // ADAPTIVE_GRAVSOFT_FORGAS: gas softening = Hsml (ghip_set_adaptive_gravsoft)
  bool grav_pending = false;       // a pair is in flight; evx[2] marks its end (see ghip_join)
  unsigned int *pair_started = nullptr;   // device word: the pair's Ewald walk has dispatched its last workgroup
  // the second half of the gas tree build (elements, moments, SphNode records, gas records) is
  // deferred to the first call that needs it, so that it runs underneath a gravity pair
  bool gas_pending = false;
  bool gas_wait_upload = false;   // ghip_upload_aos_particles done, ghip_upload_aos_gas not yet: the
                                   // deferred gas-tree work must not run on stale SphP fields
  hipEvent_t evx[4];               // pair ordering: inputs ready / Newton combined / Ewald combined /
                                   // Ewald walk kernel done (ghip_hydro waits for it)
  bool evx_ready = false;
  hipEvent_t evt[2];               // tree build: fork after the tree-order gather / join (curve_order
                                   // runs on stream2 next to the element emission on the main stream)
  // adaptive wavefront plan of the gravity walks (ghip_walk.h): per walk kind the elements
  // visited per bucket in the previous call (double-buffered) and the scratch plan arrays
  DevBuf plan_steps[3][2], plan_nsub, plan_woff, plan_wave;
  int plan_nb[3] = {-1, -1, -1}, plan_ns[3] = {-1, -1, -1}, plan_cur[3] = {0, 0, 0};
  hipStream_t plan_writer[3] = {nullptr, nullptr, nullptr};   // stream of the last walk of each kind: its
                                                               // per-bucket counts feed the next plan
  DevBuf cubtmp3;                                              // scan space of a plan built on stream3

  // ewald
  DevBuf ewtab;   // double[(EN+1)^3]: fcorrx scaled by 1/Box^2 (y, z by symmetry)
  DevBuf ewbrick; // the same values brick-tiled (GHIP_EW_DOUBLES), for the Ewald walk outside a pair
  double ew_box = 0;
  DevBuf srtab;   // float[NTAB]
  bool srtab_ready = false;

  // scan/sort temp
  DevBuf cubtmp;
  // particle-mesh force (ghip_pm.hip): hipFFT plans for pm_n^3, mesh buffers
  int pm_n = 0;
  void *pm_fwd = nullptr, *pm_inv = nullptr;
  DevBuf pm_rho, pm_k, pm_force;
  // counters (device): 8 x u64
  DevBuf counters;
  // work counters of the walks and the SPH kernels, 64 slots per kind (ghip_count.h): of the current
  // calls (each call of a kind clears its own) and of the run (ghip_run_begin clears)
  DevBuf cslots, rslots;
  ghip_stats stats;
  hipEvent_t ev[16];
  DDState dd;                // multi-GPU domain decomposition (ghip_dd.hip)
  DevBuf bh_swallow, bh_injected;   // "next" row N4 (ghip_sink.hip): P[].SwallowID u32[n],
                                    // SphP[].i.Injected_BH_Energy f64[ngas]
  int timestep_endrun = 0;   // endrun code of the last ghip_advance_timesteps failure
  bool ev_ready = false;

  // ---- asynchronous tree build (ghip_tree.hip) ----
  // A build whose particle number equals the previous build's is enqueued without the host waiting
  // for the node counts: buffers are sized by capacity, every kernel reads the sizes on the device
  // (TreeSizes).  The counts are VERIFIED -- the host waits for ev_sizes only, recorded right after the
  // counting stage, not for the walks -- before anything persistent is modified on their basis
  // (ghip_tree_verify: entry of ghip_density and of everything that joins).  A build that turns out
  // bad (more nodes than the buffers hold, key runs too long for the 32-bit sort) left nelem = 0 on
  // the device, so whatever ran on it did nothing; it is then repeated synchronously and the gravity
  // calls made since are replayed from grav_log.
  bool async = false;              // ghip_set_async: drift / kick report their errors at the next sync
  bool tree_unverified = false;
  bool tree_async_ok = true;       // GHIP_TREE_SYNC=1 switches the asynchronous build off
  hipEvent_t ev_sizes = nullptr, ev_sizes_gas = nullptr;
  bool gas_unverified = false;     // the same for the gas tree, whose whole build is deferred (ghip_finish_gas_tree)
  bool gas_async = false;          // ... and follows the gravity tree's mode
  int build_gen = 0;
  bool sort_wide = false;          // sticky: a build met key runs too long for the 32-bit sort
  bool in_recover = false;
  struct GravCall
  {
    ghip_grav_params p;
    int walk;
  };
  std::vector<GravCall> grav_log;  // gravity calls since the last tree build
  std::vector<void *> host_pins;   // ranges page-locked by ghip_pin_host
  bool hydro_early = false;        // ghip_set_hydro_release
  // ---- the gravity tree between two full builds (ghip_set_dynamic_tree, ghip_export.hip): a copy of
  // the last full build's element list whose nodes are drifted and kicked as forcetree.c:1356-1520
  // does; the gravity walks of a sub-step read it instead of the tree of the current positions ----
  bool dyn_on = false, dyn_valid = false, dyn_use = false;
  TreeDev dyn;
  DevBuf dyn_ev, dyn_dp;           // double4[nelem]: (vs, vmax) / (dp, kicked) per element
  DevBuf dyn_eh, dyn_cnt, dyn_fa;  // scratch of the moment pass (export's kernels)
  DevBuf dyn_kick;                 // double4[nelem]: a kick pass's per-element sums
  DevBuf kick_dv, kick_flag;       // f64[3][n], i32[n]: velocity changes of the last ghip_advance_timesteps
  // balance of a Newton + Ewald pair (ghip_gravity.hip, pair_balance): dynamic LDS per Newtonian
  // workgroup in use, and a ring of the last pairs' start / end events with the cap they ran under
  int pair_lds = 10240;
  hipEvent_t pc_ev[4][4] = {};
  int pc_cap[4] = {0, 0, 0, 0};        // 0: slot empty or already read
  int pc_hyd[4] = {0, 0, 0, 0};        // the hydro kernel was queued underneath that pair
  int pc_head = 0;
  bool pc_ready = false;
  float pc_cost[2] = {-1.f, -1.f};     // last measured cost of a pair under 8 KB / 10 KB
  int pc_age[2] = {0, 0};              // pairs launched since that measurement
  hipEvent_t ev_side = nullptr;    // end of ghip_gravity_to_records' work on the pair's stream

  // ---- run statistics without a host synchronisation per step (ghip_run_begin / ghip_step_begin /
  // ghip_step_end / ghip_get_run_stats): a ring of event sets, device-side accumulated counters ----
  std::vector<hipEvent_t> ev_ring;     // [slots][GHIP_NEV + 2]: phase events + step begin / end marks
  int ring_slots = 0, ring_cur = -1;   // ring_cur < 0: the fixed set ev[] is in use
  hipEvent_t *evp = nullptr;           // the event set in use (ev[] or a ring slot)
  long long run_steps = 0, run_syncs0 = 0, run_launches0 = 0, run_dens_iter = 0;
  long long n_syncs = 0;               // blocking host waits inside the library since creation
  DevBuf run_acc;                      // u64[16]: counters summed over the steps of a run
};
#define GHIP_NEV 16

// every blocking wait of the host goes through these two (they are counted: ghip_run_stats)
static inline hipError_t ghip_stream_sync(ghip_ctx *ctx, hipStream_t st)
{
  ctx->n_syncs++;
  return hipStreamSynchronize(st);
}
static inline hipError_t ghip_event_sync(ghip_ctx *ctx, hipEvent_t e)
{
  ctx->n_syncs++;
  return hipEventSynchronize(e);
}
long long ghip_launch_count(void);   // kernel launches issued through this library so far (ghip_api.hip)
int ghip_tree_verify(ghip_ctx *ctx);  // tree.hip
int ghip_gas_verify(ghip_ctx *ctx);

int ghip_fail(ghip_ctx *ctx, int code, const char *fmt, ...);
// Device error words: ints in pinned, device-visible host memory that kernels set when an internal
// invariant breaks.  Every synchronising entry point (ghip_sync, ghip_get_field, the AoS downloads,
// ghip_get_stats) calls ghip_check_device_errors after its stream sync and fails with GHIP_EDEVICE.
#define GHIP_ERRW_PLAN 0     // wavefront plan of a gravity walk exceeded its grid (ghip_walk.h)
#define GHIP_ERRW_LET 1      // a target wanted to open a pruned node of another shard's tree
#define GHIP_ERRW_TREE 2     // tree emission wrote outside the element list / malformed import
#define GHIP_ERRW_GHOST 3    // spare
#define GHIP_ERRW_DRIFT 4    // a particle was ahead of the drift target (reference: endrun(12), predict.c:148)
#define GHIP_ERRW_TIMESTEP 5 // the endrun code of a failed timestep criterion (888, 818, 112313)
#define GHIP_ERRW_COUNT 8
// (one block per process, shared by its contexts: an error raised by a kernel of one logical shard
// is seen by whichever context synchronises next)
int *ghip_errwords(void);
static inline int *ghip_errword(ghip_ctx *, int which) { return ghip_errwords() + which; }
int ghip_check_device_errors(ghip_ctx *ctx);
// The overlapped Newton+Ewald pair returns with its walks still running on their own streams, so
// that the SPH phases -- which read and write nothing the walks touch -- can be enqueued on the
// main stream underneath them.  Every other entry point first makes the main stream wait for the
// pair (GHIP_JOIN at its top).
int ghip_join(ghip_ctx *ctx);        // wait for a pair in flight AND complete a deferred gas tree
// Workgroup size of the short kernels that may run underneath a gravity pair: the walks use
// one-wavefront workgroups, and a 256-thread workgroup -- which needs four free wavefront slots in
// one CU at the same moment -- would starve behind them however high its stream's priority.
static inline int ghip_wg(const ghip_ctx *ctx) { return ctx->grav_pending ? 64 : 256; }
// pinned word the record unpack sets when a record of the gas block [0, ngas) is not Type 0
static inline int *ghip_gas_mixed_word(ghip_ctx *ctx)
{
  return reinterpret_cast<int *>(reinterpret_cast<char *>(ctx->pinned) + 256);
}
static inline unsigned long long *ghip_cslot(ghip_ctx *ctx, int kind)
{
  return reinterpret_cast<unsigned long long *>(ctx->cslots.p) + (size_t) kind * GHIP_CKIND_U64;
}
static inline unsigned long long *ghip_rslot(ghip_ctx *ctx, int kind)
{
  return reinterpret_cast<unsigned long long *>(ctx->rslots.p) + (size_t) kind * GHIP_CKIND_U64;
}
// host copy of a slot buffer added up: out[kind][which], which = 0, 1
int ghip_read_slots(ghip_ctx *ctx, DevBuf &buf, unsigned long long out[GHIP_CK_COUNT][2]);
int ghip_mark_converted_gas(ghip_ctx *ctx);
int ghip_unmark_massless_for_hydro(ghip_ctx *ctx);
int ghip_check_gas_types(ghip_ctx *ctx);   // after TYPE changed: recount, sets ctx->gas_mixed (synchronises)
int ghip_dyn_capture(ghip_ctx *ctx);                 // after a full build: copy the gravity tree, vs / vmax per node
int ghip_dyn_kick_recorded(ghip_ctx *ctx);           // force_kick_node for what ghip_advance_timesteps recorded
void ghip_dyn_release(ghip_ctx *ctx);
int ghip_fill_walk_records(ghip_ctx *ctx, TreeDev &t);
int ghip_gravity_finish_on(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac, int all_shards,
                           hipStream_t st);
int ghip_join_pair(ghip_ctx *ctx);   // wait for a pair in flight only (entry of the gravity walks)
int ghip_finish_gas_tree(ghip_ctx *ctx);   // complete a deferred gas tree (entry of the SPH phases)
#define GHIP_JOIN(ctx)                          \
  do                                            \
    {                                           \
      int j_ = ghip_join(ctx);                  \
      if(j_ != GHIP_OK)                         \
        return j_;                              \
    }                                           \
  while(0)
void ghip_pm_release(ghip_ctx *ctx);

#define HIPCHK(call)                                                                         \
  do                                                                                         \
    {                                                                                        \
      hipError_t e_ = (call);                                                                \
      if(e_ != hipSuccess)                                                                   \
        return ghip_fail(ctx, GHIP_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,        \
                         hipGetErrorString(e_));                                             \
    }                                                                                        \
  while(0)

#define GCHK(call)            \
  do                          \
    {                         \
      int r_ = (call);        \
      if(r_ != GHIP_OK)       \
        return r_;            \
    }                         \
  while(0)

int ghip_ensure(ghip_ctx *ctx, DevBuf &b, size_t bytes);
static inline int cdiv(long long a, int b) { return (int) ((a + b - 1) / b); }

template <class T> static inline T *P(DevBuf &b) { return reinterpret_cast<T *>(b.p); }
template <class T> static inline const T *P(const DevBuf &b) { return reinterpret_cast<const T *>(b.p); }

// Target sharding (multi-GPU).  The curve-ordered target list is cut into buckets of 64; rank r
// owns buckets r, r+N, r+2N, ... (every rank samples the whole volume, so per-rank cost is
// balanced without a cost model).  The list is stored rank-major, so a rank's targets are the
// contiguous range [lo, lo+cnt); `per` is the padded common slice length of the all-gathers.
static inline void ghip_shard_range(int nt, int nranks, int rank, int *lo, int *cnt, int *per)
{
  int nb = (nt + 63) / 64;
  int lastsize = nt - (nb - 1) * 64;
  int acc = 0, mine = 0, mylo = 0;
  for(int r = 0; r < nranks; r++)
    {
      int cb = (r < nb) ? (nb - r + nranks - 1) / nranks : 0;
      int c = cb * 64;
      if(cb > 0 && (nb - 1) % nranks == r)
        c -= 64 - lastsize;
      if(r == rank)
        {
          mylo = acc;
          mine = c;
        }
      acc += c;
    }
  *lo = mylo;
  *cnt = mine;
  *per = ((nb + nranks - 1) / nranks) * 64;
}

// tree.hip
int ghip_tree_build_impl(ghip_ctx *ctx);
int ghip_build_target_lists(ghip_ctx *ctx);
int ghip_gastree_refresh_hmax(ghip_ctx *ctx);
int ghip_gather_f64(ghip_ctx *ctx, int n, const int *perm, const double *src, double *dst);
int ghip_gather_f64_lim(ghip_ctx *ctx, int n, const int *perm, const double *src, int limit,
                        double *dst);
// gravity.hip
int ghip_gravity_impl(ghip_ctx *ctx, const ghip_grav_params *p, int walk);
int ghip_build_segments(ghip_ctx *ctx, TreeDev &t, bool walk_records);
// sph.hip
int ghip_density_impl(ghip_ctx *ctx, const ghip_dens_params *p);
int ghip_hydro_impl(ghip_ctx *ctx, const ghip_hydro_params *p);
int ghip_sph_fill_nodes(ghip_ctx *ctx, bool hmax_only);
// dd.hip
void ghip_dd_release(ghip_ctx *ctx);

// ---------------------------------------------------------------------------------------------
// walk segments: the element list of a tree is cut into `ns` contiguous segments and `nsub`
// wavefronts share one bucket of 64 targets (wavefront `sub` takes segments sub, sub+nsub, ...);
// see ghip_walk.h "Load balance"
// ---------------------------------------------------------------------------------------------
#define GHIP_MAXANC 24
#define GHIP_MAXSUB 8
struct WalkSeg
{
  int ns;                          // number of segments
  int nsub;                        // wavefronts per bucket
  const int *__restrict__ start;   // [ns+1] first element of each segment
  const int *__restrict__ nanc;    // [ns]
  const int *__restrict__ anc;     // [ns][GHIP_MAXANC] ancestors of start[k], root first
};

int ghip_walk_layout(const TreeDev &t, WalkSeg &sg, int table);   // returns nsub

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_wave_min_i32(int v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      int o = __shfl_xor(v, off, 64);
      v = o < v ? o : v;
    }
  return v;
}


// element link record: x = skip (element index after this subtree), y = sorted particle index
// for a particle element or -(level+1) for a node, z = first particle of the node, w = count
#define LK_IS_PARTICLE(lk) ((lk).y >= 0)

__device__ __forceinline__ double d_nearest(double x, double boxsize, double boxhalf)
{
  // forcetree.c:49 NEAREST
  return (x > boxhalf) ? (x - boxsize) : ((x < -boxhalf) ? (x + boxsize) : x);
}

__device__ __forceinline__ double d_ngb_periodic(double x, int periodic, double boxsize,
                                                 double boxhalf)
{
  // allvars.h:300-308 NGB_PERIODIC_LONG_*
  double xt = fabs(x);
  return (periodic && xt > boxhalf) ? (boxsize - xt) : xt;
}

__device__ __forceinline__ unsigned long long d_wave_sum_u64(unsigned long long v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
