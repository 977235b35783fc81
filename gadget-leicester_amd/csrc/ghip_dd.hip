// ghip_dd.hip -- multi-GPU force path: Peano-Hilbert domain decomposition with tree-node and
// ghost-particle exchange.
//
// Replaces, for a run sharded over the GPUs of one node (one process per GPU):
//   domain_Decomposition / domain_findSplit_work_balanced     domain.c:100, 378-384, 1075-1113
//   force_create_empty_nodes / force_insert_pseudo_particles /
//   force_exchange_pseudodata / force_treeupdate_pseudos       forcetree.c:384-450, 879-1075
//   the export rounds of gravity_tree / density / hydro_force   gravtree.c:175-339, density.c:193-389,
//                                                               hydra.c:274-526
//
// The reference keeps one global oct-tree whose top part is replicated; a target whose walk opens
// another rank's top-leaf is EXPORTED there, walked remotely, and the partial sum comes back --
// rounds of small messages driven by the host.  On a node of GPUs with point-to-point xGMI links the
// data goes the other way, once per phase:
//   gravity   every shard sends every other shard the part of its tree that shard can possibly need
//             (a LOCALLY ESSENTIAL TREE): walking its own tree top-down against the receiver's
//             target groups (bounding boxes + the least OldAcc inside), a node that no target of the
//             receiver can open under the reference's opening rules is sent as ONE element with its
//             moments ("pruned node"), anything that could be opened is descended, particles that
//             are reached are sent as particles.  The receiver sorts its own particles and the
//             imported elements by key and builds ONE tree over both (ghip_tree.hip): cells shared
//             by several shards are re-created from their parts, so every node a target can test
//             has the moments of the global tree, and the walk (ghip_walk.h, unchanged) makes for
//             each target exactly the opening decisions of the reference's single global tree --
//             the interaction set is the same for every number of shards, as in the reference
//             (domain.c:20-22).  If a target ever has to open a pruned node the walk raises a
//             device error instead of losing mass.
//   SPH       gas particles within the (padded) search radius of another shard's target groups, or
//             whose own smoothing sphere reaches them, are sent as GHOSTS before density(); their
//             updated records follow once between density() and hydro_force().
// Ownership is by Peano-Hilbert key range [splits[r], splits[r+1]): a cell lies wholly inside a
// shard's range or it is "shared" and always descended.
#include <hipcub/hipcub.hpp>

#include "ghip_internal.h"
#include "ghip_keys.h"

// tree.hip
int ghip_dd_build_gas_tree(ghip_ctx *ctx);
int ghip_dd_refresh_ghosts(ghip_ctx *ctx);
void ghip_dd_comm_release(ghip_ctx *ctx);
// sink.hip
int ghip_dd_sink_begin(ghip_ctx *ctx, int op);
int ghip_dd_sink_step(ghip_ctx *ctx);
int ghip_dd_pm_begin(ghip_ctx *ctx);   // ghip_pm.hip
int ghip_dd_pm_step(ghip_ctx *ctx);
extern "C" int ghip_dd_exchange(ghip_ctx *ctx);

#define DD_OP_MIGRATE 1
#define DD_OP_GRAVITY 2
#define DD_OP_DENSITY 3
#define DD_OP_HYDRO 4

// ---------------------------------------------------------------------------------------------
// set-up
// ---------------------------------------------------------------------------------------------
static int dd_store_segments(ghip_ctx *ctx, int nseg, const unsigned long long *keys, const int *owner);
extern "C" int ghip_dd_init(ghip_ctx *ctx, int rank, int nranks)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || nranks < 1 || nranks > GHIP_MAXRANKS || rank < 0 || rank >= nranks)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_init: need 0 <= rank < nranks <= %d", GHIP_MAXRANKS);
  DDState &D = ctx->dd;
  D.on = true;
  D.rank = rank;
  D.nranks = nranks;
  for(int r = 0; r <= nranks; r++)   // default: equal key ranges
    D.splits[r] = (r == nranks) ? (1ULL << 63) : ((1ULL << 63) / (unsigned long long) nranks) * r;
  D.gt_nimp = 0;
  D.nghost = 0;
  D.op = 0;
  D.x.kind = 0;
  ctx->gt.built = false;
  ctx->st.built = false;
  ctx->shard_rank = 0;   // (the replicated-source sharding of ghip_set_shard is a different mode)
  ctx->shard_n = 1;
  return ghip_dd_set_splits(ctx, D.splits);
}

extern "C" int ghip_dd_set_domain(ghip_ctx *ctx, const double corner[3], const double center[3],
                                  double len, const double soft[6])
{
  if(!ctx || !corner || !center || !soft || !(len > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_domain: bad arguments");
  for(int j = 0; j < 3; j++)
    {
      ctx->corner[j] = corner[j];
      ctx->center[j] = center[j];
    }
  ctx->dlen = len;
  for(int j = 0; j < 6; j++)
    ctx->soft[j] = soft[j];
  ctx->gt.built = false;
  ctx->st.built = false;
  return GHIP_OK;
}

// ownership of the curve: merge neighbours of one owner, drop empty pieces, keep device copies
static int dd_store_segments(ghip_ctx *ctx, int nseg, const unsigned long long *keys, const int *owner)
{
  DDState &D = ctx->dd;
  std::vector<unsigned long long> k, lo, hi;
  std::vector<int> o;
  for(int s = 0; s < nseg; s++)
    {
      if(keys[s + 1] == keys[s])
        continue;
      if(!o.empty() && o.back() == owner[s])
        continue;   // (same owner as the piece before: one piece)
      k.push_back(keys[s]);
      o.push_back(owner[s]);
    }
  if(k.empty() || k[0] != 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd: the key ranges do not start at 0");
  k.push_back(1ULL << 63);
  const int m = (int) o.size();
  for(int s = 0; s < m; s++)
    if(o[s] == D.rank)
      {
        lo.push_back(k[s]);
        hi.push_back(k[s + 1]);
      }
  D.nseg = m;
  D.nown = (int) lo.size();
  GCHK(ghip_ensure(ctx, D.segkey, (size_t) (m + 1) * 8));
  GCHK(ghip_ensure(ctx, D.segowner, (size_t) m * 4));
  GCHK(ghip_ensure(ctx, D.ownlo, (size_t) (D.nown > 0 ? D.nown : 1) * 8));
  GCHK(ghip_ensure(ctx, D.ownhi, (size_t) (D.nown > 0 ? D.nown : 1) * 8));
  HIPCHK(hipMemcpy(D.segkey.p, k.data(), (size_t) (m + 1) * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(D.segowner.p, o.data(), (size_t) m * 4, hipMemcpyHostToDevice));
  if(D.nown > 0)
    {
      HIPCHK(hipMemcpy(D.ownlo.p, lo.data(), (size_t) D.nown * 8, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(D.ownhi.p, hi.data(), (size_t) D.nown * 8, hipMemcpyHostToDevice));
    }
  return GHIP_OK;
}

extern "C" int ghip_dd_set_splits(ghip_ctx *ctx, const unsigned long long *splits)
{
  if(!ctx || !splits || !ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_splits: call ghip_dd_init first");
  DDState &D = ctx->dd;
  for(int r = 0; r <= D.nranks; r++)
    {
      if(r > 0 && splits[r] < splits[r - 1])
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_splits: keys must not decrease");
      D.splits[r] = splits[r];
    }
  if(D.splits[0] != 0 || D.splits[D.nranks] < (1ULL << 63))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_splits: the ranges must cover [0, 2^63)");
  std::vector<unsigned long long> keys(D.splits, D.splits + D.nranks + 1);
  std::vector<int> owner(D.nranks);
  for(int r = 0; r < D.nranks; r++)
    owner[r] = r;
  return dd_store_segments(ctx, D.nranks, keys.data(), owner.data());
}

extern "C" int ghip_dd_set_segments(ghip_ctx *ctx, int nseg, const unsigned long long *keys, const int *owner)
{
  if(!ctx || !keys || !owner || !ctx->dd.on || nseg < 1)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_segments: call ghip_dd_init first; nseg >= 1");
  DDState &D = ctx->dd;
  for(int s = 0; s < nseg; s++)
    {
      if(keys[s + 1] < keys[s])
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_segments: keys must not decrease");
      if(owner[s] < 0 || owner[s] >= D.nranks)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_segments: owner %d of segment %d is not a rank", owner[s], s);
    }
  if(keys[0] != 0 || keys[nseg] < (1ULL << 63))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_set_segments: the segments must cover [0, 2^63)");
  return dd_store_segments(ctx, nseg, keys, owner);
}

extern "C" int ghip_dd_set_ghost_margin(ghip_ctx *ctx, double margin)
{
  if(!ctx || !(margin >= 1.0))
    return GHIP_EINVAL;
  ctx->dd.gh_margin = margin;
  return GHIP_OK;
}

// domain_findSplit_work_balanced (domain.c:1075-1113) with equal speed factors: cut `ndomain`
// curve-ordered pieces of work into `ncpu` contiguous ranges so that the running sum follows the
// running average.  Pure host arithmetic (also what the CPU tests of the multi-rank logic call).
extern "C" int ghip_dd_find_split(int ncpu, int ndomain, const double *domainWork, int *start,
                                  int *end)
{
  if(ncpu < 1 || ndomain < ncpu || !domainWork || !start || !end)
    return GHIP_EINVAL;
  double work = 0;
  for(int i = 0; i < ndomain; i++)
    work += domainWork[i];
  const double workavg = work / ncpu;
  double work_before = 0, workavg_before = 0;
  int s = 0;
  for(int i = 0; i < ncpu; i++)
    {
      int e = s;
      work = domainWork[e];
      while((work + work_before < workavg + workavg_before) || (i == ncpu - 1 && e < ndomain - 1))
        {
          if((ndomain - e) > (ncpu - i))
            e++;
          else
            break;
          work += domainWork[e];
        }
      start[i] = s;
      end[i] = e;
      work_before += work;
      workavg_before += workavg;
      s = e + 1;
    }
  return GHIP_OK;
}

// Peano-Hilbert keys (21 bits per dimension, peano.c:300) of the resident particles in the domain
// cube of ghip_dd_set_domain / ghip_tree_build: what the decomposition orders and cuts
__global__ void k_dd_ph_keys(int n, const double *__restrict__ x, const double *__restrict__ y,
                             const double *__restrict__ z, double cx, double cy, double cz,
                             double fac, unsigned long long *__restrict__ key)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  key[i] = d_peano21(d_cell21(x[i], cx, fac), d_cell21(y[i], cy, fac), d_cell21(z[i], cz, fac));
}

extern "C" int ghip_dd_keys(ghip_ctx *ctx, unsigned long long *keys_host)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !keys_host || !(ctx->dlen > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_keys: set the domain first");
  const int n = ctx->n;
  if(n == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) n * 8));
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);
  k_dd_ph_keys<<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, x, x + n, x + 2 * (size_t) n, ctx->corner[0],
                                                      ctx->corner[1], ctx->corner[2], fac,
                                                      P<unsigned long long>(ctx->stage));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(keys_host, ctx->stage.p, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  return GHIP_OK;
}

// every local particle must lie in this shard's key range (the host's domain decomposition, or
// ghip's own migration, guarantees it): a stray would be counted in nobody's cells
// does [lo, hi) lie inside ONE of this rank's own pieces of the curve?
__device__ __forceinline__ bool d_owned_range(unsigned long long lo, unsigned long long hi, int nown,
                                              const unsigned long long *__restrict__ ownlo,
                                              const unsigned long long *__restrict__ ownhi)
{
  int a = 0, b = nown - 1, j = -1;   // largest j with ownlo[j] <= lo
  while(a <= b)
    {
      int mid = (a + b) >> 1;
      if(ownlo[mid] <= lo)
        {
          j = mid;
          a = mid + 1;
        }
      else
        b = mid - 1;
    }
  return j >= 0 && hi <= ownhi[j];
}

__global__ void k_dd_check_range(int n, const double *__restrict__ x, const double *__restrict__ y,
                                 const double *__restrict__ z, double cx, double cy, double cz,
                                 double fac, int nown, const unsigned long long *__restrict__ ownlo,
                                 const unsigned long long *__restrict__ ownhi,
                                 int *__restrict__ errword)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  // (a particle outside the domain cube: error 6, the host must give a fresh extent first)
  unsigned long long k = d_peano21(d_cell21(x[i], cx, fac, errword), d_cell21(y[i], cy, fac, errword),
                                   d_cell21(z[i], cz, fac, errword));
  if(!d_owned_range(k, k + 1, nown, ownlo, ownhi))
    *(volatile int *) errword = 5;
}

// ---------------------------------------------------------------------------------------------
// target groups
// ---------------------------------------------------------------------------------------------
// one wavefront per group of `gsz` consecutive targets of the curve-ordered list (indices of the
// gravity tree): positions in tree order, val (OldAcc or Hsml, host order) through perm
__global__ void __launch_bounds__(64)
k_dd_groups(int nt, int gsz, const int *__restrict__ tgt, const double *__restrict__ sx,
            const double *__restrict__ sy, const double *__restrict__ sz,
            const int *__restrict__ perm, const double *__restrict__ val, double margin,
            DDGroup *__restrict__ out)
{
  const int g = blockIdx.x, lane = threadIdx.x;
  const long long a0l = (long long) g * gsz;
  const int a0 = a0l > nt ? nt : (int) a0l;
  int a1 = (a0l + gsz > nt) ? nt : (int) (a0l + gsz);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  double vmin = 1e300, vmax = 0;
  for(int a = a0 + lane; a < a1; a += 64)
    {
      const int s = tgt[a];
      const double p[3] = {sx[s], sy[s], sz[s]};
      const double v = val[perm[s]];
      for(int k = 0; k < 3; k++)
        {
          lo[k] = p[k] < lo[k] ? p[k] : lo[k];
          hi[k] = p[k] > hi[k] ? p[k] : hi[k];
        }
      vmin = v < vmin ? v : vmin;
      vmax = v > vmax ? v : vmax;
    }
  for(int off = 32; off > 0; off >>= 1)
    {
      for(int k = 0; k < 3; k++)
        {
          double o = __shfl_xor(lo[k], off, 64);
          lo[k] = o < lo[k] ? o : lo[k];
          o = __shfl_xor(hi[k], off, 64);
          hi[k] = o > hi[k] ? o : hi[k];
        }
      double o = __shfl_xor(vmin, off, 64);
      vmin = o < vmin ? o : vmin;
      o = __shfl_xor(vmax, off, 64);
      vmax = o > vmax ? o : vmax;
    }
  if(lane == 0)
    {
      DDGroup G;
      if(a1 > a0)
        {
          G.cx = 0.5 * (lo[0] + hi[0]);
          G.cy = 0.5 * (lo[1] + hi[1]);
          G.cz = 0.5 * (lo[2] + hi[2]);
          // (half extents rounded up a little: the box must contain every target)
          G.ex = 0.5 * (hi[0] - lo[0]) * (1 + 1e-12) + 1e-14 * (fabs(G.cx) + 1e-300);
          G.ey = 0.5 * (hi[1] - lo[1]) * (1 + 1e-12) + 1e-14 * (fabs(G.cy) + 1e-300);
          G.ez = 0.5 * (hi[2] - lo[2]) * (1 + 1e-12) + 1e-14 * (fabs(G.cz) + 1e-300);
          G.amin = vmin;
          G.rmax = vmax * margin;
        }
      else
        {
          G.cx = G.cy = G.cz = 0;
          G.ex = G.ey = G.ez = -1;
          G.amin = 1e300;
          G.rmax = 0;
        }
      out[DD_NSUPER + g] = G;
    }
}

// super-group s = union of groups [s*DD_NSUB, (s+1)*DD_NSUB)
__global__ void __launch_bounds__(64) k_dd_supergroups(DDGroup *__restrict__ tab)
{
  const int s = blockIdx.x, lane = threadIdx.x;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  double vmin = 1e300, vmax = 0;
  if(lane < DD_NSUB)
    {
      const DDGroup G = tab[DD_NSUPER + s * DD_NSUB + lane];
      if(G.ex >= 0)
        {
          lo[0] = G.cx - G.ex;
          hi[0] = G.cx + G.ex;
          lo[1] = G.cy - G.ey;
          hi[1] = G.cy + G.ey;
          lo[2] = G.cz - G.ez;
          hi[2] = G.cz + G.ez;
          vmin = G.amin;
          vmax = G.rmax;
        }
    }
  for(int off = 32; off > 0; off >>= 1)
    {
      for(int k = 0; k < 3; k++)
        {
          double o = __shfl_xor(lo[k], off, 64);
          lo[k] = o < lo[k] ? o : lo[k];
          o = __shfl_xor(hi[k], off, 64);
          hi[k] = o > hi[k] ? o : hi[k];
        }
      double o = __shfl_xor(vmin, off, 64);
      vmin = o < vmin ? o : vmin;
      o = __shfl_xor(vmax, off, 64);
      vmax = o > vmax ? o : vmax;
    }
  if(lane == 0)
    {
      DDGroup G;
      if(hi[0] >= lo[0])
        {
          G.cx = 0.5 * (lo[0] + hi[0]);
          G.cy = 0.5 * (lo[1] + hi[1]);
          G.cz = 0.5 * (lo[2] + hi[2]);
          G.ex = 0.5 * (hi[0] - lo[0]) * (1 + 1e-12) + 1e-14 * (fabs(G.cx) + 1e-300);
          G.ey = 0.5 * (hi[1] - lo[1]) * (1 + 1e-12) + 1e-14 * (fabs(G.cy) + 1e-300);
          G.ez = 0.5 * (hi[2] - lo[2]) * (1 + 1e-12) + 1e-14 * (fabs(G.cz) + 1e-300);
          G.amin = vmin;
          G.rmax = vmax;
        }
      else
        {
          G.cx = G.cy = G.cz = 0;
          G.ex = G.ey = G.ez = -1;
          G.amin = 1e300;
          G.rmax = 0;
        }
      tab[s] = G;
    }
}

// target list `tgt` (indices of the gravity tree, curve order).  gas = false: groups carry the least
// OldAcc; gas = true: the largest smoothing length times the ghost margin.
static int build_groups(ghip_ctx *ctx, bool gas, const int *tgt, int nt)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, D.grp_own, (size_t) DD_STRIDE * sizeof(DDGroup)));
  DDGroup *tab = P<DDGroup>(D.grp_own);
  int gsz = (nt + DD_NGROUPS - 1) / DD_NGROUPS;
  if(gsz < 1)
    gsz = 1;
  k_dd_groups<<<DD_NGROUPS, 64, 0, st>>>(nt, gsz, tgt, P<double>(ctx->sx), P<double>(ctx->sy),
                                         P<double>(ctx->sz), P<int>(ctx->gt.perm),
                                         P<double>(ctx->f[gas ? GHIP_F_HSML : GHIP_F_OLDACC]),
                                         gas ? D.gh_margin_cur : 1.0, tab);
  k_dd_supergroups<<<DD_NSUPER, 64, 0, st>>>(tab);
  HIPCHK(hipGetLastError());
  // the status record behind the table: has anything gone wrong on this shard so far?  (The error,
  // if any, is kept: after the all-gather every shard fails, this one with its own message.)
  HIPCHK(ghip_stream_sync(ctx, st));
  D.local_err = ghip_check_device_errors(ctx);
  if(D.local_err != GHIP_OK)
    D.local_msg = ctx->err;
  DDGroup status;
  memset(&status, 0, sizeof(status));
  status.cx = D.local_err != GHIP_OK ? 1.0 : 0.0;
  HIPCHK(hipMemcpyAsync(tab + DD_TABLE, &status, sizeof(status), hipMemcpyHostToDevice, st));
  HIPCHK(ghip_stream_sync(ctx, st));   // (`status` lives on this frame)
  return GHIP_OK;
}

// after the all-gather of the group tables: did every shard get this far without an error?
__global__ void k_dd_collect_status(int nranks, const DDGroup *__restrict__ all, double *__restrict__ out)
{
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if(r < nranks)
    out[r] = all[(size_t) r * DD_STRIDE + DD_TABLE].cx;
}

static int check_group_status(ghip_ctx *ctx, const char *what)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks;
  GCHK(ghip_ensure(ctx, D.status_all, (size_t) GHIP_MAXRANKS * 2 * 8));
  k_dd_collect_status<<<1, 64, 0, st>>>(P_, P<DDGroup>(D.grp_all), P<double>(D.status_all));
  double flags[GHIP_MAXRANKS];
  HIPCHK(hipMemcpyAsync(flags, D.status_all.p, (size_t) P_ * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  if(D.local_err != GHIP_OK)
    return ghip_fail(ctx, D.local_err, "%s", D.local_msg.c_str());
  for(int r = 0; r < P_; r++)
    if(flags[r] != 0)
      return ghip_fail(ctx, GHIP_EDEVICE, "%s: shard %d reported an error while it prepared its target "
                       "groups (its own message says what); every shard stops here", what, r);
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// gravity: selection of the locally essential trees
// ---------------------------------------------------------------------------------------------
struct LetK
{
  double theta2;     // ErrTolTheta^2 (0: relative criterion)
  double errtol;     // ErrTolForceAcc
  double boxsize, boxhalf;
  int periodic, unequal;
  int nranks, me;
  int nown;                                   // this rank's own pieces of the curve
  const unsigned long long *ownlo, *ownhi;
};

// smallest possible squared distance between the point s and any point of the group's box
// (component-wise, with the nearest image when periodic: forcetree.c:2028-2032)
__device__ __forceinline__ double d_box_r2min(const DDGroup &G, double sx, double sy, double sz,
                                              const LetK &K)
{
  double d0 = sx - G.cx, d1 = sy - G.cy, d2 = sz - G.cz;
  if(K.periodic)
    {
      d0 = d_nearest(d0, K.boxsize, K.boxhalf);
      d1 = d_nearest(d1, K.boxsize, K.boxhalf);
      d2 = d_nearest(d2, K.boxsize, K.boxhalf);
    }
  d0 = fabs(d0) - G.ex;
  d1 = fabs(d1) - G.ey;
  d2 = fabs(d2) - G.ez;
  d0 = d0 > 0 ? d0 : 0;
  d1 = d1 > 0 ? d1 : 0;
  d2 = d2 > 0 ? d2 : 0;
  return d0 * d0 + d1 * d1 + d2 * d2;
}

// Can ANY target inside the group's box (with OldAcc >= G.amin) open this node under the rules of
// force_treeevaluate (forcetree.c:2074-2139) and its short-range / Ewald variants (which open a
// subset)?  Conservative: a 1e-9 slack absorbs the rounding of the per-target arithmetic.
__device__ __forceinline__ bool d_group_can_open(const DDGroup &G, const double4 &xm, const double4 &cl,
                                                 double aux, const LetK &K)
{
  if(G.ex < 0)
    return false;
  const double slack = 1.0 - 1.0e-9;
  const double r2 = d_box_r2min(G, xm.x, xm.y, xm.z, K) * slack;
  const double len = cl.w;
  if(K.theta2 != 0)
    {
      if(len * len >= r2 * K.theta2)   // forcetree.c:2076
        return true;
    }
  else
    {
      if(xm.w * len * len >= r2 * r2 * (K.errtol * G.amin))   // forcetree.c:2085
        return true;
      // forcetree.c:2093-2104: a target inside the 1.2*len box around the cell's centre (plain
      // coordinate differences, as the reference takes them)
      const double lim = 0.60 * len * (1.0 + 1.0e-9);
      if(fabs(cl.x - G.cx) - G.ex < lim && fabs(cl.y - G.cy) - G.ey < lim &&
         fabs(cl.z - G.cz) - G.ez < lim)
        return true;
    }
  // forcetree.c:2108-2139: a node that mixes softenings (or any node under
  // ADAPTIVE_GRAVSOFT_FORGAS) is opened by a softer target closer than the largest softening below
  if(K.unequal && aux < 0 && r2 < aux * aux)
    return true;
  return false;
}

__global__ void k_let_init(int nelem, unsigned long long allmask, unsigned long long *__restrict__ reach,
                           unsigned long long *__restrict__ sendm)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  reach[e] = (e == 0) ? allmask : 0ULL;   // element 0 is the root cell (or the only particle)
  sendm[e] = 0ULL;
}

__device__ __forceinline__ double d_bcast(double v, int j)
{
  return __shfl(v, j, 64);
}

// One level of the top-down pass: a node some rank reaches is either left as it is for that rank
// (pruned: sent as one element) or descended (its children become reachable).
// One wavefront per LET_EPW consecutive elements (the candidates of a wavefront are tested one after
// the other, so few per wavefront keep a level's launch short; all 64 lanes take part in every
// test).  Lane-parallel: which of them are nodes of this level
// that somebody reaches, and does the cell lie wholly inside this shard's key range.  Then, rank by
// rank: lane l holds super-group l of that rank's table in registers, and the candidates that rank
// reaches are tested one after the other against all 64 super-groups at once (and against the 16
// groups of up to four hit super-groups at a time).  Finally every candidate lane publishes its own
// result and marks its children.
#define LET_EPW 16
__global__ void __launch_bounds__(64)
k_let_level(int nelem, int level, const int4 *__restrict__ lk, const double4 *__restrict__ xm,
            const double4 *__restrict__ cl, const double *__restrict__ aux,
            const unsigned long long *__restrict__ skey, const DDGroup *__restrict__ groups, LetK K,
            unsigned long long *__restrict__ reach, unsigned long long *__restrict__ sendm)
{
  const int lane = threadIdx.x;
  const int e = blockIdx.x * LET_EPW + lane;
  bool cand = false;
  int4 me = make_int4(0, 0, 0, 0);
  unsigned long long r = 0;
  if(lane < LET_EPW && e < nelem)
    {
      me = lk[e];
      if(me.y == -(level + 1))
        {
          r = reach[e];
          cand = r != 0;
        }
    }
  if(__ballot(cand) == 0)
    return;
  // does the cell lie wholly inside this shard's key range?  (every octree cell is one contiguous
  // piece of the Peano-Hilbert curve)  If not, other shards hold particles of it too: its local
  // moments are partial and it must be descended for everybody.
  bool shared = true;
  double4 m4 = make_double4(0, 0, 0, 0), c4 = make_double4(0, 0, 0, 0);
  double a = 0;
  if(cand)
    {
      if(level > 0)
        {
          const int sh = 3 * (GHIP_BITS - level);
          const unsigned long long ph = d_peano_of_morton(skey[me.z]);
          const unsigned long long lo = (ph >> sh) << sh;
          const unsigned long long hi = lo + (1ULL << sh);
          shared = !d_owned_range(lo, hi, K.nown, K.ownlo, K.ownhi);
        }
      m4 = xm[e];
      c4 = cl[e];
      a = aux[e];
    }
  unsigned long long X = (cand && shared) ? r : 0ULL;
  const bool test = cand && !shared;
  for(int b = 0; b < K.nranks; b++)
    {
      unsigned long long cm = __ballot(test && ((r >> b) & 1ULL));
      if(cm == 0)
        continue;
      const DDGroup *tab = groups + (size_t) b * DD_STRIDE;
      const DDGroup Gs = tab[lane];
      while(cm)
        {
          const int j = __builtin_ctzll(cm);
          cm &= cm - 1;
          const double4 mj = make_double4(d_bcast(m4.x, j), d_bcast(m4.y, j), d_bcast(m4.z, j),
                                          d_bcast(m4.w, j));
          const double4 cj = make_double4(d_bcast(c4.x, j), d_bcast(c4.y, j), d_bcast(c4.z, j),
                                          d_bcast(c4.w, j));
          const double aj = d_bcast(a, j);
          unsigned long long sm = __ballot(d_group_can_open(Gs, mj, cj, aj, K));
          bool open = false;
          while(sm && !open)
            {
              int sg = -1;
              for(int q = 0; q < 4 && sm; q++)
                {
                  const int s0 = __builtin_ctzll(sm);
                  sm &= sm - 1;
                  if((lane >> 4) == q)
                    sg = s0;
                }
              const bool hit = sg >= 0 && d_group_can_open(tab[DD_NSUPER + sg * DD_NSUB + (lane & 15)],
                                                           mj, cj, aj, K);
              open = __ballot(hit) != 0;
            }
          if(open && lane == j)
            X |= 1ULL << b;
        }
    }
  if(cand)
    {
      sendm[e] = r & ~X;
      if(X)
        for(int c = e + 1; c < me.x;)
          {
            const int4 ck = lk[c];
            reach[c] = X;
            if(ck.y >= 0)
              sendm[c] = X;   // a particle that is reached is sent
            c = ck.x;
          }
    }
}

__global__ void k_let_single(int nelem, const int4 *__restrict__ lk,
                             const unsigned long long *__restrict__ reach,
                             unsigned long long *__restrict__ sendm)
{
  // a shard with one particle has no root cell: element 0 is that particle
  if(blockIdx.x == 0 && threadIdx.x == 0 && nelem > 0 && lk[0].y >= 0)
    sendm[0] = reach[0];
}

// ---------------------------------------------------------------------------------------------
// order-preserving selection of the items whose mask has bit b set, for every rank b at once:
// count per block of SELM_BLOCK items and rank, scan per rank, scatter the item indices
// ---------------------------------------------------------------------------------------------
#define SELM_ITEMS 16
#define SELM_BLOCK (64 * SELM_ITEMS)

__global__ void __launch_bounds__(64)
k_selm_count(int n, int nranks, const unsigned long long *__restrict__ mask, int nblk,
             int *__restrict__ blockcnt)
{
  __shared__ int cnt[GHIP_MAXRANKS];
  const int lane = threadIdx.x;
  cnt[lane] = 0;
  __syncthreads();
  const int base = blockIdx.x * SELM_BLOCK;
  for(int j = 0; j < SELM_ITEMS; j++)
    {
      const int a = base + j * 64 + lane;
      const unsigned long long m = a < n ? mask[a] : 0ULL;
      unsigned long long any = m;
      for(int off = 32; off > 0; off >>= 1)
        any |= __shfl_xor(any, off, 64);
      while(any)
        {
          const int b = __builtin_ctzll(any);
          any &= any - 1;
          const unsigned long long bal = __ballot((m >> b) & 1ULL);
          if(lane == 0)
            cnt[b] += __popcll(bal);
        }
    }
  __syncthreads();
  if(lane < nranks)
    blockcnt[(size_t) lane * nblk + blockIdx.x] = cnt[lane];
}

// one wavefront per rank: block counts -> exclusive offsets (in place), total[rank]
__global__ void __launch_bounds__(64) k_selm_scan(int nblk, int *__restrict__ blockcnt,
                                                  int *__restrict__ total)
{
  int *row = blockcnt + (size_t) blockIdx.x * nblk;
  int run = 0;
  for(int b0 = 0; b0 < nblk; b0 += 64)
    {
      const int b = b0 + threadIdx.x;
      const int c = b < nblk ? row[b] : 0;
      int incl = c;
      for(int o = 1; o < 64; o <<= 1)
        {
          const int v = __shfl_up(incl, o, 64);
          if((int) threadIdx.x >= o)
            incl += v;
        }
      if(b < nblk)
        row[b] = run + incl - c;
      run += __shfl(incl, 63, 64);
    }
  if(threadIdx.x == 0)
    total[blockIdx.x] = run;
}

struct SelOff
{
  int off[GHIP_MAXRANKS];   // first slot of each rank's list in the output
};

__global__ void __launch_bounds__(64)
k_selm_scatter(int n, int nranks, const unsigned long long *__restrict__ mask, int nblk,
               const int *__restrict__ blockoff, SelOff S, int *__restrict__ out)
{
  __shared__ int pos[GHIP_MAXRANKS];
  const int lane = threadIdx.x;
  pos[lane] = lane < nranks ? S.off[lane] + blockoff[(size_t) lane * nblk + blockIdx.x] : 0;
  __syncthreads();
  const int base = blockIdx.x * SELM_BLOCK;
  const unsigned long long below = (1ULL << lane) - 1ULL;
  for(int j = 0; j < SELM_ITEMS; j++)
    {
      const int a = base + j * 64 + lane;
      const unsigned long long m = a < n ? mask[a] : 0ULL;
      unsigned long long any = m;
      for(int off = 32; off > 0; off >>= 1)
        any |= __shfl_xor(any, off, 64);
      while(any)
        {
          const int b = __builtin_ctzll(any);
          any &= any - 1;
          const unsigned long long bal = __ballot((m >> b) & 1ULL);
          const int p0 = pos[b];
          if((m >> b) & 1ULL)
            out[p0 + __popcll(bal & below)] = a;
          __syncthreads();
          if(lane == 0)
            pos[b] = p0 + __popcll(bal);
          __syncthreads();
        }
    }
}

// lists[dest-major] of the items selected for each rank; counts/offsets (host) in scount/soff
static int multi_select(ghip_ctx *ctx, int n, const unsigned long long *mask, DevBuf &list,
                        int *scount, int *soff, int *total_out)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks;
  const int nblk = cdiv(n > 0 ? n : 1, SELM_BLOCK);
  GCHK(ghip_ensure(ctx, D.selcnt, ((size_t) P_ * nblk + GHIP_MAXRANKS) * 4));
  int *bc = P<int>(D.selcnt), *tot = bc + (size_t) P_ * nblk;
  k_selm_count<<<nblk, 64, 0, st>>>(n, P_, mask, nblk, bc);
  k_selm_scan<<<P_, 64, 0, st>>>(nblk, bc, tot);
  HIPCHK(hipGetLastError());
  int htot[GHIP_MAXRANKS];
  HIPCHK(hipMemcpyAsync(htot, tot, (size_t) P_ * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  SelOff S;
  int run = 0;
  for(int b = 0; b < GHIP_MAXRANKS; b++)
    {
      S.off[b] = run;
      if(b < P_)
        {
          scount[b] = htot[b];
          soff[b] = run;
          run += htot[b];
        }
    }
  *total_out = run;
  GCHK(ghip_ensure(ctx, list, (size_t) (run > 0 ? run : 1) * 4));
  if(run > 0)
    {
      k_selm_scatter<<<nblk, 64, 0, st>>>(n, P_, mask, nblk, bc, S, P<int>(list));
      HIPCHK(hipGetLastError());
    }
  return GHIP_OK;
}

// element list -> LetRec records
__global__ void k_let_pack(int nrec, const int *__restrict__ list, const int4 *__restrict__ lk,
                           const double4 *__restrict__ xm, const double *__restrict__ aux,
                           const unsigned long long *__restrict__ skey, LetRec *__restrict__ out)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nrec)
    return;
  const int e = list[a];
  const int4 k = lk[e];
  const double4 v = xm[e];
  LetRec r;
  r.x = v.x;
  r.y = v.y;
  r.z = v.z;
  r.m = v.w;
  r.aux = aux[e];
  r.pad = 0;
  r.spare = 0;
  if(k.y >= 0)
    {
      r.key = skey[k.y];
      r.level = 0;
    }
  else
    {
      const int L = -k.y - 1;
      const int sh = 63 - 3 * L;
      r.key = (skey[k.z] >> sh) << sh;
      r.level = L;
    }
  out[a] = r;
}

// ---------------------------------------------------------------------------------------------
// SPH: ghost selection
// ---------------------------------------------------------------------------------------------
struct GhostK
{
  double boxsize, boxhalf, margin;
  int periodic, nranks, me;
};

// gas particle (x, h): can it be a neighbour of a target in the group (r < padded search radius of
// the group), or can one of the group's targets lie inside ITS smoothing sphere (hydra.c:1266)?
__device__ __forceinline__ bool d_group_needs(const DDGroup &G, double x, double y, double z, double hj,
                                              const GhostK &K)
{
  if(G.ex < 0)
    return false;
  double d0 = x - G.cx, d1 = y - G.cy, d2 = z - G.cz;
  if(K.periodic)
    {
      d0 = d_nearest(d0, K.boxsize, K.boxhalf);
      d1 = d_nearest(d1, K.boxsize, K.boxhalf);
      d2 = d_nearest(d2, K.boxsize, K.boxhalf);
    }
  d0 = fabs(d0) - G.ex;
  d1 = fabs(d1) - G.ey;
  d2 = fabs(d2) - G.ez;
  d0 = d0 > 0 ? d0 : 0;
  d1 = d1 > 0 ? d1 : 0;
  d2 = d2 > 0 ? d2 : 0;
  const double R = (G.rmax > hj ? G.rmax : hj) * (1.0 + 1.0e-9);
  return d0 * d0 + d1 * d1 + d2 * d2 < R * R;
}

// distance test between a chunk of particles (box c +- e, largest padded radius hmax) and a group's box
__device__ __forceinline__ bool d_group_near_box(const DDGroup &G, double cx, double cy, double cz,
                                                 double ex, double ey, double ez, double hmax,
                                                 const GhostK &K)
{
  if(G.ex < 0)
    return false;
  double d0 = cx - G.cx, d1 = cy - G.cy, d2 = cz - G.cz;
  if(K.periodic)
    {
      d0 = d_nearest(d0, K.boxsize, K.boxhalf);
      d1 = d_nearest(d1, K.boxsize, K.boxhalf);
      d2 = d_nearest(d2, K.boxsize, K.boxhalf);
    }
  d0 = fabs(d0) - G.ex - ex;
  d1 = fabs(d1) - G.ey - ey;
  d2 = fabs(d2) - G.ez - ez;
  d0 = d0 > 0 ? d0 : 0;
  d1 = d1 > 0 ? d1 : 0;
  d2 = d2 > 0 ? d2 : 0;
  const double R = (G.rmax > hmax ? G.rmax : hmax) * (1.0 + 1.0e-9);
  return d0 * d0 + d1 * d1 + d2 * d2 < R * R;
}

// One wavefront per (chunk of 64 local gas particles that are neighbours in the gravity tree's
// (Morton) order, destination rank) -- `src` lists the particles as indices of that tree.  The chunk's
// bounding box is tested against the 64 super-groups of the rank by the 64 lanes at once; only for
// the super-groups it comes near are the 16 groups staged through LDS and every lane tests its own
// particle against the near ones.  Interior chunks -- nearly all of them -- cost one box test.
// mask[] must be zero on entry; the ranks' bits are OR-ed in.
__global__ void __launch_bounds__(64)
k_ghost_select(int nsrc, const int *__restrict__ src, const int *__restrict__ perm,
               const double *__restrict__ sx, const double *__restrict__ sy,
               const double *__restrict__ sz, const double *__restrict__ h,
               const DDGroup *__restrict__ groups, GhostK K, unsigned long long *__restrict__ mask,
               double *__restrict__ h0)
{
  __shared__ DDGroup sh[DD_NSUB];
  const int lane = threadIdx.x;
  const int nd = K.nranks - 1;                   // destinations per chunk
  const int chunk = blockIdx.x / nd;
  int b = blockIdx.x - chunk * nd;
  if(b >= K.me)
    b++;                                         // skip this rank itself
  const int a = chunk * 64 + lane;
  const bool valid = a < nsrc;
  int i = 0;
  double px = 0, py = 0, pz = 0, hj = 0;
  if(valid)
    {
      const int s = src[a];
      i = perm[s];
      px = sx[s];
      py = sy[s];
      pz = sz[s];
      hj = h[i] * K.margin;
      if(blockIdx.x == chunk * nd)
        h0[i] = h[i];
    }
  // chunk box around lane 0's particle (nearest-image offsets: a chunk is compact, the box may
  // straddle the periodic boundary)
  const double rx = __shfl(px, 0, 64), ry = __shfl(py, 0, 64), rz = __shfl(pz, 0, 64);
  double ox = valid ? px - rx : 0, oy = valid ? py - ry : 0, oz = valid ? pz - rz : 0;
  if(K.periodic)
    {
      ox = d_nearest(ox, K.boxsize, K.boxhalf);
      oy = d_nearest(oy, K.boxsize, K.boxhalf);
      oz = d_nearest(oz, K.boxsize, K.boxhalf);
    }
  double lo[3] = {ox, oy, oz}, hi[3] = {ox, oy, oz}, hm = hj;
  for(int off = 32; off > 0; off >>= 1)
    {
      for(int k = 0; k < 3; k++)
        {
          double o = __shfl_xor(lo[k], off, 64);
          lo[k] = o < lo[k] ? o : lo[k];
          o = __shfl_xor(hi[k], off, 64);
          hi[k] = o > hi[k] ? o : hi[k];
        }
      const double o = __shfl_xor(hm, off, 64);
      hm = o > hm ? o : hm;
    }
  const double cx = rx + 0.5 * (lo[0] + hi[0]), cy = ry + 0.5 * (lo[1] + hi[1]),
               cz = rz + 0.5 * (lo[2] + hi[2]);
  const double ex = 0.5 * (hi[0] - lo[0]) * (1 + 1e-12) + 1e-14, ey = 0.5 * (hi[1] - lo[1]) * (1 + 1e-12) + 1e-14,
               ez = 0.5 * (hi[2] - lo[2]) * (1 + 1e-12) + 1e-14;
  const DDGroup *tab = groups + (size_t) b * DD_STRIDE;
  unsigned long long sm = __ballot(d_group_near_box(tab[lane], cx, cy, cz, ex, ey, ez, hm, K));
  bool need = false;
  while(sm)
    {
      const int sg = __builtin_ctzll(sm);
      sm &= sm - 1;
      __syncthreads();
      bool near = false;
      if(lane < DD_NSUB)
        {
          const DDGroup G = tab[DD_NSUPER + sg * DD_NSUB + lane];
          sh[lane] = G;
          near = d_group_near_box(G, cx, cy, cz, ex, ey, ez, hm, K);
        }
      unsigned long long qm = __ballot(near);
      __syncthreads();
      while(qm)
        {
          const int q = __builtin_ctzll(qm);
          qm &= qm - 1;
          if(valid && !need && d_group_needs(sh[q], px, py, pz, hj, K))
            need = true;
        }
      if(__ballot(valid && !need) == 0)
        break;   // every particle of the chunk is a ghost there already
    }
  if(need)
    atomicOr(mask + i, 1ULL << b);
}

// GhostRec of local gas particle i (host order): the two records of the SPH kernels
// (ghip_tree.hip k_gather_gas)
__global__ void k_ghost_pack(int nrec, const int *__restrict__ list, int n, int ngas,
                             const double *__restrict__ pos, const double *__restrict__ mass,
                             const double *__restrict__ velpred, const double *__restrict__ h,
                             const double *__restrict__ pres, const double *__restrict__ rho,
                             const double *__restrict__ dhf, const double *__restrict__ divv,
                             const double *__restrict__ curl, const int *__restrict__ timebin,
                             const int *__restrict__ type, GhostRec *__restrict__ out)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nrec)
    return;
  const int i = list[a];
  GhostRec r;
  r.p[0] = pos[i];
  r.p[1] = pos[(size_t) n + i];
  r.p[2] = pos[2 * (size_t) n + i];
  r.p[3] = type[i] == 0 ? mass[i] : -1.0;   // (a converted particle of the gas block: nobody's neighbour)
  r.p[4] = velpred[i];
  r.p[5] = velpred[(size_t) ngas + i];
  r.p[6] = velpred[2 * (size_t) ngas + i];
  r.p[7] = h[i];
  const int tb = timebin[i];
  r.q[0] = pres[i];
  r.q[1] = rho[i];
  r.q[2] = dhf[i];
  r.q[3] = divv[i];
  r.q[4] = curl[i];
  r.q[5] = (double) (tb ? (1 << tb) : 0);   // hydra.c:966
  r.q[6] = 0;
  r.q[7] = 0;
  out[a] = r;
}

static int pack_ghosts(ghip_ctx *ctx, int total)
{
  DDState &D = ctx->dd;
  const int n = ctx->n, ng = ctx->ngas;
  GCHK(ghip_ensure(ctx, D.gh_send, (size_t) (total > 0 ? total : 1) * sizeof(GhostRec)));
  if(total == 0)
    return GHIP_OK;
  k_ghost_pack<<<cdiv(total, 256), 256, 0, ctx->stream>>>(
    total, P<int>(D.gh_list), n, ng, P<double>(ctx->f[GHIP_F_POS]), P<double>(ctx->f[GHIP_F_MASS]),
    P<double>(ctx->f[GHIP_F_VELPRED]), P<double>(ctx->f[GHIP_F_HSML]),
    P<double>(ctx->f[GHIP_F_PRESSURE]), P<double>(ctx->f[GHIP_F_DENSITY]),
    P<double>(ctx->f[GHIP_F_DHSMLFAC]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    P<double>(ctx->f[GHIP_F_CURLVEL]), P<int>(ctx->f[GHIP_F_TIMEBIN]), P<int>(ctx->f[GHIP_F_TYPE]),
    P<GhostRec>(D.gh_send));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// largest smoothing length any density evaluation of this call used, relative to the one the
// ghosts were selected with: max(right bracket, final h) / h0 over the local gas (host order)
__global__ void k_ghost_growth(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                               const double *__restrict__ hcur, const double *__restrict__ right,
                               const double *__restrict__ h0, unsigned long long *__restrict__ out)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  double r = 0;
  if(ti < nt)
    {
      const int s = tgt[ti];
      const double hm = right[s] > hcur[s] ? right[s] : hcur[s];
      r = hm / h0[perm[s]];
    }
  for(int off = 32; off > 0; off >>= 1)
    {
      double o = __shfl_xor(r, off, 64);
      r = o > r ? o : r;
    }
  if((threadIdx.x & 63) == 0 && r > 0)
    atomicMax(out, (unsigned long long) __double_as_longlong(r));   // positive doubles order like integers
}

// ---------------------------------------------------------------------------------------------
// the state machine: compute until the next exchange, exchange, continue
// ---------------------------------------------------------------------------------------------
#define set_allgather ghip_dd_set_allgather
#define set_alltoallv ghip_dd_set_alltoallv

// ---- gravity ------------------------------------------------------------------------------
static int gravity_step(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks;
  if(D.phase == 0)
    {
      // the shard's own tree (moments of the cells it owns) and its target groups
      GCHK(ghip_join_pair(ctx));
      D.gt_nimp = 0;
      if(ctx->n > 0)
        {
          const double *x = P<double>(ctx->f[GHIP_F_POS]);
          const double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);
          k_dd_check_range<<<cdiv(ctx->n, 256), 256, 0, st>>>(
            ctx->n, x, x + ctx->n, x + 2 * (size_t) ctx->n, ctx->corner[0], ctx->corner[1],
            ctx->corner[2], fac, D.nown, P<unsigned long long>(D.ownlo), P<unsigned long long>(D.ownhi),
            ghip_errword(ctx, GHIP_ERRW_TREE));
        }
      GCHK(ghip_tree_build_impl(ctx));
      GCHK(ghip_build_target_lists(ctx));
      GCHK(build_groups(ctx, false, P<int>(ctx->tg_grav), ctx->nt_grav));
      set_allgather(D, D.grp_own.p, (size_t) DD_STRIDE * sizeof(DDGroup), &D.grp_all);
      D.phase = 1;
      return 1;
    }
  if(D.phase == 1)
    {
      GCHK(check_group_status(ctx, "gravity"));   // (all shards together, see DD_STRIDE)
      // what can the others need of this tree?
      TreeDev &t = ctx->gt;
      int scount[GHIP_MAXRANKS], soff[GHIP_MAXRANKS], total = 0;
      for(int r = 0; r < GHIP_MAXRANKS; r++)
        scount[r] = soff[r] = 0;
      if(t.nelem > 0 && P_ > 1)
        {
          GCHK(ghip_ensure(ctx, D.reach, (size_t) t.nelem * 8));
          GCHK(ghip_ensure(ctx, D.sendm, (size_t) t.nelem * 8));
          LetK K;
          K.theta2 = D.gp.ErrTolTheta * D.gp.ErrTolTheta;
          K.errtol = D.gp.ErrTolForceAcc;
          K.boxsize = D.gp.BoxSize;
          K.boxhalf = 0.5 * D.gp.BoxSize;
          K.periodic = D.gp.periodic;
          K.unequal = D.gp.unequal_softenings || ctx->adaptive_gravsoft;
          K.nranks = P_;
          K.me = D.rank;
          K.nown = D.nown;
          K.ownlo = P<unsigned long long>(D.ownlo);
          K.ownhi = P<unsigned long long>(D.ownhi);
          unsigned long long all = (P_ >= 64) ? ~0ULL : ((1ULL << P_) - 1ULL);
          all &= ~(1ULL << D.rank);
          unsigned long long *reach = P<unsigned long long>(D.reach),
                             *sendm = P<unsigned long long>(D.sendm);
          k_let_init<<<cdiv(t.nelem, 256), 256, 0, st>>>(t.nelem, all, reach, sendm);
          for(int L = 0; L <= t.maxlevel; L++)
            k_let_level<<<cdiv(t.nelem, LET_EPW), 64, 0, st>>>(
              t.nelem, L, P<int4>(t.lk), P<double4>(t.xm), P<double4>(t.cl), P<double>(t.aux),
              P<unsigned long long>(t.skey), P<DDGroup>(D.grp_all), K, reach, sendm);
          k_let_single<<<1, 64, 0, st>>>(t.nelem, P<int4>(t.lk), reach, sendm);
          HIPCHK(hipGetLastError());
          GCHK(multi_select(ctx, t.nelem, sendm, D.let_list, scount, soff, &total));
        }
      GCHK(ghip_ensure(ctx, D.let_send, (size_t) (total > 0 ? total : 1) * sizeof(LetRec)));
      if(total > 0)
        {
          k_let_pack<<<cdiv(total, 256), 256, 0, st>>>(total, P<int>(D.let_list), P<int4>(t.lk),
                                                      P<double4>(t.xm), P<double>(t.aux),
                                                      P<unsigned long long>(t.skey),
                                                      P<LetRec>(D.let_send));
          HIPCHK(hipGetLastError());
        }
      D.let_sent = total;
      set_alltoallv(D, D.let_send.p, sizeof(LetRec), scount, soff, &D.let_recv);
      D.phase = 2;
      return 1;
    }
  if(D.phase == 2)
    {
      // one tree over the local particles and everything that was imported, then the walks
      D.gt_nimp = D.x.rtotal;
      GCHK(ghip_tree_build_impl(ctx));
      D.phase = 3;
      D.op = 0;
      return ghip_gravity_impl(ctx, &D.gp, D.walk);
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: gravity has no phase %d", D.phase);
}

// ---- density --------------------------------------------------------------------------------
// the local gas targets in curve order, as indices of the (merged) gravity tree
__global__ void k_flag_gas_targets(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                                   int ngas, int *__restrict__ flags)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a < nt)
    flags[a] = perm[tgt[a]] < ngas ? 1 : 0;
}

__global__ void k_mask_local_gas(int n, int ngas, const int *__restrict__ perm,
                                 unsigned long long *__restrict__ mask)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    mask[s] = perm[s] < ngas ? 1ULL : 0ULL;
}

static int density_step(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks;
  const int ng = ctx->ngas;
  if(D.phase == 0)
    {
      // groups of the gas targets (bounding box, padded search radius) from the curve order of the
      // gravity tree: positions in tree order, smoothing lengths through perm
      // (a density() that no gravity_tree() precedes -- init.c:791 -- orders the shard's own
      // particles itself: the tree of the local particles, nothing imported)
      if(!ctx->gt.built)
        {
          GCHK(ghip_join_pair(ctx));
          D.gt_nimp = 0;
          GCHK(ghip_tree_build_impl(ctx));
        }
      GCHK(ghip_build_target_lists(ctx));
      const int nt = ctx->nt_grav;
      int ngt = 0;
      GCHK(ghip_ensure(ctx, D.gas_tgt, (size_t) (nt > 0 ? nt : 1) * 4));
      if(nt > 0 && ng > 0)
        {
          GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) nt * 4));
          k_flag_gas_targets<<<cdiv(nt, 256), 256, 0, st>>>(nt, P<int>(ctx->tg_grav), P<int>(ctx->gt.perm),
                                                           ng, P<int>(ctx->dflags));
          int *dnum = reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 33);
          size_t tb = 0;
          HIPCHK(hipcub::DeviceSelect::Flagged(nullptr, tb, P<int>(ctx->tg_grav), P<int>(ctx->dflags),
                                               P<int>(D.gas_tgt), dnum, nt, st));
          GCHK(ghip_ensure(ctx, ctx->cubtmp, tb + 256));
          HIPCHK(hipcub::DeviceSelect::Flagged(ctx->cubtmp.p, tb, P<int>(ctx->tg_grav),
                                               P<int>(ctx->dflags), P<int>(D.gas_tgt), dnum, nt, st));
          HIPCHK(hipMemcpyAsync(&ngt, dnum, 4, hipMemcpyDeviceToHost, st));
          HIPCHK(ghip_stream_sync(ctx, st));
        }
      GCHK(build_groups(ctx, true, P<int>(D.gas_tgt), ngt));
      set_allgather(D, D.grp_own.p, (size_t) DD_STRIDE * sizeof(DDGroup), &D.grp_all);
      D.phase = 1;
      return 1;
    }
  if(D.phase == 1)
    {
      GCHK(check_group_status(ctx, "density"));
      // which local gas particles are ghosts where
      int total = 0;
      for(int r = 0; r < GHIP_MAXRANKS; r++)
        D.gh_scount[r] = D.gh_soff[r] = 0;
      GCHK(ghip_ensure(ctx, D.h0, (size_t) (ng > 0 ? ng : 1) * 8));
      if(ng > 0 && P_ > 1)
        {
          GCHK(ghip_ensure(ctx, D.gh_mask, (size_t) ng * 8));
          GhostK K;
          K.boxsize = D.dp.BoxSize;
          K.boxhalf = 0.5 * D.dp.BoxSize;
          K.periodic = D.dp.periodic;
          K.margin = D.gh_margin_cur;
          K.nranks = P_;
          K.me = D.rank;
          // the local gas particles in the gravity tree's order (chunks of 64 are compact)
          const int nsrc = ctx->gt.n;
          GCHK(ghip_ensure(ctx, D.sendm, (size_t) nsrc * 8));
          k_mask_local_gas<<<cdiv(nsrc, 256), 256, 0, st>>>(nsrc, ng, P<int>(ctx->gt.perm),
                                                           P<unsigned long long>(D.sendm));
          int c0[GHIP_MAXRANKS], o0[GHIP_MAXRANKS], nfound = 0;
          GCHK(multi_select(ctx, nsrc, P<unsigned long long>(D.sendm), D.gas_src, c0, o0, &nfound));
          if(nfound != ng)
            return ghip_fail(ctx, GHIP_EINVAL, "ghost selection: %d of %d gas particles in the tree",
                             nfound, ng);
          HIPCHK(hipMemsetAsync(D.gh_mask.p, 0, (size_t) ng * 8, st));
          k_ghost_select<<<cdiv(ng, 64) * (P_ - 1), 64, 0, st>>>(
            ng, P<int>(D.gas_src), P<int>(ctx->gt.perm), P<double>(ctx->sx), P<double>(ctx->sy),
            P<double>(ctx->sz), P<double>(ctx->f[GHIP_F_HSML]), P<DDGroup>(D.grp_all), K,
            P<unsigned long long>(D.gh_mask), P<double>(D.h0));
          HIPCHK(hipGetLastError());
          GCHK(multi_select(ctx, ng, P<unsigned long long>(D.gh_mask), D.gh_list, D.gh_scount,
                            D.gh_soff, &total));
        }
      else if(ng > 0)
        HIPCHK(hipMemcpyAsync(D.h0.p, ctx->f[GHIP_F_HSML].p, (size_t) ng * 8, hipMemcpyDeviceToDevice,
                              st));
      D.gh_sent = total;
      GCHK(pack_ghosts(ctx, total));
      set_alltoallv(D, D.gh_send.p, sizeof(GhostRec), D.gh_scount, D.gh_soff, &D.gh_recv);
      D.phase = 2;
      return 1;
    }
  if(D.phase == 2)
    {
      // gas tree over the local gas and the ghosts, the h iteration for the local targets
      D.nghost = D.x.rtotal;
      GCHK(ghip_dd_build_gas_tree(ctx));
      D.dens_rc = ghip_density_impl(ctx, &D.dp);
      if(D.dens_rc != GHIP_OK)
        D.dens_msg = ctx->err;
      // Every search radius used must have stayed inside the padded radius the ghosts were selected
      // with.  The reference re-exports a target at every h iteration, whatever its radius has become
      // (density.c:160-677); here the ghosts are fixed for the call, so a radius that outgrows the
      // padding -- the first density() of a run grows a poor guess by 1.26 per pass -- means: select the
      // ghosts again with a larger padding and repeat the call from the smoothing lengths it started
      // with.  The decision is taken by ALL shards on the all-gathered growth factors, so that they
      // repeat (or fail) together; a rank-local failure of the iteration travels the same way.
      double mine[2] = {0.0, 0.0};
      if(D.dens_rc == GHIP_OK && ctx->nt_gas > 0)
        {
          unsigned long long *dr = P<unsigned long long>(ctx->counters) + 34;
          HIPCHK(hipMemsetAsync(dr, 0, 8, st));
          k_ghost_growth<<<cdiv(ctx->nt_gas, 256), 256, 0, st>>>(
            ctx->nt_gas, P<int>(ctx->tg_gas), P<int>(ctx->st.perm), P<double>(ctx->dhcur),
            P<double>(ctx->dright), P<double>(D.h0), dr);
          HIPCHK(hipMemcpyAsync(&mine[0], dr, 8, hipMemcpyDeviceToHost, st));
          HIPCHK(ghip_stream_sync(ctx, st));
        }
      mine[1] = D.dens_rc != GHIP_OK ? 1.0 : 0.0;
      D.gh_growth = mine[0];
      GCHK(ghip_ensure(ctx, D.status_own, 16));
      HIPCHK(hipMemcpyAsync(D.status_own.p, mine, 16, hipMemcpyHostToDevice, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      set_allgather(D, D.status_own.p, 16, &D.status_all);
      D.phase = 25;
      return 1;
    }
  if(D.phase == 25)
    {
      double all[2 * GHIP_MAXRANKS];
      HIPCHK(hipMemcpyAsync(all, D.status_all.p, (size_t) P_ * 16, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      double worst = 0;
      int failed = -1;
      for(int r = 0; r < P_; r++)
        {
          if(all[2 * r] > worst)
            worst = all[2 * r];
          if(all[2 * r + 1] != 0 && failed < 0)
            failed = r;
        }
      if(worst > D.gh_margin_cur && P_ > 1)
        {
          // (a failed iteration of the incomplete attempt -- non-convergence next to missing ghosts --
          // is repeated like everything else)
          if(D.gh_retries >= 40)
            return ghip_fail(ctx, GHIP_ENOCONV, "density: the ghost radius grew %d times without covering "
                             "the smoothing lengths (largest growth %.3g)", D.gh_retries, worst);
          D.gh_retries++;
          D.gh_margin_cur = 1.26 * worst;   // one more pass of the reference's growth factor as head-room
          if(ng > 0)
            HIPCHK(hipMemcpyAsync(ctx->f[GHIP_F_HSML].p, D.h0.p, (size_t) ng * 8, hipMemcpyDeviceToDevice, st));
          D.phase = 0;
          return density_step(ctx);   // groups with the larger padding -> all-gather -> ...
        }
      if(failed >= 0)
        {
          if(D.dens_rc != GHIP_OK)
            return ghip_fail(ctx, D.dens_rc, "%s", D.dens_msg.c_str());
          return ghip_fail(ctx, GHIP_EDEVICE, "density: the h iteration failed on shard %d (its own message "
                           "says why); every shard stops here", failed);
        }
      // the ghosts' records as they are after density(): hydro_force needs h, rho, P, f, div, curl
      GCHK(pack_ghosts(ctx, D.gh_sent));
      set_alltoallv(D, D.gh_send.p, sizeof(GhostRec), D.gh_scount, D.gh_soff, &D.gh_recv);
      D.phase = 3;
      return 1;
    }
  if(D.phase == 3)
    {
      if(D.x.rtotal != D.nghost)
        return ghip_fail(ctx, GHIP_ECOMM, "ghost refresh: %d records, expected %d", D.x.rtotal, D.nghost);
      GCHK(ghip_dd_refresh_ghosts(ctx));
      D.phase = 4;
      D.op = 0;
      return GHIP_OK;
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: density has no phase %d", D.phase);
}

// ---- migration --------------------------------------------------------------------------------
// domain_exchange (domain.c:665-1060): after a drift some particles lie outside their shard's key
// range; each moves to the shard that owns its key, with every resident field.  One record per
// particle: 8-byte slots, the fields in enum order (ints widened), slot MIG_SLOTS-1 = 1 for gas.
#define MIG_SLOTS 40
struct MigRec
{
  unsigned long long s[MIG_SLOTS];
};
struct MigField
{
  void *p;
  int ncomp, isint, gas, slot;
};
struct MigTable
{
  int nf;
  MigField f[GHIP_F_COUNT];
};
struct MigSplits
{
  int nseg, me;
  const unsigned long long *key;   // [nseg + 1]
  const int *owner;                // [nseg]
};

void ghip_field_info(int f, int *gas, int *ncomp, int *isint);   // api.hip

static MigTable mig_table(DevBuf *bufs)
{
  MigTable T;
  T.nf = GHIP_F_COUNT;
  int slot = 0;
  for(int f = 0; f < GHIP_F_COUNT; f++)
    {
      ghip_field_info(f, &T.f[f].gas, &T.f[f].ncomp, &T.f[f].isint);
      T.f[f].p = bufs[f].p;
      T.f[f].slot = slot;
      slot += T.f[f].ncomp;
    }
  return T;   // slot <= MIG_SLOTS - 1 (checked by ghip_dd_begin)
}

__global__ void k_mig_dest(int n, const double *__restrict__ x, const double *__restrict__ y,
                           const double *__restrict__ z, double cx, double cy, double cz, double fac,
                           MigSplits S, unsigned long long *__restrict__ mask)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  // (clamped to the cube: a particle that left it goes to the owner of the boundary cell -- whose range
  // check then refuses it until the host has set a fresh extent, see d_cell21)
  const unsigned long long k = d_peano21(d_cell21(x[i], cx, fac), d_cell21(y[i], cy, fac), d_cell21(z[i], cz, fac));
  int lo = 0, hi = S.nseg - 1;   // largest s with key[s] <= k
  while(lo < hi)
    {
      int mid = (lo + hi + 1) >> 1;
      if(S.key[mid] <= k)
        lo = mid;
      else
        hi = mid - 1;
    }
  const int dest = S.owner[lo];
  mask[i] = (dest == S.me) ? 0ULL : (1ULL << dest);
}

__global__ void k_mig_pack(int nrec, const int *__restrict__ list, int n, int ngas, MigTable T,
                           MigRec *__restrict__ out)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nrec)
    return;
  const int i = list[a];
  MigRec &r = out[a];
  const bool isgas = i < ngas;
  for(int f = 0; f < T.nf; f++)
    {
      const MigField F = T.f[f];
      const size_t pitch = F.gas ? ngas : n;
      for(int c = 0; c < F.ncomp; c++)
        {
          unsigned long long v = 0;
          if(!F.gas || isgas)
            {
              if(F.isint)
                v = (unsigned long long) (unsigned int) reinterpret_cast<const int *>(F.p)[c * pitch + i];
              else
                v = reinterpret_cast<const unsigned long long *>(F.p)[c * pitch + i];
            }
          r.s[F.slot + c] = v;
        }
    }
  r.s[MIG_SLOTS - 1] = isgas ? 1ULL : 0ULL;
}

// keep / gas flags packed for ONE scan: low word counts gas, high word counts the other types
__global__ void k_mig_flags_old(int n, int ngas, const unsigned long long *__restrict__ mask,
                                unsigned long long *__restrict__ v)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    v[i] = mask[i] ? 0ULL : (i < ngas ? 1ULL : (1ULL << 32));
}

__global__ void k_mig_flags_new(int nrecv, const MigRec *__restrict__ rec,
                                unsigned long long *__restrict__ v)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(j < nrecv)
    v[j] = rec[j].s[MIG_SLOTS - 1] ? 1ULL : (1ULL << 32);
}

struct MigCounts
{
  int kg, ko, rg, ro;   // kept gas / other, received gas / other
};

__global__ void k_mig_move_old(int n, int ngas, const unsigned long long *__restrict__ mask,
                               const unsigned long long *__restrict__ rank, MigCounts C, MigTable A,
                               MigTable Bn)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n || mask[i])
    return;
  const bool isgas = i < ngas;
  const int nn = C.kg + C.ko + C.rg + C.ro, ngn = C.kg + C.rg;
  const unsigned long long rk = rank[i];
  const int j = isgas ? (int) (rk & 0xffffffffULL) : ngn + (int) (rk >> 32);
  for(int f = 0; f < A.nf; f++)
    {
      const MigField F = A.f[f], G = Bn.f[f];
      if(F.gas && !isgas)
        continue;
      const size_t po = F.gas ? ngas : n, pn = F.gas ? ngn : nn;
      for(int c = 0; c < F.ncomp; c++)
        {
          if(F.isint)
            reinterpret_cast<int *>(G.p)[c * pn + j] = reinterpret_cast<const int *>(F.p)[c * po + i];
          else
            reinterpret_cast<double *>(G.p)[c * pn + j] =
              reinterpret_cast<const double *>(F.p)[c * po + i];
        }
    }
}

__global__ void k_mig_move_new(int nrecv, const MigRec *__restrict__ rec,
                               const unsigned long long *__restrict__ rank, MigCounts C, MigTable Bn)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nrecv)
    return;
  const MigRec &r = rec[a];
  const bool isgas = r.s[MIG_SLOTS - 1] != 0;
  const int nn = C.kg + C.ko + C.rg + C.ro, ngn = C.kg + C.rg;
  const unsigned long long rk = rank[a];
  const int j = isgas ? C.kg + (int) (rk & 0xffffffffULL) : ngn + C.ko + (int) (rk >> 32);
  for(int f = 0; f < Bn.nf; f++)
    {
      const MigField G = Bn.f[f];
      if(G.gas && !isgas)
        continue;
      const size_t pn = G.gas ? ngn : nn;
      for(int c = 0; c < G.ncomp; c++)
        {
          const unsigned long long v = r.s[G.slot + c];
          if(G.isint)
            reinterpret_cast<int *>(G.p)[c * pn + j] = (int) (unsigned int) v;
          else
            reinterpret_cast<unsigned long long *>(G.p)[c * pn + j] = v;
        }
    }
}

static int scan_u64(ghip_ctx *ctx, const unsigned long long *in, unsigned long long *out, int n)
{
  size_t tb = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, n, ctx->stream));
  GCHK(ghip_ensure(ctx, ctx->cubtmp, tb + 256));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cubtmp.p, tb, in, out, n, ctx->stream));
  return GHIP_OK;
}

static int migrate_step(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks, n = ctx->n, ng = ctx->ngas;
  if(D.phase == 0)
    {
      GHIP_JOIN(ctx);
      int scount[GHIP_MAXRANKS], soff[GHIP_MAXRANKS], total = 0;
      for(int r = 0; r < GHIP_MAXRANKS; r++)
        scount[r] = soff[r] = 0;
      GCHK(ghip_ensure(ctx, D.mig_mask, (size_t) (n > 0 ? n : 1) * 8));
      if(n > 0 && P_ > 1)
        {
          MigSplits S;
          S.nseg = D.nseg;
          S.me = D.rank;
          S.key = P<unsigned long long>(D.segkey);
          S.owner = P<int>(D.segowner);
          const double *x = P<double>(ctx->f[GHIP_F_POS]);
          const double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);
          k_mig_dest<<<cdiv(n, 256), 256, 0, st>>>(n, x, x + n, x + 2 * (size_t) n, ctx->corner[0],
                                                   ctx->corner[1], ctx->corner[2], fac, S,
                                                   P<unsigned long long>(D.mig_mask));
          HIPCHK(hipGetLastError());
          GCHK(multi_select(ctx, n, P<unsigned long long>(D.mig_mask), D.mig_list, scount, soff, &total));
        }
      else if(n > 0)
        HIPCHK(hipMemsetAsync(D.mig_mask.p, 0, (size_t) n * 8, st));
      GCHK(ghip_ensure(ctx, D.mig_send, (size_t) (total > 0 ? total : 1) * sizeof(MigRec)));
      if(total > 0)
        {
          k_mig_pack<<<cdiv(total, 256), 256, 0, st>>>(total, P<int>(D.mig_list), n, ng,
                                                      mig_table(ctx->f), P<MigRec>(D.mig_send));
          HIPCHK(hipGetLastError());
        }
      D.mig_out = total;
      set_alltoallv(D, D.mig_send.p, sizeof(MigRec), scount, soff, &D.mig_recv);
      D.phase = 1;
      return 1;
    }
  if(D.phase == 1)
    {
      const int nrecv = D.x.rtotal;
      D.mig_in = nrecv;
      D.phase = 2;
      D.op = 0;
      if(nrecv == 0 && D.mig_out == 0)
        return GHIP_OK;   // nobody left, nobody came: the resident arrays stay as they are
      // new positions: kept gas, received gas, kept others, received others
      GCHK(ghip_ensure(ctx, D.mig_scan, (size_t) (n + nrecv + 2) * 16));
      unsigned long long *vo = P<unsigned long long>(D.mig_scan), *ro = vo + (n + 1),
                         *vn = ro + (n + 1), *rn = vn + (nrecv + 1);
      // (layout: flags_old[n+1], rank_old[n+1], flags_new[nrecv+1], rank_new[nrecv+1]; the extra
      // zero entry makes the last rank the total)
      HIPCHK(hipMemsetAsync(vo, 0, (size_t) (n + nrecv + 2) * 16, st));
      if(n > 0)
        k_mig_flags_old<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<unsigned long long>(D.mig_mask), vo);
      if(nrecv > 0)
        k_mig_flags_new<<<cdiv(nrecv, 256), 256, 0, st>>>(nrecv, P<MigRec>(D.mig_recv), vn);
      HIPCHK(hipGetLastError());
      GCHK(scan_u64(ctx, vo, ro, n + 1));
      GCHK(scan_u64(ctx, vn, rn, nrecv + 1));
      unsigned long long tot[2];
      HIPCHK(hipMemcpyAsync(&tot[0], ro + n, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&tot[1], rn + nrecv, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      MigCounts C;
      C.kg = (int) (tot[0] & 0xffffffffULL);
      C.ko = (int) (tot[0] >> 32);
      C.rg = (int) (tot[1] & 0xffffffffULL);
      C.ro = (int) (tot[1] >> 32);
      const int nn = C.kg + C.ko + C.rg + C.ro, ngn = C.kg + C.rg;
      for(int f = 0; f < GHIP_F_COUNT; f++)
        {
          int gas, ncomp, isint;
          ghip_field_info(f, &gas, &ncomp, &isint);
          size_t bytes = (size_t) (gas ? ngn : nn) * ncomp * (isint ? 4 : 8);
          size_t before = D.fshadow[f].cap;
          GCHK(ghip_ensure(ctx, D.fshadow[f], bytes));
          if(D.fshadow[f].cap != before)
            HIPCHK(hipMemsetAsync(D.fshadow[f].p, 0, D.fshadow[f].cap, st));
        }
      const MigTable A = mig_table(ctx->f), Bn = mig_table(D.fshadow);
      if(n > 0)
        k_mig_move_old<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<unsigned long long>(D.mig_mask), ro, C,
                                                     A, Bn);
      if(nrecv > 0)
        k_mig_move_new<<<cdiv(nrecv, 256), 256, 0, st>>>(nrecv, P<MigRec>(D.mig_recv), rn, C, Bn);
      HIPCHK(hipGetLastError());
      HIPCHK(ghip_stream_sync(ctx, st));
      for(int f = 0; f < GHIP_F_COUNT; f++)
        {
          DevBuf t = ctx->f[f];
          ctx->f[f] = D.fshadow[f];
          D.fshadow[f] = t;
        }
      ctx->n = nn;
      ctx->ngas = ngn;
      ctx->gt.built = false;
      ctx->st.built = false;
      ctx->nactive = -1;
      ctx->lists_dirty = ctx->gas_list_dirty = true;
      ctx->gas_types_unknown = true;   // (an arrival may be a converted particle of its old gas block)
      return GHIP_OK;
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: migration has no phase %d", D.phase);
}

extern "C" int ghip_dd_begin(ghip_ctx *ctx, int op, const void *params, int walk)
{
  if(!ctx || (!params && op != DD_OP_MIGRATE))
    return GHIP_EINVAL;
  DDState &D = ctx->dd;
  if(!D.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_begin: call ghip_dd_init first");
  if(!(ctx->dlen > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_begin: call ghip_dd_set_domain first");
  if(D.x.kind != 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_begin: an exchange is still pending");
  HIPCHK(hipSetDevice(ctx->device));
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  if(op == DD_OP_GRAVITY)
    {
      if(walk < 0 || walk > GHIP_WALK_NEWTON_EWALD)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_begin: unknown walk %d", walk);
      D.gp = *reinterpret_cast<const ghip_grav_params *>(params);
      D.walk = walk;
    }
  else if(op == DD_OP_DENSITY)
    {
      D.dp = *reinterpret_cast<const ghip_dens_params *>(params);
      D.gh_margin_cur = D.gh_margin;   // (grows within the call when a smoothing length outgrows it)
      D.gh_retries = 0;
      D.dens_rc = GHIP_OK;
    }
  else if(op == DD_OP_HYDRO)
    D.hp = *reinterpret_cast<const ghip_hydro_params *>(params);
  else if(op >= GHIP_DD_SINK_DENSITY && op <= GHIP_DD_BH_SWALLOW)
    {
      D.sink = *reinterpret_cast<const ghip_dd_sink_args *>(params);
      GCHK(ghip_dd_sink_begin(ctx, op));
    }
  else if(op == GHIP_DD_PM)
    {
      D.pm = *reinterpret_cast<const ghip_pm_params *>(params);
      GCHK(ghip_dd_pm_begin(ctx));
    }
  else if(op == DD_OP_MIGRATE)
    {
      static_assert(sizeof(MigRec) == MIG_SLOTS * 8, "MigRec layout");
      int slots = 0;
      for(int f = 0; f < GHIP_F_COUNT; f++)
        {
          int gas, ncomp, isint;
          ghip_field_info(f, &gas, &ncomp, &isint);
          slots += ncomp;
        }
      if(slots > MIG_SLOTS - 1)
        return ghip_fail(ctx, GHIP_EINVAL, "migration record too small for %d field slots", slots);
    }
  else
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_begin: unknown operation %d", op);
  D.op = op;
  D.phase = 0;
  D.bytes_sent[op] = 0;
  return GHIP_OK;
}

// 1: an exchange is pending (ghip_dd_exchange / ghip_dd_exchange_local), 0: the operation is
// complete, < 0: error
extern "C" int ghip_dd_step(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_EINVAL;
  DDState &D = ctx->dd;
  if(D.x.kind != 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: run the pending exchange first");
  HIPCHK(hipSetDevice(ctx->device));
  if(D.op >= GHIP_DD_SINK_DENSITY && D.op <= GHIP_DD_BH_SWALLOW)
    return ghip_dd_sink_step(ctx);
  if(D.op == GHIP_DD_PM)
    return ghip_dd_pm_step(ctx);
  if(D.op == DD_OP_MIGRATE)
    return migrate_step(ctx);
  if(D.op == DD_OP_GRAVITY)
    return gravity_step(ctx);
  if(D.op == DD_OP_DENSITY)
    return density_step(ctx);
  if(D.op == DD_OP_HYDRO)
    {
      // the ghosts' records are current since the end of density(): hydro_force is local
      D.op = 0;
      return ghip_hydro_impl(ctx, &D.hp);
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: no operation in progress");
}

// the whole operation over RCCL (one process per GPU)
extern "C" int ghip_dd_run(ghip_ctx *ctx, int op, const void *params, int walk)
{
  GCHK(ghip_dd_begin(ctx, op, params, walk));
  for(;;)
    {
      int r = ghip_dd_step(ctx);
      if(r <= 0)
        return r;
      GCHK(ghip_dd_exchange(ctx));
    }
}

extern "C" int ghip_dd_get_info(const ghip_ctx *ctx, long long out[16])
{
  if(!ctx || !out)
    return GHIP_EINVAL;
  const DDState &D = ctx->dd;
  for(int i = 0; i < 16; i++)
    out[i] = 0;
  out[0] = D.rank;
  out[1] = D.nranks;
  out[2] = D.gt_nimp;                  // elements imported into the gravity tree
  out[3] = D.let_sent;                 // elements this shard sent
  out[4] = D.nghost;                   // ghost gas particles imported
  out[5] = D.gh_sent;                  // ghosts sent
  out[6] = D.bytes_sent[DD_OP_GRAVITY];
  out[7] = D.bytes_sent[DD_OP_DENSITY];
  out[8] = (long long) (D.gh_growth * 1.0e6);
  out[9] = ctx->gt.nelem;
  out[10] = ctx->st.nelem;
  out[11] = ctx->n;
  out[12] = ctx->ngas;
  out[13] = D.mig_out;
  out[14] = D.mig_in;
  out[15] = D.bytes_sent[DD_OP_MIGRATE];
  return GHIP_OK;
}

void ghip_dd_release(ghip_ctx *ctx)
{
  if(!ctx)
    return;
  ghip_dd_comm_release(ctx);
  DDState &D = ctx->dd;
  DevBuf *bs[] = {&D.status_own, &D.status_all, &D.xstage, &D.grp_own, &D.grp_all, &D.reach, &D.sendm, &D.selcnt, &D.let_list,
                  &D.let_send, &D.let_recv, &D.src_x, &D.src_y, &D.src_z, &D.src_m, &D.src_aux,
                  &D.src_key, &D.src_lvl, &D.gh_mask, &D.gh_list, &D.gh_send, &D.gh_recv, &D.gsx,
                  &D.gsy, &D.gsz, &D.gsm, &D.gsh, &D.h0, &D.gas_tgt, &D.mig_mask, &D.mig_list,
                  &D.mig_send, &D.mig_recv, &D.mig_scan, &D.gas_src, &D.sk_send, &D.sk_all, &D.sk_part,
                  &D.sk_parts, &D.sk_work, &D.pm_all, &D.segkey, &D.segowner, &D.ownlo, &D.ownhi};
  for(DevBuf *b : bs)
    {
      if(b->p)
        (void) hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
  for(int f = 0; f < GHIP_F_COUNT; f++)
    {
      if(D.fshadow[f].p)
        (void) hipFree(D.fshadow[f].p);
      D.fshadow[f].p = nullptr;
      D.fshadow[f].cap = 0;
    }
}
