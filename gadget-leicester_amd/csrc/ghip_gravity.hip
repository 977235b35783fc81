// ghip_gravity.hip -- host drivers of the tree-walk gravity on gfx950 (kernels: ghip_walk.h).
//
// Replaces the active-list loop of gravity_tree() (gravtree.c:130-168) over
// force_treeevaluate() (forcetree.c:1797-2317), force_treeevaluate_shortrange()
// (forcetree.c:2330-2845) and force_treeevaluate_ewald_correction() (forcetree.c:2873-3204),
// the OldAcc / G post-pass (gravtree.c:381-403) and ewald_init() (forcetree.c:4402-4527).
#include <cmath>

#include <hipcub/hipcub.hpp>

#include "ghip_walk.h"

// the post-pass of gravity_tree (gravtree.c:362-403), in the reference's order:
//   comoving && !PERIODIC && !PMGRID: GravAccel += 0.5 Hubble^2 Omega0 / G * Pos   (:362-373)
//   OldAcc = |GravAccel|, under PMGRID |GravAccel + GravPM / G|                    (:375-391)
//   GravAccel *= G                                                                  (:398-403)
__global__ void k_grav_finish(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                              int n, double G, double comoving_fac, const double *__restrict__ pos,
                              const double *__restrict__ gravpm, double *__restrict__ oacc,
                              double *__restrict__ oldacc)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int i = perm[tgt[ti]];
  double a0 = oacc[i], a1 = oacc[(size_t) n + i], a2 = oacc[2 * (size_t) n + i];
  if(comoving_fac != 0)
    {
      a0 += comoving_fac * pos[i];
      a1 += comoving_fac * pos[(size_t) n + i];
      a2 += comoving_fac * pos[2 * (size_t) n + i];
    }
  double b0 = a0, b1 = a1, b2 = a2;
  if(gravpm)
    {
      b0 = a0 + gravpm[i] / G;
      b1 = a1 + gravpm[(size_t) n + i] / G;
      b2 = a2 + gravpm[2 * (size_t) n + i] / G;
    }
  oldacc[i] = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
  oacc[i] = a0 * G;
  oacc[(size_t) n + i] = a1 * G;
  oacc[2 * (size_t) n + i] = a2 * G;
}

// gravtree.c:470-483: GravAccel += OmegaLambda * Hubble^2 * Pos (vacuum energy in physical
// coordinates; non-periodic, non-comoving runs without a PM mesh), after the multiplication by G
__global__ void k_grav_vacuum(int nt, const int *__restrict__ tgt, const int *__restrict__ perm, int n,
                              double fac, const double *__restrict__ pos, double *__restrict__ oacc)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int i = perm[tgt[ti]];
  for(int j = 0; j < 3; j++)
    oacc[(size_t) j * n + i] += fac * pos[(size_t) j * n + i];
}

// softened direct summation (formula of forcetree.c:4273-4336), LDS-tiled, for accuracy checks
__global__ void __launch_bounds__(256)
k_grav_direct(int n, const double *__restrict__ sx, const double *__restrict__ sy,
              const double *__restrict__ sz, const double4 *__restrict__ xm,
              const double *__restrict__ ssoft, int nt, const int *__restrict__ tgt, GravK p,
              double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az)
{
  // sources in tree order: positions from the sorted planes, mass from xm[j].w (a per-particle
  // (x,y,z,m) array the caller gathers), softening from ssoft; 256 of them per LDS tile
  __shared__ double lx[256], ly[256], lz[256], lm[256], ls[256];
  int ti = blockIdx.x * 256 + threadIdx.x;
  bool valid = ti < nt;
  int s = valid ? tgt[ti] : 0;
  double px = sx[s], py = sy[s], pz = sz[s], hi = ssoft[s];
  double a0 = 0, a1 = 0, a2 = 0;
  for(int base = 0; base < n; base += 256)
    {
      int j = base + threadIdx.x;
      if(j < n)
        {
          lx[threadIdx.x] = sx[j];
          ly[threadIdx.x] = sy[j];
          lz[threadIdx.x] = sz[j];
          lm[threadIdx.x] = xm[j].w;
          ls[threadIdx.x] = ssoft[j];
        }
      __syncthreads();
      int lim = min(256, n - base);
      for(int q = 0; q < lim; q++)
        {
          double dx = lx[q] - px, dy = ly[q] - py, dz = lz[q] - pz;
          if(p.periodic)
            {
              dx = d_nearest(dx, p.boxsize, p.boxhalf);
              dy = d_nearest(dy, p.boxsize, p.boxhalf);
              dz = d_nearest(dz, p.boxsize, p.boxhalf);
            }
          double r2 = dx * dx + dy * dy + dz * dz;
          double h = hi;
          if(p.unequal && h < ls[q])
            h = ls[q];
          double r;
          double fac = d_grav_fac(lm[q], r2, h, h * h, r);
          a0 += dx * fac;
          a1 += dy * fac;
          a2 += dz * fac;
        }
      __syncthreads();
    }
  if(valid)
    {
      ax[ti] = a0;
      ay[ti] = a1;
      az[ti] = a2;
    }
}

__global__ void k_scatter_direct(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                                 const double *__restrict__ ax, const double *__restrict__ ay,
                                 const double *__restrict__ az, int n, double *__restrict__ oacc,
                                 int *__restrict__ ocost)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int i = perm[tgt[ti]];
  oacc[i] = ax[ti];
  oacc[(size_t) n + i] = ay[ti];
  oacc[2 * (size_t) n + i] = az[ti];
  ocost[i] = 0;
}

__global__ void k_pack_xyzm(int n, const double *__restrict__ x, const double *__restrict__ y,
                            const double *__restrict__ z, const int *__restrict__ perm,
                            const double *__restrict__ mass_host, double4 *__restrict__ out)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    out[s] = make_double4(x[s], y[s], z[s], mass_host[perm[s]]);
}

// ---------------------------------------------------------------------------------------------
// Ewald table (ewald_init / ewald_force, forcetree.c:4402-4527, 4727-4778): alpha = 2,
// |n|,|h| <= 4, one octant of (EN+1)^3 points at x = 0.5*(i,j,k)/EN, scaled by 1/BoxSize^2.
// Only the x component is tabulated: fcorry[i][j][k] = fcorrx[j][i][k] and fcorrz[i][j][k] =
// fcorrx[k][j][i] by the symmetry of the cubic lattice sums (equal up to the rounding of the
// summation order, ~1e-16), see d_ewald_interp.
// ---------------------------------------------------------------------------------------------
__global__ void k_ewald_table(double inv_box2, double *__restrict__ tab, double *__restrict__ brick)
{
  const int E1 = GHIP_EN + 1;
  int nidx = blockIdx.x * blockDim.x + threadIdx.x;
  if(nidx >= E1 * E1 * E1)
    return;
  int i = nidx / (E1 * E1), j = (nidx / E1) % E1, k = nidx % E1;
  double f0 = 0;
  if(i + j + k != 0)
    {
      const double alpha = 2.0;
      double x0 = 0.5 * ((double) i) / GHIP_EN, x1 = 0.5 * ((double) j) / GHIP_EN,
             x2 = 0.5 * ((double) k) / GHIP_EN;
      double r2 = x0 * x0 + x1 * x1 + x2 * x2;
      double rr = r2 * sqrt(r2);
      f0 += x0 / rr;
      for(int n0 = -4; n0 <= 4; n0++)
        for(int n1 = -4; n1 <= 4; n1++)
          for(int n2 = -4; n2 <= 4; n2++)
            {
              double d0 = x0 - n0, d1 = x1 - n1, d2 = x2 - n2;
              double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
              double val = erfc(alpha * r) + 2 * alpha * r / sqrt(M_PI) * exp(-alpha * alpha * r * r);
              double w = val / (r * r * r);
              f0 -= d0 * w;
            }
      for(int h0 = -4; h0 <= 4; h0++)
        for(int h1 = -4; h1 <= 4; h1++)
          for(int h2 = -4; h2 <= 4; h2++)
            {
              int hh = h0 * h0 + h1 * h1 + h2 * h2;
              if(hh > 0)
                {
                  double hdotx = x0 * h0 + x1 * h1 + x2 * h2;
                  double val = 2.0 / ((double) hh) * exp(-M_PI * M_PI * hh / (alpha * alpha)) *
                               sin(2 * M_PI * hdotx);
                  f0 -= h0 * val;
                }
            }
    }
  const double val = f0 * inv_box2;
  tab[nidx] = val;
  // brick-tiled copy (ghip_internal.h): the value's home brick, and the overlap slot of the brick before
  brick[ghip_ew_offset(i, j, k / 3, k % 3)] = val;
  if(k % 3 == 0 && k > 0)
    brick[ghip_ew_offset(i, j, k / 3 - 1, 3)] = val;
}

extern "C" int ghip_ewald_init(ghip_ctx *ctx, double BoxSize)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !(BoxSize > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_ewald_init: bad BoxSize");
  const int E1 = GHIP_EN + 1;
  const int nt = E1 * E1 * E1;
  GCHK(ghip_ensure(ctx, ctx->ewtab, (size_t) (nt + 2) * sizeof(double)));
  GCHK(ghip_ensure(ctx, ctx->ewbrick, (size_t) (GHIP_EW_DOUBLES + 2) * sizeof(double)));
  HIPCHK(hipMemsetAsync(ctx->ewbrick.p, 0, (size_t) (GHIP_EW_DOUBLES + 2) * sizeof(double), ctx->stream));
  k_ewald_table<<<cdiv(nt, 64), 64, 0, ctx->stream>>>(1.0 / (BoxSize * BoxSize),
                                                      P<double>(ctx->ewtab), P<double>(ctx->ewbrick));
  HIPCHK(hipGetLastError());
  ctx->ew_box = BoxSize;
  return GHIP_OK;
}

extern "C" int ghip_ewald_get_table(ghip_ctx *ctx, double *host)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !host || ctx->ew_box == 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_ewald_get_table: table not initialised");
  const int E1 = GHIP_EN + 1;
  const size_t nt = (size_t) E1 * E1 * E1;
  std::vector<double> tmp(nt);
  std::vector<double> brick((size_t) GHIP_EW_DOUBLES);
  HIPCHK(hipMemcpyAsync(tmp.data(), ctx->ewtab.p, nt * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(brick.data(), ctx->ewbrick.p, brick.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  // the brick-tiled copy (and its overlap slots) must hold the same numbers
  for(int i = 0; i < E1; i++)
    for(int j = 0; j < E1; j++)
      for(int k = 0; k < E1; k++)
        {
          const double v = tmp[((size_t) i * E1 + j) * E1 + k];
          if(brick[ghip_ew_offset(i, j, k / 3, k % 3)] != v ||
             (k % 3 == 0 && k > 0 && brick[ghip_ew_offset(i, j, k / 3 - 1, 3)] != v))
            return ghip_fail(ctx, GHIP_EDEVICE, "ewald table: the brick-tiled copy differs at (%d,%d,%d)",
                             i, j, k);
        }
  // the three tables of the reference (fcorrx, fcorry, fcorrz) out of the one stored
  for(int i = 0; i < E1; i++)
    for(int j = 0; j < E1; j++)
      for(int k = 0; k < E1; k++)
        {
          size_t q = ((size_t) i * E1 + j) * E1 + k;
          host[q] = tmp[q];
          host[nt + q] = tmp[((size_t) j * E1 + i) * E1 + k];
          host[2 * nt + q] = tmp[((size_t) k * E1 + j) * E1 + i];
        }
  return GHIP_OK;
}

static int prepare_tables(ghip_ctx *ctx, const ghip_grav_params *p, int walk, GravK &k)
{
  k.theta = p->ErrTolTheta;
  k.errtol = p->ErrTolForceAcc;
  k.boxsize = p->BoxSize;
  k.boxhalf = 0.5 * p->BoxSize;
  k.periodic = p->periodic;
  k.unequal = p->unequal_softenings || ctx->adaptive_gravsoft;   // the latter implies the former
  k.rcut = p->Rcut;
  k.rcut2 = p->Rcut * p->Rcut;
  k.asmthfac = (p->Asmth > 0) ? 0.5 / p->Asmth * (GHIP_NTAB / 3.0) : 0;  // forcetree.c:2378
  k.fac_intp = (p->BoxSize > 0) ? 2 * GHIP_EN / p->BoxSize : 0;
  k.debug_steps = getenv("GHIP_DEBUG_STEPS") ? 1 : 0;
  k.k1875 = 1.875;
  if(ctx->dd.on)
    {
      // where a walk reports that it had to open an imported pruned node (d_walk_errw)
      int *w = ghip_errword(ctx, GHIP_ERRW_LET);
      HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(d_walk_errw), &w, sizeof(w), 0, hipMemcpyHostToDevice,
                                    ctx->stream));
    }
  // XCD-contiguous block order: off by default -- with the adaptive plan the grid is sized for
  // the worst case and its idle tail would all land on the last XCD (measured: Ewald walk 8.4 vs
  // 7.7 ms); the L2 locality it buys was within noise (12.9 vs 13.1 ms)
  k.xcd_remap = getenv("GHIP_WALK_XCD") ? atoi(getenv("GHIP_WALK_XCD")) : 0;
  if(k.xcd_remap < 0)
    k.xcd_remap = 0;
  if(walk == GHIP_WALK_SHORTRANGE)
    {
      if(!(p->Asmth > 0) || !(p->Rcut > 0))
        return ghip_fail(ctx, GHIP_EINVAL, "shortrange walk needs Rcut, Asmth > 0");
      if(!ctx->srtab_ready)
        {
          // forcetree.c:4195-4202 (float table, as in the reference)
          float tab[GHIP_NTAB];
          for(int i = 0; i < GHIP_NTAB; i++)
            {
              double u = 3.0 / GHIP_NTAB * (i + 0.5);
              tab[i] = (float) (erfc(u) + 2.0 * u / sqrt(M_PI) * exp(-u * u));
            }
          GCHK(ghip_ensure(ctx, ctx->srtab, sizeof(tab)));
          HIPCHK(hipMemcpyAsync(ctx->srtab.p, tab, sizeof(tab), hipMemcpyHostToDevice,
                                ctx->stream));
          HIPCHK(ghip_stream_sync(ctx, ctx->stream));
          ctx->srtab_ready = true;
        }
    }
  if(walk == GHIP_WALK_EWALD)
    {
      if(!(p->BoxSize > 0))
        return ghip_fail(ctx, GHIP_EINVAL, "ewald walk needs BoxSize > 0");
      if(ctx->ew_box != p->BoxSize)
        GCHK(ghip_ewald_init(ctx, p->BoxSize));
    }
  return GHIP_OK;
}

// segment table of the gravity tree; called at the end of every tree build
int ghip_build_segments(ghip_ctx *ctx, TreeDev &t, bool walk_records)
{
  t.ns = 1;
  if(t.n == 0)
    return GHIP_OK;
  // >= 2048 elements per segment, at most 256 segments: the segment is the unit of work a
  // wavefront cannot split, and a bucket spends most of its steps in the one or two segments that
  // hold its own neighbourhood (measured at c2: 64 -> 256 segments: 12.2 -> 10.8 ms)
  // (t.nelem: exact after a synchronous build, the last verified build's after an asynchronous one --
  // any granularity is valid, the kernel cuts the list the device knows)
  int ns = t.nelem / 2048;
  if(ns < 1)
    ns = 1;
  if(ns > 256)
    ns = 256;
  if(getenv("GHIP_WALK_SEGMENTS"))
    {
      int v = atoi(getenv("GHIP_WALK_SEGMENTS"));
      if(v >= 1 && v <= 4096)
        ns = v;
    }
  // three tables: fine (ns), mid (48, or 96), coarse (16, or 32).  The fine one serves small launches (one rank's
  // share of a multi-GPU run: many wavefronts per bucket), the mid one a full-size Newtonian walk
  // (enough buckets to fill the chip, so fewer ancestor replays win), the coarse one the Ewald walk,
  // which visits ~6x fewer elements per bucket.  48 / 16 were measured with the round-2 kernels (a
  // cheaper element visit makes the ancestor replays of a segment entry relatively dearer): step at
  // c2 9.19 -> 8.58 ms, at c4's size on one GPU 81.7 -> 76.1 ms, against ns/2 = 128 and ns/4 = 64.
  // A full-size shard of a larger box (c4 cut in 8: 8192 buckets each over a merged tree) is the one
  // case measured that wants them finer: its heaviest buckets need more wavefronts than 48 segments
  // allow (slowest shard's pair 12.9 ms with 48 / 16, 11.4 ms with 96 / 32; c2 cut in 2 -- 4096
  // buckets each -- keeps 48 / 16: 5.17 against 5.40 ms).
  const bool big_shard = ctx->dd.on && ctx->n >= 6144 * 64;
  const int mid = big_shard ? 96 : 48, coarse = big_shard ? 32 : 16;
  SegTables T;
  T.ntab = 3;
  int so = 0, no = 0, total = 0;
  for(int j = 0; j < 3; j++)
    {
      int v = j == 0 ? ns : (j == 1 ? mid : coarse);
      if(j == 1 && getenv("GHIP_WALK_SEG_MID"))
        v = atoi(getenv("GHIP_WALK_SEG_MID"));
      if(j == 2 && getenv("GHIP_WALK_SEG_COARSE"))
        v = atoi(getenv("GHIP_WALK_SEG_COARSE"));
      if(v > ns)
        v = ns;
      T.ns[j] = v < 1 ? 1 : v;
      T.soff[j] = so;
      T.noff[j] = no;
      so += T.ns[j] + 1;
      no += T.ns[j];
      total += T.ns[j] + 1;
    }
  t.ns = ns;
  for(int j = 0; j < 3; j++)
    {
      t.seg_ns[j] = T.ns[j];
      t.seg_soff[j] = T.soff[j];
      t.seg_noff[j] = T.noff[j];
    }
  GCHK(ghip_ensure(ctx, t.seg_start, (size_t) so * 4));
  GCHK(ghip_ensure(ctx, t.seg_nanc, (size_t) no * 4));
  GCHK(ghip_ensure(ctx, t.seg_anc, (size_t) no * GHIP_MAXANC * 4));
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  k_build_segments<<<cdiv(total, 64), 64, 0, ctx->stream>>>(
    ts, P<int4>(t.lk), T, P<int>(t.seg_start), P<int>(t.seg_nanc), P<int>(t.seg_anc));
  HIPCHK(hipGetLastError());
  if(!walk_records)
    return GHIP_OK;
  return ghip_fill_walk_records(ctx, t);
}

// the walk's 64-byte records from the element arrays (again after the arrays changed: the drifted tree)
int ghip_fill_walk_records(ghip_ctx *ctx, TreeDev &t)
{
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  const long long ce = (long long) t.n + t.cap_nodes;   // elements the buffers hold
  if(ce + 2 >= (1LL << 26))
    return ghip_fail(ctx, GHIP_EINVAL, "the walk addresses its 64-byte element records by a 32-bit byte "
                     "offset: room for %lld elements is more than 2^26", ce);
  GCHK(ghip_ensure(ctx, t.mq, (size_t) (ce + 1) * sizeof(WalkHot)));
  GCHK(ghip_ensure(ctx, t.mq2, (size_t) (ce + 1) * sizeof(WalkCold)));
  k_fill_elems<<<cdiv(ce, 256), 256, 0, ctx->stream>>>(ts, P<double4>(t.xm), P<double4>(t.cl),
                                                      P<int4>(t.lk), P<double>(t.aux),
                                                      P<WalkHot>(t.mq), P<WalkCold>(t.mq2));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// table 0 fine, 1 mid, 2 coarse (ghip_build_segments)
int ghip_walk_layout(const TreeDev &t, WalkSeg &sg, int table)
{
  sg.ns = t.seg_ns[table];
  sg.nsub = sg.ns < GHIP_MAXSUB ? sg.ns : GHIP_MAXSUB;
  if(getenv("GHIP_WALK_SUBS"))
    {
      int v = atoi(getenv("GHIP_WALK_SUBS"));
      if(v >= 1 && v <= sg.ns)
        sg.nsub = v;
    }
  if(sg.nsub < 1)
    sg.nsub = 1;
  sg.start = P<int>(t.seg_start) + t.seg_soff[table];
  sg.nanc = P<int>(t.seg_nanc) + t.seg_noff[table];
  sg.anc = P<int>(t.seg_anc) + (size_t) t.seg_noff[table] * GHIP_MAXANC;
  return sg.nsub;
}

static int walk_layout(const TreeDev &t, int nt, WalkSeg &sg, int *nbuckets, bool ewald)
{
  *nbuckets = (nt + 63) / 64;
  static int mid_min = -1;
  if(mid_min < 0)
    {
      mid_min = 3072;   // buckets from which the walks use the coarser tables (c2: 2 shards yes, 4 no)
      if(getenv("GHIP_WALK_MID_MIN"))
        mid_min = atoi(getenv("GHIP_WALK_MID_MIN"));
    }
  // (a small launch -- one rank's share -- needs more wavefronts per bucket than the coarser table has
  // segments: one table finer for both walks)
  int table = ewald ? (*nbuckets >= mid_min ? 2 : 1) : (*nbuckets >= mid_min ? 1 : 0);
  return ghip_walk_layout(t, sg, table);
}

// per-wavefront partial results; slot 1 = second set (the Ewald walk of an overlapped pair)
struct PartialBufs
{
  double *ax, *ay, *az;
  int *cost;
};

static int ensure_partials(ghip_ctx *ctx, int nwaves, int slot, PartialBufs &pb)
{
  size_t cnt = (size_t) nwaves * 64;
  DevBuf &bx = slot ? ctx->tax2 : ctx->tax, &by = slot ? ctx->tay2 : ctx->tay,
         &bz = slot ? ctx->taz2 : ctx->taz, &bc = slot ? ctx->tcost2 : ctx->tcost;
  GCHK(ghip_ensure(ctx, bx, cnt * 8));
  GCHK(ghip_ensure(ctx, by, cnt * 8));
  GCHK(ghip_ensure(ctx, bz, cnt * 8));
  GCHK(ghip_ensure(ctx, bc, cnt * 4));
  pb.ax = P<double>(bx);
  pb.ay = P<double>(by);
  pb.az = P<double>(bz);
  pb.cost = P<int>(bc);
  return GHIP_OK;
}

// wavefront plan of one walk call (see k_plan_nsub): kind 0 Newton/short-range, 1 Ewald,
// 2 external targets (never has history)
static int build_plan(ghip_ctx *ctx, int kind, int nb, int ns, WalkPlan &plan, int slot = 0,
                      hipStream_t on = nullptr)
{
  hipStream_t st = on ? on : ctx->stream;
  DevBuf &b_nsub = slot ? ctx->plan_nsub2 : ctx->plan_nsub, &b_woff = slot ? ctx->plan_woff2 : ctx->plan_woff,
         &b_wave = slot ? ctx->plan_wave2 : ctx->plan_wave,
         &b_tmp = slot ? ctx->cubtmp2 : (st == ctx->stream3 && st != ctx->stream ? ctx->cubtmp3 : ctx->cubtmp);
  int sbase = (49152 + nb - 1) / nb;
  sbase = sbase < 8 ? 8 : (sbase > 64 ? 64 : sbase);
  if(sbase > ns)
    sbase = ns;
  if(getenv("GHIP_WALK_SUBS"))
    {
      int v = atoi(getenv("GHIP_WALK_SUBS"));
      if(v >= 1 && v <= ns)
        sbase = v;
    }
  const int maxwaves = (sbase + 2) * nb + 8;
  GCHK(ghip_ensure(ctx, b_nsub, (size_t) nb * 4));
  GCHK(ghip_ensure(ctx, b_woff, (size_t) (nb + 1) * 4));
  GCHK(ghip_ensure(ctx, b_wave, (size_t) maxwaves * 4));
  // per-bucket visits of the previous call and of this one; allocated with headroom, since growing
  // a buffer drops what it held (the history is then simply not used for this call)
  int cur = ctx->plan_cur[kind];
  bool prev_kept = ctx->plan_steps[kind][cur].p != nullptr;
  const size_t steps_bytes = ((size_t) nb + nb / 8 + 64) * 4;
  for(int k = 0; k < 2; k++)
    if(ctx->plan_steps[kind][k].cap < (size_t) nb * 4)
      {
        GCHK(ghip_ensure(ctx, ctx->plan_steps[kind][k], steps_bytes));
        if(k == cur)
          prev_kept = false;
      }
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  const bool adaptive = !(getenv("GHIP_WALK_ADAPTIVE") && atoi(getenv("GHIP_WALK_ADAPTIVE")) == 0);
  // history of the kind's previous call: usable when that call had about as many buckets (a
  // shard's particle number changes with every migration, the merged tree's segment count may too;
  // neither moves the targets along the curve by more than a few buckets)
  const int pnb = ctx->plan_nb[kind];
  int tol = nb / 16 < 8 ? 8 : nb / 16;
  if(getenv("GHIP_PLAN_EXACT") && atoi(getenv("GHIP_PLAN_EXACT")))
    tol = ctx->plan_ns[kind] == ns ? 0 : -1;
  bool have_prev = adaptive && kind < 2 && pnb > 0 && (pnb > nb ? pnb - nb : nb - pnb) <= tol && prev_kept;
  int nprev = have_prev ? (pnb < nb ? pnb : nb) : 0;
  if(getenv("GHIP_PLAN_DEBUG") && kind < 2)
    {
      static int calls[2], used[2];
      calls[kind]++;
      used[kind] += have_prev;
      if(calls[kind] % 16 == 0)
        fprintf(stderr, "ghip plan kind %d: history used in %d of %d calls (nb %d, before %d)\n", kind,
                used[kind], calls[kind], nb, pnb);
    }
  unsigned int *prev = P<unsigned int>(ctx->plan_steps[kind][cur]);
  unsigned int *out = P<unsigned int>(ctx->plan_steps[kind][cur ^ 1]);
  k_plan_nsub<<<cdiv(nb, 256), 256, 0, st>>>(nb, ns, sbase, prev, nprev, P<int>(b_nsub));
  size_t tb = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, P<int>(b_nsub),
                                          P<int>(b_woff), nb, st));
  GCHK(ghip_ensure(ctx, b_tmp, tb + 256));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, P<int>(b_nsub),
                                          P<int>(b_woff), nb, st));
  HIPCHK(hipMemsetAsync(b_wave.p, 0xff, (size_t) maxwaves * 4, st));
  k_plan_fill<<<cdiv(nb, 256), 256, 0, st>>>(nb, P<int>(b_nsub), P<int>(b_woff),
                                             maxwaves, P<int>(b_wave), out,
                                             ghip_errword(ctx, GHIP_ERRW_PLAN));
  HIPCHK(hipGetLastError());
  if(kind < 2)
    {
      ctx->plan_nb[kind] = nb;
      ctx->plan_ns[kind] = ns;
      ctx->plan_cur[kind] = cur ^ 1;
    }
  plan.nwaves = maxwaves;
  plan.wave_bucket = P<int>(b_wave);
  plan.woff = P<int>(b_woff);
  plan.nsub = P<int>(b_nsub);
  plan.steps_out = out;
  plan.started = nullptr;
  plan.started_at = 0;
  return GHIP_OK;
}

template <int MODE>
static void launch_walk(ghip_ctx *ctx, const TreeDev &t, const WalkSeg &sg, int nbuckets, int nt,
                        const int *tgt, const double *tx, const double *ty, const double *tz,
                        const double *tsoft, const double *toldacc, const GravK &k,
                        unsigned long long *counter, const WalkPlan &plan, const PartialBufs &pb,
                        hipStream_t stream)
{
  long long nthreads = (long long) plan.nwaves * 64;
  // one wavefront per workgroup: wavefronts of a bucket finish at very different times, and a
  // 256-thread workgroup would hold its four slots until the slowest one is done
  int bsize = 64;
  if(getenv("GHIP_WALK_BLOCK"))
    {
      int v = atoi(getenv("GHIP_WALK_BLOCK"));
      if(v == 64 || v == 128 || v == 256)
        bsize = v;
    }
  int blocks = cdiv(nthreads, bsize);
  blocks = (blocks + 7) & ~7;   // whole number of blocks per XCD (see k_grav_walk)
  if(k.xcd_remap > 1)           // ... and of chunks per XCD, so that the chunked order covers every wavefront
    blocks = ((blocks + 8 * k.xcd_remap - 1) / (8 * k.xcd_remap)) * (8 * k.xcd_remap);
  WalkPlan pl = plan;
  if(pl.started)
    {
      static double frac = -1;
      if(frac < 0)
        frac = getenv("GHIP_HYDRO_TRIGGER_AT") ? atof(getenv("GHIP_HYDRO_TRIGGER_AT")) : 1.0;
      pl.started_at = (int) (frac * (blocks - 1));
      if(pl.started_at > blocks - 1 || pl.started_at < 0)
        pl.started_at = blocks - 1;
    }
  // In a pair the Newtonian walk must leave room for the Ewald walk: at its full 8 wavefronts per
  // SIMD it owns the whole register file and the two kernels merely follow each other.  (Unused)
  // dynamic LDS per one-wavefront workgroup caps it: 8 KB = 20 per CU = 5 per SIMD (13.0 -> 11.4 ms
  // per step at c2 when it was introduced); 10 KB = 4 per SIMD since the round-3 instruction diet of
  // the visit made the Newtonian wavefronts quicker (same build, c2: 8.59 ms with 8 KB -- the Ewald
  // walk then ends after the Newtonian one and hydro after both -- 8.43 with 10 KB; the LDS
  // allocation granularity leaves nothing in between).
  // (which of the two suits depends on the workload: at c2 10 KB balances the pair, at c4's size 8 KB
  // does -- 75.6 against 80.3 ms per step; the choice follows the measured durations of the last pair
  // that ran under the cap in use: pair_balance below.  GHIP_PAIR_NEWTON_LDS fixes it.)
  static int lds_env = -1;
  if(lds_env < 0)
    lds_env = getenv("GHIP_PAIR_NEWTON_LDS") ? atoi(getenv("GHIP_PAIR_NEWTON_LDS")) : 0;
  const int lds_n = lds_env > 0 ? lds_env : ctx->pair_lds;
  // (only when the launch is large enough to fill the chip by itself: a small share of a
  // multi-GPU run leaves room anyway -- measured on c2's shards: 2048 buckets (4 shards) 4.0 -> 3.7 ms
  // with the cap, 1024 buckets (8 shards) 2.38 -> 2.45 ms)
  static int cap_min = -1;
  if(cap_min < 0)
    cap_min = getenv("GHIP_PAIR_CAP_MIN") ? atoi(getenv("GHIP_PAIR_CAP_MIN")) : 1536;
  // (the same handle on the pair's Ewald walk, off by default: GHIP_PAIR_EWALD_LDS)
  static int lds_e = -1;
  if(lds_e < 0)
    lds_e = getenv("GHIP_PAIR_EWALD_LDS") ? atoi(getenv("GHIP_PAIR_EWALD_LDS")) : 0;
  const size_t dyn_lds =
    (MODE == GHIP_WALK_NEWTON && stream != ctx->stream && nbuckets >= cap_min)
      ? (size_t) lds_n
      : ((MODE == GHIP_WALK_EWALD && stream != ctx->stream && nbuckets >= cap_min) ? (size_t) lds_e : 0);
#define GHIP_LAUNCH_WALK(PER, UNEQ)                                                              \
  k_grav_walk<MODE, PER, UNEQ><<<blocks, bsize, dyn_lds, stream>>>(                                \
    t.nelem, P<WalkHot>(t.mq), P<WalkCold>(t.mq2), sg, nt, tgt, tx, ty, tz, tsoft, toldacc, k,     \
    P<float>(ctx->srtab), P<double>((UNEQ) && MODE == GHIP_WALK_EWALD ? ctx->ewbrick : ctx->ewtab),  \
    pb.ax, pb.ay, pb.az, pb.cost, getenv("GHIP_WALK_NOCOUNT") ? nullptr : counter,                  \
    P<unsigned long long>(ctx->rslots) + (counter - P<unsigned long long>(ctx->cslots)), pl)
  // The Ewald walk has no softening rule; its UNEQUAL instantiation is the variant that reads the
  // brick-tiled table (the default; GHIP_EW_BRICK=0 selects the plain rows).  Alone the walk is bound
  // by the L2 requests of its gathers and the bricks cut them (6.3 -> 4.5 ms at c2); inside a pair
  // the bricks pay since the walks' scalar-instruction diet made the pair vector-issue bound
  // (step 9.90 -> 9.56 ms together with the leaner look-up; before that diet the pair was bound by
  // scalar issue and latency and the bricks' longer index arithmetic lost: DESIGN.md 4.3).
  static int brick_env = -2;
  if(brick_env == -2)
    brick_env = getenv("GHIP_EW_BRICK") ? atoi(getenv("GHIP_EW_BRICK")) : -1;
  const bool brick = brick_env >= 0 ? brick_env != 0 : true;
  const bool uneq = (MODE == GHIP_WALK_EWALD) ? brick : (bool) k.unequal;
  if(k.periodic && uneq)
    GHIP_LAUNCH_WALK(true, true);
  else if(k.periodic)
    GHIP_LAUNCH_WALK(true, false);
  else if(uneq)
    GHIP_LAUNCH_WALK(false, true);
  else
    GHIP_LAUNCH_WALK(false, false);
#undef GHIP_LAUNCH_WALK
}

static void launch_walk_any(ghip_ctx *ctx, int walk, const TreeDev &t, const WalkSeg &sg,
                            int nbuckets, int nt, const int *tgt, const double *tx,
                            const double *ty, const double *tz, const double *tsoft,
                            const double *toldacc, const GravK &k, unsigned long long *counter,
                            const WalkPlan &plan, const PartialBufs &pb, hipStream_t stream)
{
  if(walk == GHIP_WALK_NEWTON)
    launch_walk<GHIP_WALK_NEWTON>(ctx, t, sg, nbuckets, nt, tgt, tx, ty, tz, tsoft, toldacc, k,
                                  counter, plan, pb, stream);
  else if(walk == GHIP_WALK_SHORTRANGE)
    launch_walk<GHIP_WALK_SHORTRANGE>(ctx, t, sg, nbuckets, nt, tgt, tx, ty, tz, tsoft, toldacc, k,
                                      counter, plan, pb, stream);
  else
    launch_walk<GHIP_WALK_EWALD>(ctx, t, sg, nbuckets, nt, tgt, tx, ty, tz, tsoft, toldacc, k,
                                 counter, plan, pb, stream);
}

static void shard_slice(const ghip_ctx *ctx, int nt, int *lo, int *cnt)
{
  int per;
  ghip_shard_range(nt, ctx->shard_n, ctx->shard_rank, lo, cnt, &per);
}

// everything one walk launch needs; prepared on the main stream
struct WalkJob
{
  int walk;
  GravK k;
  WalkSeg sg;
  WalkPlan plan;
  PartialBufs pb;
  int nbuckets;
  unsigned long long *counter;
};

// plan_on: the stream the wavefront plan and the counter resets are enqueued on (default: the main
// stream).  The plan reads only what the previous walk of the same kind left behind, so in a sequence
// of pairs each walk's plan is built on that walk's own stream (ghip_gravity_impl), off the main
// stream's tree build.
static int prepare_job(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt, int slot,
                       WalkJob &J, hipStream_t plan_on = nullptr)
{
  hipStream_t ps = plan_on ? plan_on : ctx->stream;
  J.walk = walk;
  GCHK(prepare_tables(ctx, p, walk, J.k));
  // (a sub-step on the kept tree of the last full build walks THAT element list; the targets, their
  // order and the result mapping stay those of the tree of the current positions)
  walk_layout(ctx->dyn_use ? ctx->dyn : ctx->gt, nt, J.sg, &J.nbuckets, walk == GHIP_WALK_EWALD);
  GCHK(build_plan(ctx, walk == GHIP_WALK_EWALD ? 1 : 0, J.nbuckets, J.sg.ns, J.plan, slot, ps));
  GCHK(ensure_partials(ctx, J.plan.nwaves, slot, J.pb));
  J.counter = ghip_cslot(ctx, walk == GHIP_WALK_EWALD ? GHIP_CK_EWALD : GHIP_CK_NEWTON);
  HIPCHK(hipMemsetAsync(J.counter, 0, GHIP_CKIND_U64 * 8, ps));
  return GHIP_OK;
}

static int run_walk(ghip_ctx *ctx, const WalkJob &J, int nt, const int *tgt, hipStream_t st)
{
  int evi = (J.walk == GHIP_WALK_EWALD) ? 4 : 2;
  ctx->plan_writer[J.walk == GHIP_WALK_EWALD ? 1 : 0] = st;
  HIPCHK(hipEventRecord(ctx->evp[evi], st));
  launch_walk_any(ctx, J.walk, ctx->dyn_use ? ctx->dyn : ctx->gt, J.sg, J.nbuckets, nt, tgt, P<double>(ctx->sx),
                  P<double>(ctx->sy), P<double>(ctx->sz), P<double>(ctx->ssoft),
                  P<double>(ctx->soldacc), J.k, J.counter, J.plan, J.pb, st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evp[evi + 1], st));
  return GHIP_OK;
}

static int combine_walk(ghip_ctx *ctx, const WalkJob &J, int nt, const int *tgt, hipStream_t st)
{
  k_combine_grav<<<cdiv(nt, 256), 256, 0, st>>>(
    nt, J.plan, tgt, P<int>(ctx->gt.perm), J.pb.ax, J.pb.ay, J.pb.az, J.pb.cost, ctx->n,
    P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<int>(ctx->f[GHIP_F_GRAVCOST]),
    J.walk == GHIP_WALK_EWALD ? 1 : 0, J.k.debug_steps);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// How many Newtonian wavefronts a pair admits per SIMD (the dynamic-LDS cap of launch_walk): 8 KB = 5
// per SIMD or 10 KB = 4 per SIMD -- the allocation granularity leaves nothing in between.  Which one
// is better depends on the workload, so it is measured: every pair records its walks' start / end
// events together with the setting it ran under; finished pairs are read back here (never waited
// for) and cost max(Newtonian, Ewald), or max(Newtonian, 1.1 x Ewald) when the hydro kernel was
// queued underneath the pair -- about a tenth of the Ewald walk, it can only start when that walk drains.  The cheaper setting is used; the other one is tried
// again for one measurement every 64 pairs.  Scheduling only: the sums of a launch do not depend on it.
static int pair_balance(ghip_ctx *ctx)
{
  if(!ctx->pc_ready)
    {
      for(int i = 0; i < 4; i++)
        for(int j = 0; j < 4; j++)
          HIPCHK(hipEventCreate(&ctx->pc_ev[i][j]));
      ctx->pc_ready = true;
      return GHIP_OK;
    }
  for(int i = 0; i < 4; i++)
    {
      if(ctx->pc_cap[i] == 0)
        continue;
      hipEvent_t *e = ctx->pc_ev[i];
      if(hipEventQuery(e[1]) != hipSuccess || hipEventQuery(e[3]) != hipSuccess)
        {
          (void) hipGetLastError();   // (hipErrorNotReady is not an error)
          continue;
        }
      float tn = 0, te = 0;
      if(hipEventElapsedTime(&tn, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&te, e[2], e[3]) == hipSuccess)
        {
          const int k = ctx->pc_cap[i] > 8192 ? 1 : 0;
          // (only a hydro kernel that waits underneath the pair lengthens the Ewald side)
          const float fe = ctx->pc_hyd[i] ? 1.1f : 1.0f;
          const float c = tn > fe * te ? tn : fe * te;
          ctx->pc_cost[k] = c;
          ctx->pc_age[k] = 0;
        }
      else
        (void) hipGetLastError();
      ctx->pc_cap[i] = 0;
    }
  ctx->pc_age[0]++;
  ctx->pc_age[1]++;
  const int cur = ctx->pair_lds > 8192 ? 1 : 0, oth = cur ^ 1;
  if(ctx->pc_cost[cur] < 0)
    return GHIP_OK;   // nothing known about the setting in use yet: keep it
  int want = cur;
  // (a measurement of the other setting goes stale: after 64 pairs, after 16 when the two are within
  // 10 % of each other -- the first pairs of a run are not representative, their plans have no history)
  const bool close = ctx->pc_cost[oth] > 0 && ctx->pc_cost[oth] < 1.1f * ctx->pc_cost[cur] &&
                     ctx->pc_cost[cur] < 1.1f * ctx->pc_cost[oth];
  if(ctx->pc_cost[oth] < 0 || ctx->pc_age[oth] > (close ? 16 : 64))
    want = oth;       // (stays there until a pair under it has been measured)
  else if(ctx->pc_cost[oth] < ctx->pc_cost[cur])
    want = oth;
  ctx->pair_lds = want ? 10240 : 8192;
  static int dbg = -1;
  if(dbg < 0)
    dbg = getenv("GHIP_PAIR_DEBUG") ? 1 : 0;
  if(dbg)
    fprintf(stderr, "[pair_balance] cost 8K %.3f (age %d)  10K %.3f (age %d)  -> %d\n", ctx->pc_cost[0],
            ctx->pc_age[0], ctx->pc_cost[1], ctx->pc_age[1], ctx->pair_lds);
  return GHIP_OK;
}

int ghip_gravity_impl(ghip_ctx *ctx, const ghip_grav_params *p, int walk)
{
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity: call ghip_tree_build first");
  if(walk < 0 || walk > GHIP_WALK_NEWTON_EWALD)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity: unknown walk %d", walk);
  const bool pair = (walk == GHIP_WALK_NEWTON_EWALD);
  if(pair && !p->periodic)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity: the Newton+Ewald pair needs periodic = 1");
  // (an asynchronously built tree stays unverified here: the walks read its sizes on the device.  The
  // call is logged so that it can be replayed should the verification reject the tree.)
  if(ctx->tree_unverified)
    {
      ghip_ctx::GravCall c;
      c.p = *p;
      c.walk = walk;
      ctx->grav_log.push_back(c);
    }
  GCHK(ghip_build_target_lists(ctx));
  hipStream_t st = ctx->stream;
  int n = ctx->n;
  int lo, nt;
  shard_slice(ctx, ctx->nt_grav, &lo, &nt);
  ghip_stats &S = ctx->stats;
  if(walk == GHIP_WALK_EWALD || pair)
    S.ewald_interactions = 0;
  if(walk != GHIP_WALK_EWALD)
    {
      S.grav_interactions = 0;
      S.grav_targets = nt;
    }
  if(nt == 0 || n == 0)
    return GHIP_OK;
  const int *tgt = P<int>(ctx->tg_grav) + lo;
  WalkJob A, E;
  // In a pair each walk's plan is built on the walk's own stream, next to the tail of the tree build
  // on the main stream -- if that stream also ran the previous walk of the kind (whose per-bucket
  // counts the plan reads); after a single walk on the main stream the plan stays there.
  hipStream_t pn = (pair && ctx->plan_writer[0] == ctx->stream3) ? ctx->stream3 : nullptr;
  hipStream_t pe = (pair && ctx->plan_writer[1] == ctx->stream2) ? ctx->stream2 : nullptr;
  if(getenv("GHIP_PLAN_ON_MAIN"))
    pn = nullptr;
  GCHK(prepare_job(ctx, p, pair ? GHIP_WALK_NEWTON : walk, nt, 0, A, pn));
  if(pair)
    GCHK(prepare_job(ctx, p, GHIP_WALK_EWALD, nt, 1, E, pe));

  // OldAcc in tree order (forcetree.c:1850: aold = ErrTolForceAcc * P[target].OldAcc)
  // (multi-GPU: the tree's sources beyond the local particles are imported elements, never targets)
  GCHK(ghip_gather_f64_lim(ctx, ctx->gt.n, P<int>(ctx->gt.perm), P<double>(ctx->f[GHIP_F_OLDACC]),
                           n, P<double>(ctx->soldacc)));
  if(!pair)
    {
      GCHK(run_walk(ctx, A, nt, tgt, st));
      return combine_walk(ctx, A, nt, tgt, st);
    }
  // The two walks run on streams of their own: one is bound by fp64 issue, the other by the table
  // gathers, and a kernel's tail is filled by the other.  The Ewald sums are added after the
  // Newtonian ones are in place (forcetree.c:3190-3193).  The call returns with both in flight
  // (ghip_join): what the host enqueues next on the main stream -- the SPH phases -- runs
  // underneath them.
  hipStream_t sN = ctx->stream3, sE = ctx->stream2;
  HIPCHK(hipEventRecord(ctx->evx[0], st));
  HIPCHK(hipStreamWaitEvent(sN, ctx->evx[0], 0));
  HIPCHK(hipStreamWaitEvent(sE, ctx->evx[0], 0));
  GCHK(pair_balance(ctx));
  hipEvent_t *pc = ctx->pc_ev[ctx->pc_head];
  ctx->pc_cap[ctx->pc_head] = ctx->pair_lds;
  ctx->pc_hyd[ctx->pc_head] = 0;
  ctx->pc_head = (ctx->pc_head + 1) & 3;
  HIPCHK(hipEventRecord(pc[0], sN));
  GCHK(run_walk(ctx, A, nt, tgt, sN));
  HIPCHK(hipEventRecord(pc[1], sN));
  // (the word the hydro kernel's start waits for, see k_grav_walk: a stale 1 from the previous pair
  // would only let hydro start early, never hold it back)
  unsigned int *started = reinterpret_cast<unsigned int *>(P<unsigned long long>(ctx->counters) + 20);
  HIPCHK(hipMemsetAsync(started, 0, 4, sE));
  E.plan.started = started;
  HIPCHK(hipEventRecord(pc[2], sE));
  GCHK(run_walk(ctx, E, nt, tgt, sE));
  HIPCHK(hipEventRecord(pc[3], sE));
  HIPCHK(hipEventRecord(ctx->evx[3], sE));   // the Ewald walk's wavefront slots are free from here on
  if(A.k.debug_steps || E.k.debug_steps || getenv("GHIP_PAIR_TWO_COMBINES"))
    {
      GCHK(combine_walk(ctx, A, nt, tgt, sN));
      HIPCHK(hipEventRecord(ctx->evx[1], sN));
      HIPCHK(hipStreamWaitEvent(sE, ctx->evx[1], 0));
      GCHK(combine_walk(ctx, E, nt, tgt, sE));
    }
  else
    {
      // both sums in one launch behind the later walk (the Newtonian one, as the pair is balanced)
      HIPCHK(hipEventRecord(ctx->evx[1], sN));
      HIPCHK(hipStreamWaitEvent(sE, ctx->evx[1], 0));
      k_combine_pair<<<cdiv(nt, 256), 256, 0, sE>>>(
        nt, A.plan, E.plan, tgt, P<int>(ctx->gt.perm), A.pb.ax, A.pb.ay, A.pb.az, A.pb.cost, E.pb.ax,
        E.pb.ay, E.pb.az, E.pb.cost, ctx->n, P<double>(ctx->f[GHIP_F_GRAVACCEL]),
        P<int>(ctx->f[GHIP_F_GRAVCOST]));
      HIPCHK(hipGetLastError());
    }
  HIPCHK(hipEventRecord(ctx->evx[2], sE));
  // (the Newtonian walk's stream builds the NEXT plan in place -- after the combine has read this one)
  HIPCHK(hipStreamWaitEvent(sN, ctx->evx[2], 0));
  ctx->grav_pending = true;
  ctx->pair_started = started;
  return GHIP_OK;
}

extern "C" int ghip_gravity(ghip_ctx *ctx, const ghip_grav_params *p, int walk)
{
  if(ctx)
    GCHK(ghip_join_pair(ctx));   // (a deferred gas tree stays deferred: it is not needed here)
  if(!ctx || !p)
    return GHIP_EINVAL;
  return ghip_gravity_impl(ctx, p, walk);
}

extern "C" int ghip_gravity_ext_soft(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt,
                                     const double *pos, const int *type, const double *soft,
                                     const double *oldacc, double *acc, int *ninteractions)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p || nt < 0 || (nt > 0 && (!pos || !type || !oldacc || !acc || !ninteractions)))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_ext: bad arguments");
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_ext: call ghip_tree_build first");
  if(walk < 0 || walk > 2)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_ext: unknown walk %d", walk);
  if(nt == 0)
    return GHIP_OK;
  GravK k;
  GCHK(prepare_tables(ctx, p, walk, k));
  hipStream_t st = ctx->stream;
  // staging: x,y,z,oldacc,soft (f64[nt]) + combined acc (3 f64[nt]) + type, cost (i32[nt])
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nt * (8 * 8 + 2 * 4) + 64));
  std::vector<double> h((size_t) nt * 5);
  for(int a = 0; a < nt; a++)
    {
      if(type[a] < 0 || type[a] > 5)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_ext: target %d has type %d", a, type[a]);
      h[a] = pos[3 * (size_t) a];
      h[(size_t) nt + a] = pos[3 * (size_t) a + 1];
      h[2 * (size_t) nt + a] = pos[3 * (size_t) a + 2];
      h[3 * (size_t) nt + a] = oldacc[a];
      // the target's own softening: gravdata_in.Soft for gas under ADAPTIVE_GRAVSOFT_FORGAS
      // (forcetree.c:1875-1878), All.ForceSoftening[Type] otherwise
      h[4 * (size_t) nt + a] = (ctx->adaptive_gravsoft && type[a] == 0 && soft)
                                 ? soft[a]
                                 : p->ForceSoftening[type[a]];
    }
  double *dx = P<double>(ctx->stage), *dy = dx + nt, *dz = dy + nt, *dold = dz + nt,
         *dsoft = dold + nt, *dacc = dsoft + nt;
  int *dtype = reinterpret_cast<int *>(dacc + 3 * (size_t) nt), *dcost = dtype + nt;
  HIPCHK(hipMemcpyAsync(dx, h.data(), (size_t) nt * 40, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dtype, type, (size_t) nt * 4, hipMemcpyHostToDevice, st));
  WalkSeg sg;
  int nbuckets;
  walk_layout(ctx->dyn_use ? ctx->dyn : ctx->gt, nt, sg, &nbuckets, walk == GHIP_WALK_EWALD);
  WalkPlan plan;
  GCHK(build_plan(ctx, 2, nbuckets, sg.ns, plan));
  PartialBufs pb;
  GCHK(ensure_partials(ctx, plan.nwaves, 0, pb));
  unsigned long long *counter = ghip_cslot(ctx, GHIP_CK_EXT);
  launch_walk_any(ctx, walk, ctx->dyn_use ? ctx->dyn : ctx->gt, sg, nbuckets, nt, nullptr, dx, dy, dz, dsoft, dold, k,
                  counter, plan, pb, st);
  k_combine_grav<<<cdiv(nt, 256), 256, 0, st>>>(nt, plan, nullptr, nullptr, P<double>(ctx->tax),
                                               P<double>(ctx->tay), P<double>(ctx->taz),
                                               P<int>(ctx->tcost), nt, dacc, dcost, 0, 0);
  HIPCHK(hipGetLastError());
  std::vector<double> r((size_t) nt * 3);
  HIPCHK(hipMemcpyAsync(r.data(), dacc, (size_t) nt * 24, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(ninteractions, dcost, (size_t) nt * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int a = 0; a < nt; a++)
    {
      acc[3 * (size_t) a] = r[a];
      acc[3 * (size_t) a + 1] = r[(size_t) nt + a];
      acc[3 * (size_t) a + 2] = r[2 * (size_t) nt + a];
    }
  return GHIP_OK;
}

extern "C" int ghip_gravity_ext(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt,
                                const double *pos, const int *type, const double *oldacc,
                                double *acc, int *ninteractions)
{
  return ghip_gravity_ext_soft(ctx, p, walk, nt, pos, type, nullptr, oldacc, acc, ninteractions);
}

// all_shards != 0: every active target regardless of the shard (multi-GPU, replicated mode: after
// the all-gather of the sharded walks each rank holds all G-less accelerations)
// the post-pass of gravity_tree() on a given stream (the caller has ordered it after the walks)
int ghip_gravity_finish_on(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac, int all_shards,
                           hipStream_t st)
{
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_finish: no tree");
  if(pmgrid && !(G != 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_finish: PMGRID needs G != 0 (GravPM / G)");
  GCHK(ghip_build_target_lists(ctx));
  int lo = 0, nt = ctx->nt_grav;
  if(!all_shards)
    shard_slice(ctx, ctx->nt_grav, &lo, &nt);
  if(nt == 0)
    return GHIP_OK;
  k_grav_finish<<<cdiv(nt, 256), 256, 0, st>>>(
    nt, P<int>(ctx->tg_grav) + lo, P<int>(ctx->gt.perm), ctx->n, G, comoving_fac,
    P<double>(ctx->f[GHIP_F_POS]), pmgrid ? P<double>(ctx->f[GHIP_F_GRAVPM]) : nullptr,
    P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<double>(ctx->f[GHIP_F_OLDACC]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_gravity_finish_ex(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac,
                                      int all_shards)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx)
    return GHIP_EINVAL;
  return ghip_gravity_finish_on(ctx, G, pmgrid, comoving_fac, all_shards, ctx->stream);
}

extern "C" int ghip_gravity_finish(ghip_ctx *ctx, double G)
{
  return ghip_gravity_finish_ex(ctx, G, 0, 0.0, 0);
}

extern "C" int ghip_gravity_finish_all(ghip_ctx *ctx, double G)
{
  return ghip_gravity_finish_ex(ctx, G, 0, 0.0, 1);
}

extern "C" int ghip_gravity_vacuum_energy(ghip_ctx *ctx, double fac)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx)
    return GHIP_EINVAL;
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_vacuum_energy: no tree");
  GCHK(ghip_build_target_lists(ctx));
  int nt = ctx->nt_grav;   // every active particle, like ghip_gravity_finish_all
  if(nt == 0)
    return GHIP_OK;
  k_grav_vacuum<<<cdiv(nt, 256), 256, 0, ctx->stream>>>(
    nt, P<int>(ctx->tg_grav), P<int>(ctx->gt.perm), ctx->n, fac, P<double>(ctx->f[GHIP_F_POS]),
    P<double>(ctx->f[GHIP_F_GRAVACCEL]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_gravity_direct(ghip_ctx *ctx, const ghip_grav_params *p)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p)
    return GHIP_EINVAL;
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_direct: call ghip_tree_build first");
  GravK k;
  GCHK(prepare_tables(ctx, p, GHIP_WALK_NEWTON, k));
  GCHK(ghip_build_target_lists(ctx));
  int n = ctx->n;
  int lo, nt;
  shard_slice(ctx, ctx->nt_grav, &lo, &nt);
  if(nt == 0 || n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  const int *tgt = P<int>(ctx->tg_grav) + lo;
  GCHK(ghip_ensure(ctx, ctx->tax, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tay, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->taz, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tcost, (size_t) nt * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) n * sizeof(double4)));
  k_pack_xyzm<<<cdiv(n, 256), 256, 0, st>>>(n, P<double>(ctx->sx), P<double>(ctx->sy),
                                            P<double>(ctx->sz), P<int>(ctx->gt.perm),
                                            P<double>(ctx->f[GHIP_F_MASS]),
                                            P<double4>(ctx->stage));
  k_grav_direct<<<cdiv(nt, 256), 256, 0, st>>>(n, P<double>(ctx->sx), P<double>(ctx->sy),
                                               P<double>(ctx->sz), P<double4>(ctx->stage),
                                               P<double>(ctx->ssoft), nt, tgt, k,
                                               P<double>(ctx->tax), P<double>(ctx->tay),
                                               P<double>(ctx->taz));
  HIPCHK(hipMemsetAsync(ctx->tcost.p, 0, (size_t) nt * 4, st));
  k_scatter_direct<<<cdiv(nt, 256), 256, 0, st>>>(
    nt, tgt, P<int>(ctx->gt.perm), P<double>(ctx->tax), P<double>(ctx->tay), P<double>(ctx->taz),
    n, P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<int>(ctx->f[GHIP_F_GRAVCOST]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}
