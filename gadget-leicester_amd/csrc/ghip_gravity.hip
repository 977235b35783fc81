// ghip_gravity.hip -- Barnes-Hut tree-walk gravity on gfx950.
//
// Replaces the active-list loop of gravity_tree() (gravtree.c:130-168) over
// force_treeevaluate() (forcetree.c:1797-2317), force_treeevaluate_shortrange()
// (forcetree.c:2330-2845) and force_treeevaluate_ewald_correction() (forcetree.c:2873-3204).
//
// One wavefront (64 lanes) walks the pre-order element list for a bucket of 64 targets that are
// consecutive along the space-filling curve.  The element index is wave-uniform, so node and
// particle records arrive through the scalar cache (s_load) and are broadcast for free; every
// lane applies the reference's PER-PARTICLE opening criterion to the same node:
//   * a lane that accepts the node interacts with its monopole and remembers the node's skip
//     index -- it ignores every element below the node (my_skip);
//   * the wave descends (e+1) if ANY participating lane must open, else jumps to the skip index.
// Each lane therefore sees exactly the interaction set, in exactly the depth-first order, of the
// reference's serial walk for its particle: parity is summation-identical, not merely within a
// group-MAC bound.  No MFMA: this is irregular fp64 pairwise work.
#include <cmath>

#include "ghip_internal.h"

struct GravK
{
  double theta2;       // ErrTolTheta^2 (0: relative criterion)
  double errtol;       // ErrTolForceAcc
  double boxsize, boxhalf;
  int periodic, unequal;
  int debug_steps;     // GHIP_DEBUG_STEPS=1: GRAVCOST receives the wave's visited-element count
  double rcut, rcut2, asmthfac;  // shortrange
  double fac_intp;     // ewald: 2*EN/BoxSize
};

// softened monopole kernel, forcetree.c:2143-2171
__device__ __forceinline__ double d_grav_fac(double mass, double r2, double r, double h)
{
  if(r >= h)
    return mass / (r2 * r);
  double h_inv = 1.0 / h;
  double h3_inv = h_inv * h_inv * h_inv;
  double u = r * h_inv;
  if(u < 0.5)
    return mass * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
  return mass * h3_inv *
         (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u -
          0.066666666667 / (u * u * u));
}

// trilinear Ewald look-up, forcetree.c:3097-3170.  tab: double4 (fx,fy,fz,0) per grid point.
__device__ __forceinline__ void d_ewald_interp(const double4 *__restrict__ tab, double fac_intp,
                                               double dx, double dy, double dz, double &fx,
                                               double &fy, double &fz)
{
  const int E1 = GHIP_EN + 1;
  double sx = -1, sy = -1, sz = -1;
  if(dx < 0)
    {
      dx = -dx;
      sx = +1;
    }
  if(dy < 0)
    {
      dy = -dy;
      sy = +1;
    }
  if(dz < 0)
    {
      dz = -dz;
      sz = +1;
    }
  double u = dx * fac_intp;
  int i = (int) u;
  if(i >= GHIP_EN)
    i = GHIP_EN - 1;
  u -= i;
  double v = dy * fac_intp;
  int j = (int) v;
  if(j >= GHIP_EN)
    j = GHIP_EN - 1;
  v -= j;
  double w = dz * fac_intp;
  int k = (int) w;
  if(k >= GHIP_EN)
    k = GHIP_EN - 1;
  w -= k;
  double f1 = (1 - u) * (1 - v) * (1 - w), f2 = (1 - u) * (1 - v) * (w);
  double f3 = (1 - u) * (v) * (1 - w), f4 = (1 - u) * (v) * (w);
  double f5 = (u) * (1 - v) * (1 - w), f6 = (u) * (1 - v) * (w);
  double f7 = (u) * (v) * (1 - w), f8 = (u) * (v) * (w);
  const double4 *b = tab + ((size_t) i * E1 + j) * E1 + k;
  double4 t1 = b[0], t2 = b[1], t3 = b[E1], t4 = b[E1 + 1];
  double4 t5 = b[E1 * E1], t6 = b[E1 * E1 + 1], t7 = b[E1 * E1 + E1], t8 = b[E1 * E1 + E1 + 1];
  fx = sx * (t1.x * f1 + t2.x * f2 + t3.x * f3 + t4.x * f4 + t5.x * f5 + t6.x * f6 + t7.x * f7 +
             t8.x * f8);
  fy = sy * (t1.y * f1 + t2.y * f2 + t3.y * f3 + t4.y * f4 + t5.y * f5 + t6.y * f6 + t7.y * f7 +
             t8.y * f8);
  fz = sz * (t1.z * f1 + t2.z * f2 + t3.z * f3 + t4.z * f4 + t5.z * f5 + t6.z * f6 + t7.z * f7 +
             t8.z * f8);
}

template <int MODE>
__global__ void __launch_bounds__(GHIP_BLOCK)
k_grav_walk(int nelem, const double4 *__restrict__ xm, const double4 *__restrict__ cl,
            const int4 *__restrict__ lk, const double *__restrict__ aux, int nt,
            const int *__restrict__ tgt, const double *__restrict__ tx,
            const double *__restrict__ ty, const double *__restrict__ tz,
            const double *__restrict__ tsoft, const double *__restrict__ toldacc, GravK p,
            const float *__restrict__ srtab, const double4 *__restrict__ ewtab,
            double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
            int *__restrict__ cost, unsigned long long *__restrict__ counter)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * GHIP_BLOCK + threadIdx.x) >> 6;
  const int ti = wave * 64 + lane;
  const bool valid = ti < nt;
  const int s = valid ? (tgt ? tgt[ti] : ti) : 0;

  double pos_x = 0, pos_y = 0, pos_z = 0, h_i = 1, aold = 0;
  if(valid)
    {
      pos_x = tx[s];
      pos_y = ty[s];
      pos_z = tz[s];
      h_i = tsoft[s];
      aold = p.errtol * toldacc[s];
    }
  double acc_x = 0, acc_y = 0, acc_z = 0;
  int nint = 0;
  int my_skip = valid ? 0 : 0x7fffffff;
  unsigned int steps = 0;

  int e = 0;
  while(e < nelem)
    {
      e = __builtin_amdgcn_readfirstlane(e);
      steps++;
      const double4 v = xm[e];
      const int4 k = lk[e];
      const bool act = (e >= my_skip);
      int next;

      double dx = v.x - pos_x, dy = v.y - pos_y, dz = v.z - pos_z;
      if(MODE == GHIP_WALK_EWALD || p.periodic)
        {
          dx = d_nearest(dx, p.boxsize, p.boxhalf);
          dy = d_nearest(dy, p.boxsize, p.boxhalf);
          dz = d_nearest(dz, p.boxsize, p.boxhalf);
        }
      const double r2 = dx * dx + dy * dy + dz * dz;
      const double mass = v.w;
      double h = h_i;
      bool interact = act;

      if(LK_IS_PARTICLE(k))
        {
          next = e + 1;
          if(MODE != GHIP_WALK_EWALD && p.unequal)
            {
              double sj = aux[e];
              if(h < sj)
                h = sj;
            }
        }
      else
        {
          const double4 c = cl[e];
          const double len = c.w;
          bool open = false;
          if(act)
            {
              if(MODE == GHIP_WALK_SHORTRANGE)
                {
                  // forcetree.c:2598-2632: whole cell beyond the cut-off -> drop the branch
                  if(r2 > p.rcut2)
                    {
                      double eff = p.rcut + 0.5 * len;
                      double d0 = c.x - pos_x, d1 = c.y - pos_y, d2 = c.z - pos_z;
                      if(p.periodic)
                        {
                          d0 = d_nearest(d0, p.boxsize, p.boxhalf);
                          d1 = d_nearest(d1, p.boxsize, p.boxhalf);
                          d2 = d_nearest(d2, p.boxsize, p.boxhalf);
                        }
                      if(d0 < -eff || d0 > eff || d1 < -eff || d1 > eff || d2 < -eff || d2 > eff)
                        {
                          interact = false;
                          my_skip = k.x;
                        }
                    }
                }
              if(interact)
                {
                  // opening criterion, forcetree.c:2074-2105
                  if(p.theta2 != 0)
                    open = (len * len > r2 * p.theta2);
                  else
                    {
                      open = (mass * len * len > r2 * r2 * aold);
                      if(!open)
                        open = (fabs(c.x - pos_x) < 0.60 * len) && (fabs(c.y - pos_y) < 0.60 * len) &&
                               (fabs(c.z - pos_z) < 0.60 * len);
                    }
                  if(MODE == GHIP_WALK_EWALD)
                    {
                      // forcetree.c:3039-3088: the correction is smooth, so an "open" verdict is
                      // overridden unless the cell straddles the half-box or is large
                      if(open)
                        {
                          bool must = false;
                          double u0 = d_nearest(c.x - pos_x, p.boxsize, p.boxhalf);
                          double u1 = d_nearest(c.y - pos_y, p.boxsize, p.boxhalf);
                          double u2 = d_nearest(c.z - pos_z, p.boxsize, p.boxhalf);
                          double lim = 0.5 * (p.boxsize - len);
                          must = (fabs(u0) > lim) || (fabs(u1) > lim) || (fabs(u2) > lim) ||
                                 (len > 0.20 * p.boxsize);
                          open = must;
                        }
                    }
                  else if(p.unequal && !open)
                    {
                      // forcetree.c:2108-2124
                      double a = aux[e];
                      double ms = fabs(a);
                      if(h < ms)
                        {
                          h = ms;
                          if(r2 < h * h && a < 0)
                            open = true;
                        }
                    }
                  if(open)
                    interact = false;
                  else
                    my_skip = k.x;
                }
            }
          next = __any(open) ? e + 1 : k.x;
        }

      if(interact)
        {
          if(MODE == GHIP_WALK_EWALD)
            {
              double fx, fy, fz;
              d_ewald_interp(ewtab, p.fac_intp, dx, dy, dz, fx, fy, fz);
              acc_x += mass * fx;
              acc_y += mass * fy;
              acc_z += mass * fz;
              nint++;
            }
          else
            {
              const double r = sqrt(r2);
              double fac = d_grav_fac(mass, r2, r, h);
              if(MODE == GHIP_WALK_SHORTRANGE)
                {
                  // forcetree.c:2739-2752
                  int tabindex = (int) (p.asmthfac * r);
                  if(tabindex < GHIP_NTAB)
                    {
                      fac *= srtab[tabindex];
                      acc_x += dx * fac;
                      acc_y += dy * fac;
                      acc_z += dz * fac;
                      nint++;
                    }
                }
              else
                {
                  acc_x += dx * fac;
                  acc_y += dy * fac;
                  acc_z += dz * fac;
                  if(mass > 0)
                    nint++;
                }
            }
        }
      e = next;
    }

  if(valid)
    {
      ax[ti] = acc_x;
      ay[ti] = acc_y;
      az[ti] = acc_z;
      cost[ti] = p.debug_steps ? (int) steps : nint;
    }
  unsigned long long tot = d_wave_sum_u64((unsigned long long) nint);
  if(lane == 0 && tot)
    atomicAdd(counter, tot);
  if(lane == 0)
    atomicAdd(counter + 8, (unsigned long long) steps);
}

// results from target order to host order; EWALD adds (forcetree.c:3190-3193)
__global__ void k_scatter_grav(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                               const double *__restrict__ ax, const double *__restrict__ ay,
                               const double *__restrict__ az, const int *__restrict__ cost, int n,
                               double *__restrict__ oacc, int *__restrict__ ocost, int accumulate)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int i = perm[tgt[ti]];
  if(accumulate)
    {
      oacc[i] += ax[ti];
      oacc[(size_t) n + i] += ay[ti];
      oacc[2 * (size_t) n + i] += az[ti];
      ocost[i] += cost[ti];
    }
  else
    {
      oacc[i] = ax[ti];
      oacc[(size_t) n + i] = ay[ti];
      oacc[2 * (size_t) n + i] = az[ti];
      ocost[i] = cost[ti];
    }
}

// gravtree.c:381-403: OldAcc = |GravAccel| (G-less), then GravAccel *= G
__global__ void k_grav_finish(int nt, const int *__restrict__ tgt, const int *__restrict__ perm,
                              int n, double G, double *__restrict__ oacc,
                              double *__restrict__ oldacc)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int i = perm[tgt[ti]];
  double a0 = oacc[i], a1 = oacc[(size_t) n + i], a2 = oacc[2 * (size_t) n + i];
  oldacc[i] = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  oacc[i] = a0 * G;
  oacc[(size_t) n + i] = a1 * G;
  oacc[2 * (size_t) n + i] = a2 * G;
}

// softened direct summation (formula of forcetree.c:4273-4336), LDS-tiled, for accuracy checks
__global__ void __launch_bounds__(256)
k_grav_direct(int n, const double *__restrict__ sx, const double *__restrict__ sy,
              const double *__restrict__ sz, const double4 *__restrict__ xm,
              const int4 *__restrict__ lk, int nelem, const double *__restrict__ ssoft, int nt,
              const int *__restrict__ tgt, GravK p, double *__restrict__ ax,
              double *__restrict__ ay, double *__restrict__ az)
{
  // sources are read straight from the sorted arrays; mass comes from the element list is
  // avoided by passing per-particle mass in xm? -- kept simple: masses are gathered below
  __shared__ double lx[256], ly[256], lz[256], lm[256], ls[256];
  int ti = blockIdx.x * 256 + threadIdx.x;
  bool valid = ti < nt;
  int s = valid ? tgt[ti] : 0;
  double px = sx[s], py = sy[s], pz = sz[s], hi = ssoft[s];
  double a0 = 0, a1 = 0, a2 = 0;
  (void) lk;
  (void) nelem;
  for(int base = 0; base < n; base += 256)
    {
      int j = base + threadIdx.x;
      if(j < n)
        {
          lx[threadIdx.x] = sx[j];
          ly[threadIdx.x] = sy[j];
          lz[threadIdx.x] = sz[j];
          lm[threadIdx.x] = xm[j].w;  // caller passes a per-particle (x,y,z,m) array here
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
          double r = sqrt(r2);
          double fac = d_grav_fac(lm[q], r2, r, h);
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
// |n|,|h| <= 4, one octant of (EN+1)^3 points at x = 0.5*(i,j,k)/EN, scaled by 1/BoxSize^2
// ---------------------------------------------------------------------------------------------
__global__ void k_ewald_table(double inv_box2, double4 *__restrict__ tab)
{
  const int E1 = GHIP_EN + 1;
  int nidx = blockIdx.x * blockDim.x + threadIdx.x;
  if(nidx >= E1 * E1 * E1)
    return;
  int i = nidx / (E1 * E1), j = (nidx / E1) % E1, k = nidx % E1;
  double f0 = 0, f1 = 0, f2 = 0;
  if(i + j + k != 0)
    {
      const double alpha = 2.0;
      double x0 = 0.5 * ((double) i) / GHIP_EN, x1 = 0.5 * ((double) j) / GHIP_EN,
             x2 = 0.5 * ((double) k) / GHIP_EN;
      double r2 = x0 * x0 + x1 * x1 + x2 * x2;
      double rr = r2 * sqrt(r2);
      f0 += x0 / rr;
      f1 += x1 / rr;
      f2 += x2 / rr;
      for(int n0 = -4; n0 <= 4; n0++)
        for(int n1 = -4; n1 <= 4; n1++)
          for(int n2 = -4; n2 <= 4; n2++)
            {
              double d0 = x0 - n0, d1 = x1 - n1, d2 = x2 - n2;
              double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
              double val = erfc(alpha * r) + 2 * alpha * r / sqrt(M_PI) * exp(-alpha * alpha * r * r);
              double w = val / (r * r * r);
              f0 -= d0 * w;
              f1 -= d1 * w;
              f2 -= d2 * w;
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
                  f1 -= h1 * val;
                  f2 -= h2 * val;
                }
            }
    }
  tab[nidx] = make_double4(f0 * inv_box2, f1 * inv_box2, f2 * inv_box2, 0.0);
}

extern "C" int ghip_ewald_init(ghip_ctx *ctx, double BoxSize)
{
  if(!ctx || !(BoxSize > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_ewald_init: bad BoxSize");
  const int E1 = GHIP_EN + 1;
  const int nt = E1 * E1 * E1;
  GCHK(ghip_ensure(ctx, ctx->ewtab, (size_t) nt * sizeof(double4)));
  k_ewald_table<<<cdiv(nt, 64), 64, 0, ctx->stream>>>(1.0 / (BoxSize * BoxSize),
                                                      P<double4>(ctx->ewtab));
  HIPCHK(hipGetLastError());
  ctx->ew_box = BoxSize;
  return GHIP_OK;
}

extern "C" int ghip_ewald_get_table(ghip_ctx *ctx, double *host)
{
  if(!ctx || !host || ctx->ew_box == 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_ewald_get_table: table not initialised");
  const int E1 = GHIP_EN + 1;
  const size_t nt = (size_t) E1 * E1 * E1;
  std::vector<double> tmp(nt * 4);
  HIPCHK(hipMemcpyAsync(tmp.data(), ctx->ewtab.p, nt * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for(size_t q = 0; q < nt; q++)
    for(int c = 0; c < 3; c++)
      host[c * nt + q] = tmp[4 * q + c];
  return GHIP_OK;
}

static int prepare_tables(ghip_ctx *ctx, const ghip_grav_params *p, int walk, GravK &k)
{
  k.theta2 = p->ErrTolTheta * p->ErrTolTheta;
  k.errtol = p->ErrTolForceAcc;
  k.boxsize = p->BoxSize;
  k.boxhalf = 0.5 * p->BoxSize;
  k.periodic = p->periodic;
  k.unequal = p->unequal_softenings;
  k.rcut = p->Rcut;
  k.rcut2 = p->Rcut * p->Rcut;
  k.asmthfac = (p->Asmth > 0) ? 0.5 / p->Asmth * (GHIP_NTAB / 3.0) : 0;  // forcetree.c:2378
  k.fac_intp = (p->BoxSize > 0) ? 2 * GHIP_EN / p->BoxSize : 0;
  k.debug_steps = getenv("GHIP_DEBUG_STEPS") ? 1 : 0;
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
          HIPCHK(hipStreamSynchronize(ctx->stream));
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

template <int MODE>
static void launch_walk(ghip_ctx *ctx, const TreeDev &t, int nt, const int *tgt, const double *tx,
                        const double *ty, const double *tz, const double *tsoft,
                        const double *toldacc, const GravK &k, unsigned long long *counter)
{
  int blocks = cdiv(nt, GHIP_BLOCK);
  k_grav_walk<MODE><<<blocks, GHIP_BLOCK, 0, ctx->stream>>>(
    t.nelem, P<double4>(t.xm), P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), nt, tgt, tx, ty,
    tz, tsoft, toldacc, k, P<float>(ctx->srtab), P<double4>(ctx->ewtab), P<double>(ctx->tax),
    P<double>(ctx->tay), P<double>(ctx->taz), P<int>(ctx->tcost), counter);
}

static void shard_slice(const ghip_ctx *ctx, int nt, int *lo, int *cnt)
{
  int per = (nt + ctx->shard_n - 1) / ctx->shard_n;
  int a = ctx->shard_rank * per;
  int b = a + per;
  if(a > nt)
    a = nt;
  if(b > nt)
    b = nt;
  *lo = a;
  *cnt = b - a;
}

int ghip_gravity_impl(ghip_ctx *ctx, const ghip_grav_params *p, int walk)
{
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity: call ghip_tree_build first");
  if(walk < 0 || walk > 2)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity: unknown walk %d", walk);
  GravK k;
  GCHK(prepare_tables(ctx, p, walk, k));
  GCHK(ghip_build_target_lists(ctx));
  hipStream_t st = ctx->stream;
  int n = ctx->n;
  int lo, nt;
  shard_slice(ctx, ctx->nt_grav, &lo, &nt);
  ghip_stats &S = ctx->stats;
  if(walk == GHIP_WALK_EWALD)
    S.ewald_interactions = 0;
  else
    {
      S.grav_interactions = 0;
      S.grav_targets = nt;
    }
  if(nt == 0 || n == 0)
    return GHIP_OK;
  const int *tgt = P<int>(ctx->tg_grav) + lo;

  GCHK(ghip_ensure(ctx, ctx->tax, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tay, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->taz, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tcost, (size_t) nt * 4));
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  unsigned long long *counter = P<unsigned long long>(ctx->counters) + (walk == GHIP_WALK_EWALD ? 1 : 0);
  HIPCHK(hipMemsetAsync(counter, 0, 8, st));
  HIPCHK(hipMemsetAsync(counter + 8, 0, 8, st));

  // OldAcc in tree order (forcetree.c:1850: aold = ErrTolForceAcc * P[target].OldAcc)
  GCHK(ghip_gather_f64(ctx, n, P<int>(ctx->gt.perm), P<double>(ctx->f[GHIP_F_OLDACC]),
                        P<double>(ctx->soldacc)));

  int evi = (walk == GHIP_WALK_EWALD) ? 4 : 2;
  HIPCHK(hipEventRecord(ctx->ev[evi], st));
  const double *tx = P<double>(ctx->sx), *ty = P<double>(ctx->sy), *tz = P<double>(ctx->sz);
  const double *ts = P<double>(ctx->ssoft), *to = P<double>(ctx->soldacc);
  if(walk == GHIP_WALK_NEWTON)
    launch_walk<GHIP_WALK_NEWTON>(ctx, ctx->gt, nt, tgt, tx, ty, tz, ts, to, k, counter);
  else if(walk == GHIP_WALK_SHORTRANGE)
    launch_walk<GHIP_WALK_SHORTRANGE>(ctx, ctx->gt, nt, tgt, tx, ty, tz, ts, to, k, counter);
  else
    launch_walk<GHIP_WALK_EWALD>(ctx, ctx->gt, nt, tgt, tx, ty, tz, ts, to, k, counter);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[evi + 1], st));

  k_scatter_grav<<<cdiv(nt, 256), 256, 0, st>>>(
    nt, tgt, P<int>(ctx->gt.perm), P<double>(ctx->tax), P<double>(ctx->tay), P<double>(ctx->taz),
    P<int>(ctx->tcost), n, P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<int>(ctx->f[GHIP_F_GRAVCOST]),
    walk == GHIP_WALK_EWALD ? 1 : 0);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_gravity(ghip_ctx *ctx, const ghip_grav_params *p, int walk)
{
  if(!ctx || !p)
    return GHIP_EINVAL;
  return ghip_gravity_impl(ctx, p, walk);
}

__global__ void k_soft_of_type_g(int n, const int *__restrict__ type, double s0, double s1,
                                 double s2, double s3, double s4, double s5,
                                 double *__restrict__ out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int t = type[i];
  out[i] = (t == 0) ? s0 : (t == 1) ? s1 : (t == 2) ? s2 : (t == 3) ? s3 : (t == 4) ? s4 : s5;
}

extern "C" int ghip_gravity_ext(ghip_ctx *ctx, const ghip_grav_params *p, int walk, int nt,
                                const double *pos, const int *type, const double *oldacc,
                                double *acc, int *ninteractions)
{
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
  // staging: x,y,z,oldacc,soft (f64[nt] each) + type (i32[nt])
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nt * (5 * 8 + 4) + 64));
  std::vector<double> h((size_t) nt * 4);
  for(int a = 0; a < nt; a++)
    {
      h[a] = pos[3 * (size_t) a];
      h[(size_t) nt + a] = pos[3 * (size_t) a + 1];
      h[2 * (size_t) nt + a] = pos[3 * (size_t) a + 2];
      h[3 * (size_t) nt + a] = oldacc[a];
    }
  double *dx = P<double>(ctx->stage), *dy = dx + nt, *dz = dy + nt, *dold = dz + nt,
         *dsoft = dold + nt;
  int *dtype = reinterpret_cast<int *>(dsoft + nt);
  HIPCHK(hipMemcpyAsync(dx, h.data(), (size_t) nt * 32, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dtype, type, (size_t) nt * 4, hipMemcpyHostToDevice, st));
  k_soft_of_type_g<<<cdiv(nt, 256), 256, 0, st>>>(nt, dtype, p->ForceSoftening[0],
                                                  p->ForceSoftening[1], p->ForceSoftening[2],
                                                  p->ForceSoftening[3], p->ForceSoftening[4],
                                                  p->ForceSoftening[5], dsoft);
  GCHK(ghip_ensure(ctx, ctx->tax, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tay, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->taz, (size_t) nt * 8));
  GCHK(ghip_ensure(ctx, ctx->tcost, (size_t) nt * 4));
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  unsigned long long *counter = P<unsigned long long>(ctx->counters) + 2;
  if(walk == GHIP_WALK_NEWTON)
    launch_walk<GHIP_WALK_NEWTON>(ctx, ctx->gt, nt, nullptr, dx, dy, dz, dsoft, dold, k, counter);
  else if(walk == GHIP_WALK_SHORTRANGE)
    launch_walk<GHIP_WALK_SHORTRANGE>(ctx, ctx->gt, nt, nullptr, dx, dy, dz, dsoft, dold, k,
                                      counter);
  else
    launch_walk<GHIP_WALK_EWALD>(ctx, ctx->gt, nt, nullptr, dx, dy, dz, dsoft, dold, k, counter);
  HIPCHK(hipGetLastError());
  std::vector<double> r((size_t) nt * 3);
  HIPCHK(hipMemcpyAsync(r.data(), ctx->tax.p, (size_t) nt * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(r.data() + nt, ctx->tay.p, (size_t) nt * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(r.data() + 2 * (size_t) nt, ctx->taz.p, (size_t) nt * 8,
                        hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(ninteractions, ctx->tcost.p, (size_t) nt * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for(int a = 0; a < nt; a++)
    {
      acc[3 * (size_t) a] = r[a];
      acc[3 * (size_t) a + 1] = r[(size_t) nt + a];
      acc[3 * (size_t) a + 2] = r[2 * (size_t) nt + a];
    }
  return GHIP_OK;
}

extern "C" int ghip_gravity_finish(ghip_ctx *ctx, double G)
{
  if(!ctx)
    return GHIP_EINVAL;
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_finish: no tree");
  GCHK(ghip_build_target_lists(ctx));
  int lo, nt;
  shard_slice(ctx, ctx->nt_grav, &lo, &nt);
  if(nt == 0)
    return GHIP_OK;
  k_grav_finish<<<cdiv(nt, 256), 256, 0, ctx->stream>>>(
    nt, P<int>(ctx->tg_grav) + lo, P<int>(ctx->gt.perm), ctx->n, G,
    P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<double>(ctx->f[GHIP_F_OLDACC]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_gravity_direct(ghip_ctx *ctx, const ghip_grav_params *p)
{
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
                                               P<double>(ctx->sz), P<double4>(ctx->stage), nullptr,
                                               0, P<double>(ctx->ssoft), nt, tgt, k,
                                               P<double>(ctx->tax), P<double>(ctx->tay),
                                               P<double>(ctx->taz));
  HIPCHK(hipMemsetAsync(ctx->tcost.p, 0, (size_t) nt * 4, st));
  k_scatter_grav<<<cdiv(nt, 256), 256, 0, st>>>(
    nt, tgt, P<int>(ctx->gt.perm), P<double>(ctx->tax), P<double>(ctx->tay), P<double>(ctx->taz),
    P<int>(ctx->tcost), n, P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<int>(ctx->f[GHIP_F_GRAVCOST]),
    0);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}
