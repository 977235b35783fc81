// ghip_walk.h -- the gravity tree-walk kernel (device code), included by ghip_gravity.hip.
//
// Replaces the bodies of force_treeevaluate() (forcetree.c:1797-2317),
// force_treeevaluate_shortrange() (forcetree.c:2330-2845) and
// force_treeevaluate_ewald_correction() (forcetree.c:2873-3204).
//
// Mapping.  A BUCKET is 64 targets consecutive along the space-filling curve; a wavefront's 64
// lanes are those targets.  The tree is ONE pre-order element list (ghip_tree.hip); the element
// index a wave looks at is wave-uniform, so node/particle records arrive through the scalar
// cache (s_load_dwordx8) and are broadcast for free.  Every lane applies the reference's
// PER-PARTICLE opening criterion to the node:
//   * a lane that accepts the node interacts with its monopole and stores the node's skip index:
//     it ignores every element below that node (my_skip);
//   * the wave descends (e+1) if ANY participating lane must open, else jumps to the skip index.
// Each lane therefore gets exactly the reference's interaction set for its particle.
//
// Load balance.  With the relative criterion a few particles with tiny |a_old| open ten times
// more nodes than the median, and their whole bucket follows.  The element list is therefore cut
// into NS contiguous SEGMENTS, and S wavefronts share one bucket: wavefront `sub` takes segments
// sub, sub+S, ...  A segment can begin anywhere in the list, so a wave entering it first replays
// the opening decision at the segment's ANCESTORS (nodes whose subtree contains the segment
// start; <= 22 of them, precomputed) WITHOUT interacting -- an ancestor's interaction belongs to
// the segment that contains the ancestor's own element.  Ownership by element index makes every
// interaction happen exactly once; the S partial sums per target are added in fixed order
// afterwards (k_combine_grav), so results are deterministic.  No MFMA: irregular fp64 work.
#pragma once
#include "ghip_internal.h"

#define GHIP_MAXANC 24
#define GHIP_MAXSUB 8

struct GravK
{
  double theta;        // ErrTolTheta (0: relative criterion)
  double errtol;       // ErrTolForceAcc
  double boxsize, boxhalf;
  int periodic, unequal;
  int debug_steps;     // GHIP_DEBUG_STEPS=1: GRAVCOST receives the wave's visited-element count
  double rcut, rcut2, asmthfac;  // shortrange
  double fac_intp;     // ewald: 2*EN/BoxSize
};

struct WalkSeg
{
  int ns;                          // number of segments
  int nsub;                        // wavefronts per bucket
  const int *__restrict__ start;   // [ns+1] first element of each segment
  const int *__restrict__ nanc;    // [ns]
  const int *__restrict__ anc;     // [ns][GHIP_MAXANC] ancestors of start[k], root first
};

// 1/sqrt(x) to full fp64 precision from the hardware seed (v_rsq_f64, ~2^-23 relative) with one
// third-order step: e = 1 - x*y^2,  y <- y*(1 + e/2 + 3e^2/8)   (error O(e^3) ~ 2^-69)
__device__ __forceinline__ double d_rsqrt(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double t = x * y;
  double e = fma(-t, y, 1.0);
  double p = fma(0.375, e, 0.5);
  return fma(y * e, p, y);
}

// softened monopole kernel, forcetree.c:2143-2171.  r2 >= h^2: m / r^3 through d_rsqrt.
__device__ __forceinline__ double d_grav_fac(double mass, double r2, double h, double h2,
                                             double &r_out)
{
  if(r2 >= h2)
    {
      double rinv = d_rsqrt(r2);
      r_out = r2 * rinv;
      return mass * rinv * rinv * rinv;
    }
  double r = sqrt(r2);
  r_out = r;
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

// One tree element as the walk reads it: 128 bytes, fetched with two back-to-back
// s_load_dwordx16 so that a step costs ONE memory latency (the element list of the 64^3+64^3
// configuration is ~100 MB: it lives in the Infinity Cache, ~550 cycles away).
//   dwords  0- 7  x, y, z, mass          (node: centre of mass; particle: position)
//   dwords  8-15  cx, cy, cz, len        (node geometry; particle: len = 0)
//   dwords 16-23  len^2, mass*len^2, 0.6*len, aux   (opening-criterion operands, reference's
//                                                    operation order (mass*len)*len)
//   dwords 24-27  skip, pidx, pstart, pcount
struct __attribute__((aligned(128))) WalkElem
{
  double x, y, z, m;
  double cx, cy, cz, len;
  double len2, mlen2, len06, aux;
  int skip, pidx, pstart, pcount;
  int pad[4];
};

typedef int v16i __attribute__((ext_vector_type(16)));

struct ElemRegs
{
  v16i lo, hi;
};

__device__ __forceinline__ double d_f64(const v16i &v, int i)
{
  return __hiloint2double(v[2 * i + 1], v[2 * i]);
}

// wave-uniform element index -> both halves of the record in SGPRs, one wait
__device__ __forceinline__ void d_load_elem(const WalkElem *__restrict__ elems, int e, ElemRegs &R)
{
  const WalkElem *p = elems + e;
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(R.lo), "=&s"(R.hi)
               : "s"(p)
               : "memory");
}

struct WalkLane
{
  double pos_x, pos_y, pos_z, h_i, aold;
  double h2;   // h_i^2
  double acc_x, acc_y, acc_z;
  int nint;
  int my_skip;
};

// NEAREST (forcetree.c:49) with one compare: |x| > boxhalf ? x - copysign(box, x) : x
// (x - (-box) and x + box are the same IEEE operation, so results are identical)
__device__ __forceinline__ double d_nearest1(double x, double boxsize, double boxhalf)
{
  return (fabs(x) > boxhalf) ? x - copysign(boxsize, x) : x;
}

// One element of the list for all 64 lanes.  OWNED = false replays only the opening decision
// (ancestor of a segment).  Returns the next element index (wave-uniform).
// mq[e] = (len^2, mass*len^2, 0.6*len, aux) with the reference's operation order
// ((mass*len)*len, forcetree.c:2085), so every comparison sees the same doubles.
template <int MODE, bool PERIODIC, bool OWNED>
__device__ __forceinline__ int d_walk_element(int e, const ElemRegs &R, const GravK &p,
                                              const float *__restrict__ srtab,
                                              const double4 *__restrict__ ewtab, WalkLane &W)
{
  const double4 v = make_double4(d_f64(R.lo, 0), d_f64(R.lo, 1), d_f64(R.lo, 2), d_f64(R.lo, 3));
  const int4 k = make_int4(R.hi[8], R.hi[9], R.hi[10], R.hi[11]);
  const bool act = (e >= W.my_skip);
  int next;

  double dx = v.x - W.pos_x, dy = v.y - W.pos_y, dz = v.z - W.pos_z;
  if(MODE == GHIP_WALK_EWALD || PERIODIC)
    {
      dx = d_nearest1(dx, p.boxsize, p.boxhalf);
      dy = d_nearest1(dy, p.boxsize, p.boxhalf);
      dz = d_nearest1(dz, p.boxsize, p.boxhalf);
    }
  const double r2 = dx * dx + dy * dy + dz * dz;
  const double mass = v.w;
  double h = W.h_i, h2 = W.h2;
  bool interact = act;

  if(LK_IS_PARTICLE(k))
    {
      next = e + 1;
      if(MODE != GHIP_WALK_EWALD && p.unequal)
        {
          double sj = d_f64(R.hi, 3);
          if(h < sj)
            {
              h = sj;
              h2 = h * h;
            }
        }
    }
  else
    {
      const double4 c = make_double4(d_f64(R.lo, 4), d_f64(R.lo, 5), d_f64(R.lo, 6), d_f64(R.lo, 7));
      const double4 q = make_double4(d_f64(R.hi, 0), d_f64(R.hi, 1), d_f64(R.hi, 2), d_f64(R.hi, 3));
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
                  double d0 = c.x - W.pos_x, d1 = c.y - W.pos_y, d2 = c.z - W.pos_z;
                  if(PERIODIC)
                    {
                      d0 = d_nearest1(d0, p.boxsize, p.boxhalf);
                      d1 = d_nearest1(d1, p.boxsize, p.boxhalf);
                      d2 = d_nearest1(d2, p.boxsize, p.boxhalf);
                    }
                  if(fabs(d0) > eff || fabs(d1) > eff || fabs(d2) > eff)
                    {
                      interact = false;
                      W.my_skip = k.x;
                    }
                }
            }
          if(interact)
            {
              // opening criterion, forcetree.c:2074-2105
              if(p.theta != 0)
                open = (q.x > r2 * p.theta * p.theta);
              else
                {
                  open = (q.y > r2 * r2 * W.aold);
                  if(!open)
                    open = (fabs(c.x - W.pos_x) < q.z) && (fabs(c.y - W.pos_y) < q.z) &&
                           (fabs(c.z - W.pos_z) < q.z);
                }
              if(MODE == GHIP_WALK_EWALD)
                {
                  // forcetree.c:3039-3088: the correction is smooth, so an "open" verdict is
                  // overridden unless the cell straddles the half-box or is large
                  if(open)
                    {
                      double u0 = d_nearest1(c.x - W.pos_x, p.boxsize, p.boxhalf);
                      double u1 = d_nearest1(c.y - W.pos_y, p.boxsize, p.boxhalf);
                      double u2 = d_nearest1(c.z - W.pos_z, p.boxsize, p.boxhalf);
                      double lim = 0.5 * (p.boxsize - len);
                      open = (fabs(u0) > lim) || (fabs(u1) > lim) || (fabs(u2) > lim) ||
                             (len > 0.20 * p.boxsize);
                    }
                }
              else if(p.unequal && !open)
                {
                  // forcetree.c:2108-2124
                  double a = q.w;
                  double ms = fabs(a);
                  if(h < ms)
                    {
                      h = ms;
                      h2 = h * h;
                      if(r2 < h2 && a < 0)
                        open = true;
                    }
                }
              if(open)
                interact = false;
              else
                W.my_skip = k.x;
            }
        }
      next = __any(open) ? e + 1 : k.x;
    }

  if(OWNED && interact)
    {
      if(MODE == GHIP_WALK_EWALD)
        {
          double fx, fy, fz;
          d_ewald_interp(ewtab, p.fac_intp, dx, dy, dz, fx, fy, fz);
          W.acc_x += mass * fx;
          W.acc_y += mass * fy;
          W.acc_z += mass * fz;
          W.nint++;
        }
      else
        {
          double r;
          double fac = d_grav_fac(mass, r2, h, h2, r);
          if(MODE == GHIP_WALK_SHORTRANGE)
            {
              // forcetree.c:2739-2752
              int tabindex = (int) (p.asmthfac * r);
              if(tabindex < GHIP_NTAB)
                {
                  fac *= srtab[tabindex];
                  W.acc_x += dx * fac;
                  W.acc_y += dy * fac;
                  W.acc_z += dz * fac;
                  W.nint++;
                }
            }
          else
            {
              W.acc_x += dx * fac;
              W.acc_y += dy * fac;
              W.acc_z += dz * fac;
              if(mass > 0)
                W.nint++;
            }
        }
    }
  return next;
}

__device__ __forceinline__ int d_wave_min_i32(int v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      int o = __shfl_xor(v, off, 64);
      v = o < v ? o : v;
    }
  return v;
}

// partial results: [nsub][nt] per component.
// Grid: a multiple of 8 blocks; the block index is remapped so that each XCD (blocks are dealt
// round-robin over the 8 XCDs) works through ONE contiguous eighth of the buckets: neighbouring
// buckets read the same deep tree nodes, which then stay in that XCD's 4 MB L2.
template <int MODE, bool PERIODIC>
__global__ void __launch_bounds__(GHIP_BLOCK)
k_grav_walk(int nelem, const WalkElem *__restrict__ elems, WalkSeg sg, int nt,
            const int *__restrict__ tgt, const double *__restrict__ tx,
            const double *__restrict__ ty, const double *__restrict__ tz,
            const double *__restrict__ tsoft, const double *__restrict__ toldacc, GravK p,
            const float *__restrict__ srtab, const double4 *__restrict__ ewtab,
            double *__restrict__ pax, double *__restrict__ pay, double *__restrict__ paz,
            int *__restrict__ pcost, unsigned long long *__restrict__ counter)
{
  const int lane = threadIdx.x & 63;
  const int per_xcd = gridDim.x >> 3;
  const int lblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = (lblock * GHIP_BLOCK + threadIdx.x) >> 6;
  const int bucket = wave / sg.nsub;
  const int sub = wave - bucket * sg.nsub;
  if(bucket * 64 >= nt)
    return;
  const int ti = bucket * 64 + lane;
  const bool valid = ti < nt;
  const int s = valid ? (tgt ? tgt[ti] : ti) : 0;

  WalkLane W;
  W.pos_x = W.pos_y = W.pos_z = 0;
  W.h_i = 1;
  W.aold = 0;
  W.h2 = 1;
  if(valid)
    {
      W.pos_x = tx[s];
      W.pos_y = ty[s];
      W.pos_z = tz[s];
      W.h_i = tsoft[s];
      W.aold = p.errtol * toldacc[s];
      W.h2 = W.h_i * W.h_i;
    }
  W.acc_x = W.acc_y = W.acc_z = 0;
  W.nint = 0;
  unsigned int steps = 0;
  ElemRegs R;

  for(int kseg = sub; kseg < sg.ns; kseg += sg.nsub)
    {
      const int s0 = sg.start[kseg], s1 = sg.start[kseg + 1];
      W.my_skip = valid ? 0 : 0x7fffffff;
      // replay the opening decisions at the ancestors of this segment's first element
      const int na = sg.nanc[kseg];
      for(int a = 0; a < na; a++)
        {
          int ea = __builtin_amdgcn_readfirstlane(sg.anc[kseg * GHIP_MAXANC + a]);
          d_load_elem(elems, ea, R);
          d_walk_element<MODE, PERIODIC, false>(ea, R, p, srtab, ewtab, W);
          steps++;
        }
      int first = d_wave_min_i32(W.my_skip);   // every lane below an accepted ancestor: jump
      int e = first > s0 ? first : s0;
      while(e < s1)
        {
          e = __builtin_amdgcn_readfirstlane(e);
          steps++;
          d_load_elem(elems, e, R);
          e = d_walk_element<MODE, PERIODIC, true>(e, R, p, srtab, ewtab, W);
        }
    }

  if(valid)
    {
      const size_t o = (size_t) sub * nt + ti;
      pax[o] = W.acc_x;
      pay[o] = W.acc_y;
      paz[o] = W.acc_z;
      pcost[o] = p.debug_steps ? (int) steps : W.nint;
    }
  unsigned long long tot = d_wave_sum_u64((unsigned long long) W.nint);
  if(lane == 0 && tot)
    atomicAdd(counter, tot);
  if(lane == 0)
    atomicAdd(counter + 8, (unsigned long long) steps);
}

// segment table of a tree: equal-length slices of the element list + the ancestor chain of each
// slice's first element (one thread per segment; a pre-order descent from the root)
__global__ void k_build_segments(int nelem, const int4 *__restrict__ lk, int ns,
                                 int *__restrict__ start, int *__restrict__ nanc,
                                 int *__restrict__ anc)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k > ns)
    return;
  long long chunk = ((long long) nelem + ns - 1) / ns;
  long long s0l = (long long) k * chunk;
  int s0 = (int) (s0l > nelem ? nelem : s0l);
  if(k == ns)
    s0 = nelem;
  start[k] = s0;
  if(k == ns)
    return;
  int na = 0, e = 0;
  while(e < s0)
    {
      int4 q = lk[e];
      if(!LK_IS_PARTICLE(q) && q.x > s0)
        {
          if(na < GHIP_MAXANC)
            anc[k * GHIP_MAXANC + na] = e;
          na++;
          e = e + 1;
        }
      else
        e = q.x;
    }
  nanc[k] = na < GHIP_MAXANC ? na : GHIP_MAXANC;
}

// sum the per-wavefront partial results of each target in fixed order and scatter to host order;
// EWALD adds to the stored values (forcetree.c:3190-3193)
__global__ void k_combine_grav(int nt, int nsub, const int *__restrict__ tgt,
                               const int *__restrict__ perm, const double *__restrict__ pax,
                               const double *__restrict__ pay, const double *__restrict__ paz,
                               const int *__restrict__ pcost, int n, double *__restrict__ oacc,
                               int *__restrict__ ocost, int accumulate, int debug_max)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  double a0 = 0, a1 = 0, a2 = 0;
  int c = 0;
  for(int s = 0; s < nsub; s++)
    {
      size_t o = (size_t) s * nt + ti;
      a0 += pax[o];
      a1 += pay[o];
      a2 += paz[o];
      c = debug_max ? (pcost[o] > c ? pcost[o] : c) : c + pcost[o];
    }
  int i = perm ? perm[tgt[ti]] : ti;
  if(accumulate)
    {
      oacc[i] += a0;
      oacc[(size_t) n + i] += a1;
      oacc[2 * (size_t) n + i] += a2;
      ocost[i] += c;
    }
  else
    {
      oacc[i] = a0;
      oacc[(size_t) n + i] = a1;
      oacc[2 * (size_t) n + i] = a2;
      ocost[i] = c;
    }
}

// the walk's 128-byte element records from the tree arrays (see WalkElem)
__global__ void k_fill_elems(int nelem, const double4 *__restrict__ xm,
                             const double4 *__restrict__ cl, const int4 *__restrict__ lk,
                             const double *__restrict__ aux, WalkElem *__restrict__ out)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  double4 v = xm[e], c = cl[e];
  int4 k = lk[e];
  WalkElem r;
  r.x = v.x;
  r.y = v.y;
  r.z = v.z;
  r.m = v.w;
  r.cx = c.x;
  r.cy = c.y;
  r.cz = c.z;
  r.len = c.w;
  if(LK_IS_PARTICLE(k))
    r.len2 = r.mlen2 = r.len06 = 0.0;
  else
    {
      r.len2 = c.w * c.w;
      r.mlen2 = v.w * c.w * c.w;
      r.len06 = 0.60 * c.w;
    }
  r.aux = aux[e];
  r.skip = k.x;
  r.pidx = k.y;
  r.pstart = k.z;
  r.pcount = k.w;
  r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0;
  out[e] = r;
}
