// ghip_tree.hip -- device oct-tree build for the gfx950 force path.
//
// Replaces force_treebuild()/force_treebuild_single()/force_update_node_recursive()
// (forcetree.c:67-872).  The reference inserts particles one by one into an oct-tree whose cells
// are fixed by DomainCorner/DomainLen and the Morton bits of each particle (forcetree.c:181-217)
// and then threads it with nextnode/sibling links in depth-first order (forcetree.c:468-872).
// Here the same cells are derived in parallel from the sorted Morton keys:
//   * a cell at level L exists for every maximal run of >= 2 sorted keys sharing their top 3L
//     bits -- exactly the internal nodes the reference's insertion creates (single-child chains
//     included), minus its empty / single-particle top-level nodes, which never contribute to a
//     force (forcetree.c:1996-2004);
//   * the depth-first ("threaded") order is the order of the sorted keys, so the tree is stored
//     as ONE pre-order element list (nodes and particles interleaved) in which "nextnode" is
//     simply e+1 and "sibling" is a stored skip index.  A wavefront walks it with a wave-uniform
//     element index, i.e. with scalar loads.
// Children are visited in Morton-octant order (x lowest bit), the order of Nodes[].u.suns[].
#include <hipcub/hipcub.hpp>

#include "ghip_internal.h"

// ---------------------------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long d_spread3(unsigned long long v)
{
  // spread the low 21 bits of v so that bit b lands at bit 3b
  v &= 0x1fffffULL;
  v = (v | (v << 32)) & 0x001f00000000ffffULL;
  v = (v | (v << 16)) & 0x001f0000ff0000ffULL;
  v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
  v = (v | (v << 2)) & 0x1249249249249249ULL;
  return v;
}

// peano.c:320-333 morton_key: x is the lowest bit of each triplet
__device__ __forceinline__ unsigned long long d_morton21(int x, int y, int z)
{
  return d_spread3((unsigned long long) x) | (d_spread3((unsigned long long) y) << 1) |
         (d_spread3((unsigned long long) z) << 2);
}

__global__ void k_morton_from_pos(int n, const double *__restrict__ x, const double *__restrict__ y,
                                  const double *__restrict__ z, double cx, double cy, double cz,
                                  double fac, unsigned long long *__restrict__ key,
                                  int *__restrict__ idx)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  // forcetree.c:181-185: (int) ((Pos - DomainCorner) * DomainFac)
  int ix = (int) ((x[i] - cx) * fac);
  int iy = (int) ((y[i] - cy) * fac);
  int iz = (int) ((z[i] - cz) * fac);
  key[i] = d_morton21(ix, iy, iz);
  idx[i] = i;
}

__global__ void k_morton_from_ints(int n, const int *__restrict__ x, const int *__restrict__ y,
                                   const int *__restrict__ z, int bits,
                                   unsigned long long *__restrict__ key)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  unsigned long long m = d_morton21(x[i], y[i], z[i]);
  if(bits < 21)
    m &= ((1ULL << (3 * bits)) - 1ULL);
  key[i] = m;
}

// peano.c:300-316 peano_hilbert_key.  The reference drives a 48-state table; the same curve is
// generated here from its definition: base octant order ph_base (octant = 4*xbit + 2*ybit + zbit)
// and the cube symmetry carried by each sub-cell (signed axis permutation, out[i] =
// in[perm[i]] ^ flip[i]).  Orientation g starts as the identity; per level:
// local = g(octant); digit = ph_base[local]; g <- child[local] o g.
__constant__ unsigned char c_ph_base[8] = {0, 7, 1, 6, 3, 4, 2, 5};
__constant__ unsigned char c_ph_perm[8][3] = {{0, 2, 1}, {0, 2, 1}, {2, 1, 0}, {2, 1, 0},
                                              {0, 1, 2}, {0, 1, 2}, {2, 1, 0}, {2, 1, 0}};
__constant__ unsigned char c_ph_flip[8][3] = {{0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {1, 0, 1},
                                              {1, 1, 0}, {1, 1, 0}, {0, 0, 0}, {1, 0, 1}};

// the same curve for one 21-bit triplet (target bucketing: 64 Peano-Hilbert-consecutive particles
// fill a box 2.7x smaller than 64 Morton-consecutive ones, so the lanes of a wavefront agree on
// more opening decisions)
__device__ __forceinline__ unsigned long long d_peano21(int x, int y, int z)
{
  int perm0 = 0, perm1 = 1, perm2 = 2, f0 = 0, f1 = 0, f2 = 0;
  unsigned long long k = 0;
  for(int b = GHIP_BITS - 1; b >= 0; b--)
    {
      int bit[3] = {(x >> b) & 1, (y >> b) & 1, (z >> b) & 1};
      int w0 = bit[perm0] ^ f0, w1 = bit[perm1] ^ f1, w2 = bit[perm2] ^ f2;
      int local = w0 * 4 + w1 * 2 + w2;
      k = (k << 3) | c_ph_base[local];
      int p[3] = {perm0, perm1, perm2}, f[3] = {f0, f1, f2};
      int a0 = c_ph_perm[local][0], a1 = c_ph_perm[local][1], a2 = c_ph_perm[local][2];
      perm0 = p[a0];
      perm1 = p[a1];
      perm2 = p[a2];
      f0 = f[a0] ^ c_ph_flip[local][0];
      f1 = f[a1] ^ c_ph_flip[local][1];
      f2 = f[a2] ^ c_ph_flip[local][2];
    }
  return k;
}

// the leading `levels` (<= 10) digits of the same key
__device__ __forceinline__ unsigned int d_peano_top(int x, int y, int z, int levels)
{
  int perm0 = 0, perm1 = 1, perm2 = 2, f0 = 0, f1 = 0, f2 = 0;
  unsigned int k = 0;
  for(int b = GHIP_BITS - 1; b >= GHIP_BITS - levels; b--)
    {
      int bit[3] = {(x >> b) & 1, (y >> b) & 1, (z >> b) & 1};
      int w0 = bit[perm0] ^ f0, w1 = bit[perm1] ^ f1, w2 = bit[perm2] ^ f2;
      int local = w0 * 4 + w1 * 2 + w2;
      k = (k << 3) | c_ph_base[local];
      int p[3] = {perm0, perm1, perm2}, f[3] = {f0, f1, f2};
      int a0 = c_ph_perm[local][0], a1 = c_ph_perm[local][1], a2 = c_ph_perm[local][2];
      perm0 = p[a0];
      perm1 = p[a1];
      perm2 = p[a2];
      f0 = f[a0] ^ c_ph_flip[local][0];
      f1 = f[a1] ^ c_ph_flip[local][1];
      f2 = f[a2] ^ c_ph_flip[local][2];
    }
  return k;
}

__global__ void k_peano_from_ints(int n, const int *__restrict__ x, const int *__restrict__ y,
                                  const int *__restrict__ z, int bits,
                                  unsigned long long *__restrict__ key)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int v[3] = {x[i], y[i], z[i]};
  int perm0 = 0, perm1 = 1, perm2 = 2, f0 = 0, f1 = 0, f2 = 0;
  unsigned long long k = 0;
  for(int b = bits - 1; b >= 0; b--)
    {
      int bit[3] = {(v[0] >> b) & 1, (v[1] >> b) & 1, (v[2] >> b) & 1};
      int w0 = bit[perm0] ^ f0, w1 = bit[perm1] ^ f1, w2 = bit[perm2] ^ f2;
      int local = w0 * 4 + w1 * 2 + w2;
      k = (k << 3) | c_ph_base[local];
      int p[3] = {perm0, perm1, perm2}, f[3] = {f0, f1, f2};
      int a0 = c_ph_perm[local][0], a1 = c_ph_perm[local][1], a2 = c_ph_perm[local][2];
      perm0 = p[a0];
      perm1 = p[a1];
      perm2 = p[a2];
      f0 = f[a0] ^ c_ph_flip[local][0];
      f1 = f[a1] ^ c_ph_flip[local][1];
      f2 = f[a2] ^ c_ph_flip[local][2];
    }
  key[i] = k;
}

extern "C" int ghip_peano_hilbert_keys(ghip_ctx *ctx, int n, const int *x, const int *y,
                                       const int *z, int bits, unsigned long long *keys)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || n < 0 || bits < 1 || bits > 21)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_peano_hilbert_keys: bad arguments");
  if(n == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) n * 3 * sizeof(int) + 8 + (size_t) n * 8));
  int *dx = P<int>(ctx->stage), *dy = dx + n, *dz = dy + n;
  unsigned long long *dk = reinterpret_cast<unsigned long long *>(
    reinterpret_cast<char *>(ctx->stage.p) + (((size_t) 3 * n * sizeof(int) + 7) & ~(size_t) 7));
  HIPCHK(hipMemcpyAsync(dx, x, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dy, y, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dz, z, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  k_peano_from_ints<<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, dx, dy, dz, bits, dk);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(keys, dk, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return GHIP_OK;
}

extern "C" int ghip_morton_keys(ghip_ctx *ctx, int n, const int *x, const int *y, const int *z,
                                int bits, unsigned long long *keys)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || n < 0 || bits < 1 || bits > 21)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_morton_keys: bad arguments");
  if(n == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) n * 3 * sizeof(int) + 8 + (size_t) n * 8));
  int *dx = P<int>(ctx->stage), *dy = dx + n, *dz = dy + n;
  unsigned long long *dk = reinterpret_cast<unsigned long long *>(
    reinterpret_cast<char *>(ctx->stage.p) + (((size_t) 3 * n * sizeof(int) + 7) & ~(size_t) 7));
  HIPCHK(hipMemcpyAsync(dx, x, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dy, y, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dz, z, (size_t) n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  k_morton_from_ints<<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, dx, dy, dz, bits, dk);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(keys, dk, (size_t) n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// structure from sorted keys
// ---------------------------------------------------------------------------------------------
// number of leading 3-bit digits two 63-bit keys share
__device__ __forceinline__ int d_common_levels(unsigned long long a, unsigned long long b)
{
  unsigned long long x = a ^ b;
  if(x == 0)
    return GHIP_BITS;
  return (__clzll((long long) x) - 1) / 3;
}

__global__ void k_prefix_levels(int n, const unsigned long long *__restrict__ skey,
                                int *__restrict__ cpl, int *__restrict__ maxlevel)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = -1;
  if(i < n)
    {
      c = (i + 1 < n) ? d_common_levels(skey[i], skey[i + 1]) : -1;
      cpl[i] = c;
    }
  // deepest node level of the tree = largest shared-digit count: one same-address access per
  // block of 1024 (per wavefront they cost 45 us at 5e5 particles)
  for(int off = 32; off > 0; off >>= 1)
    {
      int o = __shfl_xor(c, off, 64);
      c = o > c ? o : c;
    }
  __shared__ int wmax[16];
  if((threadIdx.x & 63) == 0)
    wmax[threadIdx.x >> 6] = c;
  __syncthreads();
  if(threadIdx.x == 0)
    {
      int nw = (blockDim.x + 63) >> 6;
      for(int w = 1; w < nw; w++)
        c = wmax[w] > c ? wmax[w] : c;
      if(c > *(volatile int *) maxlevel)
        atomicMax(maxlevel, c);
    }
}

__global__ void k_node_counts(int n, const int *__restrict__ cpl, int *__restrict__ cnt)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int cprev = (i > 0) ? cpl[i - 1] : -1;
  int d = cpl[i] - cprev;
  cnt[i] = d > 0 ? d : 0;
}

// one thread per particle: emits the nodes that START at this particle (levels cprev+1..c_i, in
// increasing depth = pre-order) followed by the particle itself.
__global__ void k_emit_elements(int n, int nelem, const unsigned long long *__restrict__ skey,
                                const int *__restrict__ cpl, const int *__restrict__ cnt,
                                const int *__restrict__ nb, const double *__restrict__ px,
                                const double *__restrict__ py, const double *__restrict__ pz,
                                const double *__restrict__ pm, const double *__restrict__ paux,
                                double ccx, double ccy, double ccz, double dlen,
                                double4 *__restrict__ xm, double4 *__restrict__ cl,
                                int4 *__restrict__ lk, double *__restrict__ aux)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int cprev = (i > 0) ? cpl[i - 1] : -1;
  int ci = cpl[i];
  int base = i + nb[i];
  unsigned long long ki = skey[i];

  if(ci > cprev)
    {
      // geometry of the level-(cprev+1) cell by the reference's recurrence (forcetree.c:263-279):
      // len halves, centre moves by a quarter of the parent's len per level
      double len = dlen, cx = ccx, cy = ccy, cz = ccz;
      for(int l = 1; l <= cprev + 1; l++)
        {
          int digit = (int) ((ki >> (63 - 3 * l)) & 7);
          double lenhalf = 0.25 * len;
          cx = (digit & 1) ? cx + lenhalf : cx - lenhalf;
          cy = (digit & 2) ? cy + lenhalf : cy - lenhalf;
          cz = (digit & 4) ? cz + lenhalf : cz - lenhalf;
          len = 0.5 * len;
        }
      for(int L = cprev + 1; L <= ci; L++)
        {
          int e = base + (L - cprev - 1);
          int sh = 63 - 3 * L;
          unsigned long long prefix = ki >> sh;
          // end of the cell's particle range: gallop, then bisect (most cells hold a handful of
          // particles, so the doubling phase ends after two or three probes)
          int lo = i + 1, hi = n;
          for(int step = 1; lo + step < n; step <<= 1)
            {
              if((skey[lo + step] >> sh) == prefix)
                lo = lo + step + 1;
              else
                {
                  hi = lo + step;
                  break;
                }
            }
          while(lo < hi)
            {
              int mid = (lo + hi) >> 1;
              if((skey[mid] >> sh) == prefix)
                lo = mid + 1;
              else
                hi = mid;
            }
          int end = lo;
          int skip = (end >= n) ? nelem : end + nb[end];
          lk[e] = make_int4(skip, -(L + 1), i, end - i);
          cl[e] = make_double4(cx, cy, cz, len);
          xm[e] = make_double4(0, 0, 0, 0);
          aux[e] = 0;
          if(L < ci)
            {
              int digit = (int) ((ki >> (63 - 3 * (L + 1))) & 7);
              double lenhalf = 0.25 * len;
              cx = (digit & 1) ? cx + lenhalf : cx - lenhalf;
              cy = (digit & 2) ? cy + lenhalf : cy - lenhalf;
              cz = (digit & 4) ? cz + lenhalf : cz - lenhalf;
              len = 0.5 * len;
            }
        }
    }
  int pe = base + cnt[i];
  double x = px[i], y = py[i], z = pz[i];
  lk[pe] = make_int4(pe + 1, i, i, 1);
  xm[pe] = make_double4(x, y, z, pm[i]);
  cl[pe] = make_double4(x, y, z, 0.0);
  aux[pe] = paux[i];
}

// multipole pass of one level (force_update_node_recursive, forcetree.c:468-872): children are
// the elements reached from e+1 by following skip links until the node's own skip.
// GRAV: aux = largest ForceSoftening below, negated when the node mixes softenings
//       (BITFLAG_MAX_SOFTENING_TYPE / BITFLAG_MIXED_SOFTENINGS_IN_NODE, forcetree.c:612-700)
// GAS : aux = hmax (Extnodes[].hmax, forcetree.c:593, 676)
template <bool GRAV, bool MOMENTS>
__global__ void k_node_level(int nelem, int level, double4 *__restrict__ xm,
                             const double4 *__restrict__ cl, const int4 *__restrict__ lk,
                             double *__restrict__ aux, bool adaptive)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y != -(level + 1))
    return;
  double mass = 0, sx = 0, sy = 0, sz = 0;
  double amax = 0;
  bool aset = false, mixed = false;
  for(int c = e + 1; c < me.x;)
    {
      int4 ck = lk[c];
      if(MOMENTS)
        {
          double4 v = xm[c];
          mass += v.w;
          sx += v.w * v.x;
          sy += v.w * v.y;
          sz += v.w * v.z;
        }
      double a = aux[c];
      if(GRAV)
        {
          bool cmixed = a < 0;
          a = fabs(a);
          mixed |= cmixed;
          if(!aset)
            {
              amax = a;
              aset = true;
            }
          else if(a > amax)
            {
              amax = a;
              mixed = true;
            }
          else if(a < amax)
            mixed = true;
        }
      else
        {
          if(a > amax)
            amax = a;
        }
      c = ck.x;
    }
  if(MOMENTS)
    {
      double4 c4 = cl[e];
      if(mass != 0)
        {
          sx /= mass;
          sy /= mass;
          sz /= mass;
        }
      else
        {
          sx = c4.x;
          sy = c4.y;
          sz = c4.z;
        }
      xm[e] = make_double4(sx, sy, sz, mass);
    }
  // ADAPTIVE_GRAVSOFT_FORGAS: NODE.maxsoft opens the node for every target inside it, mixed or not
  // (forcetree.c:2125-2139)
  aux[e] = (GRAV && (mixed || adaptive)) ? -amax : amax;
}

// ---------------------------------------------------------------------------------------------
// gathers into tree order
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_f64(int n, const int *__restrict__ perm, const double *__restrict__ src,
                             double *__restrict__ dst)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    dst[s] = src[perm[s]];
}

// gas records in gas-tree order: gp = x,y,z,m,vx,vy,vz,h ; gq = P,rho,f,divv,curl,timestep,0,0
__global__ void k_gather_gas(int ng, const int *__restrict__ perm, const double *__restrict__ x,
                             const double *__restrict__ y, const double *__restrict__ z,
                             const double *__restrict__ m, const double *__restrict__ vx,
                             const double *__restrict__ vy, const double *__restrict__ vz,
                             const double *__restrict__ h, const double *__restrict__ pres,
                             const double *__restrict__ rho, const double *__restrict__ dhf,
                             const double *__restrict__ divv, const double *__restrict__ curl,
                             const int *__restrict__ timebin, double *__restrict__ gp,
                             double *__restrict__ gq)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= ng)
    return;
  int i = perm[s];
  double *r = gp + (size_t) 8 * s;
  r[0] = x[i];
  r[1] = y[i];
  r[2] = z[i];
  r[3] = m[i];
  r[4] = vx[i];
  r[5] = vy[i];
  r[6] = vz[i];
  r[7] = h[i];
  double *q = gq + (size_t) 8 * s;
  int tb = timebin[i];
  q[0] = pres[i];
  q[1] = rho[i];
  q[2] = dhf[i];
  q[3] = divv[i];
  q[4] = curl[i];
  q[5] = (double) (tb ? (1 << tb) : 0);  // hydra.c:966 timestep
  q[6] = 0;
  q[7] = 0;
}

int ghip_gather_f64(ghip_ctx *ctx, int n, const int *perm, const double *src, double *dst)
{
  if(n <= 0)
    return GHIP_OK;
  k_gather_f64<<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, perm, src, dst);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

__global__ void k_mark_active(int nact, const int *__restrict__ act, const int *__restrict__ iperm,
                              int limit, int *__restrict__ flags)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nact)
    return;
  int i = act[a];
  if(i >= 0 && i < limit)
    flags[iperm[i]] = 1;
}

// ---------------------------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------------------------
// ---- two-step key sort: a radix sort of (top 32 key bits, index) pairs -- half the passes and
// two thirds of the bytes of a 64-bit sort -- then the rare runs of equal top bits (particles
// closer than 2^-11 of the domain) are ordered by their full keys in place
__global__ void k_key_hi(int n, int shift, const unsigned long long *__restrict__ key,
                         unsigned int *__restrict__ hi)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    hi[i] = (unsigned int) (key[i] >> shift);
}

__global__ void k_gather_keys(int n, const int *__restrict__ perm,
                              const unsigned long long *__restrict__ key,
                              unsigned long long *__restrict__ skey)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    skey[s] = key[perm[s]];
}

// one thread per run start: insertion sort by (full key, index) -- the order a stable 64-bit
// sort gives.  Runs longer than `maxrun` are left alone and reported (the caller falls back).
__global__ void k_fix_runs(int n, int maxrun, unsigned long long *__restrict__ skey,
                           int *__restrict__ perm, int *__restrict__ longest)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  unsigned int h = (unsigned int) (skey[s] >> 31);
  if(s > 0 && (unsigned int) (skey[s - 1] >> 31) == h)
    return;   // not a run start
  int e = s + 1;
  while(e < n && (unsigned int) (skey[e] >> 31) == h)
    e++;
  int len = e - s;
  if(len == 1)
    return;
  if(len > maxrun)
    {
      atomicMax(longest, len);
      return;
    }
  for(int a = s + 1; a < e; a++)
    {
      unsigned long long k = skey[a];
      int p = perm[a];
      int b = a - 1;
      while(b >= s && (skey[b] > k || (skey[b] == k && perm[b] > p)))
        {
          skey[b + 1] = skey[b];
          perm[b + 1] = perm[b];
          b--;
        }
      skey[b + 1] = k;
      perm[b + 1] = p;
    }
}

__global__ void k_iota_tree(int n, int *__restrict__ a)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    a[i] = i;
}

__global__ void k_gather_i32(int n, const int *__restrict__ idx, const int *__restrict__ src,
                             int *__restrict__ dst)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    dst[i] = src[idx[i]];
}

static int cub_tmp(ghip_ctx *ctx, size_t bytes)
{
  return ghip_ensure(ctx, ctx->cubtmp, bytes + 256);
}

// ---------------------------------------------------------------------------------------------
// tree construction driver.  The build is latency-bound (dozens of launches of a few us each),
// so the steps are arranged for few launches and ONE host round trip for both trees:
//   sort (gravity tree)  ->  gas order by compaction  ->  node counts of both trees  ->  read
//   back {maxlevel, nnodes} of both  ->  emit + moments of both  ->  curve order of the targets
// ---------------------------------------------------------------------------------------------
__global__ void k_gather5(int n, const int *__restrict__ perm, const double *__restrict__ x,
                          const double *__restrict__ y, const double *__restrict__ z,
                          const double *__restrict__ m, const double *__restrict__ a,
                          double *__restrict__ ox, double *__restrict__ oy,
                          double *__restrict__ oz, double *__restrict__ om,
                          double *__restrict__ oa, int *__restrict__ iperm)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  int i = perm[s];
  ox[s] = x[i];
  oy[s] = y[i];
  oz[s] = z[i];
  om[s] = m[i];
  oa[s] = a[i];
  iperm[i] = s;
}

// out[0] = deepest level, out[1] = number of nodes, out[2] = longest unsorted key run
__global__ void k_tree_info(int n, const int *__restrict__ nb, const int *__restrict__ cnt,
                            const int *__restrict__ dinfo, int *__restrict__ out)
{
  out[0] = dinfo[0];
  out[1] = nb[n - 1] + cnt[n - 1];
  out[2] = dinfo[1];
}

// gas flags in gravity-tree order (gas = host index below ngas)
__global__ void k_gas_flags(int n, int ngas, const int *__restrict__ perm, int *__restrict__ flag)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    flag[s] = perm[s] < ngas ? 1 : 0;
}

// the gas tree's sorted order is the gravity tree's with the other types removed (same keys,
// same tie order)
__global__ void k_gas_compact(int n, int ngas, const int *__restrict__ perm,
                              const unsigned long long *__restrict__ skey,
                              const int *__restrict__ rank, int *__restrict__ gperm,
                              unsigned long long *__restrict__ gskey)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  int i = perm[s];
  if(i < ngas)
    {
      int r = rank[s];
      gperm[r] = i;
      gskey[r] = skey[s];
    }
}

// flags of the curve-ordered gravity-tree indices, and their compaction to gas-tree indices
__global__ void k_gas_flags_ph(int n, int ngas, const int *__restrict__ phorder,
                               const int *__restrict__ perm, int *__restrict__ flag)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(j < n)
    flag[j] = perm[phorder[j]] < ngas ? 1 : 0;
}

__global__ void k_gas_compact_ph(int n, int ngas, const int *__restrict__ phorder,
                                 const int *__restrict__ perm, const int *__restrict__ rank,
                                 const int *__restrict__ pos, int *__restrict__ gphorder)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(j >= n)
    return;
  int s = phorder[j];
  if(perm[s] < ngas)
    gphorder[pos[j]] = rank[s];
}

__global__ void k_peano_hi(int n, int levels, const double *__restrict__ x,
                           const double *__restrict__ y, const double *__restrict__ z, double cx,
                           double cy, double cz, double fac, unsigned int *__restrict__ hi,
                           int *__restrict__ idx)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  hi[i] = d_peano_top((int) ((x[i] - cx) * fac), (int) ((y[i] - cy) * fac),
                      (int) ((z[i] - cz) * fac), levels);
  idx[i] = i;
}

static int exclusive_sum(ghip_ctx *ctx, const int *in, int *out, int n, hipStream_t on = nullptr)
{
  hipStream_t st = on ? on : ctx->stream;
  size_t tb = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, n, st));
  GCHK(cub_tmp(ctx, tb));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cubtmp.p, tb, in, out, n, st));
  return GHIP_OK;
}

static int *tree_dinfo(ghip_ctx *ctx, bool gas)
{
  return reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 48) + (gas ? 2 : 0);
}

// Morton keys of the gravity tree's particles, sorted -> t.skey, t.perm
static int sort_by_key(ghip_ctx *ctx, TreeDev &t, int n, const double *x, const double *y,
                       const double *z, bool wide)
{
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, t.key, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, t.skey, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, t.idx, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.perm, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.phkey, (size_t) n * 8));
  double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);  // DomainFac, domain.c:2012
  k_morton_from_pos<<<cdiv(n, 256), 256, 0, st>>>(n, x, y, z, ctx->corner[0], ctx->corner[1],
                                                  ctx->corner[2], fac,
                                                  P<unsigned long long>(t.key), P<int>(t.idx));
  HIPCHK(hipGetLastError());
  size_t tb = 0;
  if(!wide)
    {
      // top 32 key bits first, then the (rare) runs of equal top bits by their full keys
      unsigned int *hi_in = P<unsigned int>(t.phkey), *hi_out = hi_in + n;
      k_key_hi<<<cdiv(n, 256), 256, 0, st>>>(n, 31, P<unsigned long long>(t.key), hi_in);
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, hi_in, hi_out, P<int>(t.idx),
                                                P<int>(t.perm), n, 0, 32, st));
      GCHK(cub_tmp(ctx, tb));
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cubtmp.p, tb, hi_in, hi_out, P<int>(t.idx),
                                                P<int>(t.perm), n, 0, 32, st));
      k_gather_keys<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(t.perm), P<unsigned long long>(t.key),
                                                  P<unsigned long long>(t.skey));
      k_fix_runs<<<cdiv(n, 256), 256, 0, st>>>(n, 32, P<unsigned long long>(t.skey),
                                               P<int>(t.perm), tree_dinfo(ctx, false) + 1);
      HIPCHK(hipGetLastError());
    }
  else
    {
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, P<unsigned long long>(t.key),
                                                P<unsigned long long>(t.skey), P<int>(t.idx),
                                                P<int>(t.perm), n, 0, 63, st));
      GCHK(cub_tmp(ctx, tb));
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cubtmp.p, tb, P<unsigned long long>(t.key),
                                                P<unsigned long long>(t.skey), P<int>(t.idx),
                                                P<int>(t.perm), n, 0, 63, st));
    }
  return GHIP_OK;
}

// gas tree order out of the gravity tree's; leaves rank[] (gravity-tree index -> gas-tree index)
// in ctx->dtgt_b for the curve order below
static int gas_order_from_gravity_tree(ghip_ctx *ctx)
{
  TreeDev &g = ctx->gt, &t = ctx->st;
  int n = ctx->n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, t.skey, (size_t) ng * 8));
  GCHK(ghip_ensure(ctx, t.perm, (size_t) ng * 4));
  GCHK(ghip_ensure(ctx, ctx->dtgt_a, (size_t) n * 4 + 16));
  GCHK(ghip_ensure(ctx, ctx->dtgt_b, (size_t) n * 4 + 16));
  k_gas_flags<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<int>(g.perm), P<int>(ctx->dtgt_a));
  GCHK(exclusive_sum(ctx, P<int>(ctx->dtgt_a), P<int>(ctx->dtgt_b), n));
  k_gas_compact<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<int>(g.perm),
                                              P<unsigned long long>(g.skey), P<int>(ctx->dtgt_b),
                                              P<int>(t.perm), P<unsigned long long>(t.skey));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// common prefix levels, nodes per particle and their scan; {maxlevel, nnodes, longest run} -> hout
static int count_nodes(ghip_ctx *ctx, TreeDev &t, int n, bool gas, int *hout)
{
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, t.cpl, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.cnt, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.nb, (size_t) (n + 1) * 4));
  int *dinfo = tree_dinfo(ctx, gas);
  k_prefix_levels<<<cdiv(n, 1024), 1024, 0, st>>>(n, P<unsigned long long>(t.skey), P<int>(t.cpl),
                                                  dinfo);
  k_node_counts<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(t.cpl), P<int>(t.cnt));
  HIPCHK(hipGetLastError());
  GCHK(exclusive_sum(ctx, P<int>(t.cnt), P<int>(t.nb), n));
  k_tree_info<<<1, 1, 0, st>>>(n, P<int>(t.nb), P<int>(t.cnt), dinfo, hout);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// element list + moments.  Sorted particle data goes to ox,oy,oz,om,oa; iperm is filled too.
static int emit_tree(ghip_ctx *ctx, TreeDev &t, int n, const int *hinfo, const double *x,
                     const double *y, const double *z, const double *m, const double *aux,
                     double *ox, double *oy, double *oz, double *om, double *oa, bool grav,
                     hipEvent_t *fork_after_gather = nullptr)
{
  hipStream_t st = ctx->stream;
  t.n = n;
  t.nnodes = hinfo[1];
  t.nelem = n + t.nnodes;
  t.maxlevel = hinfo[0] < GHIP_BITS ? hinfo[0] : GHIP_BITS;
  GCHK(ghip_ensure(ctx, t.xm, (size_t) t.nelem * sizeof(double4)));
  GCHK(ghip_ensure(ctx, t.cl, (size_t) t.nelem * sizeof(double4)));
  GCHK(ghip_ensure(ctx, t.lk, (size_t) t.nelem * sizeof(int4)));
  GCHK(ghip_ensure(ctx, t.aux, (size_t) t.nelem * sizeof(double)));
  k_gather5<<<cdiv(n, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(n, P<int>(t.perm), x, y, z, m, aux, ox, oy, oz, om, oa,
                                          P<int>(t.iperm));
  if(fork_after_gather)
    HIPCHK(hipEventRecord(*fork_after_gather, st));
  k_emit_elements<<<cdiv(n, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
    n, t.nelem, P<unsigned long long>(t.skey), P<int>(t.cpl), P<int>(t.cnt), P<int>(t.nb), ox, oy,
    oz, om, oa, ctx->center[0], ctx->center[1], ctx->center[2], ctx->dlen, P<double4>(t.xm),
    P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux));
  HIPCHK(hipGetLastError());
  for(int L = t.maxlevel; L >= 0; L--)
    {
      if(grav)
        k_node_level<true, true><<<cdiv(t.nelem, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
          t.nelem, L, P<double4>(t.xm), P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux),
          ctx->adaptive_gravsoft);
      else
        k_node_level<false, true><<<cdiv(t.nelem, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
          t.nelem, L, P<double4>(t.xm), P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), false);
    }
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// Peano-Hilbert order of the tree-order particles: the order in which targets are bucketed.
// The order only has to make 64 consecutive targets compact, so the sort runs on the leading
// ceil(log8 n) + 3 digits of the key (at most 10: they fit a 32-bit sort).
static int curve_order(ghip_ctx *ctx, hipStream_t st)
{
  TreeDev &g = ctx->gt, &t = ctx->st;
  int n = ctx->n, ng = ctx->ngas;
  GCHK(ghip_ensure(ctx, g.phorder, (size_t) n * 4));
  if(ng > 0)
    GCHK(ghip_ensure(ctx, t.phorder, (size_t) ng * 4));
  if(getenv("GHIP_TARGET_ORDER") && !strcmp(getenv("GHIP_TARGET_ORDER"), "morton"))
    {
      k_iota_tree<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(g.phorder));
      if(ng > 0)
        k_iota_tree<<<cdiv(ng, 256), 256, 0, st>>>(ng, P<int>(t.phorder));
      HIPCHK(hipGetLastError());
      return GHIP_OK;
    }
  int levels = 3;
  for(long long m = n; m > 1; m >>= 3)
    levels++;
  if(levels > 10)
    levels = 10;
  double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);
  unsigned int *hi_in = P<unsigned int>(g.phkey), *hi_out = hi_in + n;
  k_peano_hi<<<cdiv(n, 256), 256, 0, st>>>(n, levels, P<double>(ctx->sx), P<double>(ctx->sy),
                                           P<double>(ctx->sz), ctx->corner[0], ctx->corner[1],
                                           ctx->corner[2], fac, hi_in, P<int>(g.idx));
  HIPCHK(hipGetLastError());
  // (a 32-bit hipcub sort with begin_bit != 0 mis-sorts on this ROCm: the digits sit at bit 0)
  size_t tb = 0;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, hi_in, hi_out, P<int>(g.idx),
                                            P<int>(g.phorder), n, 0, 3 * levels, st));
  GCHK(cub_tmp(ctx, tb));
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cubtmp.p, tb, hi_in, hi_out, P<int>(g.idx),
                                            P<int>(g.phorder), n, 0, 3 * levels, st));
  if(ng > 0)
    {
      // the gas targets in the same order: compaction of the list above (rank[] = dtgt_b)
      GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) n * 4));
      int *flag = P<int>(ctx->dflags), *pos = P<int>(ctx->dtgt_a);
      k_gas_flags_ph<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<int>(g.phorder), P<int>(g.perm), flag);
      GCHK(exclusive_sum(ctx, flag, pos, n, st));
      k_gas_compact_ph<<<cdiv(n, 256), 256, 0, st>>>(n, ng, P<int>(g.phorder), P<int>(g.perm),
                                                     P<int>(ctx->dtgt_b), pos, P<int>(t.phorder));
      HIPCHK(hipGetLastError());
    }
  return GHIP_OK;
}

// hsml != nullptr: ADAPTIVE_GRAVSOFT_FORGAS, a gas particle is softened with its Hsml
// (forcetree.c:705-716, 1851-1856, 2038-2058)
__global__ void k_soft_of_type(int n, const int *__restrict__ type, double s0, double s1, double s2,
                               double s3, double s4, double s5, const double *__restrict__ hsml,
                               double *__restrict__ out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int t = type[i];
  out[i] = (t == 0) ? (hsml ? hsml[i] : s0)
                    : (t == 1) ? s1 : (t == 2) ? s2 : (t == 3) ? s3 : (t == 4) ? s4 : s5;
}

static void tree_reset(TreeDev &t, int n)
{
  t.n = n;
  t.nnodes = 0;
  t.nelem = n;
  t.built = (n == 0);
}

int ghip_tree_build_impl(ghip_ctx *ctx)
{
  int n = ctx->n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double *y = x + n, *z = y + n;
  const double *m = P<double>(ctx->f[GHIP_F_MASS]);
  const double *h = P<double>(ctx->f[GHIP_F_HSML]);
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  tree_reset(ctx->gt, n);
  tree_reset(ctx->st, ng);
  ctx->gas_pending = false;
  ctx->lists_dirty = true;
  ctx->stats.tree_nodes = ctx->stats.gastree_nodes = 0;
  if(n == 0)
    {
      HIPCHK(hipEventRecord(ctx->ev[1], st));
      return GHIP_OK;
    }
  int *hinfo = reinterpret_cast<int *>(ctx->pinned);  // [0..2] gravity tree, [4..6] gas tree
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));

  // per-particle softening in host order (aux of the gravity tree's particle elements)
  GCHK(ghip_ensure(ctx, ctx->ssoft, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, ctx->soldacc, (size_t) n * 8));
  double *tmp_soft = P<double>(ctx->soldacc);  // scratch until the first gravity call
  k_soft_of_type<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(ctx->f[GHIP_F_TYPE]), ctx->soft[0],
                                               ctx->soft[1], ctx->soft[2], ctx->soft[3],
                                               ctx->soft[4], ctx->soft[5],
                                               ctx->adaptive_gravsoft ? P<double>(ctx->f[GHIP_F_HSML])
                                                                      : nullptr,
                                               tmp_soft);

  bool wide = getenv("GHIP_SORT64") && atoi(getenv("GHIP_SORT64")) == 1;
  for(;;)
    {
      HIPCHK(hipMemsetAsync(tree_dinfo(ctx, false), 0, 16, st));
      GCHK(sort_by_key(ctx, ctx->gt, n, x, y, z, wide));
      GCHK(count_nodes(ctx, ctx->gt, n, false, hinfo));
      if(ng > 0)
        {
          GCHK(gas_order_from_gravity_tree(ctx));
          GCHK(count_nodes(ctx, ctx->st, ng, true, hinfo + 4));
        }
      HIPCHK(hipStreamSynchronize(st));   // the one host round trip of the build
      if(wide || hinfo[2] == 0)
        break;
      wide = true;   // strongly clustered input: long runs of equal top key bits, sort full keys
    }

  GCHK(ghip_ensure(ctx, ctx->sx, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, ctx->sy, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, ctx->sz, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, ctx->gt.iperm, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) n * 5 * sizeof(double)));
  GCHK(emit_tree(ctx, ctx->gt, n, hinfo, x, y, z, m, tmp_soft, P<double>(ctx->sx),
                 P<double>(ctx->sy), P<double>(ctx->sz), P<double>(ctx->stage),
                 P<double>(ctx->ssoft), true, &ctx->evt[0]));
  ctx->gt.built = true;
  // The curve order of the targets needs the tree-order positions only (the gather at the head of
  // emit_tree): it runs on a second stream next to the element emission, the moment passes and the
  // walk records.  Both chains are strings of short launches, so side by side they take the longer
  // of the two instead of the sum.
  // (the main stream's remaining launches are enqueued first: the host needs longer to enqueue the
  // curve order's ~25 launches than the device needs to run the main chain)
  GCHK(ghip_build_segments(ctx, ctx->gt, true));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->evt[0], 0));
  GCHK(curve_order(ctx, ctx->stream2));
  HIPCHK(hipEventRecord(ctx->evt[1], ctx->stream2));
  HIPCHK(hipStreamWaitEvent(st, ctx->evt[1], 0));

  // gas tree over host indices [0, ngas): same cells, gas only; aux = Hsml.  Its order, node
  // counts and curve order are known; the rest of its build is deferred to the first call that
  // needs it (ghip_finish_gas_tree), which normally is ghip_density -- enqueued while a gravity
  // pair is in flight, so that this work runs underneath the walks.
  if(ng > 0)
    {
      for(int q = 0; q < 3; q++)
        ctx->gas_hinfo[q] = hinfo[4 + q];
      ctx->st.nnodes = hinfo[5];
      ctx->st.nelem = ng + hinfo[5];
      ctx->gas_pending = true;
    }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  ctx->stats.tree_nodes = ctx->gt.nnodes;
  ctx->stats.gastree_nodes = ctx->st.nnodes;
  return GHIP_OK;
}

int ghip_finish_gas_tree(ghip_ctx *ctx)
{
  if(!ctx->gas_pending)
    return GHIP_OK;
  ctx->gas_pending = false;
  const int n = ctx->n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double *y = x + n, *z = y + n;
  const double *m = P<double>(ctx->f[GHIP_F_MASS]);
  const double *h = P<double>(ctx->f[GHIP_F_HSML]);
  GCHK(ghip_ensure(ctx, ctx->st.iperm, (size_t) ng * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) ng * 5 * sizeof(double)));
  double *s = P<double>(ctx->stage);
  GCHK(emit_tree(ctx, ctx->st, ng, ctx->gas_hinfo, x, y, z, m, h, s, s + ng, s + 2 * (size_t) ng,
                 s + 3 * (size_t) ng, s + 4 * (size_t) ng, false));
  ctx->st.built = true;
  GCHK(ghip_sph_fill_nodes(ctx, false));
  GCHK(ghip_ensure(ctx, ctx->gp, (size_t) ng * 64));
  GCHK(ghip_ensure(ctx, ctx->gq, (size_t) ng * 64));
  const double *vp = P<double>(ctx->f[GHIP_F_VELPRED]);
  k_gather_gas<<<cdiv(ng, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
    ng, P<int>(ctx->st.perm), x, y, z, m, vp, vp + ng, vp + 2 * (size_t) ng, h,
    P<double>(ctx->f[GHIP_F_PRESSURE]), P<double>(ctx->f[GHIP_F_DENSITY]),
    P<double>(ctx->f[GHIP_F_DHSMLFAC]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    P<double>(ctx->f[GHIP_F_CURLVEL]), P<int>(ctx->f[GHIP_F_TIMEBIN]), P<double>(ctx->gp),
    P<double>(ctx->gq));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// refresh Extnodes[].hmax of the gas tree from the current smoothing lengths in gp[].h
// (force_update_hmax, forcetree.c:1661-1786; recomputed exactly instead of only raised)
__global__ void k_aux_from_gp(int nelem, const int4 *__restrict__ lk,
                              const double *__restrict__ gp, double *__restrict__ aux)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y >= 0)
    aux[e] = gp[(size_t) 8 * me.y + 7];
}

int ghip_gastree_refresh_hmax(ghip_ctx *ctx)
{
  TreeDev &t = ctx->st;
  if(t.n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  HIPCHK(hipEventRecord(ctx->ev[8], st));
  k_aux_from_gp<<<cdiv(t.nelem, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(t.nelem, P<int4>(t.lk), P<double>(ctx->gp),
                                                   P<double>(t.aux));
  for(int L = t.maxlevel; L >= 0; L--)
    k_node_level<false, false><<<cdiv(t.nelem, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
      t.nelem, L, P<double4>(t.xm), P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), false);
  HIPCHK(hipGetLastError());
  GCHK(ghip_sph_fill_nodes(ctx, true));
  HIPCHK(hipEventRecord(ctx->ev[9], st));
  return GHIP_OK;
}

// target lists: tree-order positions of the active particles, ascending (so that 64
// consecutive targets are spatial neighbours), sliced for this rank's shard
static int make_list(ghip_ctx *ctx, TreeDev &t, int host_limit, DevBuf &list, int *count)
{
  hipStream_t st = ctx->stream;
  int n = t.n;
  *count = 0;
  if(n == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, list, (size_t) n * 4));
  if(ctx->nactive < 0)
    {
      HIPCHK(hipMemcpyAsync(list.p, t.phorder.p, (size_t) n * 4, hipMemcpyDeviceToDevice, st));
      *count = n;
      return GHIP_OK;
    }
  if(ctx->nactive == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, ctx->dtgt_a, (size_t) n * 4 + 16));
  HIPCHK(hipMemsetAsync(ctx->dflags.p, 0, (size_t) n * 4, st));
  k_mark_active<<<cdiv(ctx->nactive, 256), 256, 0, st>>>(ctx->nactive, P<int>(ctx->act_host_idx),
                                                          P<int>(t.iperm), host_limit,
                                                          P<int>(ctx->dflags));
  // flags in Peano-Hilbert order, then an order-preserving selection of the PH-ordered indices
  k_gather_i32<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(t.phorder), P<int>(ctx->dflags),
                                             P<int>(ctx->dtgt_a));
  HIPCHK(hipGetLastError());
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  int *dnum = reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 32);
  size_t tb = 0;
  HIPCHK(hipcub::DeviceSelect::Flagged(nullptr, tb, P<int>(t.phorder), P<int>(ctx->dtgt_a),
                                       P<int>(list), dnum, n, st));
  GCHK(cub_tmp(ctx, tb));
  HIPCHK(hipcub::DeviceSelect::Flagged(ctx->cubtmp.p, tb, P<int>(t.phorder),
                                       P<int>(ctx->dtgt_a), P<int>(list), dnum, n, st));
  HIPCHK(hipMemcpyAsync(count, dnum, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return GHIP_OK;
}

struct ShardLoT
{
  int lo[GHIP_MAXRANKS + 1];
};

// curve order -> rank-major order: position j of rank r's slice holds bucket (j/64)*N + r
__global__ void k_shard_permute(int nt, int nranks, ShardLoT L, const int *__restrict__ src,
                                int *__restrict__ dst)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(j >= nt)
    return;
  int r = 0;
  while(r + 1 < nranks && j >= L.lo[r + 1])
    r++;
  int a = j - L.lo[r];
  int gb = (a >> 6) * nranks + r;
  dst[j] = src[gb * 64 + (a & 63)];
}

static int permute_for_shards(ghip_ctx *ctx, DevBuf &list, int nt)
{
  if(ctx->shard_n <= 1 || nt == 0)
    return GHIP_OK;
  ShardLoT L;
  for(int r = 0; r <= ctx->shard_n; r++)
    {
      int l = nt, c, p2;
      if(r < ctx->shard_n)
        ghip_shard_range(nt, ctx->shard_n, r, &l, &c, &p2);
      L.lo[r] = l;
    }
  GCHK(ghip_ensure(ctx, ctx->dtgt_b, (size_t) nt * 4 + 16));
  HIPCHK(hipMemcpyAsync(ctx->dtgt_b.p, list.p, (size_t) nt * 4, hipMemcpyDeviceToDevice,
                        ctx->stream));
  k_shard_permute<<<cdiv(nt, 256), 256, 0, ctx->stream>>>(nt, ctx->shard_n, L,
                                                          P<int>(ctx->dtgt_b), P<int>(list));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

int ghip_build_target_lists(ghip_ctx *ctx)
{
  if(!ctx->lists_dirty)
    return GHIP_OK;
  if(ctx->nactive >= 0)
    GCHK(ghip_finish_gas_tree(ctx));   // an active subset is marked through st.iperm
  GCHK(make_list(ctx, ctx->gt, ctx->n, ctx->tg_grav, &ctx->nt_grav));
  GCHK(make_list(ctx, ctx->st, ctx->ngas, ctx->tg_gas, &ctx->nt_gas));
  GCHK(permute_for_shards(ctx, ctx->tg_grav, ctx->nt_grav));
  GCHK(permute_for_shards(ctx, ctx->tg_gas, ctx->nt_gas));
  ctx->lists_dirty = false;
  return GHIP_OK;
}
