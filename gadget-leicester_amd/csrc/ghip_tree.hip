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
#include "ghip_keys.h"

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
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
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
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
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

// slvl (multi-GPU only, else nullptr): level of source i when it is an imported pruned node
// ("pseudo-leaf", see k_emit_elements), 0 for a particle
__global__ void k_prefix_levels(int n, const unsigned long long *__restrict__ skey,
                                const int *__restrict__ slvl, int *__restrict__ cpl,
                                int *__restrict__ maxlevel)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = -1;
  if(i < n)
    {
      c = (i + 1 < n) ? d_common_levels(skey[i], skey[i + 1]) : -1;
      cpl[i] = c;
      if(slvl && slvl[i] - 1 > c)
        c = slvl[i] - 1;   // its chain of single-child ancestors reaches down to level - 1
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

// k_prefix_levels and k_node_counts in one launch: the shared-digit counts with both neighbours come
// straight from the keys
__global__ void k_prefix_counts(int n, const unsigned long long *__restrict__ skey,
                                const int *__restrict__ slvl, int *__restrict__ cpl,
                                int *__restrict__ cnt, int *__restrict__ maxlevel,
                                int *__restrict__ errword)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int c = -1;
  if(i < n)
    {
      const unsigned long long k = skey[i];
      const int craw = (i + 1 < n) ? d_common_levels(k, skey[i + 1]) : -1;
      const int cprev = (i > 0) ? d_common_levels(skey[i - 1], k) : -1;
      cpl[i] = craw;
      int ceff = craw;
      if(slvl && slvl[i] > 0)
        {
          const int L = slvl[i];   // (see k_node_counts)
          if(craw >= L || cprev >= L)
            *(volatile int *) errword = 3;
          if(L - 1 > ceff)
            ceff = L - 1;
        }
      const int d = ceff - cprev;
      cnt[i] = d > 0 ? d : 0;
      c = ceff;
    }
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

__global__ void k_node_counts(int n, const int *__restrict__ cpl, const int *__restrict__ slvl,
                              int *__restrict__ cnt, int *__restrict__ errword)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int cprev = (i > 0) ? cpl[i - 1] : -1;
  int ceff = cpl[i];
  if(slvl && slvl[i] > 0)
    {
      // a pruned node of level L stands for >= 2 particles: the cells of levels .. L-1 that hold
      // nothing else exist too (single-child chains, like the insertion tree's).  No other source
      // may lie inside its cell.
      const int L = slvl[i];
      if(cpl[i] >= L || cprev >= L)
        *(volatile int *) errword = 3;
      if(L - 1 > ceff)
        ceff = L - 1;
    }
  int d = ceff - cprev;
  cnt[i] = d > 0 ? d : 0;
}

// one thread per particle: emits the nodes that START at this particle (levels cprev+1..c_i, in
// increasing depth = pre-order) followed by the particle itself.
__global__ void k_emit_elements(int n, const TreeSizes *__restrict__ ts,
                                const unsigned long long *__restrict__ skey,
                                const int *__restrict__ cpl, const int *__restrict__ cnt,
                                const int *__restrict__ nb, const double *__restrict__ px,
                                const double *__restrict__ py, const double *__restrict__ pz,
                                const double *__restrict__ pm, const double *__restrict__ paux,
                                double ccx, double ccy, double ccz, double dlen,
                                double4 *__restrict__ xm, double4 *__restrict__ cl,
                                int4 *__restrict__ lk, double *__restrict__ aux,
                                const int *__restrict__ slvl)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nelem = ts->nelem;   // 0: the counting stage found the tree unusable (TreeSizes.bad)
  if(i >= n || nelem == 0)
    return;
  int cprev = (i > 0) ? cpl[i - 1] : -1;
  int ci = cpl[i];
  const int plevel = slvl ? slvl[i] : 0;   // > 0: imported pruned node (multi-GPU)
  if(plevel > 0 && plevel - 1 > ci)
    ci = plevel - 1;
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
  if(plevel > 0)
    {
      // A node of another shard's tree that no target of this shard opens, imported as a leaf:
      // its moments (centre of mass, mass, softening word) come with it, its geometry is the
      // level-`plevel` cell of its key.  lk.y = -(level+1) - 32 keeps the moment passes off it
      // (they address nodes by -(level+1)); every other consumer sees a node (lk.y < 0) whose
      // skip link is e+1.
      double len = dlen, cx = ccx, cy = ccy, cz = ccz;
      for(int l = 1; l <= plevel; l++)
        {
          int digit = (int) ((ki >> (63 - 3 * l)) & 7);
          double lenhalf = 0.25 * len;
          cx = (digit & 1) ? cx + lenhalf : cx - lenhalf;
          cy = (digit & 2) ? cy + lenhalf : cy - lenhalf;
          cz = (digit & 4) ? cz + lenhalf : cz - lenhalf;
          len = 0.5 * len;
        }
      lk[pe] = make_int4(pe + 1, -(plevel + 1) - 32, i, 1);
      xm[pe] = make_double4(x, y, z, pm[i]);
      cl[pe] = make_double4(cx, cy, cz, len);
      aux[pe] = paux[i];
      return;
    }
  lk[pe] = make_int4(pe + 1, i, i, 1);
  xm[pe] = make_double4(x, y, z, pm[i]);
  cl[pe] = make_double4(x, y, z, 0.0);
  aux[pe] = paux[i];
}

// multipole pass (force_update_node_recursive, forcetree.c:468-872): the children of node e are the
// elements reached from e+1 by following skip links until the node's own skip, summed in that
// (octant) order.
// GRAV: aux = largest ForceSoftening below, negated when the node mixes softenings
//       (BITFLAG_MAX_SOFTENING_TYPE / BITFLAG_MIXED_SOFTENINGS_IN_NODE, forcetree.c:612-700)
// GAS : aux = hmax (Extnodes[].hmax, forcetree.c:593, 676)
template <bool GRAV, bool MOMENTS>
__device__ __forceinline__ void d_node_moments(int e, const int4 me, double4 *xm, const double4 *cl,
                                               const int4 *lk, double *aux, bool adaptive)
{
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

// The bottom-up pass in TWO launches (the level-by-level form took one launch per tree level: ~10
// dependent launches per tree and pass, each reading the links of every element).  The element list
// is in pre-order, so a node's subtree is the contiguous range [e, skip).
//   k_moments_local: a workgroup owns a chunk of MOM_CH consecutive elements.  Every node whose
//     subtree ends inside the chunk -- all but a few thousand -- has all its descendants in the chunk:
//     the workgroup computes them level by level, deepest first, with a barrier between levels.  The
//     nodes that reach beyond their chunk (the ancestors of the chunk boundaries) are appended to
//     per-level lists.
//   k_moments_upper: ONE workgroup takes those lists level by level, deepest first.
// Children are summed in list (octant) order as ever, so the numbers are those of the level passes.
#define MOM_CH 2048
#define MOM_LEVELS (GHIP_BITS + 1)

// BLOCK: 256 threads, or one wavefront for a pass that runs underneath a gravity pair (ghip_wg)
template <bool GRAV, bool MOMENTS, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_moments_local(const TreeSizes *__restrict__ ts, double4 *xm, const double4 *cl, const int4 *lk,
                double *aux, bool adaptive, int *__restrict__ upper, int ucap, int *__restrict__ ucount)
{
  const int nelem = ts->nelem;
  const int c0 = blockIdx.x * MOM_CH;
  if(c0 >= nelem)
    return;
  const int c1 = (c0 + MOM_CH < nelem) ? c0 + MOM_CH : nelem;
  __shared__ unsigned int levels;           // levels at which this chunk owns nodes
  __shared__ signed char lev[MOM_CH];       // level of the node this chunk owns at each slot, -1: none
  if(threadIdx.x == 0)
    levels = 0;
  __syncthreads();
  for(int q = threadIdx.x; q < MOM_CH; q += BLOCK)
    {
      const int e = c0 + q;
      int mine = -1;
      if(e < c1)
        {
          const int4 k = lk[e];
          if(k.y < 0 && k.y >= -MOM_LEVELS)   // a node of level -k.y - 1 (not a particle, not a pruned leaf)
            {
              const int L = -k.y - 1;
              if(k.x > c1)
                {
                  const int slot = atomicAdd(&ucount[L], 1);
                  if(slot < ucap)
                    upper[L * ucap + slot] = e;
                }
              else
                {
                  mine = L;
                  atomicOr(&levels, 1u << L);
                }
            }
        }
      lev[q] = (signed char) mine;
    }
  __syncthreads();
  unsigned int todo = levels;
  while(todo)
    {
      const int L = 31 - __clz(todo);   // deepest level first
      todo &= ~(1u << L);
      for(int q = threadIdx.x; q < MOM_CH; q += BLOCK)
        if(lev[q] == L)
          d_node_moments<GRAV, MOMENTS>(c0 + q, lk[c0 + q], xm, cl, lk, aux, adaptive);
      __threadfence_block();
      __syncthreads();
    }
}

template <bool GRAV, bool MOMENTS, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_moments_upper(const TreeSizes *__restrict__ ts, double4 *xm, const double4 *cl, const int4 *lk,
                double *aux, bool adaptive, const int *__restrict__ upper, int ucap, int *ucount)
{
  const bool usable = ts->nelem > 0;
  for(int L = GHIP_BITS; L >= 0; L--)
    {
      const int nu = ucount[L];   // (the same for every thread: the branch is uniform)
      if(nu == 0 || !usable)
        continue;
      for(int k = threadIdx.x; k < nu && k < ucap; k += BLOCK)
        {
          const int e = upper[L * ucap + k];
          d_node_moments<GRAV, MOMENTS>(e, lk[e], xm, cl, lk, aux, adaptive);
        }
      __threadfence_block();
      __syncthreads();
    }
  __syncthreads();
  if(threadIdx.x < MOM_LEVELS)   // the lists are left empty for the next pass
    ucount[threadIdx.x] = 0;
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

// the same when sources beyond `limit` (imported elements of a multi-GPU tree) have no entry in src
__global__ void k_gather_f64_lim(int n, const int *__restrict__ perm, const double *__restrict__ src,
                                 int limit, double *__restrict__ dst)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    {
      int i = perm[s];
      dst[s] = i < limit ? src[i] : 0.0;
    }
}

int ghip_gather_f64_lim(ghip_ctx *ctx, int n, const int *perm, const double *src, int limit,
                        double *dst)
{
  if(n <= 0)
    return GHIP_OK;
  k_gather_f64_lim<<<cdiv(n, 256), 256, 0, ctx->stream>>>(n, perm, src, limit, dst);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
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

// Radix sort of (32-bit key, index) pairs on bits [0, end_bit).  Below 2^20 keys the library picks a
// block sort followed by ~19 merge passes (~125 us at 5e5 keys, all dependent ~5 us launches).  Its
// one-sweep radix sort (GHIP_SORT_ONESWEEP=1) needs fewer launches but was measured slower here:
// a histogram pass, its scan, and per 8-bit digit two clears + one 25 us pass = 157 us.
// tmp: scratch of the stream the sort runs on (ctx->cubtmp: main stream, ctx->cubtmp2: stream2)
static int sort_pairs_u32(ghip_ctx *ctx, const unsigned int *kin, unsigned int *kout, const int *vin,
                          int *vout, int n, int end_bit, hipStream_t st, DevBuf &tmp)
{
  static int onesweep = -1;
  if(onesweep < 0)
    onesweep = (getenv("GHIP_SORT_ONESWEEP") && atoi(getenv("GHIP_SORT_ONESWEEP")) == 1) ? 1 : 0;
  size_t tb = 0;
  if(onesweep && n > 4096)
    {
      using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::default_config, 0>;
      HIPCHK(rocprim::radix_sort_pairs<cfg>(nullptr, tb, kin, kout, vin, vout, (size_t) n, 0u,
                                            (unsigned int) end_bit, st));
      GCHK(ghip_ensure(ctx, tmp, tb + 256));
      HIPCHK(rocprim::radix_sort_pairs<cfg>(tmp.p, tb, kin, kout, vin, vout, (size_t) n, 0u,
                                            (unsigned int) end_bit, st));
      return GHIP_OK;
    }
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, n, 0, end_bit, st));
  GCHK(ghip_ensure(ctx, tmp, tb + 256));
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, kin, kout, vin, vout, n, 0, end_bit, st));
  return GHIP_OK;
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

// the sizes of the tree, to the device and to the pinned host mirror (TreeSizes): dinfo[0] = deepest
// level, dinfo[1] = longest key run the 32-bit sort left unsorted
__global__ void k_tree_info(int n, const int *__restrict__ nb, const int *__restrict__ cnt,
                            int *__restrict__ dinfo, int cap_nodes, int wide, int gen,
                            TreeSizes *__restrict__ dsz, TreeSizes *__restrict__ hsz)
{
  TreeSizes s;
  s.nnodes = nb[n - 1] + cnt[n - 1];
  s.maxlevel = dinfo[0];
  s.longrun = dinfo[1];
  s.bad = (s.nnodes > cap_nodes) ? 1 : ((!wide && s.longrun != 0) ? 2 : 0);
  s.nelem = s.bad ? 0 : n + s.nnodes;
  s.n = n;
  s.gen = gen;
  s.pad = 0;
  *dsz = s;
  *hsz = s;
  dinfo[0] = dinfo[1] = 0;   // (the consumer leaves the words zeroed for the next build: no memset in front of it)
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

// Exclusive scan of ints out of one-wavefront workgroups, for scans that run underneath a gravity
// pair: the library's single-pass scan uses 256-thread workgroups, which wait behind the walks for
// four free wavefront slots on one CU (see ghip_wg).  Three launches: per-block sums, their scan by
// one wavefront, the blocks' local scans.
#define WSCAN_ITEMS 16
#define WSCAN_BLOCK (64 * WSCAN_ITEMS)
__global__ void __launch_bounds__(64) k_wscan_sums(int n, const int *__restrict__ in, int *__restrict__ bsum)
{
  const int base = blockIdx.x * WSCAN_BLOCK + threadIdx.x * WSCAN_ITEMS;
  int c = 0;
  for(int j = 0; j < WSCAN_ITEMS; j++)
    c += (base + j < n) ? in[base + j] : 0;
  for(int o = 32; o > 0; o >>= 1)
    c += __shfl_down(c, o, 64);
  if(threadIdx.x == 0)
    bsum[blockIdx.x] = c;
}

__global__ void __launch_bounds__(64) k_wscan_blocks(int nblk, int *__restrict__ bsum)
{
  int run = 0;
  for(int b0 = 0; b0 < nblk; b0 += 64)
    {
      const int b = b0 + threadIdx.x;
      const int c = b < nblk ? bsum[b] : 0;
      int incl = c;
      for(int o = 1; o < 64; o <<= 1)
        {
          const int v = __shfl_up(incl, o, 64);
          if((int) threadIdx.x >= o)
            incl += v;
        }
      if(b < nblk)
        bsum[b] = run + incl - c;
      run += __shfl(incl, 63, 64);
    }
}

__global__ void __launch_bounds__(64) k_wscan_apply(int n, const int *__restrict__ in,
                                                    const int *__restrict__ boff, int *__restrict__ out)
{
  const int base = blockIdx.x * WSCAN_BLOCK + threadIdx.x * WSCAN_ITEMS;
  int v[WSCAN_ITEMS], c = 0;
  for(int j = 0; j < WSCAN_ITEMS; j++)
    {
      v[j] = (base + j < n) ? in[base + j] : 0;
      c += v[j];
    }
  int incl = c;
  for(int o = 1; o < 64; o <<= 1)
    {
      const int u = __shfl_up(incl, o, 64);
      if((int) threadIdx.x >= o)
        incl += u;
    }
  int run = boff[blockIdx.x] + incl - c;
  for(int j = 0; j < WSCAN_ITEMS; j++)
    {
      if(base + j < n)
        out[base + j] = run;
      run += v[j];
    }
}

static int exclusive_sum(ghip_ctx *ctx, const int *in, int *out, int n, hipStream_t on = nullptr)
{
  hipStream_t st = on ? on : ctx->stream;
  if(ctx->grav_pending && st == ctx->stream)
    {
      const int nblk = cdiv(n, WSCAN_BLOCK);
      GCHK(ghip_ensure(ctx, ctx->cubtmp, (size_t) (nblk + 2) * 4));
      int *bsum = P<int>(ctx->cubtmp);
      k_wscan_sums<<<nblk, 64, 0, st>>>(n, in, bsum);
      k_wscan_blocks<<<1, 64, 0, st>>>(nblk, bsum);
      k_wscan_apply<<<nblk, 64, 0, st>>>(n, in, bsum, out);
      HIPCHK(hipGetLastError());
      return GHIP_OK;
    }
  DevBuf &tmp = (st == ctx->stream) ? ctx->cubtmp : ctx->cubtmp2;   // scratch of the stream it runs on
  size_t tb = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, n, st));
  GCHK(ghip_ensure(ctx, tmp, tb + 256));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, in, out, n, st));
  return GHIP_OK;
}

static int *tree_dinfo(ghip_ctx *ctx, bool gas)
{
  return reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 48) + (gas ? 2 : 0);
}

// Morton keys of the gravity tree's particles, sorted -> t.skey, t.perm
// the keys imported elements carry replace the ones computed from their coordinates (a pruned
// node's key is its cell's prefix; its centre of mass may round onto the cell's face)
__global__ void k_override_keys(int first, int n, const unsigned long long *__restrict__ given,
                                unsigned long long *__restrict__ key)
{
  int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    key[i] = given[i];
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

// per-particle softening, Morton key, index and the key's top 32 bits in one launch (the head of a
// single-GPU gravity-tree build: k_soft_of_type + k_morton_from_pos + k_key_hi)
__global__ void k_keys_fused(int n, const int *__restrict__ type, double s0, double s1, double s2,
                             double s3, double s4, double s5, const double *__restrict__ hsml,
                             const double *__restrict__ x, const double *__restrict__ y,
                             const double *__restrict__ z, double cx, double cy, double cz, double fac,
                             double *__restrict__ soft, unsigned long long *__restrict__ key,
                             int *__restrict__ idx, unsigned int *__restrict__ hi)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const int t = type[i];
  soft[i] = (t == 0) ? (hsml ? hsml[i] : s0) : (t == 1) ? s1 : (t == 2) ? s2 : (t == 3) ? s3 : (t == 4) ? s4 : s5;
  const int ix = (int) ((x[i] - cx) * fac);   // forcetree.c:181-185
  const int iy = (int) ((y[i] - cy) * fac);
  const int iz = (int) ((z[i] - cz) * fac);
  const unsigned long long k = d_morton21(ix, iy, iz);
  key[i] = k;
  idx[i] = i;
  hi[i] = (unsigned int) (k >> 31);
}

// type != nullptr (single-GPU gravity tree, 32-bit sort): the softenings (-> soft_out) are formed by the
// same launch as the keys
static int sort_by_key(ghip_ctx *ctx, TreeDev &t, int n, const double *x, const double *y,
                       const double *z, bool wide, int nlocal = -1,
                       const unsigned long long *given_keys = nullptr, bool gas = false,
                       const int *type = nullptr, double *soft_out = nullptr)
{
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, t.key, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, t.skey, (size_t) n * 8));
  GCHK(ghip_ensure(ctx, t.idx, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.perm, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.phkey, (size_t) n * 8));
  double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);  // DomainFac, domain.c:2012
  const bool fused = type && soft_out && !wide && !given_keys;
  if(type && soft_out && !fused)
    k_soft_of_type<<<cdiv(n, 256), 256, 0, st>>>(n, type, ctx->soft[0], ctx->soft[1], ctx->soft[2],
                                                 ctx->soft[3], ctx->soft[4], ctx->soft[5],
                                                 ctx->adaptive_gravsoft ? P<double>(ctx->f[GHIP_F_HSML]) : nullptr,
                                                 soft_out);
  if(fused)
    k_keys_fused<<<cdiv(n, 256), 256, 0, st>>>(
      n, type, ctx->soft[0], ctx->soft[1], ctx->soft[2], ctx->soft[3], ctx->soft[4], ctx->soft[5],
      ctx->adaptive_gravsoft ? P<double>(ctx->f[GHIP_F_HSML]) : nullptr, x, y, z, ctx->corner[0],
      ctx->corner[1], ctx->corner[2], fac, soft_out, P<unsigned long long>(t.key), P<int>(t.idx),
      P<unsigned int>(t.phkey));
  else
    k_morton_from_pos<<<cdiv(n, 256), 256, 0, st>>>(n, x, y, z, ctx->corner[0], ctx->corner[1],
                                                    ctx->corner[2], fac,
                                                    P<unsigned long long>(t.key), P<int>(t.idx));
  if(given_keys && nlocal >= 0 && nlocal < n)
    k_override_keys<<<cdiv(n - nlocal, 256), 256, 0, st>>>(nlocal, n, given_keys,
                                                           P<unsigned long long>(t.key));
  HIPCHK(hipGetLastError());
  size_t tb = 0;
  if(!wide)
    {
      // top 32 key bits first, then the (rare) runs of equal top bits by their full keys
      unsigned int *hi_in = P<unsigned int>(t.phkey), *hi_out = hi_in + n;
      if(!fused)
        k_key_hi<<<cdiv(n, 256), 256, 0, st>>>(n, 31, P<unsigned long long>(t.key), hi_in);
      GCHK(sort_pairs_u32(ctx, hi_in, hi_out, P<int>(t.idx), P<int>(t.perm), n, 32, st, ctx->cubtmp));
      k_gather_keys<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(t.perm), P<unsigned long long>(t.key),
                                                  P<unsigned long long>(t.skey));
      k_fix_runs<<<cdiv(n, 256), 256, 0, st>>>(n, 32, P<unsigned long long>(t.skey),
                                               P<int>(t.perm), tree_dinfo(ctx, gas) + 1);
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
  const int wg = ghip_wg(ctx);
  k_gas_flags<<<cdiv(n, wg), wg, 0, st>>>(n, ng, P<int>(g.perm), P<int>(ctx->dtgt_a));
  GCHK(exclusive_sum(ctx, P<int>(ctx->dtgt_a), P<int>(ctx->dtgt_b), n));
  k_gas_compact<<<cdiv(n, wg), wg, 0, st>>>(n, ng, P<int>(g.perm),
                                              P<unsigned long long>(g.skey), P<int>(ctx->dtgt_b),
                                              P<int>(t.perm), P<unsigned long long>(t.skey));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// common prefix levels, nodes per particle and their scan; the tree's sizes -> t.dsz (device) and
// t.hsz (pinned host mirror).  cap_nodes: nodes the element buffers hold (more: TreeSizes.bad = 1).
// src_lvl (multi-GPU): levels of the imported pruned nodes in source order, gathered to sorted order
static int count_nodes(ghip_ctx *ctx, TreeDev &t, int n, bool gas, int cap_nodes, bool wide,
                       const int *src_lvl = nullptr)
{
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, t.cpl, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.cnt, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, t.nb, (size_t) (n + 1) * 4));
  int *dinfo = tree_dinfo(ctx, gas);
  const int *slvl = nullptr;
  if(src_lvl)
    {
      GCHK(ghip_ensure(ctx, t.slvl, (size_t) n * 4));
      k_gather_i32<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(t.perm), src_lvl, P<int>(t.slvl));
      slvl = P<int>(t.slvl);
    }
  const int wgp = ctx->grav_pending ? 64 : 1024, wg = ghip_wg(ctx);
  k_prefix_counts<<<cdiv(n, wgp), wgp, 0, st>>>(n, P<unsigned long long>(t.skey), slvl, P<int>(t.cpl),
                                                P<int>(t.cnt), dinfo, ghip_errword(ctx, GHIP_ERRW_TREE));
  HIPCHK(hipGetLastError());
  (void) wg;
  GCHK(exclusive_sum(ctx, P<int>(t.cnt), P<int>(t.nb), n));
  k_tree_info<<<1, 1, 0, st>>>(n, P<int>(t.nb), P<int>(t.cnt), dinfo, cap_nodes, wide ? 1 : 0,
                               ctx->build_gen, P<TreeSizes>(t.dsz), t.hsz);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// element buffers of a tree of n sources for up to cap_nodes nodes (one padding record: the walks
// touch e+1 and the skip link of every element they load)
static int tree_ensure_elements(ghip_ctx *ctx, TreeDev &t, int n, int cap_nodes)
{
  const size_t ce = (size_t) n + (size_t) cap_nodes + 1;
  GCHK(ghip_ensure(ctx, t.xm, ce * sizeof(double4)));
  GCHK(ghip_ensure(ctx, t.cl, ce * sizeof(double4)));
  GCHK(ghip_ensure(ctx, t.lk, ce * sizeof(int4)));
  GCHK(ghip_ensure(ctx, t.aux, ce * sizeof(double)));
  // per-level lists of the nodes that reach beyond their chunk (k_moments_local), and their counts
  const size_t ucap = ce / MOM_CH + 2;
  GCHK(ghip_ensure(ctx, t.father, ucap * MOM_LEVELS * 4));
  const size_t before = t.arrived.cap;
  GCHK(ghip_ensure(ctx, t.arrived, MOM_LEVELS * 4));
  if(t.arrived.cap != before)   // the counts start at zero and every pass leaves them at zero
    HIPCHK(hipMemsetAsync(t.arrived.p, 0, t.arrived.cap, ctx->stream));
  t.cap_nodes = cap_nodes;
  return GHIP_OK;
}

// what the counting stage found, taken over by the host (after ev_sizes): exact node count, deepest
// level; the buffers grow when the count demands it (with slack: the next builds of the same particle
// set then run without the host waiting for their counts)
static int tree_adopt_sizes(ghip_ctx *ctx, TreeDev &t, bool grow)
{
  TreeSizes z;   // (the mirror is written by the device: read it field by field, not cached)
  {
    const volatile int *v = reinterpret_cast<const volatile int *>(t.hsz);
    int *w = reinterpret_cast<int *>(&z);
    for(size_t q = 0; q < sizeof(TreeSizes) / sizeof(int); q++)
      w[q] = v[q];
  }
  if(z.gen != ctx->build_gen || z.n != t.n)
    return ghip_fail(ctx, GHIP_EDEVICE, "tree sizes of build %d expected, the device reported build %d "
                     "(%d sources for %d)", ctx->build_gen, z.gen, z.n, t.n);
  t.nnodes = z.nnodes;
  t.nelem = t.n + z.nnodes;
  t.maxlevel = z.maxlevel < GHIP_BITS ? z.maxlevel : GHIP_BITS;
  if(z.bad == 0)
    {
      t.last_n = t.n;
      t.last_nnodes = z.nnodes;
    }
  if(grow)
    {
      // (always: the buffers hold n + cap_nodes elements, and the number of sources changes too --
      // a shard's merged tree has more of them than its own tree)
      int cap = t.cap_nodes;
      if(cap < z.nnodes + z.nnodes / 16 + 256)
        cap = z.nnodes + z.nnodes / 4 + 4096;
      GCHK(tree_ensure_elements(ctx, t, t.n, cap));
    }
  return GHIP_OK;
}

template <bool GRAV, bool MOMENTS>
static int moment_pass(ghip_ctx *ctx, TreeDev &t, bool adaptive)
{
  hipStream_t st = ctx->stream;
  const size_t ce = (size_t) t.n + (size_t) t.cap_nodes + 1;
  const int ucap = (int) (ce / MOM_CH + 2);
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  const int nblk = cdiv((long long) ce, MOM_CH);
#define MOM_ARGS ts, P<double4>(t.xm), P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), adaptive, \
                 P<int>(t.father), ucap, P<int>(t.arrived)
  if(ctx->grav_pending)
    {
      // underneath a gravity pair: one-wavefront workgroups (see ghip_wg)
      k_moments_local<GRAV, MOMENTS, 64><<<nblk, 64, 0, st>>>(MOM_ARGS);
      k_moments_upper<GRAV, MOMENTS, 64><<<1, 64, 0, st>>>(MOM_ARGS);
    }
  else
    {
      k_moments_local<GRAV, MOMENTS, 256><<<nblk, 256, 0, st>>>(MOM_ARGS);
      k_moments_upper<GRAV, MOMENTS, 1024><<<1, 1024, 0, st>>>(MOM_ARGS);
    }
#undef MOM_ARGS
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// element list + moments.  Sorted particle data goes to ox,oy,oz,om,oa; iperm is filled too.  Grids
// are sized by the buffers' capacity, the kernels read the element count on the device.
static int emit_tree(ghip_ctx *ctx, TreeDev &t, int n, const double *x, const double *y,
                     const double *z, const double *m, const double *aux, double *ox, double *oy,
                     double *oz, double *om, double *oa, bool grav,
                     hipEvent_t *fork_after_gather = nullptr, const int *slvl = nullptr)
{
  hipStream_t st = ctx->stream;
  const int wg = ghip_wg(ctx);
  const int ce = n + t.cap_nodes;
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  k_gather5<<<cdiv(n, wg), wg, 0, st>>>(n, P<int>(t.perm), x, y, z, m, aux, ox, oy, oz, om, oa,
                                        P<int>(t.iperm));
  if(fork_after_gather)
    HIPCHK(hipEventRecord(*fork_after_gather, st));
  k_emit_elements<<<cdiv(n, wg), wg, 0, st>>>(
    n, ts, P<unsigned long long>(t.skey), P<int>(t.cpl), P<int>(t.cnt), P<int>(t.nb), ox, oy,
    oz, om, oa, ctx->center[0], ctx->center[1], ctx->center[2], ctx->dlen, P<double4>(t.xm),
    P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), slvl);
  HIPCHK(hipGetLastError());
  if(grav)
    return moment_pass<true, true>(ctx, t, ctx->adaptive_gravsoft);
  return moment_pass<false, true>(ctx, t, false);
}

// Peano-Hilbert order of the tree-order particles: the order in which targets are bucketed.
// The order only has to make 64 consecutive targets compact, so the sort runs on the leading
// ceil(log8 n) + 3 digits of the key (at most 10: they fit a 32-bit sort).
static int curve_order(ghip_ctx *ctx, hipStream_t st)
{
  TreeDev &g = ctx->gt, &t = ctx->st;
  // (multi-GPU: the gravity tree holds imported elements too, and the shard's gas tree -- local gas
  // plus ghosts -- gets its own order from ghip_dd.hip)
  int n = g.n;
  GCHK(ghip_ensure(ctx, g.phorder, (size_t) n * 4));
  if(getenv("GHIP_TARGET_ORDER") && !strcmp(getenv("GHIP_TARGET_ORDER"), "morton"))
    {
      k_iota_tree<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(g.phorder));
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
  DevBuf &tmp = (st == ctx->stream) ? ctx->cubtmp : ctx->cubtmp2;
  GCHK(sort_pairs_u32(ctx, hi_in, hi_out, P<int>(g.idx), P<int>(g.phorder), n, 3 * levels, st, tmp));
  return GHIP_OK;
}

// the gas targets in the gravity targets' order: compaction of that list (rank[] = dtgt_b, left by
// gas_order_from_gravity_tree)
static int gas_curve_order(ghip_ctx *ctx)
{
  TreeDev &g = ctx->gt, &t = ctx->st;
  const int n = g.n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  const int wg = ghip_wg(ctx);
  GCHK(ghip_ensure(ctx, t.phorder, (size_t) ng * 4));
  if(getenv("GHIP_TARGET_ORDER") && !strcmp(getenv("GHIP_TARGET_ORDER"), "morton"))
    {
      k_iota_tree<<<cdiv(ng, wg), wg, 0, st>>>(ng, P<int>(t.phorder));
      HIPCHK(hipGetLastError());
      return GHIP_OK;
    }
  GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) n * 4));
  int *flag = P<int>(ctx->dflags), *pos = P<int>(ctx->dtgt_a);
  k_gas_flags_ph<<<cdiv(n, wg), wg, 0, st>>>(n, ng, P<int>(g.phorder), P<int>(g.perm), flag);
  GCHK(exclusive_sum(ctx, flag, pos, n, st));
  k_gas_compact_ph<<<cdiv(n, wg), wg, 0, st>>>(n, ng, P<int>(g.phorder), P<int>(g.perm),
                                               P<int>(ctx->dtgt_b), pos, P<int>(t.phorder));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

static void tree_reset(TreeDev &t, int n)
{
  t.n = n;
  t.nnodes = 0;
  t.nelem = n;
  t.built = (n == 0);
}

// sources of a multi-GPU gravity tree: the local particles followed by the imported elements
// (ghip_dd.hip fills dd.src_* from the field arrays and the received LetRec records)
__global__ void k_concat_sources(int n, int nimp, const double *__restrict__ x,
                                 const double *__restrict__ y, const double *__restrict__ z,
                                 const double *__restrict__ m, const double *__restrict__ soft,
                                 const LetRec *__restrict__ imp, double *__restrict__ ox,
                                 double *__restrict__ oy, double *__restrict__ oz,
                                 double *__restrict__ om, double *__restrict__ oa,
                                 unsigned long long *__restrict__ okey, int *__restrict__ olvl)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n + nimp)
    return;
  if(i < n)
    {
      ox[i] = x[i];
      oy[i] = y[i];
      oz[i] = z[i];
      om[i] = m[i];
      oa[i] = soft[i];
      okey[i] = 0;
      olvl[i] = 0;
    }
  else
    {
      const LetRec r = imp[i - n];
      ox[i] = r.x;
      oy[i] = r.y;
      oz[i] = r.z;
      om[i] = r.m;
      oa[i] = r.aux;
      okey[i] = r.key;
      olvl[i] = r.level;
    }
}

int ghip_tree_build_impl(ghip_ctx *ctx)
{
  int n = ctx->n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double *y = x + n, *z = y + n;
  const double *m = P<double>(ctx->f[GHIP_F_MASS]);
  // multi-GPU: the tree's sources are the local particles [0, n) plus the elements imported from
  // the other shards [n, n + nimp); targets are local particles only.  The gas tree of a shard
  // (local gas + ghosts) is built by ghip_dd.hip, not here.
  const bool dd = ctx->dd.on;
  const int nimp = dd ? ctx->dd.gt_nimp : 0;
  const int nsrc = n + nimp;
  HIPCHK(hipEventRecord(ctx->evp[0], st));
  tree_reset(ctx->gt, nsrc);
  tree_reset(ctx->st, dd ? 0 : ng);
  if(dd)
    ctx->st.built = false;
  ctx->gas_pending = false;
  ctx->lists_dirty = ctx->gas_list_dirty = true;
  ctx->stats.tree_nodes = ctx->stats.gastree_nodes = 0;
  if(nsrc == 0)
    {
      HIPCHK(hipEventRecord(ctx->evp[1], st));
      return GHIP_OK;
    }
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));

  // per-particle softening in host order (aux of the gravity tree's particle elements)
  GCHK(ghip_ensure(ctx, ctx->ssoft, (size_t) nsrc * 8));
  GCHK(ghip_ensure(ctx, ctx->soldacc, (size_t) nsrc * 8));
  double *tmp_soft = P<double>(ctx->soldacc);  // scratch until the first gravity call
  if(n > 0 && nimp > 0)   // (without imports the softenings come out of the key kernel, sort_by_key)
    k_soft_of_type<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(ctx->f[GHIP_F_TYPE]), ctx->soft[0],
                                                 ctx->soft[1], ctx->soft[2], ctx->soft[3],
                                                 ctx->soft[4], ctx->soft[5],
                                                 ctx->adaptive_gravsoft ? P<double>(ctx->f[GHIP_F_HSML])
                                                                        : nullptr,
                                                 tmp_soft);
  const unsigned long long *given_keys = nullptr;
  const int *src_lvl = nullptr;
  if(nimp > 0)
    {
      DDState &D = ctx->dd;
      GCHK(ghip_ensure(ctx, D.src_x, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_y, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_z, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_m, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_aux, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_key, (size_t) nsrc * 8));
      GCHK(ghip_ensure(ctx, D.src_lvl, (size_t) nsrc * 4));
      k_concat_sources<<<cdiv(nsrc, 256), 256, 0, st>>>(
        n, nimp, x, y, z, m, tmp_soft, P<LetRec>(D.let_recv), P<double>(D.src_x), P<double>(D.src_y),
        P<double>(D.src_z), P<double>(D.src_m), P<double>(D.src_aux),
        P<unsigned long long>(D.src_key), P<int>(D.src_lvl));
      HIPCHK(hipGetLastError());
      x = P<double>(D.src_x);
      y = P<double>(D.src_y);
      z = P<double>(D.src_z);
      m = P<double>(D.src_m);
      tmp_soft = P<double>(D.src_aux);
      given_keys = P<unsigned long long>(D.src_key);
      src_lvl = P<int>(D.src_lvl);
    }

  // Asynchronous build: the particle number is the last verified build's, so the element buffers
  // already hold that build's nodes with slack.  Everything below is enqueued without waiting for the
  // counts; ghip_tree_verify checks them later (see ghip_internal.h).  Otherwise -- a first build, a
  // changed particle number, multi-GPU shards whose imports change every step -- the host waits for
  // the counting stage once (ev_sizes) and sizes the buffers exactly.
  const bool use_gas = ng > 0 && !dd;
  TreeDev &G = ctx->gt, &S = ctx->st;
  bool async = ctx->tree_async_ok && !dd && !ctx->in_recover && G.last_n == nsrc && G.cap_nodes > 0 &&
               (!use_gas || (S.last_n == ng && S.cap_nodes > 0));
  if(async)
    {
      // (when the last verified count has eaten most of the slack: grow now rather than fail the build)
      GCHK(tree_ensure_elements(ctx, G, nsrc, G.last_nnodes + G.last_nnodes / 16 + 256 > G.cap_nodes
                                                ? G.last_nnodes + G.last_nnodes / 4 + 4096
                                                : G.cap_nodes));
    }
  ctx->build_gen++;
  ctx->grav_log.clear();
  ctx->tree_unverified = false;
  ctx->gas_unverified = false;
  ctx->gas_async = async;
  bool wide = ctx->sort_wide || (getenv("GHIP_SORT64") && atoi(getenv("GHIP_SORT64")) == 1);
  for(;;)
    {
      GCHK(sort_by_key(ctx, G, nsrc, x, y, z, wide, n, given_keys, false,
                       nimp == 0 ? P<int>(ctx->f[GHIP_F_TYPE]) : nullptr, tmp_soft));
      GCHK(count_nodes(ctx, G, nsrc, false, async ? G.cap_nodes : 0x7fffffff, wide, src_lvl));
      HIPCHK(hipEventRecord(ctx->ev_sizes, st));
      if(async)
        break;
      HIPCHK(ghip_event_sync(ctx, ctx->ev_sizes));   // the one host round trip of a synchronous build
      if(wide || G.hsz->bad != 2)
        break;
      wide = true;   // strongly clustered input: long runs of equal top key bits, sort full keys ...
      ctx->sort_wide = true;   // ... from now on
    }
  if(async)
    {
      // host copies: the last verified build's, until ghip_tree_verify brings this build's
      G.nnodes = G.last_nnodes;
      G.nelem = nsrc + G.nnodes;
      ctx->tree_unverified = true;
    }
  else
    GCHK(tree_adopt_sizes(ctx, G, true));

  GCHK(ghip_ensure(ctx, ctx->sx, (size_t) nsrc * 8));
  GCHK(ghip_ensure(ctx, ctx->sy, (size_t) nsrc * 8));
  GCHK(ghip_ensure(ctx, ctx->sz, (size_t) nsrc * 8));
  GCHK(ghip_ensure(ctx, ctx->gt.iperm, (size_t) nsrc * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nsrc * 5 * sizeof(double)));
  GCHK(emit_tree(ctx, ctx->gt, nsrc, x, y, z, m, tmp_soft, P<double>(ctx->sx),
                 P<double>(ctx->sy), P<double>(ctx->sz), P<double>(ctx->stage),
                 P<double>(ctx->ssoft), true, &ctx->evt[0],
                 src_lvl ? P<int>(ctx->gt.slvl) : nullptr));
  ctx->gt.built = true;
  // The curve order of the targets needs the tree-order positions only (the gather at the head of
  // emit_tree): it runs on a second stream next to the element emission, the moment pass and the
  // walk records.  Both chains are strings of short launches, so side by side they take the longer
  // of the two instead of the sum.
  // (the main stream's remaining launches are enqueued first: the host needs longer to enqueue the
  // curve order's ~25 launches than the device needs to run the main chain)
  GCHK(ghip_build_segments(ctx, ctx->gt, true));
  HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->evt[0], 0));
  GCHK(curve_order(ctx, ctx->stream2));
  HIPCHK(hipEventRecord(ctx->evt[1], ctx->stream2));
  HIPCHK(hipStreamWaitEvent(st, ctx->evt[1], 0));

  // gas tree over host indices [0, ngas): same cells, gas only; aux = Hsml.  Its order is the
  // gravity tree's with the other types removed.  All of its build is deferred to the first call that
  // needs it (ghip_finish_gas_tree), which normally is ghip_density -- enqueued while a gravity pair
  // is in flight, so that this work runs underneath the walks instead of in front of them.
  if(use_gas)
    ctx->gas_pending = true;
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evp[1], st));
  ctx->stats.tree_nodes = ctx->gt.nnodes;
  ctx->stats.gastree_nodes = ctx->st.nnodes;
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: the gas tree of a shard = its own gas particles [0, ngas) + the ghosts imported from
// the other shards [ngas, ngas + nghost) (ghip_dd.hip).  Ghosts are sources only.
// ---------------------------------------------------------------------------------------------
__global__ void k_concat_gas_sources(int ng, int nghost, int n, const double *__restrict__ pos,
                                     const double *__restrict__ mass, const double *__restrict__ h,
                                     const GhostRec *__restrict__ gh, double *__restrict__ ox,
                                     double *__restrict__ oy, double *__restrict__ oz,
                                     double *__restrict__ om, double *__restrict__ oh)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= ng + nghost)
    return;
  if(i < ng)
    {
      ox[i] = pos[i];
      oy[i] = pos[(size_t) n + i];
      oz[i] = pos[2 * (size_t) n + i];
      om[i] = mass[i];
      oh[i] = h[i];
    }
  else
    {
      const GhostRec &r = gh[i - ng];
      ox[i] = r.p[0];
      oy[i] = r.p[1];
      oz[i] = r.p[2];
      om[i] = r.p[3];
      oh[i] = r.p[7];
    }
}

// the ghosts' records (gp, gq) in gas-tree order, straight from the received GhostRec
__global__ void k_place_ghosts(int nsrc, int ng, const int *__restrict__ perm,
                               const GhostRec *__restrict__ gh, double *__restrict__ gp,
                               double *__restrict__ gq)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= nsrc)
    return;
  const int i = perm[s];
  if(i < ng)
    return;
  const GhostRec &r = gh[i - ng];
  for(int c = 0; c < 8; c++)
    {
      gp[(size_t) 8 * s + c] = r.p[c];
      gq[(size_t) 8 * s + c] = r.q[c];
    }
}

// k_gather_gas for the local part of a shard's gas tree (sources >= ng are ghosts)
__global__ void k_gather_gas_local(int nsrc, int ng, const int *__restrict__ perm,
                                   const double *__restrict__ x, const double *__restrict__ y,
                                   const double *__restrict__ z, const double *__restrict__ m,
                                   const double *__restrict__ vx, const double *__restrict__ vy,
                                   const double *__restrict__ vz, const double *__restrict__ h,
                                   const double *__restrict__ pres, const double *__restrict__ rho,
                                   const double *__restrict__ dhf, const double *__restrict__ divv,
                                   const double *__restrict__ curl, const int *__restrict__ timebin,
                                   double *__restrict__ gp, double *__restrict__ gq)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= nsrc)
    return;
  int i = perm[s];
  if(i >= ng)
    return;
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

int ghip_dd_build_gas_tree(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  if(ctx->gas_types_unknown)
    {
      ctx->gas_types_unknown = false;
      GCHK(ghip_check_gas_types(ctx));
    }
  const int n = ctx->n, ng = ctx->ngas, nghost = D.nghost, nsg = ng + nghost;
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->st;
  tree_reset(t, nsg);
  ctx->gas_pending = false;
  ctx->lists_dirty = ctx->gas_list_dirty = true;
  ctx->stats.gastree_nodes = 0;
  if(nsg == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, D.gsx, (size_t) nsg * 8));
  GCHK(ghip_ensure(ctx, D.gsy, (size_t) nsg * 8));
  GCHK(ghip_ensure(ctx, D.gsz, (size_t) nsg * 8));
  GCHK(ghip_ensure(ctx, D.gsm, (size_t) nsg * 8));
  GCHK(ghip_ensure(ctx, D.gsh, (size_t) nsg * 8));
  double *gx = P<double>(D.gsx), *gy = P<double>(D.gsy), *gz = P<double>(D.gsz),
         *gm = P<double>(D.gsm), *gh = P<double>(D.gsh);
  k_concat_gas_sources<<<cdiv(nsg, 256), 256, 0, st>>>(
    ng, nghost, n, P<double>(ctx->f[GHIP_F_POS]), P<double>(ctx->f[GHIP_F_MASS]),
    P<double>(ctx->f[GHIP_F_HSML]), P<GhostRec>(D.gh_recv), gx, gy, gz, gm, gh);
  HIPCHK(hipGetLastError());
  // (a shard's ghosts change every step: synchronous sizes, like the shard's gravity trees)
  ctx->build_gen++;
  bool wide = ctx->sort_wide || (getenv("GHIP_SORT64") && atoi(getenv("GHIP_SORT64")) == 1);
  for(;;)
    {
      GCHK(sort_by_key(ctx, t, nsg, gx, gy, gz, wide, -1, nullptr, true));
      GCHK(count_nodes(ctx, t, nsg, true, 0x7fffffff, wide));
      HIPCHK(ghip_stream_sync(ctx, st));
      if(wide || t.hsz->bad != 2)
        break;
      wide = true;
      ctx->sort_wide = true;
    }
  GCHK(tree_adopt_sizes(ctx, t, true));
  GCHK(ghip_ensure(ctx, t.iperm, (size_t) nsg * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nsg * 5 * sizeof(double)));
  double *s = P<double>(ctx->stage);
  GCHK(emit_tree(ctx, t, nsg, gx, gy, gz, gm, gh, s, s + nsg, s + 2 * (size_t) nsg,
                 s + 3 * (size_t) nsg, s + 4 * (size_t) nsg, false));
  t.built = true;
  GCHK(ghip_sph_fill_nodes(ctx, false));
  GCHK(ghip_ensure(ctx, ctx->gp, (size_t) nsg * 64));
  GCHK(ghip_ensure(ctx, ctx->gq, (size_t) nsg * 64));
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double *vp = P<double>(ctx->f[GHIP_F_VELPRED]);
  if(ng > 0)
    k_gather_gas_local<<<cdiv(nsg, 256), 256, 0, st>>>(
      nsg, ng, P<int>(t.perm), x, x + n, x + 2 * (size_t) n, P<double>(ctx->f[GHIP_F_MASS]), vp,
      vp + ng, vp + 2 * (size_t) ng, P<double>(ctx->f[GHIP_F_HSML]),
      P<double>(ctx->f[GHIP_F_PRESSURE]), P<double>(ctx->f[GHIP_F_DENSITY]),
      P<double>(ctx->f[GHIP_F_DHSMLFAC]), P<double>(ctx->f[GHIP_F_DIVVEL]),
      P<double>(ctx->f[GHIP_F_CURLVEL]), P<int>(ctx->f[GHIP_F_TIMEBIN]), P<double>(ctx->gp),
      P<double>(ctx->gq));
  if(nghost > 0)
    k_place_ghosts<<<cdiv(nsg, 256), 256, 0, st>>>(nsg, ng, P<int>(t.perm), P<GhostRec>(D.gh_recv),
                                                  P<double>(ctx->gp), P<double>(ctx->gq));
  HIPCHK(hipGetLastError());
  t.n = nsg;
  GCHK(ghip_mark_converted_gas(ctx));   // (a ghost that is converted on its owner arrives marked)
  // curve order of the sources (the targets are selected from it): leading digits of the
  // Peano-Hilbert key of the tree-order positions, as curve_order does for the gravity tree
  GCHK(ghip_ensure(ctx, t.phorder, (size_t) nsg * 4));
  int levels = 3;
  for(long long m = nsg; m > 1; m >>= 3)
    levels++;
  if(levels > 10)
    levels = 10;
  const double fac = 1.0 / ctx->dlen * (double) (1ULL << GHIP_BITS);
  unsigned int *hi_in = P<unsigned int>(t.phkey), *hi_out = hi_in + nsg;
  k_peano_hi<<<cdiv(nsg, 256), 256, 0, st>>>(nsg, levels, s, s + nsg, s + 2 * (size_t) nsg,
                                             ctx->corner[0], ctx->corner[1], ctx->corner[2], fac,
                                             hi_in, P<int>(t.idx));
  HIPCHK(hipGetLastError());
  GCHK(sort_pairs_u32(ctx, hi_in, hi_out, P<int>(t.idx), P<int>(t.phorder), nsg, 3 * levels, st, ctx->cubtmp));
  ctx->stats.gastree_nodes = t.nnodes;
  return GHIP_OK;
}

// the ghosts' records as their owners hold them after density() (same order as the first import)
int ghip_dd_refresh_ghosts(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  TreeDev &t = ctx->st;
  if(!t.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghost refresh without a gas tree");
  if(D.nghost > 0)
    {
      k_place_ghosts<<<cdiv(t.n, 256), 256, 0, ctx->stream>>>(t.n, ctx->ngas, P<int>(t.perm),
                                                             P<GhostRec>(D.gh_recv),
                                                             P<double>(ctx->gp), P<double>(ctx->gq));
      HIPCHK(hipGetLastError());
      GCHK(ghip_mark_converted_gas(ctx));   // (the refreshed records arrive with their masses)
    }
  return ghip_gastree_refresh_hmax(ctx);
}

// the deferred build of the gas tree: its order out of the gravity tree's, node counts, elements,
// moments (hmax), SphNode records, gas records, curve order of the gas targets
static int build_gas_tree(ghip_ctx *ctx, bool async)
{
  const int n = ctx->n, ng = ctx->ngas;
  hipStream_t st = ctx->stream;
  TreeDev &S = ctx->st;
  const double *x = P<double>(ctx->f[GHIP_F_POS]);
  const double *y = x + n, *z = y + n;
  const double *m = P<double>(ctx->f[GHIP_F_MASS]);
  const double *h = P<double>(ctx->f[GHIP_F_HSML]);
  S.n = ng;
  if(async)
    GCHK(tree_ensure_elements(ctx, S, ng, S.last_nnodes + S.last_nnodes / 16 + 256 > S.cap_nodes
                                            ? S.last_nnodes + S.last_nnodes / 4 + 4096
                                            : S.cap_nodes));
  GCHK(gas_order_from_gravity_tree(ctx));
  GCHK(count_nodes(ctx, S, ng, true, async ? S.cap_nodes : 0x7fffffff, true));
  HIPCHK(hipEventRecord(ctx->ev_sizes_gas, st));
  if(async)
    {
      S.nnodes = S.last_nnodes;
      S.nelem = ng + S.nnodes;
      ctx->gas_unverified = true;
    }
  else
    {
      HIPCHK(ghip_event_sync(ctx, ctx->ev_sizes_gas));
      GCHK(tree_adopt_sizes(ctx, S, true));
      ctx->gas_unverified = false;
    }
  GCHK(ghip_ensure(ctx, S.iperm, (size_t) ng * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) ng * 5 * sizeof(double)));
  double *s = P<double>(ctx->stage);
  GCHK(emit_tree(ctx, S, ng, x, y, z, m, h, s, s + ng, s + 2 * (size_t) ng, s + 3 * (size_t) ng,
                 s + 4 * (size_t) ng, false));
  S.built = true;
  GCHK(ghip_sph_fill_nodes(ctx, false));
  GCHK(ghip_ensure(ctx, ctx->gp, (size_t) ng * 64));
  GCHK(ghip_ensure(ctx, ctx->gq, (size_t) ng * 64));
  const double *vp = P<double>(ctx->f[GHIP_F_VELPRED]);
  k_gather_gas<<<cdiv(ng, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
    ng, P<int>(S.perm), x, y, z, m, vp, vp + ng, vp + 2 * (size_t) ng, h,
    P<double>(ctx->f[GHIP_F_PRESSURE]), P<double>(ctx->f[GHIP_F_DENSITY]),
    P<double>(ctx->f[GHIP_F_DHSMLFAC]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    P<double>(ctx->f[GHIP_F_CURLVEL]), P<int>(ctx->f[GHIP_F_TIMEBIN]), P<double>(ctx->gp),
    P<double>(ctx->gq));
  HIPCHK(hipGetLastError());
  GCHK(ghip_mark_converted_gas(ctx));
  GCHK(gas_curve_order(ctx));
  ctx->gas_list_dirty = true;
  ctx->stats.gastree_nodes = S.nnodes;
  return GHIP_OK;
}

int ghip_finish_gas_tree(ghip_ctx *ctx)
{
  if(!ctx->gas_pending || ctx->gas_wait_upload)
    return GHIP_OK;
  ctx->gas_pending = false;
  return build_gas_tree(ctx, ctx->gas_async && ctx->st.last_n == ctx->ngas && ctx->st.cap_nodes > 0);
}

// the gas tree's counterpart of ghip_tree_verify: before the h iteration modifies smoothing lengths
int ghip_gas_verify(ghip_ctx *ctx)
{
  if(!ctx->gas_unverified)
    return GHIP_OK;
  ctx->gas_unverified = false;
  HIPCHK(ghip_event_sync(ctx, ctx->ev_sizes_gas));
  GCHK(tree_adopt_sizes(ctx, ctx->st, false));
  ctx->stats.gastree_nodes = ctx->st.nnodes;
  if(ctx->st.hsz->bad != 0)
    {
      // nothing has used the tree yet (nelem = 0 on the device): once more, with the host waiting for
      // the count and the buffers grown to it
      HIPCHK(ghip_stream_sync(ctx, ctx->stream));
      return build_gas_tree(ctx, false);
    }
  return GHIP_OK;
}

// refresh Extnodes[].hmax of the gas tree from the current smoothing lengths in gp[].h
// (force_update_hmax, forcetree.c:1661-1786; recomputed exactly instead of only raised)
__global__ void k_aux_from_gp(const TreeSizes *__restrict__ ts, const int4 *__restrict__ lk,
                              const double *__restrict__ gp, double *__restrict__ aux)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= ts->nelem)
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
  const int wg = ghip_wg(ctx), ce = t.n + t.cap_nodes;
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  HIPCHK(hipEventRecord(ctx->evp[8], st));
  k_aux_from_gp<<<cdiv(ce, wg), wg, 0, st>>>(ts, P<int4>(t.lk), P<double>(ctx->gp), P<double>(t.aux));
  HIPCHK(hipGetLastError());
  GCHK((moment_pass<false, false>(ctx, t, false)));
  GCHK(ghip_sph_fill_nodes(ctx, true));
  HIPCHK(hipEventRecord(ctx->evp[9], st));
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// verification of an asynchronously built tree (see ghip_internal.h)
// ---------------------------------------------------------------------------------------------
static int tree_recover(ghip_ctx *ctx)
{
  // Whatever ran on the bad tree did nothing (nelem = 0 on the device).  Drain it, repeat the build
  // synchronously -- the buffers grow / the sort goes wide as the counts demand -- and replay the
  // gravity calls made since.
  ctx->in_recover = true;
  const std::vector<ghip_ctx::GravCall> log = ctx->grav_log;
  int r = GHIP_OK;
  if(hipStreamSynchronize(ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream2) != hipSuccess ||
     hipStreamSynchronize(ctx->stream3) != hipSuccess)
    r = ghip_fail(ctx, GHIP_EHIP, "tree recovery: stream synchronisation failed");
  ctx->n_syncs += 3;
  ctx->grav_pending = false;
  if(ctx->gt.hsz->bad == 2)
    ctx->sort_wide = true;
  if(r == GHIP_OK)
    r = ghip_tree_build_impl(ctx);
  for(size_t k = 0; r == GHIP_OK && k < log.size(); k++)
    {
      if(k > 0)
        r = ghip_join_pair(ctx);
      if(r == GHIP_OK)
        r = ghip_gravity_impl(ctx, &log[k].p, log[k].walk);
    }
  ctx->in_recover = false;
  return r;
}

int ghip_tree_verify(ghip_ctx *ctx)
{
  if(!ctx->tree_unverified)
    return GHIP_OK;
  ctx->tree_unverified = false;
  HIPCHK(ghip_event_sync(ctx, ctx->ev_sizes));   // the counting stage only: the walks may still be running
  GCHK(tree_adopt_sizes(ctx, ctx->gt, false));
  ctx->stats.tree_nodes = ctx->gt.nnodes;
  if(ctx->gt.hsz->bad != 0)
    return tree_recover(ctx);
  return GHIP_OK;
}

// flags of the tree-order sources that are local particles (multi-GPU)
__global__ void k_flag_local(int n, int limit, const int *__restrict__ perm, int *__restrict__ flags)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < n)
    flags[s] = perm[s] < limit ? 1 : 0;
}

// A record of the gas block [0, ngas) whose Type is not 0 any more (a particle converted since the last
// rearrange_particle_sequence(): the reference rearranges at domain decompositions only, domain.c:128,
// and its loops test P[].Type) is neither an SPH target (density.c:1049, hydra.c:184) nor anybody's
// neighbour (ngb.c:93, 213).  It stays in the gas tree; its record is marked with a negative mass, which
// the neighbour loops skip, and the target lists leave it out.
// (massless != 0, the -DBLACK_HOLES / -DDUST builds: a swallowed gas particle, Mass == 0, is skipped by
// the neighbour loops as well -- density.c:831-834, hydra.c:1235-1238 -- while it stays a target; a ghost
// arrives with its mass and is marked here)
__global__ void k_mark_converted(int nsrc, int ngas, int massless, const int *__restrict__ perm,
                                 const int *__restrict__ type, double *__restrict__ gp)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= nsrc)
    return;
  const int i = perm[s];
  if((i < ngas && type[i] != 0) || (massless && gp[(size_t) 8 * s + 3] == 0))
    gp[(size_t) 8 * s + 3] = -1.0;
}

__global__ void k_unflag_converted(int n, int limit, const int *__restrict__ perm, const int *__restrict__ type,
                                   int *__restrict__ flags)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  const int i = perm[s];
  if(i < limit && type[i] != 0)
    flags[s] = 0;
}

int ghip_mark_converted_gas(ghip_ctx *ctx)
{
  TreeDev &t = ctx->st;
  if((!ctx->gas_mixed && !ctx->skip_massless) || t.n == 0)
    return GHIP_OK;
  const int wg = ghip_wg(ctx);
  k_mark_converted<<<cdiv(t.n, wg), wg, 0, ctx->stream>>>(t.n, ctx->ngas, ctx->skip_massless ? 1 : 0,
                                                          P<int>(t.perm), P<int>(ctx->f[GHIP_F_TYPE]),
                                                          P<double>(ctx->gp));
  HIPCHK(hipGetLastError());
  ctx->massless_marked = ctx->skip_massless != 0;
  return GHIP_OK;
}

// -DDUST without -DBLACK_HOLES: hydro_force() sums the massless gas that density() skipped (hydra.c:1235
// is under BLACK_HOLES alone): the mark comes off the records that are gas
__global__ void k_unmark_massless(int nsrc, int ngas, const int *__restrict__ perm, const int *__restrict__ type,
                                  double *__restrict__ gp)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= nsrc)
    return;
  const int i = perm[s];
  if(gp[(size_t) 8 * s + 3] < 0 && (i >= ngas || type[i] == 0))
    gp[(size_t) 8 * s + 3] = 0.0;
}

int ghip_unmark_massless_for_hydro(ghip_ctx *ctx)
{
  TreeDev &t = ctx->st;
  if(!ctx->massless_marked || ctx->skip_massless != 1 || t.n == 0)
    return GHIP_OK;
  const int wg = ghip_wg(ctx);
  k_unmark_massless<<<cdiv(t.n, wg), wg, 0, ctx->stream>>>(t.n, ctx->ngas, P<int>(t.perm),
                                                           P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->gp));
  HIPCHK(hipGetLastError());
  ctx->massless_marked = false;
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
  const bool imports = n > host_limit;   // multi-GPU: sources beyond the local particles are never targets
  const bool mixed = (&t == &ctx->st) && ctx->gas_mixed;   // converted particles in the gas block: no targets
  if(ctx->nactive < 0 && !imports && !mixed)
    {
      HIPCHK(hipMemcpyAsync(list.p, t.phorder.p, (size_t) n * 4, hipMemcpyDeviceToDevice, st));
      *count = n;
      return GHIP_OK;
    }
  if(ctx->nactive == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) n * 4));
  GCHK(ghip_ensure(ctx, ctx->dtgt_a, (size_t) n * 4 + 16));
  if(ctx->nactive < 0)
    k_flag_local<<<cdiv(n, 256), 256, 0, st>>>(n, host_limit, P<int>(t.perm), P<int>(ctx->dflags));
  else
    {
      HIPCHK(hipMemsetAsync(ctx->dflags.p, 0, (size_t) n * 4, st));
      k_mark_active<<<cdiv(ctx->nactive, 256), 256, 0, st>>>(ctx->nactive, P<int>(ctx->act_host_idx),
                                                              P<int>(t.iperm), host_limit,
                                                              P<int>(ctx->dflags));
    }
  if(mixed)
    k_unflag_converted<<<cdiv(n, 256), 256, 0, st>>>(n, host_limit, P<int>(t.perm), P<int>(ctx->f[GHIP_F_TYPE]),
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
  HIPCHK(ghip_stream_sync(ctx, st));
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
  if(ctx->nactive >= 0 && (ctx->lists_dirty || ctx->gas_list_dirty))
    GCHK(ghip_finish_gas_tree(ctx));   // an active subset is marked through st.iperm
  if(ctx->lists_dirty)
    {
      GCHK(make_list(ctx, ctx->gt, ctx->n, ctx->tg_grav, &ctx->nt_grav));
      GCHK(permute_for_shards(ctx, ctx->tg_grav, ctx->nt_grav));
      ctx->lists_dirty = false;
    }
  // the gas list follows once the (deferred) gas tree exists: the gravity walks do not need it
  if(ctx->gas_list_dirty && !ctx->gas_pending)
    {
      if(ctx->dd.on && !ctx->st.built)
        ctx->nt_gas = 0;   // multi-GPU: the gas list is made when the shard's gas tree exists (ghip_dd.hip)
      else
        GCHK(make_list(ctx, ctx->st, ctx->ngas, ctx->tg_gas, &ctx->nt_gas));
      GCHK(permute_for_shards(ctx, ctx->tg_gas, ctx->nt_gas));
      ctx->gas_list_dirty = false;
    }
  return GHIP_OK;
}
