// ghip_keys.h -- space-filling-curve keys on the device (peano.c:300-358), shared by the tree
// build (ghip_tree.hip) and the domain decomposition (ghip_dd.hip).
#pragma once
#include "ghip_internal.h"

// ---------------------------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned long long d_spread3(unsigned long long v)
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
static __device__ __forceinline__ unsigned long long d_morton21(int x, int y, int z)
{
  return d_spread3((unsigned long long) x) | (d_spread3((unsigned long long) y) << 1) |
         (d_spread3((unsigned long long) z) << 2);
}

// peano.c:300-316 peano_hilbert_key.  The reference drives a 48-state table; the same curve is
// generated here from its definition: base octant order ph_base (octant = 4*xbit + 2*ybit + zbit)
// and the cube symmetry carried by each sub-cell (signed axis permutation, out[i] =
// in[perm[i]] ^ flip[i]).  Orientation g starts as the identity; per level:
// local = g(octant); digit = ph_base[local]; g <- child[local] o g.
static __constant__ unsigned char c_ph_base[8] = {0, 7, 1, 6, 3, 4, 2, 5};
static __constant__ unsigned char c_ph_perm[8][3] = {{0, 2, 1}, {0, 2, 1}, {2, 1, 0}, {2, 1, 0},
                                              {0, 1, 2}, {0, 1, 2}, {2, 1, 0}, {2, 1, 0}};
static __constant__ unsigned char c_ph_flip[8][3] = {{0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {1, 0, 1},
                                              {1, 1, 0}, {1, 1, 0}, {0, 0, 0}, {1, 0, 1}};

// the same curve for one 21-bit triplet (target bucketing: 64 Peano-Hilbert-consecutive particles
// fill a box 2.7x smaller than 64 Morton-consecutive ones, so the lanes of a wavefront agree on
// more opening decisions)
static __device__ __forceinline__ unsigned long long d_peano21(int x, int y, int z)
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

// Integer coordinate of a position inside the domain cube, (int) ((x - corner) * fac) as the reference
// forms it (forcetree.c:181-185, domain.c:1090) -- clamped to the cube's 2^21 cells.  A particle of a
// non-periodic run that has drifted outside the cube handed to ghip_dd_set_domain (or sits exactly on
// its upper face) would otherwise yield a negative or 22-bit integer and, from it, a garbage key: it
// could be sent to an arbitrary rank and pass that rank's range check.  Clamped it stays with the rank
// that owns the boundary cell; *outside (if given) is raised so that the caller can insist on a fresh
// extent (the reference recomputes it at every decomposition, domain.c:1972-2014).
static __device__ __forceinline__ int d_cell21(double x, double corner, double fac, int *outside = nullptr)
{
  const double c = (x - corner) * fac;
  int i = (int) c;
  const int top = (1 << GHIP_BITS) - 1;
  if(!(c >= 0) || i > top)
    {
      if(outside)
        *(volatile int *) outside = 6;
      i = !(c >= 0) ? 0 : top;
    }
  return i;
}

// the leading `levels` (<= 10) digits of the same key
static __device__ __forceinline__ unsigned int d_peano_top(int x, int y, int z, int levels)
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


// inverse of d_spread3: collect every third bit of v (bit 3b -> bit b)
static __device__ __forceinline__ int d_compact3(unsigned long long v)
{
  v &= 0x1249249249249249ULL;
  v = (v | (v >> 2)) & 0x10c30c30c30c30c3ULL;
  v = (v | (v >> 4)) & 0x100f00f00f00f00fULL;
  v = (v | (v >> 8)) & 0x001f0000ff0000ffULL;
  v = (v | (v >> 16)) & 0x001f00000000ffffULL;
  v = (v | (v >> 32)) & 0x1fffffULL;
  return (int) v;
}

// Peano-Hilbert key of the cell a 63-bit Morton key lies in (all 21 levels)
static __device__ __forceinline__ unsigned long long d_peano_of_morton(unsigned long long mk)
{
  return d_peano21(d_compact3(mk), d_compact3(mk >> 1), d_compact3(mk >> 2));
}
