// ghip_walk.h -- the gravity tree-walk kernel (device code), included by ghip_gravity.hip.
//
// Replaces the bodies of force_treeevaluate() (forcetree.c:1797-2317),
// force_treeevaluate_shortrange() (forcetree.c:2330-2845) and
// force_treeevaluate_ewald_correction() (forcetree.c:2873-3204).
//
// Mapping.  A BUCKET is 64 targets consecutive along the space-filling curve; a wavefront's 64
// lanes are those targets.  The tree is ONE pre-order element list (ghip_tree.hip); the element
// index a wave looks at is wave-uniform, so element records arrive through the scalar path
// (s_load_dwordx16) and are broadcast for free.  Every lane applies the reference's
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
// afterwards (k_combine_grav), so results are deterministic.
//
// Latency.  The walk is a pointer chase: the next element is known only after the vote.  The
// record a step needs is ONE 64-byte scalar load (the "hot" half: monopole, opening-criterion
// operands, links); the "cold" half (cell centre, len) is fetched only when a lane may sit
// inside the cell or a special walk needs it.  Occupancy (registers) hides the rest.
// No MFMA: irregular fp64 work.
#pragma once
#include "ghip_internal.h"
#include "ghip_count.h"

#ifndef GHIP_WALK_WAVES
#define GHIP_WALK_WAVES 8   // wavefronts per SIMD the register allocator must leave room for
#endif
// instruction diet of the element visit, bit by bit (for A/B builds; all on by default):
//   1  the distance first, ONE compare "is any lane further than half a box away" in front of the
//      three per-axis nearest-image compares
//   2  interaction count: "mass > 0" is a property of the (wave-uniform) record: scalar test, one add
//   4  unsoftened interaction: the skip index is taken over inside the interaction's execution region
//   8  the constant 1.5 of the reciprocal-cube step stays in a register pair
#ifndef GHIP_WALK_OPT
#define GHIP_WALK_OPT 15
#endif


struct GravK
{
  double theta;        // ErrTolTheta (0: relative criterion)
  double errtol;       // ErrTolForceAcc
  double boxsize, boxhalf;
  int periodic, unequal;
  int debug_steps;     // GHIP_DEBUG_STEPS=1: GRAVCOST receives the wave's visited-element count
  int xcd_remap;       // GHIP_WALK_XCD=1: XCD-contiguous block order (off: measured slower)
  double rcut, rcut2, asmthfac;  // shortrange
  double fac_intp;     // ewald: 2*EN/BoxSize
  double k1875;        // 1.875 (a kernel argument lives in scalar registers: see d_mass_over_r3)
};

// multi-GPU: the device error word (pinned host memory) a wavefront sets when a lane has to open an
// imported pruned node.  A device-side pointer, not a kernel argument: it is read only on that
// (never taken) path, so the walk loop carries no register for it.
__device__ int *d_walk_errw = nullptr;

// hot half of an element (64 B):  x, y, z, mass | (mass*len)*len, len*len | skip, flags | aux
//   flags (the word called pidx below): sign bit = node, bit 0 = "mass > 0" (the interaction counts,
//   forcetree.c:2214) -- both are properties of the record, so the walk tests them on the scalar unit
//   node: centre of mass, opening-criterion operands in the reference's operation order
//   (forcetree.c:2085), aux = max softening below (negative: mixed softenings)
//   particle: position, 0, 0, aux = its softening
// cold half (64 B): cx, cy, cz, len | 0.6*len | spare
struct __attribute__((aligned(64))) WalkHot
{
  double x, y, z, m;
  double mlen2, len2;
  int skip, pidx;
  double aux;
};
struct __attribute__((aligned(64))) WalkCold
{
  double cx, cy, cz, len;
  double len06;
  double spare[3];
};

// a 64-byte record in 16 consecutive SGPRs, as eight doubles: every double of a record is an
// aligned register pair as it arrives (a record of sixteen ints would have to be re-assembled into
// doubles with scalar shifts and ors)
typedef double v16i __attribute__((ext_vector_type(8)));

// two neighbouring entries of the Ewald table (8-byte aligned; the hardware takes 16-byte
// global loads at any dword address)
struct __attribute__((aligned(8))) EwPair
{
  double lo, hi;
};
__device__ __forceinline__ EwPair d_ldpair(const double *p)
{
  return *reinterpret_cast<const EwPair *>(p);
}

__device__ __forceinline__ double d_f64(const v16i &v, int i)
{
  return v[i];
}
// the two 32-bit halves of double i
__device__ __forceinline__ int d_lo32(const v16i &v, int i)
{
  return __double2loint(v[i]);
}
__device__ __forceinline__ int d_hi32(const v16i &v, int i)
{
  return __double2hiint(v[i]);
}

// Scalar loads of 64-byte records (wave-uniform index).  Load and wait are ONE asm statement:
// a scalar load writes its destination asynchronously, and between two asm statements the
// register allocator would be free to copy or spill a destination that is still in flight.
// The address is the list's base (an SGPR pair for the whole kernel) plus a 32-bit byte offset
// e * 64 in one SGPR -- the instruction adds them (and an immediate), where a 64-bit pointer per
// record costs four scalar instructions: the walk issues more scalar than vector instructions,
// and the scalar unit is the first to fill (ghip_build_segments refuses lists beyond 2^26
// records).
__device__ __forceinline__ unsigned int d_rec_off(int e)
{
  return (unsigned int) __builtin_amdgcn_readfirstlane(e) << 6;
}
template <class T>
__device__ __forceinline__ void d_load1(const T *__restrict__ base, int e, v16i &R)
{
  const unsigned int off = d_rec_off(e);
  asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(R) : "s"(base), "s"(off) : "memory");
}
// two records in flight together (the Ewald walk's two segment cursors)
template <class T>
__device__ __forceinline__ void d_load2(const T *__restrict__ base, int ea, v16i &A, int eb, v16i &B)
{
  const unsigned int oa = d_rec_off(ea), ob = d_rec_off(eb);
  asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(A), "=&s"(B)
               : "s"(base), "s"(oa), "s"(ob)
               : "memory");
}
// Touch a record so that its cache line is on its way into the scalar cache: one dword into a
// scratch SGPR.  This is the one deliberately asynchronous load: the scratch register is an in/out
// operand of every later wait (d_load1_touch, d_drain_touch), nothing else may use it in between,
// and tests/test_abi_and_host.py checks the generated code for exactly that.
// d_touch_next: the record after the one at byte offset `off` (immediate offset: no address arithmetic)
template <class T>
__device__ __forceinline__ void d_touch_next(const T *__restrict__ base, unsigned int off, int &scratch)
{
  asm volatile("s_load_dword %0, %1, %2 offset:0x40" : "=s"(scratch) : "s"(base), "s"(off) : "memory");
}
template <class T>
__device__ __forceinline__ void d_touch(const T *__restrict__ base, int e, int &scratch)
{
  const unsigned int off = d_rec_off(e);
  asm volatile("s_load_dword %0, %1, %2" : "=s"(scratch) : "s"(base), "s"(off) : "memory");
}
// load the record at byte offset `off` and wait for it and for the outstanding touches
template <class T>
__device__ __forceinline__ void d_load1_touch(const T *__restrict__ base, unsigned int off, v16i &R,
                                              int &s1, int &s2)
{
  asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(R), "+s"(s1), "+s"(s2)
               : "s"(base), "s"(off)
               : "memory");
}
__device__ __forceinline__ void d_drain_touch(int &s1, int &s2)
{
  asm volatile("s_waitcnt lgkmcnt(0) ; drain-touches" : "+s"(s1), "+s"(s2) : : "memory");
}

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

// m / r^3 for the unsoftened interaction: y0 = v_rsq_f64(x), e = 1 - x*y0^2,
// m*y^3 = m*y0^3 * (1 + 3e/2 + 15e^2/8 + O(e^3))   -- the same third-order step as d_rsqrt, cubed (7
// operations where d_rsqrt and three multiplications take 8)
__device__ __forceinline__ double d_mass_over_r3(double mass, double x, double k15 = 1.5,
                                                 double k1875 = 1.875)
{
  double y = __builtin_amdgcn_rsq(x);
  double y2 = y * y;
  double e = fma(-x, y2, 1.0);
  // (neither constant is an inline constant of the instruction set.  Kept by the caller in a vector
  // and a scalar register pair they cost nothing per visit; as literals they cost two moves.)
  double p = fma(k1875, e, k15);
  double c = (mass * y) * y2;
  return fma(c * e, p, c);
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

// trilinear Ewald look-up, forcetree.c:3097-3170, from the one-component table: plain rows of EN+1
// values, or (BRICK) the brick-tiled copy.
template <bool BRICK>
__device__ __forceinline__ void d_ewald_interp(const double *__restrict__ tab, double fac_intp,
                                               double dx, double dy, double dz, double &fx,
                                               double &fy, double &fz)
{
  const int E1 = GHIP_EN + 1;
  // |d| for the look-up (a free source modifier), the sign put back at the end: -1 for d >= 0
  const double sx = dx < 0 ? 1.0 : -1.0, sy = dy < 0 ? 1.0 : -1.0, sz = dz < 0 ? 1.0 : -1.0;
  dx = fabs(dx);
  dy = fabs(dy);
  dz = fabs(dz);
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
  // one table serves the three components: by the cubic symmetry of the lattice sum
  // fy(x,y,z) = fx(y,x,z) and fz(x,y,z) = fx(z,y,x), so tab holds fx only (2.2 MB instead of
  // 6.6 MB: it stays in the 4 MB L2 of an XCD).  Two neighbouring entries along the last index
  // come with one 16-byte load.  Weights: f1..f8 = W[a][b][c] for the corner (i+a, j+b, k+c).
  EwPair x00, x01, x10, x11, y00, y01, y10, y11, z00, z01, z10, z11;
  if(!BRICK)
    {
      // 32-bit byte offsets from the (wave-uniform) table pointer: the loads take them as they are
      // (scalar base + vector offset), the row +1 in the second index through the instruction's
      // immediate offset
      const unsigned int ui = (unsigned int) i, uj = (unsigned int) j, uk = (unsigned int) k;
      const unsigned int R1 = E1 * 8u, R2 = E1 * E1 * 8u;
      const unsigned int ox = (ui * E1 + uj) * R1 + uk * 8u;
      const unsigned int oy = (uj * E1 + ui) * R1 + uk * 8u;
      const unsigned int oz = (uk * E1 + uj) * R1 + ui * 8u;
      const unsigned int ox2 = ox + R2, oy2 = oy + R2, oz2 = oz + R2;
      const char *tb = reinterpret_cast<const char *>(tab);
#define EW_LD(off, imm) d_ldpair(reinterpret_cast<const double *>(tb + (size_t) (off) + (imm)))
      x00 = EW_LD(ox, 0), x01 = EW_LD(ox, R1), x10 = EW_LD(ox2, 0), x11 = EW_LD(ox2, R1);
      y00 = EW_LD(oy, 0), y01 = EW_LD(oy, R1), y10 = EW_LD(oy2, 0), y11 = EW_LD(oy2, R1);
      z00 = EW_LD(oz, 0), z01 = EW_LD(oz, R1), z10 = EW_LD(oz2, 0), z11 = EW_LD(oz2, R1);
#undef EW_LD
    }
  else
    {
      // brick-tiled (ghip_internal.h): 2 x 2 rows x 3 cells of the last index per 128-byte line, so
      // the 64 lanes of a wavefront -- whose cells differ by a few units in every index -- fall into
      // fewer lines than with rows of 65 values.  Byte offset of a value: A(first index) +
      // B(second index) + C(last index).
      const unsigned int SA = GHIP_EW_NB * GHIP_EW_NKB * 128, SB = GHIP_EW_NKB * 128;
      const unsigned int ai = (i >> 1) * SA + (i & 1) * 64, ai1 = ((i + 1) >> 1) * SA + ((i + 1) & 1) * 64;
      const unsigned int aj = (j >> 1) * SA + (j & 1) * 64, aj1 = ((j + 1) >> 1) * SA + ((j + 1) & 1) * 64;
      const unsigned int ak = (k >> 1) * SA + (k & 1) * 64, ak1 = ((k + 1) >> 1) * SA + ((k + 1) & 1) * 64;
      const unsigned int bi = (i >> 1) * SB + (i & 1) * 32, bi1 = ((i + 1) >> 1) * SB + ((i + 1) & 1) * 32;
      const unsigned int bj = (j >> 1) * SB + (j & 1) * 32, bj1 = ((j + 1) >> 1) * SB + ((j + 1) & 1) * 32;
      // last index: brick v/3, slot v%3  ->  (v/3)*16 + v - 3*(v/3) = (v/3)*13 + v   (v < 128)
      const unsigned int ck = ((((unsigned int) k * 43u) >> 7) * 13u + (unsigned int) k) * 8u;
      const unsigned int ci = ((((unsigned int) i * 43u) >> 7) * 13u + (unsigned int) i) * 8u;
      const char *tb = reinterpret_cast<const char *>(tab);
#define EW_LD(off) d_ldpair(reinterpret_cast<const double *>(tb + (off)))
      // x: T[i+a][j+b][k..k+1];  y: T[j+b][i+a][k..k+1] (y<first index offset><second index offset>);
      // z: T[k+c][j+b][i..i+1]
      x00 = EW_LD(ai + bj + ck), x01 = EW_LD(ai + bj1 + ck), x10 = EW_LD(ai1 + bj + ck),
      x11 = EW_LD(ai1 + bj1 + ck);
      y00 = EW_LD(aj + bi + ck), y01 = EW_LD(aj + bi1 + ck), y10 = EW_LD(aj1 + bi + ck),
      y11 = EW_LD(aj1 + bi1 + ck);
      z00 = EW_LD(ak + bj + ci), z01 = EW_LD(ak + bj1 + ci), z10 = EW_LD(ak1 + bj + ci),
      z11 = EW_LD(ak1 + bj1 + ci);
#undef EW_LD
    }
  // fx: rows (a,b), pair over c
  fx = sx * (x00.lo * f1 + x00.hi * f2 + x01.lo * f3 + x01.hi * f4 + x10.lo * f5 + x10.hi * f6 +
             x11.lo * f7 + x11.hi * f8);
  // fy = sum T[j+b][i+a][k+c] W[a][b][c]: rows (b,a), pair over c
  fy = sy * (y00.lo * f1 + y00.hi * f2 + y10.lo * f3 + y10.hi * f4 + y01.lo * f5 + y01.hi * f6 +
             y11.lo * f7 + y11.hi * f8);
  // fz = sum T[k+c][j+b][i+a] W[a][b][c]: rows (c,b), pair over a
  fz = sz * (z00.lo * f1 + z10.lo * f2 + z01.lo * f3 + z11.lo * f4 + z00.hi * f5 + z10.hi * f6 +
             z01.hi * f7 + z11.hi * f8);
}

// per-lane target state shared by both in-flight segments
struct WalkLane
{
  double pos_x, pos_y, pos_z, h_i, aold;
  double h2;   // h_i^2
  double k15;  // 1.5, opaque to the compiler (see GHIP_WALK_OPT bit 8)
  double acc_x, acc_y, acc_z;
  int nint;
};

// NEAREST (forcetree.c:49) with one compare: |x| > boxhalf ? x - copysign(box, x) : x
// (x - (-box) and x + box are the same IEEE operation, so results are identical)
__device__ __forceinline__ double d_nearest1(double x, double boxsize, double boxhalf)
{
  return (fabs(x) > boxhalf) ? x - copysign(boxsize, x) : x;
}

// the same under the execution mask (compare, sign transfer, subtract: 3 vector instructions
// where the select form takes 5; the empty asm keeps the compiler from if-converting it back)
__device__ __forceinline__ double d_nearest1_masked(double x, double boxsize, double boxhalf)
{
  if(fabs(x) > boxhalf)
    {
      x -= copysign(boxsize, x);
      asm volatile("" : "+v"(x));
    }
  return x;
}

// Lane masks.  Every per-lane decision of the walk is held as a 64-bit mask in scalar registers:
// a vector compare writes its result there directly (ballot of a compare), the masks are combined
// on the scalar unit, a vote is a scalar compare with zero, and a mask becomes the execution mask
// of a conditional block without any vector instruction (inverse ballot).
typedef unsigned long long lmask;
#define D_BAL(cond) __builtin_amdgcn_ballot_w64(cond)
#define D_LANE(mask) __builtin_amdgcn_inverse_ballot_w64(mask)

// my_skip = max(my_skip, skip) for the lanes of the current execution mask, in place: as one
// instruction on one register (written as a select or a plain max the compiler keeps two copies of
// the index alive and moves them twice per visit)
__device__ __forceinline__ void d_skip_to(int &my_skip, int skip)
{
  asm volatile("v_max_i32 %0, %1, %0" : "+v"(my_skip) : "s"(skip));
}

// One element of the list for all 64 lanes.  H = hot record (already in SGPRs).  OWNED = false
// replays only the opening decision (ancestor of a segment).  Returns the next element index
// (wave-uniform).  All 64 lanes are active here (lanes without a target carry my_skip = INT_MAX).
template <int MODE, bool PERIODIC, bool UNEQUAL, bool OWNED, bool REL>
__device__ __forceinline__ int d_walk_element(int e, const v16i &H,
                                              const WalkCold *__restrict__ cold, const GravK &p,
                                              const float *__restrict__ srtab,
                                              const double *__restrict__ ewtab, WalkLane &W,
                                              int &my_skip)
{
  const double ex = d_f64(H, 0), ey = d_f64(H, 1), ez = d_f64(H, 2), mass = d_f64(H, 3);
  const double mlen2 = d_f64(H, 4), len2 = d_f64(H, 5);
  const int skip = d_lo32(H, 6), pidx = d_hi32(H, 6);
  const double aux = d_f64(H, 7);
  const lmask act = D_BAL(e >= my_skip);
  int next;

  double dx = ex - W.pos_x, dy = ey - W.pos_y, dz = ez - W.pos_z;
  double r2;
  if((GHIP_WALK_OPT & 1) && (MODE == GHIP_WALK_EWALD || PERIODIC))
    {
      // No lane needs a nearest-image shift unless its distance reaches half a box: |d_k| > boxhalf
      // implies r2 >= boxhalf^2 (the sum only grows and rounding is monotone).  One compare in the
      // common case; otherwise the per-axis shifts under their masks and the distance once more.
      r2 = dx * dx + dy * dy + dz * dz;
      if(D_BAL(r2 >= p.boxhalf * p.boxhalf) != 0)
        {
          if(D_LANE(D_BAL(fabs(dx) > p.boxhalf)))
            {
              dx -= copysign(p.boxsize, dx);
              asm volatile("" : "+v"(dx));
            }
          if(D_LANE(D_BAL(fabs(dy) > p.boxhalf)))
            {
              dy -= copysign(p.boxsize, dy);
              asm volatile("" : "+v"(dy));
            }
          if(D_LANE(D_BAL(fabs(dz) > p.boxhalf)))
            {
              dz -= copysign(p.boxsize, dz);
              asm volatile("" : "+v"(dz));
            }
          r2 = dx * dx + dy * dy + dz * dz;
        }
    }
  else if(MODE == GHIP_WALK_EWALD || PERIODIC)
    {
      // nearest image (forcetree.c:49): three compares into lane masks, ONE scalar test for "no lane
      // on any axis" -- the common case -- and only otherwise the per-axis shifts under their masks
      const lmask wx = D_BAL(fabs(dx) > p.boxhalf), wy = D_BAL(fabs(dy) > p.boxhalf),
                  wz = D_BAL(fabs(dz) > p.boxhalf);
      if((wx | wy | wz) != 0)
        {
          if(D_LANE(wx))
            {
              dx -= copysign(p.boxsize, dx);
              asm volatile("" : "+v"(dx));
            }
          if(D_LANE(wy))
            {
              dy -= copysign(p.boxsize, dy);
              asm volatile("" : "+v"(dy));
            }
          if(D_LANE(wz))
            {
              dz -= copysign(p.boxsize, dz);
              asm volatile("" : "+v"(dz));
            }
        }
      r2 = dx * dx + dy * dy + dz * dz;
    }
  else
    r2 = dx * dx + dy * dy + dz * dz;
  double h = W.h_i, h2 = W.h2;
  lmask interact;

  if(pidx >= 0)
    {
      next = e + 1;
      interact = act;
      if(MODE != GHIP_WALK_EWALD && UNEQUAL)
        {
          if(h < aux)
            {
              h = aux;
              h2 = h * h;
            }
        }
    }
  else
    {
      // first part of the criterion needs only the hot record (forcetree.c:2074-2091)
      lmask gt, rel;
      // (REL = "ErrTolTheta == 0": the kernel runs one of two copies of the traversal, so that the
      // choice of the criterion costs no scalar instruction per element)
      if(REL)
        {
          gt = D_BAL(mlen2 > r2 * r2 * W.aold);
          rel = ~0ull;
        }
      else
        {
          gt = D_BAL(len2 > r2 * p.theta * p.theta);
          rel = 0ull;
        }
      lmask open = act & gt;
      lmask far = act & ~gt & rel;   // passed the distance test, box test still pending
      lmask drop = 0;                // short-range walk: whole cell beyond the cut-off
      // the cold half (cell centre, len) is needed for:
      //  * the "inside the 1.2*len box" rule of the relative criterion (forcetree.c:2093-2104):
      //    only possible when r2 < 3.63*len^2 (centre of mass inside the cell, box half-width
      //    0.6*len: |x - s| < (0.6 + 0.5)*sqrt(3)*len), tested conservatively with 4*len^2
      //    (formed exactly by adding 2 to the exponent, on the scalar unit)
      //  * the short-range cut-off pruning and the Ewald override
      const double len2x4 = __hiloint2double(d_hi32(H, 5) + 0x00200000, d_lo32(H, 5));
      const lmask boxcand = far & D_BAL(r2 < len2x4);
      lmask cutcand = 0;
      if(MODE == GHIP_WALK_SHORTRANGE)
        cutcand = act & D_BAL(r2 > p.rcut2);
      lmask need_cold = boxcand;
      if(MODE == GHIP_WALK_SHORTRANGE)
        need_cold |= cutcand;
      if(MODE == GHIP_WALK_EWALD)
        need_cold |= open;
      if(need_cold != 0)
        {
          v16i C;
          d_load1(cold, e, C);
          const double cx = d_f64(C, 0), cy = d_f64(C, 1), cz = d_f64(C, 2), len = d_f64(C, 3);
          const double len06 = d_f64(C, 4);
          const double c0 = cx - W.pos_x, c1 = cy - W.pos_y, c2 = cz - W.pos_z;
          if(MODE == GHIP_WALK_SHORTRANGE)
            {
              // forcetree.c:2598-2632: whole cell beyond the cut-off -> drop the branch.  The
              // reference tests this BEFORE the opening criterion.
              if(cutcand != 0)
                {
                  double eff = p.rcut + 0.5 * len;
                  double d0 = c0, d1 = c1, d2 = c2;
                  if(PERIODIC)
                    {
                      d0 = d_nearest1(d0, p.boxsize, p.boxhalf);
                      d1 = d_nearest1(d1, p.boxsize, p.boxhalf);
                      d2 = d_nearest1(d2, p.boxsize, p.boxhalf);
                    }
                  drop = cutcand &
                         (D_BAL(fabs(d0) > eff) | D_BAL(fabs(d1) > eff) | D_BAL(fabs(d2) > eff));
                }
              open &= ~drop;
              far &= ~drop;
            }
          if(far != 0)
            open |= far & D_BAL(fabs(c0) < len06) & D_BAL(fabs(c1) < len06) &
                    D_BAL(fabs(c2) < len06);
          if(MODE == GHIP_WALK_EWALD)
            {
              // forcetree.c:3039-3088: the correction is smooth, so an "open" verdict is
              // overridden unless the cell straddles the half-box or is large
              if(open != 0 && !(len > 0.20 * p.boxsize))
                {
                  double u0 = d_nearest1(c0, p.boxsize, p.boxhalf);
                  double u1 = d_nearest1(c1, p.boxsize, p.boxhalf);
                  double u2 = d_nearest1(c2, p.boxsize, p.boxhalf);
                  double lim = 0.5 * (p.boxsize - len);
                  open &= D_BAL(fabs(u0) > lim) | D_BAL(fabs(u1) > lim) | D_BAL(fabs(u2) > lim);
                }
            }
        }
      if(MODE != GHIP_WALK_EWALD && UNEQUAL)
        {
          // forcetree.c:2108-2124
          const double ms = fabs(aux);
          const lmask up = act & ~drop & ~open & D_BAL(h < ms);
          if(up != 0)
            {
              if(D_LANE(up))
                {
                  h = ms;
                  h2 = ms * ms;
                }
              if(aux < 0)
                open |= up & D_BAL(r2 < ms * ms);
            }
        }
      interact = act & ~drop & ~open;
      // (the unsoftened Newtonian interaction below takes the skip index over inside its own execution
      // region; a lane inside its softening length goes through the general block and is handled here)
      if(!((GHIP_WALK_OPT & 4) && OWNED && MODE == GHIP_WALK_NEWTON) || (interact & D_BAL(r2 < h2)) != 0)
        if(D_LANE(interact | drop))
          d_skip_to(my_skip, skip);   // (= skip: my_skip <= e < skip for these lanes)
      next = skip;
      if(open != 0)
        {
          next = e + 1;
          // multi-GPU: a node imported from another shard as a leaf (skip == e + 1: nothing below
          // it came along) must never be opened -- the sender proved that no target here can.  If
          // one does, the locally essential tree was incomplete: hard error, not silently lost
          // mass.  (A cell of this shard's own holds >= 2 particles: its skip is >= e + 3.)
          // (a real branch on the scalar compare first: the pointer is a global load, and a
          // combined condition would wait for it at every opened node)
          if(skip == next)
            {
              int *w = *(int *volatile *) &d_walk_errw;
              if(w)
                *(volatile int *) w = 1;
            }
        }
    }

  // Newtonian walk, nobody inside its softening length (the common case, one scalar test): the plain
  // m / r^3 under ONE execution mask -- the general block below switches masks for the two branches
  // of the softened kernel (forcetree.c:2143-2171)
  if(OWNED && MODE == GHIP_WALK_NEWTON && (interact & D_BAL(r2 < h2)) == 0)
    {
      // mass > 0 as a scalar test on the record's words (a mass is never negative): the high word
      // positive, or a denormal
      // "mass > 0": bit 0 of the record's flag word (k_fill_elems)
      const int counted = pidx & 1;
      // the skip index, taken over by the interacting lanes: of a node > e >= their my_skip; of a
      // particle e + 1, which changes nothing (the next element is e + 1 or beyond)
      const int sk = skip;
      if(D_LANE(interact))
        {
          const double fac = (GHIP_WALK_OPT & 8) ? d_mass_over_r3(mass, r2, W.k15, p.k1875)
                                                 : d_mass_over_r3(mass, r2);
          W.acc_x += dx * fac;
          W.acc_y += dy * fac;
          W.acc_z += dz * fac;
          if(GHIP_WALK_OPT & 2)
            W.nint += counted;
          else
            W.nint += (mass > 0) ? 1 : 0;
          if(GHIP_WALK_OPT & 4)
            d_skip_to(my_skip, sk);
        }
      return next;
    }
  if(OWNED && D_LANE(interact))
    {
      if(MODE == GHIP_WALK_EWALD)
        {
          double fx, fy, fz;
          d_ewald_interp<UNEQUAL>(ewtab, p.fac_intp, dx, dy, dz, fx, fy, fz);   // (UNEQUAL: brick table)
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

// which wavefront works for which bucket (see "adaptive split" below)
struct WalkPlan
{
  int nwaves;                            // wavefronts launched
  const int *__restrict__ wave_bucket;   // [nwaves] bucket of each wavefront, -1: idle
  const int *__restrict__ woff;          // [nbuckets] first wavefront of the bucket
  const int *__restrict__ nsub;          // [nbuckets] wavefronts sharing the bucket
  unsigned int *__restrict__ steps_out;  // [nbuckets] elements visited (summed over wavefronts)
  unsigned int *started;                 // != nullptr: set to 1 when workgroup started_at of the grid starts
  int started_at;
};

// cursor over the segments one wavefront owns through one of its two slots
struct SegCursor
{
  int kseg;      // current segment (>= ns: exhausted)
  int e, s1;     // current element, end of segment
};

// enter segment `c.kseg`: replay the ancestors, position the cursor at the first element any lane
// still needs.  Returns false when the slot has no segment left.
template <int MODE, bool PERIODIC, bool UNEQUAL, bool REL>
__device__ __forceinline__ bool d_enter_segment(SegCursor &c, int stride, const WalkSeg &sg,
                                                const WalkHot *__restrict__ hot,
                                                const WalkCold *__restrict__ cold, const GravK &p,
                                                const float *__restrict__ srtab,
                                                const double *__restrict__ ewtab, bool valid,
                                                WalkLane &W, int &my_skip, unsigned int &steps)
{
  while(c.kseg < sg.ns)
    {
      const int s0 = __builtin_amdgcn_readfirstlane(sg.start[c.kseg]);
      c.s1 = __builtin_amdgcn_readfirstlane(sg.start[c.kseg + 1]);
      my_skip = valid ? 0 : 0x7fffffff;
      const int na = __builtin_amdgcn_readfirstlane(sg.nanc[c.kseg]);
      for(int a = 0; a < na; a++)
        {
          int ea = __builtin_amdgcn_readfirstlane(sg.anc[c.kseg * GHIP_MAXANC + a]);
          v16i H;
          d_load1(hot, ea, H);
          d_walk_element<MODE, PERIODIC, UNEQUAL, false, REL>(ea, H, cold, p, srtab, ewtab, W, my_skip);
          steps++;
        }
      int first = d_wave_min_i32(my_skip);   // every lane below an accepted ancestor: jump
      c.e = __builtin_amdgcn_readfirstlane(first > s0 ? first : s0);
      if(c.e < c.s1)
        return true;
      c.kseg += stride;
    }
  return false;
}

// the traversal of one wavefront: its segments of the element list for its bucket
template <int MODE, bool PERIODIC, bool UNEQUAL, bool REL>
__device__ __forceinline__ void d_walk_run(const WalkSeg &sg, int sub, const WalkHot *__restrict__ hot,
                                           const WalkCold *__restrict__ cold, const GravK &p,
                                           const float *__restrict__ srtab,
                                           const double *__restrict__ ewtab, bool valid, WalkLane &W,
                                           unsigned int &steps)
{
  if(MODE != GHIP_WALK_EWALD)
    {
      // this wavefront owns segments sub, sub+S, sub+2S, ...  (advancing two segments at once was
      // measured for the Newtonian walk: no gain, +9 VGPRs -- that walk is issue-bound)
      SegCursor A;
      int skipA = 0;
      A.kseg = sub;
      const int stride = sg.nsub;
      bool liveA = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(A, stride, sg, hot, cold, p, srtab,
                                                            ewtab, valid, W, skipA, steps);
      // Both possible successors of an element -- e+1 (descend / next particle) and its skip
      // link -- are touched as soon as its record has arrived, so the next step's 64-byte load
      // finds its line in the scalar cache instead of paying the L2 latency of the pointer
      // chase.  (the list carries one padding record: e+1 and skip are always readable)
      int t1 = 0, t2 = 0;
      while(liveA)
        {
          v16i HA;
          const unsigned int offA = d_rec_off(A.e);
          d_load1_touch(hot, offA, HA, t1, t2);
          d_touch_next(hot, offA, t1);
          d_touch(hot, d_lo32(HA, 6), t2);
          steps++;
          A.e = __builtin_amdgcn_readfirstlane(
            d_walk_element<MODE, PERIODIC, UNEQUAL, true, REL>(A.e, HA, cold, p, srtab, ewtab, W, skipA));
          if(A.e >= A.s1)
            {
              // drain the touches first: their scratch registers must not be live while a load
              // into them is in flight across code the register allocator is free to spill in
              d_drain_touch(t1, t2);
              A.kseg += stride;
              liveA = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(A, stride, sg, hot, cold, p, srtab,
                                                               ewtab, valid, W, skipA, steps);
            }
        }
      d_drain_touch(t1, t2);
    }
  else
    {
      // Ewald walk: bound by the table gathers (12 x 16 B per lane and interaction through the
      // vector-memory path).  Two of the wavefront's segments advance together (slot A: sub,
      // sub+2S, ...; slot B: sub+S, sub+3S, ...) so that two elements' gathers are in flight.
      SegCursor A, B;
      int skipA = 0, skipB = 0;
      A.kseg = sub;
      B.kseg = sub + sg.nsub;
      const int stride = 2 * sg.nsub;
      bool liveA = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(A, stride, sg, hot, cold, p, srtab, ewtab, valid,
                                                   W, skipA, steps);
      bool liveB = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(B, stride, sg, hot, cold, p, srtab, ewtab, valid,
                                                   W, skipB, steps);
      while(liveA && liveB)
        {
          v16i HA, HB;
          d_load2(hot, A.e, HA, B.e, HB);
          steps += 2;
          A.e = __builtin_amdgcn_readfirstlane(
            d_walk_element<MODE, PERIODIC, UNEQUAL, true, REL>(A.e, HA, cold, p, srtab, ewtab, W, skipA));
          B.e = __builtin_amdgcn_readfirstlane(
            d_walk_element<MODE, PERIODIC, UNEQUAL, true, REL>(B.e, HB, cold, p, srtab, ewtab, W, skipB));
          if(A.e >= A.s1)
            {
              A.kseg += stride;
              liveA = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(A, stride, sg, hot, cold, p, srtab, ewtab,
                                                      valid, W, skipA, steps);
            }
          if(B.e >= B.s1)
            {
              B.kseg += stride;
              liveB = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(B, stride, sg, hot, cold, p, srtab, ewtab,
                                                      valid, W, skipB, steps);
            }
        }
      if(liveB)
        {
          A = B;
          skipA = skipB;
          liveA = true;
        }
      while(liveA)
        {
          v16i HA;
          d_load1(hot, A.e, HA);
          steps++;
          A.e = __builtin_amdgcn_readfirstlane(
            d_walk_element<MODE, PERIODIC, UNEQUAL, true, REL>(A.e, HA, cold, p, srtab, ewtab, W, skipA));
          if(A.e >= A.s1)
            {
              A.kseg += stride;
              liveA = d_enter_segment<MODE, PERIODIC, UNEQUAL, REL>(A, stride, sg, hot, cold, p, srtab, ewtab,
                                                      valid, W, skipA, steps);
            }
        }
    }

}

// partial results: [nsub][nt] per component.
// Grid: a multiple of 8 blocks; the block index is remapped so that each XCD (blocks are dealt
// round-robin over the 8 XCDs) works through ONE contiguous eighth of the buckets: neighbouring
// buckets read the same deep tree nodes, which then stay in that XCD's 4 MB L2.
template <int MODE, bool PERIODIC, bool UNEQUAL>
__global__ void __launch_bounds__(GHIP_BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == GHIP_WALK_EWALD ? 4 : GHIP_WALK_WAVES, 8)))
k_grav_walk(int nelem, const WalkHot *__restrict__ hot, const WalkCold *__restrict__ cold,
            WalkSeg sg, int nt, const int *__restrict__ tgt, const double *__restrict__ tx,
            const double *__restrict__ ty, const double *__restrict__ tz,
            const double *__restrict__ tsoft, const double *__restrict__ toldacc, GravK p,
            const float *__restrict__ srtab, const double *__restrict__ ewtab,
            double *__restrict__ pax, double *__restrict__ pay, double *__restrict__ paz,
            int *__restrict__ pcost, unsigned long long *__restrict__ counter,
            unsigned long long *__restrict__ acc, WalkPlan plan)
{
  const int lane = threadIdx.x & 63;
  // "every workgroup of this grid has been dispatched": from here on the kernel only drains, and the
  // wavefront slots it gives back stay free -- the main stream's hydro kernel waits for this word
  // (hipStreamWaitValue32 in ghip_hydro) instead of for the kernel's end
  if(plan.started && blockIdx.x == plan.started_at && threadIdx.x == 0)
    __hip_atomic_store(plan.started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // xcd_remap 1: XCD x walks the x-th eighth of the wavefront list; G > 1: the list is dealt to the XCDs
  // in chunks of G consecutive wavefronts (XCD x takes chunks x, x + 8, ...): the buckets resident on an
  // XCD are neighbours, the work is still spread evenly
  const int per_xcd = gridDim.x >> 3;
  int lblock = blockIdx.x;
  if(p.xcd_remap == 1)
    lblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  else if(p.xcd_remap > 1)
    {
      const int j = blockIdx.x >> 3, x = blockIdx.x & 7, G = p.xcd_remap;
      lblock = ((j / G) * 8 + x) * G + (j % G);
    }
  const int wave = __builtin_amdgcn_readfirstlane((lblock * (int) blockDim.x + threadIdx.x) >> 6);
  if(wave >= plan.nwaves)
    return;
  const int bucket = __builtin_amdgcn_readfirstlane(plan.wave_bucket[wave]);
  if(bucket < 0 || bucket * 64 >= nt)
    return;
  const int sub = wave - __builtin_amdgcn_readfirstlane(plan.woff[bucket]);
  sg.nsub = __builtin_amdgcn_readfirstlane(plan.nsub[bucket]);   // wavefronts sharing THIS bucket
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
  W.k15 = 1.5;
  asm volatile("" : "+v"(W.k15));   // (not rematerialised inside the loop: two moves per visit otherwise)
  unsigned int steps = 0;

  // two copies of the traversal, one per opening criterion (a uniform choice: no divergence)
  if(p.theta == 0)
    d_walk_run<MODE, PERIODIC, UNEQUAL, true>(sg, sub, hot, cold, p, srtab, ewtab, valid, W, steps);
  else
    d_walk_run<MODE, PERIODIC, UNEQUAL, false>(sg, sub, hot, cold, p, srtab, ewtab, valid, W, steps);

  if(valid)
    {
      const size_t o = (size_t) wave * 64 + lane;   // wave-major partials
      pax[o] = W.acc_x;
      pay[o] = W.acc_y;
      paz[o] = W.acc_z;
      pcost[o] = p.debug_steps ? (int) steps : W.nint;
    }
  unsigned long long tot = d_wave_sum_u64((unsigned long long) W.nint);
  // counter: this call's totals (reset by the next call of the kind); acc: the run's (ghip_run_begin),
  // kept by the kernel itself -- the next call's reset is enqueued on this walk's stream and may
  // overtake whatever the main stream would do with the counters at the end of a step
  // (64 slots each, ghip_count.h: no address that every wavefront of the launch hits)
  if(lane == 0 && tot && counter)
    {
      d_count(counter, 0, tot);
      d_count(acc, 0, tot);
    }
  if(lane == 0)
    {
      if(counter)
        {
          d_count(counter, 1, (unsigned long long) steps);
          d_count(acc, 1, (unsigned long long) steps);
        }
      atomicAdd(plan.steps_out + bucket, steps);   // cost model of the next call's plan
    }
}

// ---- adaptive split: how many wavefronts share each bucket ----------------------------------
// The previous call of the same walk recorded the elements visited per bucket.  A bucket gets
// floor(S*steps/mean)+1 wavefronts (2..64, never more than there are segments), so every
// wavefront walks about the same number of elements and the kernel has no long tail.  S is 8
// for a full-size launch and grows (up to one wavefront per segment) when there are few buckets
// -- one rank's share of a multi-GPU run -- so that the chip still sees ~50 000 wavefronts: a
// wavefront is a serial chain of dependent loads, only their number hides the latency.  The sum
// is bounded by S+2 per bucket on average, which is what the grid is sized for.
__global__ void k_plan_nsub(int nb, int ns, int sbase, const unsigned int *__restrict__ steps_prev,
                            int nprev, int *__restrict__ nsub)
{
  // nprev > 0: the previous call of the kind left the visits of its first nprev buckets (the
  // caller passes min(nb, its bucket count): a shard's particle number moves by a few buckets
  // from step to step, the curve order of the targets does not).  Their sum is taken here, by
  // every block again (<= 32 loads per thread), so that sum(nsub) <= (S+2) nb holds by
  // construction: buckets with history share S*nprev, the others get S.
  __shared__ unsigned long long part[4];
  unsigned long long tot = 0;
  if(nprev > 0)
    {
      for(int k = threadIdx.x; k < nprev; k += blockDim.x)
        tot += steps_prev[k];
      for(int o = 32; o > 0; o >>= 1)
        tot += __shfl_xor(tot, o);
      if((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = tot;
      __syncthreads();
      tot = part[0] + part[1] + part[2] + part[3];
    }
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= nb)
    return;
  int s = ns < sbase ? ns : sbase;
  if(b < nprev && tot > 0)
    {
      double mean = (double) tot / (double) nprev;
      s = (int) ((double) sbase * (double) steps_prev[b] / mean) + 1;
      s = s < 2 ? 2 : s;
      s = s > 64 ? 64 : s;
      if(s > ns)
        s = ns;
    }
  nsub[b] = s < 1 ? 1 : s;
}

// The grid and the partial-sum buffers are sized for maxwaves = (S+2)*nb + 8 wavefronts, which
// bounds sum(nsub) because k_plan_nsub normalises with the sum of the very steps_prev[] entries it uses.
// Should that invariant ever break, wavefronts beyond maxwaves would
// not exist while k_combine_grav still summed their slots: the plan is therefore checked here and
// an overflow is a hard error (GHIP_E_PLAN in the context's device error word, reported by the
// next synchronising entry point) instead of silently wrong forces.
__global__ void k_plan_fill(int nb, const int *__restrict__ nsub, const int *__restrict__ woff,
                            int maxwaves, int *__restrict__ wave_bucket,
                            unsigned int *__restrict__ steps_out, int *__restrict__ errword)
{
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= nb)
    return;
  int o = woff[b], s = nsub[b];
  if(o + s > maxwaves)
    *(volatile int *) errword = 1;
  for(int q = 0; q < s; q++)
    if(o + q < maxwaves)
      wave_bucket[o + q] = b;
  steps_out[b] = 0;
}

// segment tables of a tree: equal-length slices of the element list + the ancestor chain of each
// slice's first element (one thread per segment; a pre-order descent from the root).  Up to three
// tables of different granularity are built by one launch; table j has ns[j] segments and lives at
// start + soff[j] (ns[j] + 1 entries), nanc + noff[j], anc + noff[j] * GHIP_MAXANC.
struct SegTables
{
  int ntab;
  int ns[3], soff[3], noff[3];
};

__global__ void k_build_segments(const TreeSizes *__restrict__ ts, const int4 *__restrict__ lk,
                                 SegTables T, int *__restrict__ start, int *__restrict__ nanc,
                                 int *__restrict__ anc)
{
  const int nelem = ts->nelem;   // (0 for an unusable tree: every segment is empty then)
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  int j = 0;
  while(j < T.ntab && g > T.ns[j])
    {
      g -= T.ns[j] + 1;
      j++;
    }
  if(j >= T.ntab)
    return;
  const int ns = T.ns[j], k = g;
  start += T.soff[j];
  nanc += T.noff[j];
  anc += (size_t) T.noff[j] * GHIP_MAXANC;
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
__global__ void k_combine_grav(int nt, WalkPlan plan, const int *__restrict__ tgt,
                               const int *__restrict__ perm, const double *__restrict__ pax,
                               const double *__restrict__ pay, const double *__restrict__ paz,
                               const int *__restrict__ pcost, int n, double *__restrict__ oacc,
                               int *__restrict__ ocost, int accumulate, int debug_max)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  const int bucket = ti >> 6, lane = ti & 63;
  const int w0 = plan.woff[bucket];
  int nsub = plan.nsub[bucket];
  if(w0 + nsub > plan.nwaves)   // plan overflow (reported by k_plan_fill): stay inside the buffers
    nsub = plan.nwaves > w0 ? plan.nwaves - w0 : 0;
  double a0 = 0, a1 = 0, a2 = 0;
  int c = 0;
  for(int s = 0; s < nsub; s++)
    {
      size_t o = (size_t) (w0 + s) * 64 + lane;
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

// The same for a Newton + Ewald pair in one pass: GravAccel = (Newtonian sum) + (Ewald sum), as the
// two calls leave it (the second one adds its sum to the first one's, forcetree.c:3190-3193) -- one
// launch behind the later of the two walks instead of two in a row.
__global__ void k_combine_pair(int nt, WalkPlan pn, WalkPlan pe, const int *__restrict__ tgt,
                               const int *__restrict__ perm, const double *__restrict__ nax,
                               const double *__restrict__ nay, const double *__restrict__ naz,
                               const int *__restrict__ ncost, const double *__restrict__ eax,
                               const double *__restrict__ eay, const double *__restrict__ eaz,
                               const int *__restrict__ ecost, int n, double *__restrict__ oacc,
                               int *__restrict__ ocost)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  const int bucket = ti >> 6, lane = ti & 63;
  double a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
  int c = 0, d = 0;
  {
    const int w0 = pn.woff[bucket];
    int nsub = pn.nsub[bucket];
    if(w0 + nsub > pn.nwaves)
      nsub = pn.nwaves > w0 ? pn.nwaves - w0 : 0;
    for(int s = 0; s < nsub; s++)
      {
        size_t o = (size_t) (w0 + s) * 64 + lane;
        a0 += nax[o];
        a1 += nay[o];
        a2 += naz[o];
        c += ncost[o];
      }
  }
  {
    const int w0 = pe.woff[bucket];
    int nsub = pe.nsub[bucket];
    if(w0 + nsub > pe.nwaves)
      nsub = pe.nwaves > w0 ? pe.nwaves - w0 : 0;
    for(int s = 0; s < nsub; s++)
      {
        size_t o = (size_t) (w0 + s) * 64 + lane;
        b0 += eax[o];
        b1 += eay[o];
        b2 += eaz[o];
        d += ecost[o];
      }
  }
  int i = perm ? perm[tgt[ti]] : ti;
  oacc[i] = a0 + b0;
  oacc[(size_t) n + i] = a1 + b1;
  oacc[2 * (size_t) n + i] = a2 + b2;
  ocost[i] = c + d;
}

// the walk's hot/cold element records from the tree arrays (see WalkHot / WalkCold)
__global__ void k_fill_elems(const TreeSizes *__restrict__ ts, const double4 *__restrict__ xm,
                             const double4 *__restrict__ cl, const int4 *__restrict__ lk,
                             const double *__restrict__ aux, WalkHot *__restrict__ hot,
                             WalkCold *__restrict__ cold)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= ts->nelem)
    return;
  double4 v = xm[e], c = cl[e];
  int4 k = lk[e];
  WalkHot h;
  h.x = v.x;
  h.y = v.y;
  h.z = v.z;
  h.m = v.w;
  WalkCold q;
  q.cx = c.x;
  q.cy = c.y;
  q.cz = c.z;
  q.len = c.w;
  if(LK_IS_PARTICLE(k))
    {
      h.mlen2 = 0.0;
      h.len2 = 0.0;
      q.len06 = 0.0;
    }
  else
    {
      h.mlen2 = v.w * c.w * c.w;   // (mass*len)*len, forcetree.c:2085
      h.len2 = c.w * c.w;
      q.len06 = 0.60 * c.w;
    }
  h.skip = k.x;
  h.pidx = (LK_IS_PARTICLE(k) ? 0 : (int) 0x80000000) | (v.w > 0 ? 1 : 0);   // flags: node, mass > 0
  h.aux = aux[e];
  q.spare[0] = q.spare[1] = q.spare[2] = 0.0;
  hot[e] = h;
  cold[e] = q;
}
