// ghip_sph.hip -- SPH density and hydro-force neighbour sums on gfx950.
//
// Replaces density()/density_evaluate() (density.c:89-704, 711-1029), hydro_force()/
// hydro_evaluate() (hydra.c:145-813, 822-1995) and the range searches they call,
// ngb_treefind_variable()/ngb_treefind_pairs() (ngb.c:169-297, 32-160).
//
// The reference first collects a neighbour list per target (a pointer-chasing range search) and
// then loops over it.  Here one wavefront serves a bucket of 64 targets that are consecutive
// along the space-filling curve and walks the GAS tree's pre-order element list with a
// wave-uniform index: a node is descended if ANY lane's search sphere overlaps it (the
// reference's own node test, per lane).  Node records (64 B) arrive through the scalar path; the
// candidate gas records of a node with <= 64 particles are fetched with one coalesced vector
// load into LDS and read back as broadcast ds_reads (one memory latency per 64 candidates), and
// each lane applies the reference's exact acceptance test (r2 < h_i^2, or r2 < h_i^2 ||
// r2 < h_j^2 for pairs) before accumulating.  When the target list is short, several wavefronts
// share a bucket, each taking every n-th candidate batch; their partial sums are added in fixed
// order.  Neighbour SETS are geometric, so they are identical to the reference's; only the
// summation order differs.

#include "ghip_internal.h"

// allvars.h:247-253
#define KERNEL_COEFF_1 2.546479089470
#define KERNEL_COEFF_2 15.278874536822
#define KERNEL_COEFF_3 45.836623610466
#define KERNEL_COEFF_4 30.557749073644
#define KERNEL_COEFF_5 5.092958178941
#define KERNEL_COEFF_6 (-15.278874536822)
#define NORM_COEFF 4.188790204786
#define NUMDIMS 3
#define GAMMA (7. / 5.)  // allvars.h:64 (this fork: 7/5, not 5/3)
#define GAMMA_MINUS1 (GAMMA - 1)
#define FACT1 0.366025403785  // allvars.h:310

struct BoxK
{
  double boxsize, boxhalf;
  int periodic;
};

// node test of ngb_treefind_* (ngb.c:276-289 / 136-151): true if the node must be opened
__device__ __forceinline__ bool d_node_overlaps(const double4 c, double dist, double px, double py,
                                                double pz, const BoxK b)
{
  const double len = c.w;
  dist += 0.5 * len;
  double dx = d_ngb_periodic(c.x - px, b.periodic, b.boxsize, b.boxhalf);
  if(dx > dist)
    return false;
  double dy = d_ngb_periodic(c.y - py, b.periodic, b.boxsize, b.boxhalf);
  if(dy > dist)
    return false;
  double dz = d_ngb_periodic(c.z - pz, b.periodic, b.boxsize, b.boxhalf);
  if(dz > dist)
    return false;
  dist += FACT1 * len;
  return !(dx * dx + dy * dy + dz * dz > dist * dist);
}

__device__ __forceinline__ double d_wrap(double d, const BoxK b)
{
  // density.c:838-851 / hydra.c:1251-1264: d > boxhalf -> d - box, d < -boxhalf -> d + box.
  // Written as one compare on |d| and a subtraction of copysign(box, d) under the execution mask
  // (the same IEEE operations; 2 vector instructions per axis in the common no-wrap case, the
  // empty asm keeps the compiler from turning it back into compare+select chains)
  if(b.periodic)
    {
      if(fabs(d) > b.boxhalf)
        {
          d -= copysign(b.boxsize, d);
          asm volatile("" : "+v"(d));
        }
    }
  return d;
}

// Bounds of a bucket's targets around the first target: half extents per axis (nearest-image
// offsets) and the largest search radius.  A staged candidate can be a neighbour of SOME lane only
// if |wrap(x_j - c)| <= extent + radius on every axis (triangle inequality of the periodic
// distance), so the lane that stages candidate j tests it once against the bucket, and the 64-lane
// loop runs over the survivors only -- typically 60 % of a leaf's particles.
struct BucketBox
{
  double cx, cy, cz, ex, ey, ez, hmax;
};

__device__ __forceinline__ double d_wave_max_f64(double v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      double o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
  return v;
}

__device__ __forceinline__ double d_first_lane_f64(double v)
{
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ BucketBox d_bucket_box(bool valid, double px, double py, double pz,
                                                  double h, const BoxK b)
{
  BucketBox B;
  B.cx = d_first_lane_f64(px);   // lane 0 of a bucket always holds a target
  B.cy = d_first_lane_f64(py);
  B.cz = d_first_lane_f64(pz);
  B.ex = d_wave_max_f64(valid ? fabs(d_wrap(px - B.cx, b)) : 0.0);
  B.ey = d_wave_max_f64(valid ? fabs(d_wrap(py - B.cy, b)) : 0.0);
  B.ez = d_wave_max_f64(valid ? fabs(d_wrap(pz - B.cz, b)) : 0.0);
  B.hmax = d_wave_max_f64(valid ? h : 0.0);
  return B;
}

// mask of the staged candidates (lane l holds candidate l) that can be within max(hmax, hj) of
// any target of the bucket
// (jm < 0: a record of the gas block that is not gas any more -- a particle converted since the last
// rearrange_particle_sequence(), marked by k_mark_converted -- is nobody's neighbour, ngb.c:213, 93)
__device__ __forceinline__ unsigned long long d_cull_batch(const BucketBox &B, bool staged, double jx,
                                                            double jy, double jz, double jm, double hj,
                                                            const BoxK b)
{
  const double H = B.hmax > hj ? B.hmax : hj;
  bool pass = staged && jm >= 0 && fabs(d_wrap(jx - B.cx, b)) <= B.ex + H &&
              fabs(d_wrap(jy - B.cy, b)) <= B.ey + H && fabs(d_wrap(jz - B.cz, b)) <= B.ez + H;
  return __builtin_amdgcn_ballot_w64(pass);
}

// ---------------------------------------------------------------------------------------------
// density
// ---------------------------------------------------------------------------------------------
// 1/sqrt(x) to full fp64 precision from the hardware seed (as the gravity walk's d_rsqrt)
__device__ __forceinline__ double d_rsqrt_sph(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double t = x * y;
  double e = fma(-t, y, 1.0);
  double p = fma(0.375, e, 0.5);
  return fma(y * e, p, y);
}

struct DensAcc
{
  double rho, wnum, dhsml, divv, rx, ry, rz;
  int nn;
};

// r8: the candidate's record (x,y,z,m,vx,vy,vz,h) -- a wave-uniform global address (scalar
// loads) or a slot of the LDS staging area (broadcast ds_read)
__device__ __forceinline__ void d_density_pair(const double *r8, bool valid,
                                               double px, double py, double pz, double vx,
                                               double vy, double vz, double h2, double hinv,
                                               double hinv3, double hinv4, double h3, const BoxK b,
                                               DensAcc &A)
{
  const double jx = r8[0], jy = r8[1], jz = r8[2], mass_j = r8[3];
  const double jvx = r8[4], jvy = r8[5], jvz = r8[6];
  double dx = d_wrap(px - jx, b), dy = d_wrap(py - jy, b), dz = d_wrap(pz - jz, b);
  double r2 = dx * dx + dy * dy + dz * dz;
  if(valid && r2 < h2)
    {
      A.nn++;
      // r and 1/r from one reciprocal square root (the target itself is among the candidates: r = 0);
      // the reference's sqrt and its divisions by r and by hinv3 become multiplications
      const double rinv = r2 > 0 ? d_rsqrt_sph(r2) : 0.0;
      const double r = r2 * rinv;
      double u = r * hinv, wk, dwk;
      if(u < 0.5)
        {
          wk = hinv3 * (KERNEL_COEFF_1 + KERNEL_COEFF_2 * (u - 1) * u * u);
          dwk = hinv4 * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4);
        }
      else
        {
          wk = hinv3 * KERNEL_COEFF_5 * (1.0 - u) * (1.0 - u) * (1.0 - u);
          dwk = hinv4 * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
        }
      A.rho += mass_j * wk;
      A.wnum += NORM_COEFF * wk * h3;
      A.dhsml += -mass_j * (NUMDIMS * hinv * wk + u * dwk);
      if(r > 0)
        {
          double fac = mass_j * dwk * rinv;
          double dvx = vx - jvx, dvy = vy - jvy, dvz = vz - jvz;
          A.divv += -fac * (dx * dvx + dy * dvy + dz * dvz);
          A.rx += fac * (dz * dvy - dy * dvz);
          A.ry += fac * (dx * dvz - dz * dvx);
          A.rz += fac * (dy * dvx - dx * dvy);
        }
    }
}

// One element of the gas tree as the SPH walks read it (64 B, one s_load_dwordx16):
//   cx, cy, cz, len | hmax | skip, pidx, pstart, pcount | pad
struct __attribute__((aligned(64))) SphNode
{
  double cx, cy, cz, len;
  double hmax;
  int skip, pidx, pstart, pcount;
  int pad[2];
};
typedef int v16i_s __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double d_f64s(const v16i_s &v, int i)
{
  return __hiloint2double(v[2 * i + 1], v[2 * i]);
}

__device__ __forceinline__ void d_load_sphnode(const SphNode *__restrict__ base, int e, v16i_s &R)
{
  unsigned long long a = reinterpret_cast<unsigned long long>(base + e);
  unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int) a);
  unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int) (a >> 32));
  const SphNode *p = reinterpret_cast<const SphNode *>(((unsigned long long) hi << 32) | lo);
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(R) : "s"(p) : "memory");
}

#define SPH_STAGE 64   // candidates staged through LDS per round (= lanes of the staging load)
#ifndef SPH_LEAF
#define SPH_LEAF 256   // a node with at most this many particles is swept flat, SPH_STAGE at a time
#endif


// up to CS staged candidates of one batch at a time: candidate slot c takes the c-th lowest set bit
// of `live` (wave-uniform, scalar unit); a lane of slot cs reads ITS candidate from LDS.  Returns the
// lane's candidate index or -1.
template <int CS>
__device__ __forceinline__ int d_next_candidates(unsigned long long &live, int cs)
{
  int mine = -1;
#pragma unroll
  for(int c = 0; c < CS; c++)
    {
      const int j = live ? __builtin_ctzll(live) : -1;
      live &= live - 1;   // (0 & anything = 0)
      if(CS == 1 || cs == c)
        mine = j;
    }
  return mine;
}

// sum over the candidate slots of a target (lanes t, t + TG, t + 2 TG, ...): fixed order
template <int TG>
__device__ __forceinline__ double d_slot_sum(double v)
{
#pragma unroll
  for(int off = TG; off < 64; off <<= 1)
    v += __shfl_xor(v, off, 64);
  return v;
}
template <int TG>
__device__ __forceinline__ double d_slot_max(double v)
{
#pragma unroll
  for(int off = TG; off < 64; off <<= 1)
    {
      const double o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
  return v;
}

// density_evaluate (density.c:711-1029, mode 0) for a bucket of TG curve-consecutive targets.
// One wavefront per workgroup = TG targets x CS = 64/TG candidate slots: lane l works for target
// l % TG on the candidates of slot l / TG.  Node records come through the scalar path; when a node
// with <= SPH_LEAF gas particles overlaps any target's search sphere its particle records
// (contiguous in gas-tree order) are fetched 64 at a time with ONE coalesced vector load -- lane l
// loads candidate l -- into LDS and culled against the bucket's bounds; the survivors are then taken
// CS at a time, slot c testing survivor c against its TG targets with the exact test r2 < h_i^2
// (a per-lane ds_read, broadcast within a slot).  A smaller bucket has a tighter bound (fewer
// survivors per target) and CS survivors share one pass of the pair arithmetic: at TG = 16 a
// candidate test accepts 4 x as often as with 64 targets per wavefront.  The slots' partial sums are
// added in fixed order at the end.
// Workgroups are dealt to the 8 XCDs in turn (workgroup b runs on XCD b % 8), so with the plain order
// every XCD's resident buckets are spread over the whole curve and its 4 MB L2 sees the whole gas set
// (hydro at c2: 1.5 GB of fabric traffic per launch for 1.0 GB of algorithmic bytes, L2 hit rate 46 %).
// With xcd != 0 XCD x walks the x-th eighth of the buckets in order: logical index = (b % 8) * per + b / 8
// (the grid is padded to a multiple of 8; indices past the end do nothing).
__device__ __forceinline__ int d_sph_block(int xcd, int nblocks)
{
  const int b = blockIdx.x;
  if(!xcd)
    return b < nblocks ? b : -1;
  const int per = (nblocks + 7) >> 3;
  const int lb = (b & 7) * per + (b >> 3);
  return lb < nblocks ? lb : -1;
}

template <int TG>
__global__ void __launch_bounds__(64)
k_density(const TreeSizes *__restrict__ ts, const SphNode *__restrict__ nodes, const double *__restrict__ gp, int nt,
          int nsub, const int *__restrict__ tgt, const double *__restrict__ hcur, BoxK b,
          double *__restrict__ prho, double *__restrict__ pnum, double *__restrict__ pdh,
          double *__restrict__ pdiv, double *__restrict__ prot,
          unsigned long long *__restrict__ counter, unsigned long long *__restrict__ racc, int xcd,
          int nblocks)
{
  constexpr int CS = 64 / TG;
  __shared__ double4 sh[SPH_STAGE][2];
  const int lane = threadIdx.x;
  const int tl = lane & (TG - 1), cs = lane / TG;
  const int lb = d_sph_block(xcd, nblocks);
  if(lb < 0)
    return;
  const int bucket = lb / nsub;
  const int sub = lb - bucket * nsub;   // this wavefront takes every nsub-th batch
  int batch = 0;
  const int ti = bucket * TG + tl;
  const bool valid = ti < nt;
  const int s = valid ? tgt[ti] : 0;
  double px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0, h = 1;
  if(valid)
    {
      const double *r8 = gp + (size_t) 8 * s;
      px = r8[0];
      py = r8[1];
      pz = r8[2];
      vx = r8[4];
      vy = r8[5];
      vz = r8[6];
      h = hcur[s];
    }
  const double h2 = h * h, hinv = 1.0 / h;
  const double hinv3 = hinv * hinv * hinv, hinv4 = hinv3 * hinv, h3 = h * h * h;
  DensAcc A = {0, 0, 0, 0, 0, 0, 0, 0};
  const BucketBox BB = d_bucket_box(valid, px, py, pz, h, b);

  const int nelem = __builtin_amdgcn_readfirstlane(ts->nelem);   // the gas tree's elements, as the device counted them
  int e = 0;
  while(e < nelem)
    {
      e = __builtin_amdgcn_readfirstlane(e);
      v16i_s N;
      d_load_sphnode(nodes, e, N);
      const int skip = N[10], pidx = N[11], pstart = N[12], pcount = N[13];
      if(pidx >= 0)
        {
          if((batch++ % nsub) == sub && gp[(size_t) 8 * pidx + 3] >= 0)
            d_density_pair(gp + (size_t) 8 * pidx, valid && cs == 0, px, py, pz, vx, vy, vz, h2, hinv,
                           hinv3, hinv4, h3, b, A);
          e = e + 1;
        }
      else
        {
          const double4 c = make_double4(d_f64s(N, 0), d_f64s(N, 1), d_f64s(N, 2), d_f64s(N, 3));
          bool open = valid && d_node_overlaps(c, h, px, py, pz, b);
          if(__any(open))
            {
              if(pcount <= SPH_LEAF)
                {
                  for(int r0 = 0; r0 < pcount; r0 += SPH_STAGE)
                    if((batch++ % nsub) == sub)
                      {
                        const bool staged = r0 + lane < pcount;
                        __syncthreads();   // previous round fully consumed
                        double4 c0 = make_double4(0, 0, 0, 0);
                        if(staged)
                          {
                            const double4 *src = reinterpret_cast<const double4 *>(
                              gp + (size_t) 8 * (pstart + r0 + lane));
                            c0 = src[0];
                            sh[lane][0] = c0;
                            sh[lane][1] = src[1];
                          }
                        unsigned long long live = d_cull_batch(BB, staged, c0.x, c0.y, c0.z, c0.w, 0.0, b);
                        __syncthreads();
                        while(live)
                          {
                            const int j = d_next_candidates<CS>(live, cs);
                            d_density_pair(reinterpret_cast<const double *>(&sh[j < 0 ? 0 : j][0]),
                                           valid && j >= 0, px, py, pz, vx, vy, vz, h2, hinv, hinv3,
                                           hinv4, h3, b, A);
                          }
                      }
                  e = skip;
                }
              else
                e = e + 1;
            }
          else
            e = skip;
        }
    }
  const double rho = d_slot_sum<TG>(A.rho), wnum = d_slot_sum<TG>(A.wnum), dh = d_slot_sum<TG>(A.dhsml);
  const double dv = d_slot_sum<TG>(A.divv), rx = d_slot_sum<TG>(A.rx), ry = d_slot_sum<TG>(A.ry);
  const double rz = d_slot_sum<TG>(A.rz);
  if(valid && cs == 0)
    {
      const size_t o = (size_t) sub * nt + ti;
      const size_t plane = (size_t) nsub * nt;
      prho[o] = rho;
      pnum[o] = wnum;
      pdh[o] = dh;
      pdiv[o] = dv;
      prot[o] = rx;
      prot[plane + o] = ry;
      prot[2 * plane + o] = rz;
    }
  unsigned long long tot = d_wave_sum_u64((unsigned long long) A.nn);
  if(lane == 0 && tot)
    {
      d_count(counter, 0, tot);
      d_count(racc, 0, tot);
    }
}

// SphNode records from the gas tree's arrays; k_sph_nodes_hmax refreshes only hmax
__global__ void k_fill_sph_nodes(const TreeSizes *__restrict__ ts, const double4 *__restrict__ cl,
                                 const int4 *__restrict__ lk, const double *__restrict__ aux,
                                 SphNode *__restrict__ out)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= ts->nelem)
    return;
  double4 c = cl[e];
  int4 k = lk[e];
  SphNode r;
  r.cx = c.x;
  r.cy = c.y;
  r.cz = c.z;
  r.len = c.w;
  r.hmax = aux[e];
  r.skip = k.x;
  r.pidx = k.y;
  r.pstart = k.z;
  r.pcount = k.w;
  r.pad[0] = r.pad[1] = 0;
  out[e] = r;
}

__global__ void k_sph_nodes_hmax(const TreeSizes *__restrict__ ts, const double *__restrict__ aux,
                                 SphNode *__restrict__ nodes)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e < ts->nelem)
    nodes[e].hmax = aux[e];
}

int ghip_sph_fill_nodes(ghip_ctx *ctx, bool hmax_only)
{
  TreeDev &t = ctx->st;
  if(t.n == 0)
    return GHIP_OK;
  const int ce = t.n + t.cap_nodes;   // elements the buffers hold; the kernels read the count on the device
  const TreeSizes *ts = P<TreeSizes>(t.dsz);
  GCHK(ghip_ensure(ctx, t.mq, (size_t) (ce + 1) * sizeof(SphNode)));
  if(hmax_only)
    k_sph_nodes_hmax<<<cdiv(ce, ghip_wg(ctx)), ghip_wg(ctx), 0, ctx->stream>>>(ts, P<double>(t.aux),
                                                                             P<SphNode>(t.mq));
  else
    k_fill_sph_nodes<<<cdiv(ce, ghip_wg(ctx)), ghip_wg(ctx), 0, ctx->stream>>>(
      ts, P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux), P<SphNode>(t.mq));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

__global__ void k_dens_init(int nt, const int *__restrict__ tgt, const double *__restrict__ gp,
                            double *__restrict__ hcur, double *__restrict__ left,
                            double *__restrict__ right)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int s = tgt[ti];
  hcur[s] = gp[(size_t) 8 * s + 7];
  left[s] = 0;   // density.c:127
  right[s] = 0;
}

struct DensFin
{
  double desnumngb, maxdev, minhsml;
  int ti_current;
  double timebase;
};

// finalisation + smoothing-length update, density.c:434-652, one thread per evaluated target
__global__ void k_dens_finalize(int nt, int nsub, const int *__restrict__ tgt,
                                const int *__restrict__ perm, int n, int ngas, DensFin P,
                                const double *__restrict__ srho,
                                const double *__restrict__ snum, const double *__restrict__ sdh,
                                const double *__restrict__ sdiv, const double *__restrict__ srot,
                                double *__restrict__ hcur, double *__restrict__ left,
                                double *__restrict__ right, const double *__restrict__ entropy,
                                const double *__restrict__ dtentropy,
                                const int *__restrict__ timebin, const int *__restrict__ tibeg,
                                double *__restrict__ gp, double *__restrict__ gq,
                                double *__restrict__ o_hsml, double *__restrict__ o_numngb,
                                double *__restrict__ o_density, double *__restrict__ o_dhf,
                                double *__restrict__ o_divv, double *__restrict__ o_curl,
                                double *__restrict__ o_pres, int *__restrict__ redo)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  const int s = tgt[ti];
  const int i = perm[s];
  double h = hcur[s];
  double rho = 0, numngb = 0, dhf = 0, divv = 0, r0 = 0, r1 = 0, r2 = 0;
  const size_t plane = (size_t) nsub * nt;
  for(int q = 0; q < nsub; q++)   // fixed order: deterministic
    {
      const size_t o = (size_t) q * nt + ti;
      rho += srho[o];
      numngb += snum[o];
      dhf += sdh[o];
      divv += sdiv[o];
      r0 += srot[o];
      r1 += srot[plane + o];
      r2 += srot[2 * plane + o];
    }
  double curl = 0;
  if(rho > 0)
    {
      dhf *= h / (NUMDIMS * rho);
      if(dhf > -0.9)
        dhf = 1 / (1 + dhf);
      else
        dhf = 1;
      curl = sqrt(r0 * r0 + r1 * r1 + r2 * r2) / rho;
      divv /= rho;
    }
  int tb = timebin[i];
  int dt_step = (tb ? (1 << tb) : 0);
  double dt_entr = (P.ti_current - (tibeg[i] + dt_step / 2)) * P.timebase;
  double pres = (entropy[i] + dtentropy[i] * dt_entr) * pow(rho, GAMMA);

  o_numngb[i] = numngb;
  o_density[i] = rho;
  o_dhf[i] = dhf;
  o_divv[i] = divv;
  o_curl[i] = curl;
  o_pres[i] = pres;
  double *q = gq + (size_t) 8 * s;
  q[0] = pres;
  q[1] = rho;
  q[2] = dhf;
  q[3] = divv;
  q[4] = curl;

  int again = 0;
  double hnew = h;
  if(numngb < (P.desnumngb - P.maxdev) ||
     (numngb > (P.desnumngb + P.maxdev) && h > (1.01 * P.minhsml)))
    {
      again = 1;
      double L = left[s], R = right[s];
      if(L > 0 && R > 0 && (R - L) < 1.0e-3 * L)
        again = 0;  // density.c:566-573 "this one should be ok"
      else
        {
          if(numngb < (P.desnumngb - P.maxdev))
            L = (h > L) ? h : L;
          else
            {
              if(R != 0)
                {
                  if(h < R)
                    R = h;
                }
              else
                R = h;
            }
          if(R > 0 && L > 0)
            hnew = pow(0.5 * (pow(L, 3) + pow(R, 3)), 1.0 / 3);
          else
            {
              if(R == 0 && L > 0)
                {
                  if(fabs(numngb - P.desnumngb) < 0.5 * P.desnumngb)
                    {
                      double fac = 1 - (numngb - P.desnumngb) / (NUMDIMS * numngb) * dhf;
                      hnew = (fac < 1.26) ? h * fac : h * 1.26;
                    }
                  else
                    hnew = h * 1.26;
                }
              if(R > 0 && L == 0)
                {
                  if(fabs(numngb - P.desnumngb) < 0.5 * P.desnumngb)
                    {
                      double fac = 1 - (numngb - P.desnumngb) / (NUMDIMS * numngb) * dhf;
                      hnew = (fac > 1 / 1.26) ? h * fac : h / 1.26;
                    }
                  else
                    hnew = h / 1.26;
                }
            }
          if(hnew < P.minhsml)
            hnew = P.minhsml;
          left[s] = L;
          right[s] = R;
        }
    }
  hcur[s] = hnew;
  gp[(size_t) 8 * s + 7] = hnew;
  o_hsml[i] = hnew;
  redo[ti] = again;
}

// workgroups a density / hydro launch should at least consist of (buckets are shared by up to
// GHIP_MAXSUB workgroups to get there)
static int ghip_sph_target_waves()
{
  static int v = -1;
  if(v < 0)
    {
      v = 8192;
      if(getenv("GHIP_SPH_WAVES") && atoi(getenv("GHIP_SPH_WAVES")) > 0)
        v = atoi(getenv("GHIP_SPH_WAVES"));
    }
  return v;
}

// targets per wavefront of the SPH kernels (the other 64/TG lanes of a target are candidate slots)
static int ghip_sph_tg()
{
  static int v = -1;
  if(v < 0)
    {
      v = 16;
      const char *e = getenv("GHIP_SPH_TG");
      if(e && (atoi(e) == 64 || atoi(e) == 32 || atoi(e) == 16 || atoi(e) == 8))
        v = atoi(e);
    }
  return v;
}

#define SPH_LAUNCH(KERNEL, TGV, GRID, ST, ...)                                   \
  do                                                                             \
    {                                                                            \
      if((TGV) == 64)                                                            \
        KERNEL<64><<<(GRID), 64, 0, (ST)>>>(__VA_ARGS__);                        \
      else if((TGV) == 32)                                                       \
        KERNEL<32><<<(GRID), 64, 0, (ST)>>>(__VA_ARGS__);                        \
      else if((TGV) == 16)                                                       \
        KERNEL<16><<<(GRID), 64, 0, (ST)>>>(__VA_ARGS__);                        \
      else                                                                       \
        KERNEL<8><<<(GRID), 64, 0, (ST)>>>(__VA_ARGS__);                         \
    }                                                                            \
  while(0)

// XCD-contiguous bucket order of the SPH kernels (d_sph_block); GHIP_SPH_XCD=0 selects the plain order
static int ghip_sph_xcd()
{
  static int v = -1;
  if(v < 0)
    v = getenv("GHIP_SPH_XCD") ? atoi(getenv("GHIP_SPH_XCD")) : 1;
  return v;
}

static void shard_slice(const ghip_ctx *ctx, int nt, int *lo, int *cnt)
{
  int per;
  ghip_shard_range(nt, ctx->shard_n, ctx->shard_rank, lo, cnt, &per);
}

static int dens_alloc(ghip_ctx *ctx)
{
  // (arrays indexed by gas-tree position: a multi-GPU shard's gas tree also holds its ghosts)
  size_t ng = (size_t) (ctx->ngas > 0 ? ctx->ngas : 1);
  if((size_t) ctx->st.n > ng)
    ng = (size_t) ctx->st.n;
  GCHK(ghip_ensure(ctx, ctx->dleft, ng * 8));
  GCHK(ghip_ensure(ctx, ctx->dright, ng * 8));
  // per-wavefront partial sums: [nsub][targets] planes
  size_t np_ = ng * GHIP_MAXSUB;
  if(getenv("GHIP_WALK_SUBS"))
    np_ = ng * 64;
  GCHK(ghip_ensure(ctx, ctx->drho, np_ * 8));
  GCHK(ghip_ensure(ctx, ctx->dnumngb, np_ * 8));
  GCHK(ghip_ensure(ctx, ctx->ddhsml, np_ * 8));
  GCHK(ghip_ensure(ctx, ctx->ddivv, np_ * 8));
  GCHK(ghip_ensure(ctx, ctx->drot, np_ * 8 * 3));
  GCHK(ghip_ensure(ctx, ctx->dhcur, ng * 8));
  GCHK(ghip_ensure(ctx, ctx->dflags, ng * 4));
  GCHK(ghip_ensure(ctx, ctx->dtgt_a, ng * 4 + 16));
  GCHK(ghip_ensure(ctx, ctx->dtgt_b, ng * 4 + 16));
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  return GHIP_OK;
}

// Order-preserving selection of the flagged entries of the target list (the not yet converged
// particles of an h-iteration, density.c:566-660).  Three small launches of one-wavefront
// workgroups instead of a library select: its single-pass look-back kernel needs 256-thread
// workgroups to make progress in order, and underneath a gravity pair -- where the density
// iterations normally run -- they wait milliseconds for four free wave slots on one CU.
#define SEL_ITEMS 16
#define SEL_BLOCK (64 * SEL_ITEMS)

__global__ void __launch_bounds__(64) k_sel_count(int n, const int *__restrict__ flags,
                                                  int *__restrict__ blockcnt)
{
  const int base = blockIdx.x * SEL_BLOCK;
  int c = 0;
  for(int j = 0; j < SEL_ITEMS; j++)
    {
      const int a = base + j * 64 + threadIdx.x;
      c += (a < n && flags[a] != 0) ? 1 : 0;
    }
  for(int o = 32; o > 0; o >>= 1)
    c += __shfl_down(c, o, 64);
  if(threadIdx.x == 0)
    blockcnt[blockIdx.x] = c;
}

// counts -> exclusive offsets (in place), total to *total; one wavefront
__global__ void __launch_bounds__(64) k_sel_scan(int nblk, int *__restrict__ blockcnt,
                                                 int *__restrict__ total)
{
  int run = 0;
  for(int b0 = 0; b0 < nblk; b0 += 64)
    {
      const int b = b0 + threadIdx.x;
      const int c = b < nblk ? blockcnt[b] : 0;
      int incl = c;
      for(int o = 1; o < 64; o <<= 1)
        {
          const int v = __shfl_up(incl, o, 64);
          if((int) threadIdx.x >= o)
            incl += v;
        }
      if(b < nblk)
        blockcnt[b] = run + incl - c;
      run += __shfl(incl, 63, 64);
    }
  if(threadIdx.x == 0)
    *total = run;
}

__global__ void __launch_bounds__(64) k_sel_scatter(int n, const int *__restrict__ in,
                                                    const int *__restrict__ flags,
                                                    const int *__restrict__ blockoff,
                                                    int *__restrict__ out)
{
  const int base = blockIdx.x * SEL_BLOCK;
  int off = blockoff[blockIdx.x];
  const unsigned long long below = (1ull << threadIdx.x) - 1ull;
  for(int j = 0; j < SEL_ITEMS; j++)
    {
      const int a = base + j * 64 + threadIdx.x;
      const bool keep = a < n && flags[a] != 0;
      const unsigned long long m = __ballot(keep);
      if(keep)
        out[off + __popcll(m & below)] = in[a];
      off += __popcll(m);
    }
}

int ghip_density_impl(ghip_ctx *ctx, const ghip_dens_params *p)
{
  // the first phase of a step that modifies persistent state (smoothing lengths): an asynchronously
  // built tree is verified first.  The host waits for the build's counting stage only; normally it
  // is long over, and the gravity walks keep running underneath.
  GCHK(ghip_tree_verify(ctx));
  GCHK(ghip_finish_gas_tree(ctx));
  if(ctx->gas_wait_upload)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_density: ghip_upload_aos_particles was not followed by "
                     "ghip_upload_aos_gas");
  GCHK(ghip_gas_verify(ctx));   // (the gas tree's node count, like the gravity tree's above)
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_density: call ghip_tree_build first");
  GCHK(ghip_build_target_lists(ctx));
  ghip_stats &S = ctx->stats;
  S.dens_neighbours = 0;
  S.dens_target_evals = 0;
  S.dens_iterations = 0;
  int ng = ctx->ngas, n = ctx->n;
  int lo, nt;
  shard_slice(ctx, ctx->nt_gas, &lo, &nt);
  if(ng == 0 || nt == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->st;
  GCHK(dens_alloc(ctx));
  double *hcur = P<double>(ctx->dhcur);
  unsigned long long *counter = ghip_cslot(ctx, GHIP_CK_DENS);
  unsigned long long *racc = ghip_rslot(ctx, GHIP_CK_DENS);
  int *dnum = reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 32);
  HIPCHK(hipMemsetAsync(counter, 0, GHIP_CKIND_U64 * 8, st));
  HIPCHK(hipEventRecord(ctx->evp[6], st));

  int *cur = P<int>(ctx->dtgt_a), *nxt = P<int>(ctx->dtgt_b);
  HIPCHK(hipMemcpyAsync(cur, P<int>(ctx->tg_gas) + lo, (size_t) nt * 4, hipMemcpyDeviceToDevice,
                        st));
  k_dens_init<<<cdiv(nt, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(nt, cur, P<double>(ctx->gp), hcur,
                                             P<double>(ctx->dleft), P<double>(ctx->dright));
  BoxK b = {p->BoxSize, 0.5 * p->BoxSize, p->periodic};
  DensFin F = {p->DesNumNgb, p->MaxNumNgbDeviation, p->MinGasHsml, p->Ti_Current,
               p->Timebase_interval};
  int maxiter = p->MaxIter > 0 ? p->MaxIter : 150;
  int ncur = nt, iter = 0;
  GCHK(ghip_ensure(ctx, ctx->cubtmp, ((size_t) cdiv(nt, SEL_BLOCK) + 2) * 4));
  int *seloff = P<int>(ctx->cubtmp);

  while(ncur > 0)
    {
      // wavefronts per bucket: enough of them to occupy the chip when the target list is short
      // (late h-iterations, one rank's share of a multi-GPU run); each takes every nsub-th batch
      const int tgw = ghip_sph_tg();
      const int nbk = (ncur + tgw - 1) / tgw;
      int nsub = (ghip_sph_target_waves() + nbk - 1) / nbk;
      nsub = nsub < 1 ? 1 : (nsub > GHIP_MAXSUB ? GHIP_MAXSUB : nsub);
      SPH_LAUNCH(k_density, tgw, (nbk * nsub + 7) & ~7, st, P<TreeSizes>(t.dsz), P<SphNode>(t.mq), P<double>(ctx->gp), ncur,
                 nsub, cur, hcur, b, P<double>(ctx->drho), P<double>(ctx->dnumngb),
                 P<double>(ctx->ddhsml), P<double>(ctx->ddivv), P<double>(ctx->drot), counter, racc,
                 ghip_sph_xcd(), nbk * nsub);
      k_dens_finalize<<<cdiv(ncur, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
        ncur, nsub, cur, P<int>(t.perm), n, ng, F, P<double>(ctx->drho), P<double>(ctx->dnumngb),
        P<double>(ctx->ddhsml), P<double>(ctx->ddivv), P<double>(ctx->drot), hcur,
        P<double>(ctx->dleft), P<double>(ctx->dright), P<double>(ctx->f[GHIP_F_ENTROPY]),
        P<double>(ctx->f[GHIP_F_DTENTROPY]), P<int>(ctx->f[GHIP_F_TIMEBIN]),
        P<int>(ctx->f[GHIP_F_TI_BEGSTEP]), P<double>(ctx->gp), P<double>(ctx->gq),
        P<double>(ctx->f[GHIP_F_HSML]), P<double>(ctx->f[GHIP_F_NUMNGB]),
        P<double>(ctx->f[GHIP_F_DENSITY]), P<double>(ctx->f[GHIP_F_DHSMLFAC]),
        P<double>(ctx->f[GHIP_F_DIVVEL]), P<double>(ctx->f[GHIP_F_CURLVEL]),
        P<double>(ctx->f[GHIP_F_PRESSURE]), P<int>(ctx->dflags));
      HIPCHK(hipGetLastError());
      S.dens_target_evals += ncur;
      // the targets that are not yet converged, in list order (see k_sel_count)
      const int nblk = cdiv(ncur, SEL_BLOCK);
      k_sel_count<<<nblk, 64, 0, st>>>(ncur, P<int>(ctx->dflags), seloff);
      k_sel_scan<<<1, 64, 0, st>>>(nblk, seloff, dnum);
      int left = 0;
      HIPCHK(hipMemcpyAsync(&left, dnum, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      GCHK(ghip_check_device_errors(ctx));   // (also what an asynchronous drift / kick deferred)
      if(left > 0)
        {
          k_sel_scatter<<<nblk, 64, 0, st>>>(ncur, cur, P<int>(ctx->dflags), seloff, nxt);
          HIPCHK(hipGetLastError());
        }
      ncur = left;
      int *tmp = cur;
      cur = nxt;
      nxt = tmp;
      if(ncur > 0)
        {
          iter++;
          if(iter > maxiter)
            return ghip_fail(ctx, GHIP_ENOCONV,
                             "density: %d particles not converged after %d h-iterations "
                             "(reference: endrun(1155), density.c:669-674)", ncur, maxiter);
        }
    }
  HIPCHK(hipEventRecord(ctx->evp[7], st));
  S.dens_iterations = iter;
  return GHIP_OK;
}

extern "C" int ghip_density(ghip_ctx *ctx, const ghip_dens_params *p)
{
  if(!ctx || !p)
    return GHIP_EINVAL;
  return ghip_density_impl(ctx, p);
}

extern "C" int ghip_update_hmax(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_EINVAL;
  GCHK(ghip_tree_verify(ctx));
  GCHK(ghip_finish_gas_tree(ctx));
  GCHK(ghip_gas_verify(ctx));
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_update_hmax: no tree");
  return ghip_gastree_refresh_hmax(ctx);
}

extern "C" int ghip_density_evaluate(ghip_ctx *ctx, const ghip_dens_params *p, int target,
                                     double h, double out7[7])
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p || !out7)
    return GHIP_EINVAL;
  GCHK(ghip_finish_gas_tree(ctx));
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_density_evaluate: no tree");
  if(target < 0 || target >= ctx->ngas)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_density_evaluate: target %d is not a gas particle",
                     target);
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->st;
  GCHK(dens_alloc(ctx));
  double *hcur = P<double>(ctx->dhcur);
  const int nsub = 1;
  int s = 0;
  HIPCHK(hipMemcpyAsync(&s, P<int>(t.iperm) + target, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  int *cur = P<int>(ctx->dtgt_a);
  HIPCHK(hipMemcpyAsync(cur, &s, 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(hcur + s, &h, 8, hipMemcpyHostToDevice, st));
  BoxK b = {p->BoxSize, 0.5 * p->BoxSize, p->periodic};
  unsigned long long *counter = ghip_cslot(ctx, GHIP_CK_DENS1);
  k_density<8><<<nsub, 64, 0, st>>>(P<TreeSizes>(t.dsz), P<SphNode>(t.mq), P<double>(ctx->gp), 1, nsub, cur, hcur, b,
                              P<double>(ctx->drho), P<double>(ctx->dnumngb),
                              P<double>(ctx->ddhsml), P<double>(ctx->ddivv), P<double>(ctx->drot),
                              counter, ghip_rslot(ctx, GHIP_CK_DENS1), 0, nsub);
  HIPCHK(hipGetLastError());
  // nt == 1: partial q of component c sits at [q] (rot: [c*nsub + q]); sum in fixed order
  double part[7][GHIP_MAXSUB * 8];
  if(nsub > GHIP_MAXSUB * 8)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_density_evaluate: too many sub-walks");
  HIPCHK(hipMemcpyAsync(part[0], ctx->drho.p, (size_t) nsub * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(part[1], ctx->dnumngb.p, (size_t) nsub * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(part[2], ctx->ddhsml.p, (size_t) nsub * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(part[3], ctx->ddivv.p, (size_t) nsub * 8, hipMemcpyDeviceToHost, st));
  for(int c = 0; c < 3; c++)
    HIPCHK(hipMemcpyAsync(part[4 + c], P<double>(ctx->drot) + (size_t) c * nsub,
                          (size_t) nsub * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int c = 0; c < 7; c++)
    {
      double acc = 0;
      for(int q = 0; q < nsub; q++)
        acc += part[c][q];
      out7[c] = acc;
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// neighbour list for one search centre (compat surface of ngb_treefind_variable / _pairs)
// ---------------------------------------------------------------------------------------------
__global__ void k_ngb_find(int nelem, const double4 *__restrict__ cl, const int4 *__restrict__ lk,
                           const double *__restrict__ aux, const double *__restrict__ gp,
                           const int *__restrict__ perm, double cx, double cy, double cz,
                           double hsml, int pairs, BoxK b, int cap, int *__restrict__ list,
                           int *__restrict__ count)
{
  if(threadIdx.x != 0 || blockIdx.x != 0)
    return;
  int num = 0;
  int e = 0;
  while(e < nelem)
    {
      const int4 k = lk[e];
      if(LK_IS_PARTICLE(k))
        {
          const double *r8 = gp + (size_t) 8 * k.y;
          double dist = hsml;
          if(pairs && r8[7] > dist)
            dist = r8[7];
          e = e + 1;
          if(r8[3] < 0)   // converted: P[p].Type > 0 (ngb.c:213)
            continue;
          double dx = d_ngb_periodic(r8[0] - cx, b.periodic, b.boxsize, b.boxhalf);
          if(dx > dist)
            continue;
          double dy = d_ngb_periodic(r8[1] - cy, b.periodic, b.boxsize, b.boxhalf);
          if(dy > dist)
            continue;
          double dz = d_ngb_periodic(r8[2] - cz, b.periodic, b.boxsize, b.boxhalf);
          if(dz > dist)
            continue;
          if(dx * dx + dy * dy + dz * dz > dist * dist)
            continue;
          if(num < cap)
            list[num] = perm[k.y];
          num++;
        }
      else
        {
          double dist = hsml;
          if(pairs && aux[e] > dist)
            dist = aux[e];
          e = d_node_overlaps(cl[e], dist, cx, cy, cz, b) ? e + 1 : k.x;
        }
    }
  *count = num;
}

extern "C" int ghip_ngb_treefind(ghip_ctx *ctx, const double center[3], double hsml, int pairs,
                                 int periodic, double boxsize, int *ngblist, int cap, int *nfound)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !center || !nfound || cap < 0 || (cap > 0 && !ngblist))
    return GHIP_EINVAL;
  GCHK(ghip_finish_gas_tree(ctx));
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_ngb_treefind: no tree");
  *nfound = 0;
  TreeDev &t = ctx->st;
  if(t.n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) (cap + 4) * 4));
  int *dlist = P<int>(ctx->stage) + 4, *dcount = P<int>(ctx->stage);
  BoxK b = {boxsize, 0.5 * boxsize, periodic};
  k_ngb_find<<<1, 64, 0, st>>>(t.nelem, P<double4>(t.cl), P<int4>(t.lk), P<double>(t.aux),
                               P<double>(ctx->gp), P<int>(t.perm), center[0], center[1], center[2],
                               hsml, pairs, b, cap, dlist, dcount);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(nfound, dcount, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  int ncopy = *nfound < cap ? *nfound : cap;
  if(ncopy > 0)
    {
      HIPCHK(hipMemcpyAsync(ngblist, dlist, (size_t) ncopy * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
    }
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// hydro
// ---------------------------------------------------------------------------------------------
struct HydK
{
  double visc_const, hubble_a2, fac_mu, fac_vsic_fix, timebase;
  int comoving, raw;
};

struct HydAcc
{
  double ax, ay, az, dtent, maxsig;
  int np;
};

struct HydTgt
{
  double px, py, pz, vx, vy, vz, h_i, h_i2, mass, rho, f1, p_over_rho2_i, soundspeed_i, timestep;
  double hinv_i, hinv4_i;   // 1 / h_i and its fourth power, as d_hydro_pair used to form them per pair
};

// What depends on the candidate alone (hydra.c:1321-1326, 1440-1448), formed ONCE per candidate --
// by the lane that stages it, or on the spot for a single-particle element -- instead of by every
// lane of every pair: p_over_rho2_j = P_j / rho_j^2, soundspeed_j = sqrt(GAMMA p_over_rho2_j rho_j),
// 1 / h_j.  Same operations in the same order as the per-pair form: identical numbers.
struct HydCand
{
  double p_over_rho2, soundspeed, hinv, f2;
};
__device__ __forceinline__ HydCand d_hydro_candidate(double pres_j, double rho_j, double h_j,
                                                     double divv_j, double curl_j, double fac_mu)
{
  HydCand C;
  C.p_over_rho2 = pres_j / (rho_j * rho_j);
  C.soundspeed = sqrt(GAMMA * C.p_over_rho2 * rho_j);
  C.hinv = 1.0 / h_j;
  // the Balsara factor of the candidate (hydra.c:1548-1551)
  C.f2 = fabs(divv_j) / (fabs(divv_j) + curl_j + 0.0001 * C.soundspeed / fac_mu * C.hinv);
  return C;
}


// r8: (x,y,z,m,vx,vy,vz,h) of the candidate; q8: (p_over_rho2, rho, dhsml factor, f2, -, timestep,
// soundspeed, 1/h) -- slots 0, 3, 6, 7 as d_hydro_candidate leaves them
__device__ __forceinline__ void d_hydro_pair(const double *r8, const double *q8, bool valid,
                                             const HydTgt &T, const HydK &K, const BoxK b,
                                             HydAcc &A)
{
  const double jx = r8[0], jy = r8[1], jz = r8[2], mass_j = r8[3];
  const double jvx = r8[4], jvy = r8[5], jvz = r8[6], h_j = r8[7];
  double dx = d_wrap(T.px - jx, b), dy = d_wrap(T.py - jy, b), dz = d_wrap(T.pz - jz, b);
  double r2 = dx * dx + dy * dy + dz * dz;
  // hydra.c:1266-1269
  if(valid && (r2 < T.h_i2 || r2 < h_j * h_j) && r2 > 0)
    {
      A.np++;
      double p_over_rho2_j = q8[0];
      const double rho_j = q8[1], dhf_j = q8[2], f2 = q8[3], ts_j = q8[5];
      const double soundspeed_j = q8[6], hinv_j = q8[7];
      // (the reference takes sqrt(r2) and divides by r three times: one reciprocal square root here)
      const double rinv = d_rsqrt_sph(r2);
      const double r = r2 * rinv;
      double dvx = T.vx - jvx, dvy = T.vy - jvy, dvz = T.vz - jvz;
      double vdotr = dx * dvx + dy * dvy + dz * dvz;
      double vdotr2 = K.comoving ? vdotr + K.hubble_a2 * r2 : vdotr;
      double dwk_i = 0, dwk_j = 0;
      if(r2 < T.h_i2)
        {
          const double u = r * T.hinv_i;
          dwk_i = (u < 0.5) ? T.hinv4_i * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4)
                            : T.hinv4_i * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
        }
      if(r2 < h_j * h_j)
        {
          const double hinv4 = hinv_j * hinv_j * hinv_j * hinv_j;
          const double u = r * hinv_j;
          dwk_j = (u < 0.5) ? hinv4 * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4)
                            : hinv4 * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
        }
      double vsig = T.soundspeed_i + soundspeed_j;
      if(vsig > A.maxsig)
        A.maxsig = vsig;
      double visc = 0;
      if(vdotr2 < 0)
        {
          // hydra.c:1512-1594
          double mu_ij = K.fac_mu * vdotr2 * rinv;
          vsig -= 3 * mu_ij;
          if(vsig > A.maxsig)
            A.maxsig = vsig;
          double rho_ij = 0.5 * (T.rho + rho_j);
          visc = 0.25 * K.visc_const * vsig * (-mu_ij) / rho_ij * (T.f1 + f2);
          double tmax = (T.timestep > ts_j) ? T.timestep : ts_j;
          double dt = 2 * tmax * K.timebase;
          if(dt > 0 && (dwk_i + dwk_j) < 0)
            {
              double lim = 0.5 * K.fac_vsic_fix * vdotr2 /
                           (0.5 * (T.mass + mass_j) * (dwk_i + dwk_j) * r * dt);
              if(lim < visc)
                visc = lim;
            }
        }
      p_over_rho2_j *= dhf_j;
      double hfc_visc = 0.5 * mass_j * visc * (dwk_i + dwk_j) * rinv;
      double hfc = hfc_visc + mass_j * (T.p_over_rho2_i * dwk_i + p_over_rho2_j * dwk_j) * rinv;
      A.ax += -hfc * dx;
      A.ay += -hfc * dy;
      A.az += -hfc * dz;
      A.dtent += 0.5 * hfc_visc * vdotr2;
    }
}

// hydro_evaluate (hydra.c:822-1995, mode 0) for a bucket of TG curve-consecutive targets x 64/TG
// candidate slots, same structure as k_density: scalar node records, candidate records (gp + gq,
// 128 B) staged through LDS 64 at a time, exact per-lane acceptance r2 < h_i^2 || r2 < h_j^2.  Node
// pruning uses max(hmax_node, h_i) like ngb_treefind_pairs (ngb.c:136).  Outputs: [5][nt] planes.
template <int TG>
__global__ void __launch_bounds__(64)
k_hydro(const TreeSizes *__restrict__ ts, const SphNode *__restrict__ nodes, const double *__restrict__ gp,
        const double *__restrict__ gq, int nt, int nsub, const int *__restrict__ tgt, BoxK b, HydK K,
        double *__restrict__ part, unsigned long long *__restrict__ counter,
        unsigned long long *__restrict__ racc, int xcd, int nblocks)
{
  constexpr int CS = 64 / TG;
  __shared__ double4 sh[SPH_STAGE][4];
  const int lane = threadIdx.x;
  const int tl = lane & (TG - 1), cs = lane / TG;
  const int lb = d_sph_block(xcd, nblocks);
  if(lb < 0)
    return;
  const int bucket = lb / nsub;
  const int sub = lb - bucket * nsub;
  int batch = 0;
  const int ti = bucket * TG + tl;
  const bool valid = ti < nt;
  const int s = valid ? tgt[ti] : 0;
  HydTgt T = {0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 0, 0, 0, 1, 1};
  if(valid)
    {
      const double *r8 = gp + (size_t) 8 * s;
      const double *q8 = gq + (size_t) 8 * s;
      T.px = r8[0];
      T.py = r8[1];
      T.pz = r8[2];
      T.mass = r8[3] < 0 ? 0.0 : r8[3];   // (a swallowed target's record is marked like a dead neighbour's)
      T.vx = r8[4];
      T.vy = r8[5];
      T.vz = r8[6];
      T.h_i = r8[7];
      T.h_i2 = T.h_i * T.h_i;
      T.hinv_i = 1.0 / T.h_i;
      T.hinv4_i = T.hinv_i * T.hinv_i * T.hinv_i * T.hinv_i;
      double pres = q8[0];
      T.rho = q8[1];
      double dhf = q8[2], divv = q8[3], curl = q8[4];
      T.timestep = q8[5];
      // hydra.c:958-975, 1154-1156
      T.soundspeed_i = sqrt(GAMMA * pres / T.rho);
      T.f1 = fabs(divv) / (fabs(divv) + curl + 0.0001 * T.soundspeed_i / T.h_i / K.fac_mu);
      T.p_over_rho2_i = pres / (T.rho * T.rho);
      T.p_over_rho2_i *= dhf;
    }
  HydAcc A = {0, 0, 0, 0, 0, 0};
  const BucketBox BB = d_bucket_box(valid, T.px, T.py, T.pz, T.h_i, b);

  const int nelem = __builtin_amdgcn_readfirstlane(ts->nelem);
  int e = 0;
  while(e < nelem)
    {
      e = __builtin_amdgcn_readfirstlane(e);
      v16i_s N;
      d_load_sphnode(nodes, e, N);
      const int skip = N[10], pidx = N[11], pstart = N[12], pcount = N[13];
      if(pidx >= 0)
        {
          if((batch++ % nsub) == sub && gp[(size_t) 8 * pidx + 3] >= 0)
            {
              const double *qs = gq + (size_t) 8 * pidx;
              const HydCand C = d_hydro_candidate(qs[0], qs[1], gp[(size_t) 8 * pidx + 7], qs[3], qs[4],
                                                  K.fac_mu);
              const double q8[8] = {C.p_over_rho2, qs[1], qs[2], C.f2, 0.0, qs[5], C.soundspeed, C.hinv};
              d_hydro_pair(gp + (size_t) 8 * pidx, q8, valid && cs == 0, T, K, b, A);
            }
          e = e + 1;
        }
      else
        {
          const double4 c = make_double4(d_f64s(N, 0), d_f64s(N, 1), d_f64s(N, 2), d_f64s(N, 3));
          const double hm = d_f64s(N, 4);            // Extnodes[].hmax
          double dist = (hm > T.h_i) ? hm : T.h_i;   // ngb.c:136
          bool open = valid && d_node_overlaps(c, dist, T.px, T.py, T.pz, b);
          if(__any(open))
            {
              if(pcount <= SPH_LEAF)
                {
                  for(int r0 = 0; r0 < pcount; r0 += SPH_STAGE)
                    if((batch++ % nsub) == sub)
                      {
                        const bool staged = r0 + lane < pcount;
                        __syncthreads();
                        double4 c0 = make_double4(0, 0, 0, 0), c1 = make_double4(0, 0, 0, 0);
                        if(staged)
                          {
                            const double4 *s0 =
                              reinterpret_cast<const double4 *>(gp + (size_t) 8 * (pstart + r0 + lane));
                            const double4 *s1 =
                              reinterpret_cast<const double4 *>(gq + (size_t) 8 * (pstart + r0 + lane));
                            c0 = s0[0];
                            c1 = s0[1];
                            double4 q0 = s1[0], q1 = s1[1];   // (P, rho, f, divv) (curl, timestep, -, -)
                            const HydCand C = d_hydro_candidate(q0.x, q0.y, c1.w, q0.w, q1.x, K.fac_mu);
                            q0.x = C.p_over_rho2;
                            q0.w = C.f2;
                            q1.z = C.soundspeed;
                            q1.w = C.hinv;
                            sh[lane][0] = c0;
                            sh[lane][1] = c1;
                            sh[lane][2] = q0;
                            sh[lane][3] = q1;
                          }
                        // (pairs: the candidate's own smoothing length c1.w counts too, hydra.c:1266)
                        unsigned long long live = d_cull_batch(BB, staged, c0.x, c0.y, c0.z, c0.w, c1.w, b);
                        __syncthreads();
                        while(live)
                          {
                            const int j = d_next_candidates<CS>(live, cs);
                            const int jj = j < 0 ? 0 : j;
                            d_hydro_pair(reinterpret_cast<const double *>(&sh[jj][0]),
                                         reinterpret_cast<const double *>(&sh[jj][2]), valid && j >= 0, T,
                                         K, b, A);
                          }
                      }
                  e = skip;
                }
              else
                e = e + 1;
            }
          else
            e = skip;
        }
    }
  const double ax = d_slot_sum<TG>(A.ax), ay = d_slot_sum<TG>(A.ay), az = d_slot_sum<TG>(A.az);
  const double de = d_slot_sum<TG>(A.dtent), ms = d_slot_max<TG>(A.maxsig);
  if(valid && cs == 0)
    {
      const size_t plane = (size_t) nsub * nt;
      const size_t o = (size_t) sub * nt + ti;
      part[o] = ax;
      part[plane + o] = ay;
      part[2 * plane + o] = az;
      part[3 * plane + o] = de;
      part[4 * plane + o] = ms;
    }
  unsigned long long tot = d_wave_sum_u64((unsigned long long) A.np);
  if(lane == 0 && tot)
    {
      d_count(counter, 0, tot);
      d_count(racc, 0, tot);
    }
}

// fixed-order sum of the partial results + the entropy-rate conversion of hydro_force
// (hydra.c:583), scattered to host order
__global__ void k_hydro_combine(int nt, int nsub, const int *__restrict__ tgt,
                                const int *__restrict__ perm, const double *__restrict__ gq,
                                const double *__restrict__ part, HydK K, int ngas,
                                double *__restrict__ o_acc, double *__restrict__ o_dtent,
                                double *__restrict__ o_maxsig)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  const size_t plane = (size_t) nsub * nt;
  double ax = 0, ay = 0, az = 0, de = 0, ms = 0;
  for(int q = 0; q < nsub; q++)
    {
      const size_t o = (size_t) q * nt + ti;
      ax += part[o];
      ay += part[plane + o];
      az += part[2 * plane + o];
      de += part[3 * plane + o];
      double m = part[4 * plane + o];
      ms = m > ms ? m : ms;
    }
  const int s = tgt[ti];
  const int i = perm[s];
  const double rho = gq[(size_t) 8 * s + 1];
  o_acc[i] = ax;
  o_acc[(size_t) ngas + i] = ay;
  o_acc[2 * (size_t) ngas + i] = az;
  o_dtent[i] = K.raw ? de : de * (GAMMA_MINUS1 / (K.hubble_a2 * pow(rho, GAMMA_MINUS1)));
  o_maxsig[i] = ms;
}

int ghip_hydro_impl(ghip_ctx *ctx, const ghip_hydro_params *p)
{
  GCHK(ghip_tree_verify(ctx));
  GCHK(ghip_finish_gas_tree(ctx));
  GCHK(ghip_gas_verify(ctx));
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_hydro: call ghip_tree_build first");
  GCHK(ghip_build_target_lists(ctx));
  ghip_stats &S = ctx->stats;
  S.hydro_pairs = 0;
  int ng = ctx->ngas;
  int lo, nt;
  shard_slice(ctx, ctx->nt_gas, &lo, &nt);
  S.hydro_targets = nt;
  if(ng == 0 || nt == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->st;
  GCHK(ghip_unmark_massless_for_hydro(ctx));   // (-DDUST without -DBLACK_HOLES)
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  unsigned long long *counter = ghip_cslot(ctx, GHIP_CK_HYDRO);
  HIPCHK(hipMemsetAsync(counter, 0, GHIP_CKIND_U64 * 8, st));
  BoxK b = {p->BoxSize, 0.5 * p->BoxSize, p->periodic};
  HydK K = {p->ArtBulkViscConst, p->hubble_a2, p->fac_mu, p->fac_vsic_fix, p->Timebase_interval,
            p->ComovingIntegrationOn, p->raw_dtentropy};
  // Underneath a gravity pair the hydro kernel (128 VGPRs) only fits where an Ewald wavefront would
  // sit, and while both walks run it slows them by more than it gains (11.4 vs 11.2 ms per step at
  // c2): it starts when the Ewald walk has drained and fills the tail of the Newtonian walk instead.
  // "Drained" is taken one wavefront generation early: the walk sets a word when its LAST workgroup
  // starts, and the stream waits for that word -- from then on the slots the Ewald walk gives back stay
  // free, and hydro's wavefronts move in while the last Ewald ones finish (GHIP_HYDRO_TRIGGER=drain:
  // the kernel's end event instead).
  // (ghip_set_hydro_release(ctx, 1): at once -- for a host whose download of the SPH results then also
  // runs underneath the walks, which is worth more than the 0.2 ms)
  if(ctx->grav_pending)
    ctx->pc_hyd[(ctx->pc_head + 3) & 3] = 1;   // (the pair launched last: its cost includes this kernel)
  if(ctx->grav_pending && !ctx->hydro_early && !getenv("GHIP_HYDRO_EARLY"))
    {
      static int drain = -1;
      if(drain < 0)
        drain = getenv("GHIP_HYDRO_TRIGGER") && !strcmp(getenv("GHIP_HYDRO_TRIGGER"), "drain");
      if(drain || !ctx->pair_started)
        HIPCHK(hipStreamWaitEvent(st, ctx->evx[3], 0));
      else
        HIPCHK(hipStreamWaitValue32(st, ctx->pair_started, 1, hipStreamWaitValueGte, 0xffffffffu));
    }
  HIPCHK(hipEventRecord(ctx->evp[10], st));
  const int tgw = ghip_sph_tg();
  const int nbk = (nt + tgw - 1) / tgw;
  int nsub = (ghip_sph_target_waves() + nbk - 1) / nbk;
  nsub = nsub < 1 ? 1 : (nsub > GHIP_MAXSUB ? GHIP_MAXSUB : nsub);
  GCHK(ghip_ensure(ctx, ctx->hpart, (size_t) 5 * nsub * nt * 8));
  SPH_LAUNCH(k_hydro, tgw, (nbk * nsub + 7) & ~7, st, P<TreeSizes>(t.dsz), P<SphNode>(t.mq), P<double>(ctx->gp),
             P<double>(ctx->gq), nt, nsub, P<int>(ctx->tg_gas) + lo, b, K, P<double>(ctx->hpart),
             counter, ghip_rslot(ctx, GHIP_CK_HYDRO), ghip_sph_xcd(), nbk * nsub);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evp[11], st));
  k_hydro_combine<<<cdiv(nt, ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(
    nt, nsub, P<int>(ctx->tg_gas) + lo, P<int>(t.perm), P<double>(ctx->gq),
    P<double>(ctx->hpart), K, ng, P<double>(ctx->f[GHIP_F_HYDROACCEL]),
    P<double>(ctx->f[GHIP_F_DTENTROPY]), P<double>(ctx->f[GHIP_F_MAXSIGNALVEL]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_set_hydro_release(ghip_ctx *ctx, int early)
{
  if(!ctx)
    return GHIP_EINVAL;
  ctx->hydro_early = early != 0;
  return GHIP_OK;
}

extern "C" int ghip_hydro(ghip_ctx *ctx, const ghip_hydro_params *p)
{
  if(!ctx || !p)
    return GHIP_EINVAL;
  return ghip_hydro_impl(ctx, p);
}
