// ghip_sink.hip -- "next" row N4: the sink (black-hole) neighbour passes and the per-particle part
// of cooling_and_starformation, for the reference's shipped flag bundle (BLACK_HOLES, SWALLOWGAS,
// ACCRETION_RADIUS, ACCRETION_DENSITY, ACCRETION_OF_DUST_ONLY, BH_MERGERS_WITHIN_H,
// BH_THERMALFEEDBACK + TMP_FEEDBACK, DUST, COOLING, SFR, BH_FORM; Makefile:8-9, 67-72, 92-96, 124,
// 199-204).
//
// Replaces the particle loops of
//   density() for Type-5 targets                    density.c:125-704 (BLACK_HOLES branches),
//                                                   density_evaluate :763-765, 831-834, 879-886, 972-978
//   blackhole_evaluate / ngb_treefind_blackhole     blackhole.c:794-1190, 1351-1474
//   blackhole_evaluate_swallow                      blackhole.c:1201-1346
//   cooling_and_starformation (per particle)        sfr_eff.c:82-947
// What stays scalar host code: blackhole_accretion()'s per-sink bookkeeping (accretion-disc
// reservoir, Mdot, adding the accreted mass and momentum: blackhole.c:133-300, 680-760), the
// conversion of a flagged gas particle into a sink (sfr_eff.c:606-640: needs the GSL stream) and the
// cooling function DoCooling (cooling.c) -- SURVEY 8(f) N4.
//
// Sinks are few (hundreds at c5) against millions of candidates: one WAVEFRONT per sink walks the
// tree with a wave-uniform element index; a cell of <= 256 particles that overlaps the search
// sphere is swept flat, 64 candidates at a time, one per lane.  Victims are marked by scatter:
// atomicMax on SwallowID (the largest ID wins, deterministic), atomicAdd on the injected energy.
#include "ghip_internal.h"

#define KERNEL_COEFF_1 2.546479089470
#define KERNEL_COEFF_2 15.278874536822
#define KERNEL_COEFF_5 5.092958178941
#define NORM_COEFF 4.188790204786
#define SINK_FACT1 0.366025403785   // allvars.h:310
#define SINK_GAMMA (7. / 5.)        // allvars.h:64
#define SINK_GAMMA_MINUS1 (SINK_GAMMA - 1)
#define SINK_LEAF 256

struct SinkBox
{
  double boxsize, boxhalf;
  int periodic;
};

__device__ __forceinline__ double d_sink_wrap(double d, const SinkBox &b)
{
  if(b.periodic)
    {
      if(d > b.boxhalf)
        d -= b.boxsize;
      if(d < -b.boxhalf)
        d += b.boxsize;
    }
  return d;
}

// node test of ngb_treefind_blackhole / ngb_treefind_variable (blackhole.c:1453-1466, ngb.c:276-289)
__device__ __forceinline__ bool d_sink_overlaps(double cx, double cy, double cz, double len, double h,
                                                double px, double py, double pz, const SinkBox &b)
{
  double dist = h + 0.5 * len;
  double dx = d_ngb_periodic(cx - px, b.periodic, b.boxsize, b.boxhalf);
  if(dx > dist)
    return false;
  double dy = d_ngb_periodic(cy - py, b.periodic, b.boxsize, b.boxhalf);
  if(dy > dist)
    return false;
  double dz = d_ngb_periodic(cz - pz, b.periodic, b.boxsize, b.boxhalf);
  if(dz > dist)
    return false;
  dist += SINK_FACT1 * len;
  return !(dx * dx + dy * dy + dz * dz > dist * dist);
}

__device__ __forceinline__ double d_sink_kernel(double r, double h)
{
  const double hinv = 1 / h, hinv3 = hinv * hinv * hinv, u = r * hinv;
  if(u < 0.5)
    return hinv3 * (KERNEL_COEFF_1 + KERNEL_COEFF_2 * (u - 1) * u * u);
  return hinv3 * KERNEL_COEFF_5 * (1.0 - u) * (1.0 - u) * (1.0 - u);
}

// fixed-order sum over the 64 lanes (valid in every lane)
__device__ __forceinline__ double d_wave_sum_f64(double v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// density for sink targets: one fixed-h evaluation per launch (gas tree: SphNode list + gp records)
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(64))) SinkSphNode   // = SphNode of ghip_sph.hip
{
  double cx, cy, cz, len;
  double hmax;
  int skip, pidx, pstart, pcount;
  int pad[2];
};

struct SinkRec   // what a pass needs of its sink (host fills it from the resident fields' values)
{
  double x, y, z, vx, vy, vz, mass, h, mdot, rho;
  unsigned int id;
  int timebin, index, pad;
};

// out: [6][ns] planes: rho, weighted numngb, sum m w A, sum m w v[3].  local_only (a multi-GPU shard,
// whose gas tree also holds ghosts): candidates are this shard's own gas particles only -- the other
// shards add theirs
__global__ void __launch_bounds__(64)
k_sink_density(int ns, const int *__restrict__ slot, const SinkRec *__restrict__ recs,
               const double *__restrict__ sh, int nelem, const SinkSphNode *__restrict__ nodes,
               const double *__restrict__ gp, const int *__restrict__ perm,
               const double *__restrict__ entropy, int ngas, int local_only, SinkBox b,
               double *__restrict__ out)
{
  const int a = slot[blockIdx.x], lane = threadIdx.x;
  const double px = recs[a].x, py = recs[a].y, pz = recs[a].z;
  const double h = sh[a], h2 = h * h;
  double rho = 0, wn = 0, se = 0, g0 = 0, g1 = 0, g2 = 0;
  const double hinv = 1.0 / h, hinv3 = hinv * hinv * hinv;
  int e = 0;
  while(e < nelem)
    {
      const SinkSphNode N = nodes[e];   // wave-uniform address
      int first, count;
      if(N.pidx >= 0)
        {
          first = N.pidx;
          count = 1;
          e = e + 1;
        }
      else
        {
          if(!d_sink_overlaps(N.cx, N.cy, N.cz, N.len, h, px, py, pz, b))
            {
              e = N.skip;
              continue;
            }
          if(N.pcount > SINK_LEAF)
            {
              e = e + 1;
              continue;
            }
          first = N.pstart;
          count = N.pcount;
          e = N.skip;
        }
      for(int p0 = first; p0 < first + count; p0 += 64)
        {
          const int p = p0 + lane;
          if(p >= first + count)
            continue;
          const double *r8 = gp + (size_t) 8 * p;
          const double mass_j = r8[3];
          if(mass_j <= 0)   // density.c:831-834; < 0: a converted particle of the gas block, not gas
            continue;
          const double dx = d_sink_wrap(px - r8[0], b), dy = d_sink_wrap(py - r8[1], b),
                       dz = d_sink_wrap(pz - r8[2], b);
          const double r2 = dx * dx + dy * dy + dz * dz;
          if(r2 < h2)
            {
              const int j = perm[p];
              if(local_only && j >= ngas)
                continue;
              const double wk = d_sink_kernel(sqrt(r2), h);
              rho += mass_j * wk;
              wn += NORM_COEFF * wk / hinv3;
              g0 += mass_j * wk * r8[4];
              g1 += mass_j * wk * r8[5];
              g2 += mass_j * wk * r8[6];
              se += mass_j * wk * (j < ngas ? entropy[j] : 0.0);
            }
        }
    }
  rho = d_wave_sum_f64(rho);
  wn = d_wave_sum_f64(wn);
  se = d_wave_sum_f64(se);
  g0 = d_wave_sum_f64(g0);
  g1 = d_wave_sum_f64(g1);
  g2 = d_wave_sum_f64(g2);
  if(lane == 0)
    {
      out[a] = rho;
      out[(size_t) ns + a] = wn;
      out[2 * (size_t) ns + a] = se;
      out[3 * (size_t) ns + a] = g0;
      out[4 * (size_t) ns + a] = g1;
      out[5 * (size_t) ns + a] = g2;
    }
}

// the sinks' own state from the resident fields (a few 8-byte reads each) + the per-sink inputs
static int gather_sinks(ghip_ctx *ctx, int nsink, const int *idx, const unsigned int *id,
                        const double *mdot, const double *rho, std::vector<SinkRec> &S)
{
  hipStream_t st = ctx->stream;
  const size_t n = (size_t) ctx->n;
  S.resize(nsink);
  std::vector<int> tb(nsink > 0 ? nsink : 1);
  for(int a = 0; a < nsink; a++)
    {
      const int i = idx[a];
      if(i < 0 || i >= ctx->n)
        return ghip_fail(ctx, GHIP_EINVAL, "sink index %d out of range", i);
      const double *pos = P<double>(ctx->f[GHIP_F_POS]), *vel = P<double>(ctx->f[GHIP_F_VEL]);
      HIPCHK(hipMemcpyAsync(&S[a].x, pos + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].y, pos + n + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].z, pos + 2 * n + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].vx, vel + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].vy, vel + n + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].vz, vel + 2 * n + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].mass, P<double>(ctx->f[GHIP_F_MASS]) + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&S[a].h, P<double>(ctx->f[GHIP_F_HSML]) + i, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&tb[a], P<int>(ctx->f[GHIP_F_TIMEBIN]) + i, 4, hipMemcpyDeviceToHost, st));
      S[a].id = id ? id[a] : 0u;
      S[a].mdot = mdot ? mdot[a] : 0.0;
      S[a].rho = rho ? rho[a] : 1.0;
      S[a].index = i;
      S[a].pad = 0;
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int a = 0; a < nsink; a++)
    S[a].timebin = tb[a];
  return GHIP_OK;
}

// the h iteration of density() for non-gas targets (density.c:521-532, 548-551, 559-652 without the
// Newton step, which needs SphP[].h.DhsmlDensityFactor: gas only, :613, 629), for sinks `todo[0..ncur)`
// whose kernel sums of this pass are in sums[k * ns + a] (rho, numngb, entropy, gasvel[3]).  Returns the
// number of sinks that need another pass (compacted to the front of `todo`).
static int sink_iterate(const ghip_dens_params *p, double ngb_factor, int ns, const double *sums,
                        int ncur, SinkIter &I)
{
  const double desnumngb = p->DesNumNgb * ngb_factor;
  int left = 0;
  for(int q = 0; q < ncur; q++)
    {
      const int a = I.todo[q];
      const double rho = sums[a], wn = sums[(size_t) ns + a];
      double h = I.h[a];
      I.numngb[a] = wn;
      I.rho[a] = rho;
      for(int k = 0; k < 4; k++)
        {
          double v = sums[(size_t) (2 + k) * ns + a];
          if(rho > 0)
            v /= rho;
          if(k == 0)
            I.entropy[a] = v;
          else
            I.gasvel[3 * (size_t) a + k - 1] = v;
        }
      if(wn < (desnumngb - p->MaxNumNgbDeviation) ||
         (wn > (desnumngb + p->MaxNumNgbDeviation) && h > (1.01 * p->MinGasHsml)))
        {
          if(I.left[a] > 0 && I.right[a] > 0 && (I.right[a] - I.left[a]) < 1.0e-3 * I.left[a])
            continue;   // "this one should be ok"
          if(wn < (desnumngb - p->MaxNumNgbDeviation))
            I.left[a] = h > I.left[a] ? h : I.left[a];
          else
            {
              if(I.right[a] != 0)
                {
                  if(h < I.right[a])
                    I.right[a] = h;
                }
              else
                I.right[a] = h;
            }
          if(I.right[a] > 0 && I.left[a] > 0)
            h = pow(0.5 * (pow(I.left[a], 3) + pow(I.right[a], 3)), 1.0 / 3);
          else
            {
              if(I.right[a] == 0 && I.left[a] > 0)
                h *= 1.26;
              if(I.right[a] > 0 && I.left[a] == 0)
                h /= 1.26;
            }
          if(h < p->MinGasHsml)
            h = p->MinGasHsml;
          I.h[a] = h;
          I.todo[left++] = a;
        }
    }
  return left;
}

static void sink_iter_init(SinkIter &I, int ns)
{
  I.h.assign(ns, 0.0);
  I.left.assign(ns, 0.0);
  I.right.assign(ns, 0.0);
  I.numngb.assign(ns, 0.0);
  I.rho.assign(ns, 0.0);
  I.entropy.assign(ns, 0.0);
  I.gasvel.assign((size_t) 3 * ns, 0.0);
  I.todo.resize(ns);
  for(int a = 0; a < ns; a++)
    I.todo[a] = a;
}

// one pass: kernel sums of the sinks todo[0..ncur) at their current h, into dout [6][ns]
static int sink_density_pass(ghip_ctx *ctx, const ghip_dens_params *p, int ns, const SinkRec *drecs,
                             double *dh, int *dslot, double *dout, const SinkIter &I, int ncur,
                             int local_only)
{
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->st;
  SinkBox b = {p->BoxSize, 0.5 * p->BoxSize, p->periodic};
  HIPCHK(hipMemcpyAsync(dh, I.h.data(), (size_t) ns * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dslot, I.todo.data(), (size_t) ncur * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(dout, 0, (size_t) ns * 48, st));
  if(t.nelem > 0)
    k_sink_density<<<ncur, 64, 0, st>>>(ns, dslot, drecs, dh, t.nelem,
                                        reinterpret_cast<const SinkSphNode *>(t.mq.p), P<double>(ctx->gp),
                                        P<int>(t.perm), P<double>(ctx->f[GHIP_F_ENTROPY]), ctx->ngas,
                                        local_only, b, dout);
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_sink_density(ghip_ctx *ctx, const ghip_dens_params *p, double ngb_factor,
                                 int nsink, const int *sink_idx, double *hsml, double *numngb,
                                 double *bh_density, double *bh_entropy, double *bh_gasvel,
                                 int *iterations)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p || nsink < 0 || (nsink > 0 && (!sink_idx || !hsml || !numngb || !bh_density ||
                                              !bh_entropy || !bh_gasvel)))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_sink_density: bad arguments");
  if(iterations)
    *iterations = 0;
  if(nsink == 0)
    return GHIP_OK;
  GCHK(ghip_finish_gas_tree(ctx));
  if(!ctx->st.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_sink_density: call ghip_tree_build first");
  if(ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_sink_density: on a multi-GPU shard use GHIP_DD_SINK_DENSITY");
  hipStream_t st = ctx->stream;
  std::vector<SinkRec> S;
  GCHK(gather_sinks(ctx, nsink, sink_idx, nullptr, nullptr, nullptr, S));
  // staging: SinkRec[ns] | h[ns] | out[6][ns] | slot[ns]
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nsink * (sizeof(SinkRec) + 8 + 48 + 4) + 256));
  SinkRec *drecs = P<SinkRec>(ctx->stage);
  double *dh = reinterpret_cast<double *>(drecs + nsink), *dout = dh + nsink;
  int *dslot = reinterpret_cast<int *>(dout + 6 * (size_t) nsink);
  HIPCHK(hipMemcpyAsync(drecs, S.data(), (size_t) nsink * sizeof(SinkRec), hipMemcpyHostToDevice, st));
  SinkIter I;
  sink_iter_init(I, nsink);
  for(int a = 0; a < nsink; a++)
    I.h[a] = hsml[a];
  std::vector<double> hout((size_t) nsink * 6);
  const int maxiter = p->MaxIter > 0 ? p->MaxIter : 150;
  int ncur = nsink, iter = 0;
  while(ncur > 0)
    {
      GCHK(sink_density_pass(ctx, p, nsink, drecs, dh, dslot, dout, I, ncur, 0));
      HIPCHK(hipMemcpyAsync(hout.data(), dout, (size_t) nsink * 48, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      ncur = sink_iterate(p, ngb_factor, nsink, hout.data(), ncur, I);
      if(ncur > 0)
        {
          iter++;
          if(iter > maxiter)
            return ghip_fail(ctx, GHIP_ENOCONV, "sink density: %d sinks not converged after %d "
                             "h-iterations (reference: endrun(1155))", ncur, maxiter);
        }
    }
  for(int a = 0; a < nsink; a++)
    {
      hsml[a] = I.h[a];
      numngb[a] = I.numngb[a];
      bh_density[a] = I.rho[a];
      bh_entropy[a] = I.entropy[a];
      for(int k = 0; k < 3; k++)
        bh_gasvel[3 * (size_t) a + k] = I.gasvel[3 * (size_t) a + k];
      // PPP[].Hsml of the sink on the device as well (blackhole_evaluate reads it)
      HIPCHK(hipMemcpyAsync(P<double>(ctx->f[GHIP_F_HSML]) + sink_idx[a], &I.h[a], 8,
                            hipMemcpyHostToDevice, st));
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  if(iterations)
    *iterations = iter;
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// the two black-hole passes over the gravity tree (all particle types)
// ---------------------------------------------------------------------------------------------
struct BhK
{
  SinkBox b;
  double ascale, dt_fac, smbh, inner, sinkb, softb, critdens, fbcoeff, unitmass;
  int dust, dust_only, acc_density;
};

// walk the gravity tree around sink S; F(p, valid) is called by all 64 lanes with lane-own candidate p
template <class F>
__device__ __forceinline__ void d_sink_walk(const SinkRec &S, int nelem, const int4 *__restrict__ lk,
                                            const double4 *__restrict__ cl, const SinkBox &b, F &&f)
{
  const int lane = threadIdx.x;
  int e = 0;
  while(e < nelem)
    {
      const int4 k = lk[e];   // wave-uniform
      int first, count;
      if(k.y >= 0)
        {
          first = k.y;
          count = 1;
          e = e + 1;
        }
      else
        {
          const double4 c = cl[e];
          if(!d_sink_overlaps(c.x, c.y, c.z, c.w, S.h, S.x, S.y, S.z, b))
            {
              e = k.x;
              continue;
            }
          if(k.w > SINK_LEAF)
            {
              e = e + 1;
              continue;
            }
          first = k.z;
          count = k.w;
          e = k.x;
        }
      for(int p0 = first; p0 < first + count; p0 += 64)
        f(p0 + lane, p0 + lane < first + count);
    }
}

// blackhole_evaluate (blackhole.c:794-1190) for the shipped bundle.  Per neighbour j within Hsml of
// the sink (both with mass > 0):
//   sink (:935-1001)   closer than the accretion boundary (InnerBoundary for the central object,
//                      mass > 0.95 SMBHmass, else SofteningBndry) and not heavier: claimed
//   dust (:1004-1036)  closer than InnerBoundary / SinkBoundary and bound (e_tot <= 0): claimed
//   gas  (:1040-1160)  the central object claims gas inside InnerBoundary; other sinks claim gas
//                      inside SinkBoundary only without ACCRETION_OF_DUST_ONLY (bound, or denser than
//                      CritDensity with ACCRETION_DENSITY); the thermal feedback of the smaller sinks
//                      is spread with the kernel weight (:1114-1150)
// Two deliberate deviations from the reference text: (1) a victim claimed by several sinks goes to
// the LARGEST ID for all three kinds -- the reference says so for gas (:1068, 1088, 1106) and lets
// the last sink of the active list win for dust and sinks (:984-988, 1028: order dependent);
// (2) e_tot of a dust grain uses the grain's own r = sqrt(r2) -- the reference reads `r` there
// (:1023) before any assignment in that iteration.
__global__ void __launch_bounds__(64)
k_bh_evaluate(int ns, const SinkRec *__restrict__ sinks, int nelem, const int4 *__restrict__ lk,
              const double4 *__restrict__ cl, const double *__restrict__ sx,
              const double *__restrict__ sy, const double *__restrict__ sz,
              const int *__restrict__ perm, int n, int ngas, const int *__restrict__ type,
              const double *__restrict__ mass, const double *__restrict__ vel,
              const double *__restrict__ gasdens, BhK K, unsigned int *__restrict__ swallow,
              double *__restrict__ injected)
{
  const SinkRec S = sinks[blockIdx.x];
  if(!(S.mass > 0))
    return;
  const double h2 = S.h * S.h;
  const bool central = S.mass > 0.95 * K.smbh;
  const double dt = (S.timebin ? (double) (1 << S.timebin) : 0.0) * K.dt_fac;
  double energy = 0.;   // blackhole.c:1114-1150 TMP_FEEDBACK: the smaller sinks only
  if(S.mass < 0.95 * K.smbh)
    energy = K.fbcoeff * pow(S.mass * K.unitmass, 0.6667) * S.mdot * K.unitmass * dt;
  d_sink_walk(S, nelem, lk, cl, K.b, [&](int p, bool valid) {
    if(!valid)
      return;
    const int j = perm[p];
    if(j >= n)
      return;
    const int ty = type[j];
    if(ty != 0 && ty != 5 && !(K.dust && ty == 2))   // blackhole.c:1372-1380
      return;
    const double mj = mass[j];
    if(!(mj > 0))
      return;
    const double dx = d_sink_wrap(S.x - sx[p], K.b), dy = d_sink_wrap(S.y - sy[p], K.b),
                 dz = d_sink_wrap(S.z - sz[p], K.b);
    const double r2 = dx * dx + dy * dy + dz * dz;
    if(!(r2 < h2))
      return;
    const double w0 = vel[j] - S.vx, w1 = vel[(size_t) n + j] - S.vy, w2 = vel[2 * (size_t) n + j] - S.vz;
    const double vrel = sqrt(w0 * w0 + w1 * w1 + w2 * w2) / K.ascale;
    bool claim = false;
    if(ty == 5 && r2 > 0)
      {
        const double bnd = central ? K.inner : K.softb;
        if(!(pow(r2, 0.5) > bnd) && mj <= S.mass)
          claim = true;
      }
    if(ty == 2 && r2 > 0)
      {
        const double bnd = central ? K.inner : K.sinkb;
        if(pow(r2, 0.5) < bnd && (vrel * vrel / 2. - S.mass / (sqrt(r2) + 1.e-20)) <= 0.)
          claim = true;
      }
    if(ty == 0)
      {
        const double r = sqrt(r2);
        const double wk = d_sink_kernel(r, S.h);
        const double etotal = vrel * vrel / 2. - S.mass / (r + 1.e-20);
        if(central)
          {
            if(r < K.inner)
              claim = true;
          }
        else if(!K.dust_only && r < K.sinkb)
          {
            if(K.acc_density ? (gasdens[j] >= K.critdens) : (etotal < 0.))
              claim = true;
          }
        if(j < ngas)
          atomicAdd(injected + j, energy * mj * wk / S.rho);
      }
    if(claim)
      atomicMax(swallow + j, S.id);
  });
}

// blackhole_evaluate_swallow (blackhole.c:1201-1346).  out: [9][ns] planes: mass, bh mass, dust mass,
// momentum[3], counts gas / sinks / dust
__global__ void __launch_bounds__(64)
k_bh_swallow(int ns, const SinkRec *__restrict__ sinks, int nelem, const int4 *__restrict__ lk,
             const double4 *__restrict__ cl, const double *__restrict__ sx,
             const double *__restrict__ sy, const double *__restrict__ sz,
             const int *__restrict__ perm, int n, const int *__restrict__ type,
             const double *__restrict__ mass_in, const double *__restrict__ vel,
             const double *__restrict__ pbh, BhK K, const unsigned int *__restrict__ swallow,
             int *__restrict__ victim, double *__restrict__ out)
{
  const int a = blockIdx.x;
  const SinkRec S = sinks[a];
  const double h2 = S.h * S.h;
  double am = 0, ab = 0, ad = 0, m0 = 0, m1 = 0, m2 = 0, c0 = 0, c1 = 0, c2 = 0;
  d_sink_walk(S, nelem, lk, cl, K.b, [&](int p, bool valid) {
    if(!valid)
      return;
    const int j = perm[p];
    if(j >= n)
      return;
    const int ty = type[j];
    if(ty != 0 && ty != 5 && !(K.dust && ty == 2))
      return;
    if(swallow[j] != S.id)
      return;
    // the search sphere of ngb_treefind_blackhole: r2 <= h^2 (blackhole.c:1390-1391)
    const double dx = d_sink_wrap(S.x - sx[p], K.b), dy = d_sink_wrap(S.y - sy[p], K.b),
                 dz = d_sink_wrap(S.z - sz[p], K.b);
    if(dx * dx + dy * dy + dz * dz > h2)
      return;
    const double mj = mass_in[j];
    am += mj;
    m0 += mj * vel[j];
    m1 += mj * vel[(size_t) n + j];
    m2 += mj * vel[2 * (size_t) n + j];
    if(ty == 5)
      {
        ab += pbh[j];
        c1 += 1;
      }
    else if(ty == 2)
      {
        ad += mj;
        c2 += 1;
      }
    else
      c0 += 1;
    victim[j] = 1;   // its mass (and BH_Mass) is set to zero once every sink has read it
  });
  double v[9] = {am, ab, ad, m0, m1, m2, c0, c1, c2};
  for(int q = 0; q < 9; q++)
    {
      const double s = d_wave_sum_f64(v[q]);
      if(threadIdx.x == 0)
        out[(size_t) q * ns + a] = s;
    }
}

__global__ void k_bh_zero_victims(int n, const int *__restrict__ victim, double *__restrict__ mass)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if(j < n && victim[j])
    mass[j] = 0;
}

__global__ void k_bh_scatter(int ns, const SinkRec *__restrict__ sinks, const double *__restrict__ v,
                             double *__restrict__ dst)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a < ns)
    dst[sinks[a].index] = v[a];
}

static int sink_buffers(ghip_ctx *ctx)
{
  const size_t n = (size_t) (ctx->n > 0 ? ctx->n : 1), ng = (size_t) (ctx->ngas > 0 ? ctx->ngas : 1);
  size_t before = ctx->bh_swallow.cap;
  GCHK(ghip_ensure(ctx, ctx->bh_swallow, n * 4));
  if(ctx->bh_swallow.cap != before)
    HIPCHK(hipMemsetAsync(ctx->bh_swallow.p, 0, ctx->bh_swallow.cap, ctx->stream));
  before = ctx->bh_injected.cap;
  GCHK(ghip_ensure(ctx, ctx->bh_injected, ng * 8));
  if(ctx->bh_injected.cap != before)
    HIPCHK(hipMemsetAsync(ctx->bh_injected.p, 0, ctx->bh_injected.cap, ctx->stream));
  return GHIP_OK;
}

static BhK bh_kparams(const ghip_bh_params *p)
{
  BhK K;
  K.b.boxsize = p->BoxSize;
  K.b.boxhalf = 0.5 * p->BoxSize;
  K.b.periodic = p->periodic;
  K.ascale = p->ascale;
  K.dt_fac = p->dt_fac;
  K.smbh = p->SMBHmass;
  K.inner = p->InnerBoundary;
  K.sinkb = p->SinkBoundary;
  K.softb = p->SofteningBndry;
  K.critdens = p->CritDensity;
  K.fbcoeff = p->FeedbackCoeff;
  K.unitmass = p->UnitMass_in_g;
  K.dust = p->dust;
  K.dust_only = p->accretion_of_dust_only;
  K.acc_density = p->accretion_density;
  return K;
}

static int check_bh_call(ghip_ctx *ctx, const ghip_bh_params *p, int nsink, const void *idx,
                         const void *id, const char *who)
{
  if(!ctx || !p || nsink < 0 || (nsink > 0 && (!idx || !id)))
    return ghip_fail(ctx, GHIP_EINVAL, "%s: bad arguments", who);
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "%s: call ghip_tree_build first", who);
  if(ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "%s: not available on a multi-GPU shard", who);
  return GHIP_OK;
}

extern "C" int ghip_sink_reset(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  GCHK(sink_buffers(ctx));
  HIPCHK(hipMemsetAsync(ctx->bh_swallow.p, 0, ctx->bh_swallow.cap, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->bh_injected.p, 0, ctx->bh_injected.cap, ctx->stream));
  return GHIP_OK;
}

extern "C" int ghip_blackhole_evaluate(ghip_ctx *ctx, const ghip_bh_params *p, int nsink,
                                       const int *sink_idx, const unsigned int *sink_id,
                                       const double *bh_mdot, const double *bh_density)
{
  if(ctx)
    GHIP_JOIN(ctx);
  GCHK(check_bh_call(ctx, p, nsink, sink_idx, sink_id, "ghip_blackhole_evaluate"));
  if(nsink > 0 && (!bh_mdot || !bh_density))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_blackhole_evaluate: bad arguments");
  GCHK(sink_buffers(ctx));
  if(nsink == 0)
    return GHIP_OK;
  std::vector<SinkRec> S;
  GCHK(gather_sinks(ctx, nsink, sink_idx, sink_id, bh_mdot, bh_density, S));
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->gt;
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nsink * sizeof(SinkRec) + 256));
  HIPCHK(hipMemcpyAsync(ctx->stage.p, S.data(), (size_t) nsink * sizeof(SinkRec),
                        hipMemcpyHostToDevice, st));
  k_bh_evaluate<<<nsink, 64, 0, st>>>(nsink, P<SinkRec>(ctx->stage), t.nelem, P<int4>(t.lk),
                                      P<double4>(t.cl), P<double>(ctx->sx), P<double>(ctx->sy),
                                      P<double>(ctx->sz), P<int>(t.perm), ctx->n, ctx->ngas,
                                      P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->f[GHIP_F_MASS]),
                                      P<double>(ctx->f[GHIP_F_VEL]), P<double>(ctx->f[GHIP_F_DENSITY]),
                                      bh_kparams(p), P<unsigned int>(ctx->bh_swallow),
                                      P<double>(ctx->bh_injected));
  HIPCHK(hipGetLastError());
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}

extern "C" int ghip_blackhole_swallow(ghip_ctx *ctx, const ghip_bh_params *p, int nsink,
                                      const int *sink_idx, const unsigned int *sink_id,
                                      double *sink_bh_mass, double *acc_mass, double *acc_bhmass,
                                      double *acc_dustmass, double *acc_momentum,
                                      long long counts[3])
{
  if(ctx)
    GHIP_JOIN(ctx);
  GCHK(check_bh_call(ctx, p, nsink, sink_idx, sink_id, "ghip_blackhole_swallow"));
  if(nsink > 0 && (!sink_bh_mass || !acc_mass || !acc_bhmass || !acc_dustmass || !acc_momentum))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_blackhole_swallow: bad arguments");
  if(counts)
    counts[0] = counts[1] = counts[2] = 0;
  GCHK(sink_buffers(ctx));
  if(nsink == 0)
    return GHIP_OK;
  std::vector<SinkRec> S;
  GCHK(gather_sinks(ctx, nsink, sink_idx, sink_id, nullptr, nullptr, S));
  hipStream_t st = ctx->stream;
  TreeDev &t = ctx->gt;
  const size_t n = (size_t) ctx->n;
  // staging: SinkRec[ns] | bh mass of the sinks [ns] | out [9][ns] ; particle-indexed: bh mass [n], victim [n]
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nsink * (sizeof(SinkRec) + 80) + n * 12 + 512));
  SinkRec *dS = P<SinkRec>(ctx->stage);
  double *dbh = reinterpret_cast<double *>(dS + nsink), *dout = dbh + nsink, *dpbh = dout + 9 * (size_t) nsink;
  int *dvict = reinterpret_cast<int *>(dpbh + n);
  HIPCHK(hipMemcpyAsync(dS, S.data(), (size_t) nsink * sizeof(SinkRec), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dbh, sink_bh_mass, (size_t) nsink * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(dpbh, 0, n * 12, st));
  k_bh_scatter<<<cdiv(nsink, 64), 64, 0, st>>>(nsink, dS, dbh, dpbh);
  k_bh_swallow<<<nsink, 64, 0, st>>>(nsink, dS, t.nelem, P<int4>(t.lk), P<double4>(t.cl),
                                     P<double>(ctx->sx), P<double>(ctx->sy), P<double>(ctx->sz),
                                     P<int>(t.perm), ctx->n, P<int>(ctx->f[GHIP_F_TYPE]),
                                     P<double>(ctx->f[GHIP_F_MASS]), P<double>(ctx->f[GHIP_F_VEL]), dpbh,
                                     bh_kparams(p), P<unsigned int>(ctx->bh_swallow), dvict, dout);
  k_bh_zero_victims<<<cdiv((long long) n, 256), 256, 0, st>>>((int) n, dvict,
                                                             P<double>(ctx->f[GHIP_F_MASS]));
  HIPCHK(hipGetLastError());
  std::vector<double> out((size_t) 9 * nsink);
  std::vector<int> vict(nsink);
  HIPCHK(hipMemcpyAsync(out.data(), dout, (size_t) 9 * nsink * 8, hipMemcpyDeviceToHost, st));
  for(int a = 0; a < nsink; a++)
    HIPCHK(hipMemcpyAsync(&vict[a], dvict + sink_idx[a], 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int a = 0; a < nsink; a++)
    {
      acc_mass[a] = out[a];
      acc_bhmass[a] = out[(size_t) nsink + a];
      acc_dustmass[a] = out[2 * (size_t) nsink + a];
      for(int k = 0; k < 3; k++)
        acc_momentum[3 * (size_t) a + k] = out[(size_t) (3 + k) * nsink + a];
      if(counts)
        for(int k = 0; k < 3; k++)
          counts[k] += (long long) (out[(size_t) (6 + k) * nsink + a] + 0.5);
      if(vict[a])
        sink_bh_mass[a] = 0;   // P[j].BH_Mass = 0 of a swallowed sink (blackhole.c:1274)
    }
  ctx->gt.built = false;   // masses changed: the tree's moments are stale
  ctx->st.built = false;
  return GHIP_OK;
}

extern "C" int ghip_sink_get_marks(ghip_ctx *ctx, unsigned int *swallow_id, double *injected_energy)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  GCHK(sink_buffers(ctx));
  hipStream_t st = ctx->stream;
  if(swallow_id && ctx->n > 0)
    HIPCHK(hipMemcpyAsync(swallow_id, ctx->bh_swallow.p, (size_t) ctx->n * 4, hipMemcpyDeviceToHost, st));
  if(injected_energy && ctx->ngas > 0)
    HIPCHK(hipMemcpyAsync(injected_energy, ctx->bh_injected.p, (size_t) ctx->ngas * 8,
                          hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}

extern "C" int ghip_sink_set_marks(ghip_ctx *ctx, const unsigned int *swallow_id,
                                   const double *injected_energy)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  GCHK(sink_buffers(ctx));
  hipStream_t st = ctx->stream;
  if(swallow_id && ctx->n > 0)
    HIPCHK(hipMemcpyAsync(ctx->bh_swallow.p, swallow_id, (size_t) ctx->n * 4, hipMemcpyHostToDevice, st));
  if(injected_energy && ctx->ngas > 0)
    HIPCHK(hipMemcpyAsync(ctx->bh_injected.p, injected_energy, (size_t) ctx->ngas * 8,
                          hipMemcpyHostToDevice, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// cooling_and_starformation, per active gas particle (sfr_eff.c:82-947), the deterministic part
// that is active for the shipped bundle, with the cooling function (DoCooling, cooling.c) as identity:
//   flag  (:226-229, 459-462)  the particle qualifies for conversion into a sink
//                              (Density >= CritPhysDensity in code units) unless its mass is 0
//   unew  (:486-488)           max(MinEgySpec, (A + dA/dt dt) / (gamma-1) rho^(gamma-1))
//         (:509-531)           + Injected_BH_Energy / Mass, capped at 5e9 K; the injection is consumed
//   dA/dt (:582-594)           (unew (gamma-1) / rho^(gamma-1) - A) / dt, floor -0.5 A / dt
// ---------------------------------------------------------------------------------------------
__global__ void k_cooling_sf(int nact, const int *__restrict__ act, int ngas,
                             const int *__restrict__ type, const double *__restrict__ mass,
                             const int *__restrict__ timebin, double timebase, double critdens,
                             double minegy, double u_to_temp, const double *__restrict__ density,
                             const double *__restrict__ entropy, double *__restrict__ dtentropy,
                             double *__restrict__ injected, int *__restrict__ flag_sink)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nact)
    return;
  const int i = act ? act[a] : a;
  if(i < 0 || i >= ngas || type[i] != 0)
    return;
  const int tb = timebin[i];
  const double dt = (tb ? (double) (1 << tb) : 0.0) * timebase;
  int flag = 1;
  if(density[i] >= critdens)
    flag = 0;
  if(mass[i] == 0)
    flag = 1;
  flag_sink[i] = !flag;
  if(flag == 1)
    {
      double unew = (entropy[i] + dtentropy[i] * dt) / SINK_GAMMA_MINUS1 * pow(density[i], SINK_GAMMA_MINUS1);
      if(unew < minegy)
        unew = minegy;
      const double inj = injected[i];
      if(inj)
        {
          if(mass[i] != 0)
            unew += inj / mass[i];
          if(u_to_temp * unew > 5.0e9)
            unew = 5.0e9 / u_to_temp;
          injected[i] = 0;
        }
      if(tb && dt > 0)
        {
          double d = (unew * SINK_GAMMA_MINUS1 / pow(density[i], SINK_GAMMA_MINUS1) - entropy[i]) / dt;
          if(d < -0.5 * entropy[i] / dt)
            d = -0.5 * entropy[i] / dt;
          dtentropy[i] = d;
        }
    }
}

extern "C" int ghip_cooling_and_starformation(ghip_ctx *ctx, double Timebase_interval,
                                              double CritPhysDensity_code, double MinEgySpec,
                                              double u_to_temp_fac, int *flag_sink_host)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  GCHK(sink_buffers(ctx));
  const int ng = ctx->ngas;
  if(ng == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  const int nact = ctx->nactive < 0 ? ng : ctx->nactive;
  GCHK(ghip_ensure(ctx, ctx->dflags, (size_t) ng * 4));
  HIPCHK(hipMemsetAsync(ctx->dflags.p, 0, (size_t) ng * 4, st));
  if(nact > 0)
    k_cooling_sf<<<cdiv(nact, 256), 256, 0, st>>>(
      nact, ctx->nactive < 0 ? nullptr : P<int>(ctx->act_host_idx), ng, P<int>(ctx->f[GHIP_F_TYPE]),
      P<double>(ctx->f[GHIP_F_MASS]), P<int>(ctx->f[GHIP_F_TIMEBIN]), Timebase_interval,
      CritPhysDensity_code, MinEgySpec, u_to_temp_fac, P<double>(ctx->f[GHIP_F_DENSITY]),
      P<double>(ctx->f[GHIP_F_ENTROPY]), P<double>(ctx->f[GHIP_F_DTENTROPY]),
      P<double>(ctx->bh_injected), P<int>(ctx->dflags));
  HIPCHK(hipGetLastError());
  if(flag_sink_host)
    HIPCHK(hipMemcpyAsync(flag_sink_host, ctx->dflags.p, (size_t) ng * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// the sink passes on a multi-GPU shard (GHIP_DD_SINK_DENSITY / _BH_EVALUATE / _BH_SWALLOW)
//
// The reference exports the few sink targets to the tasks whose domains their search spheres touch
// (density.c:384-470, blackhole.c:310-600) and adds the partial results back on the home task.
// Here every shard learns ALL sinks (96-byte records, an all-to-all-v that includes the sender),
// runs each pass for all of them over ITS OWN particles, and the per-sink partial sums are
// all-gathered and added in rank order on every shard: the h iteration then takes the same decisions
// everywhere without another exchange, and a victim, local to exactly one shard, sees every claim.
// ---------------------------------------------------------------------------------------------
int ghip_dd_sink_begin(ghip_ctx *ctx, int op)
{
  GHIP_JOIN(ctx);
  DDState &D = ctx->dd;
  const ghip_dd_sink_args &A = D.sink;
  if(A.nsink < 0 || (A.nsink > 0 && !A.sink_idx))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd sinks: bad arguments");
  if(op == GHIP_DD_SINK_DENSITY)
    {
      if(!A.dens || (A.nsink > 0 && (!A.hsml || !A.numngb || !A.bh_density || !A.bh_entropy || !A.bh_gasvel)))
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd sink density: bad arguments");
      if(!ctx->st.built)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd sink density: run GHIP_DD_DENSITY of this step first");
    }
  else
    {
      if(!A.bh || (A.nsink > 0 && !A.sink_id))
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd black holes: bad arguments");
      if(op == GHIP_DD_BH_EVALUATE && A.nsink > 0 && (!A.bh_mdot || !A.bh_density_in))
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd black holes: bad arguments");
      if(op == GHIP_DD_BH_SWALLOW && A.nsink > 0 &&
         (!A.sink_bh_mass || !A.acc_mass || !A.acc_bhmass || !A.acc_dustmass || !A.acc_momentum))
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd black holes: bad arguments");
      if(!ctx->gt.built)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd black holes: run GHIP_DD_GRAVITY of this step first");
    }
  return GHIP_OK;
}

// one density pass of the sinks still iterating, over this shard's own gas; the partial sums go
// to everybody
static int dd_sink_density_pass(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  const int ns = D.sk_total;
  GCHK(ghip_ensure(ctx, D.sk_part, (size_t) ns * 48));
  GCHK(ghip_ensure(ctx, D.sk_work, (size_t) ns * 12 + 64));
  double *dh = P<double>(D.sk_work);
  int *dslot = reinterpret_cast<int *>(dh + ns);
  GCHK(sink_density_pass(ctx, D.sink.dens, ns, P<SinkRec>(D.sk_all), dh, dslot, P<double>(D.sk_part),
                         D.sk_it, D.sk_ncur, 1));
  ghip_dd_set_allgather(D, D.sk_part.p, (size_t) ns * 48, &D.sk_parts);
  return 1;
}

// rank-ordered sum of the all-gathered partial planes [nranks][k][ns]
static int dd_sink_sum_parts(ghip_ctx *ctx, int k, std::vector<double> &parts, std::vector<double> &sums)
{
  DDState &D = ctx->dd;
  const size_t per = (size_t) k * D.sk_total;
  parts.resize(per * D.nranks);
  sums.assign(per, 0.0);
  HIPCHK(hipMemcpyAsync(parts.data(), D.sk_parts.p, per * D.nranks * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  for(int r = 0; r < D.nranks; r++)
    for(size_t q = 0; q < per; q++)
      sums[q] += parts[(size_t) r * per + q];
  return GHIP_OK;
}

int ghip_dd_sink_step(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  const ghip_dd_sink_args &A = D.sink;
  hipStream_t st = ctx->stream;
  const int op = D.op, nloc = A.nsink;
  if(D.phase == 0)
    {
      // this shard's sinks to every shard (itself included: the receive buffer is the rank-ordered
      // list of all sinks)
      std::vector<SinkRec> S;
      GCHK(gather_sinks(ctx, nloc, A.sink_idx, A.sink_id, op == GHIP_DD_BH_EVALUATE ? A.bh_mdot : nullptr,
                        op == GHIP_DD_BH_EVALUATE ? A.bh_density_in : nullptr, S));
      if(op == GHIP_DD_SINK_DENSITY)
        for(int a = 0; a < nloc; a++)
          S[a].h = A.hsml[a];
      GCHK(ghip_ensure(ctx, D.sk_send, (size_t) (nloc > 0 ? nloc : 1) * sizeof(SinkRec)));
      if(nloc > 0)
        HIPCHK(hipMemcpyAsync(D.sk_send.p, S.data(), (size_t) nloc * sizeof(SinkRec), hipMemcpyHostToDevice, st));
      HIPCHK(ghip_stream_sync(ctx, st));   // S goes out of scope
      int scount[GHIP_MAXRANKS], soff[GHIP_MAXRANKS];
      for(int r = 0; r < D.nranks; r++)
        {
          scount[r] = nloc;
          soff[r] = 0;
        }
      ghip_dd_set_alltoallv(D, D.sk_send.p, sizeof(SinkRec), scount, soff, &D.sk_all);
      D.phase = 1;
      return 1;
    }
  if(D.phase == 1)
    {
      D.sk_total = D.x.rtotal;
      D.sk_off = D.x.roff[D.rank];
      const int ns = D.sk_total;
      if(D.x.rcount[D.rank] != nloc)
        return ghip_fail(ctx, GHIP_ECOMM, "ghip_dd sinks: the exchange lost this shard's own records");
      if(op == GHIP_DD_BH_SWALLOW && A.counts)
        A.counts[0] = A.counts[1] = A.counts[2] = 0;
      if(ns == 0)
        {
          D.op = 0;
          return 0;
        }
      if(op == GHIP_DD_SINK_DENSITY)
        {
          std::vector<SinkRec> all(ns);
          HIPCHK(hipMemcpyAsync(all.data(), D.sk_all.p, (size_t) ns * sizeof(SinkRec), hipMemcpyDeviceToHost, st));
          HIPCHK(ghip_stream_sync(ctx, st));
          sink_iter_init(D.sk_it, ns);
          for(int a = 0; a < ns; a++)
            D.sk_it.h[a] = all[a].h;
          D.sk_ncur = ns;
          D.sk_iter = 0;
          D.phase = 2;
          return dd_sink_density_pass(ctx);
        }
      GCHK(sink_buffers(ctx));
      TreeDev &t = ctx->gt;
      const size_t n = (size_t) ctx->n;
      if(op == GHIP_DD_BH_EVALUATE)
        {
          k_bh_evaluate<<<ns, 64, 0, st>>>(ns, P<SinkRec>(D.sk_all), t.nelem, P<int4>(t.lk), P<double4>(t.cl),
                                           P<double>(ctx->sx), P<double>(ctx->sy), P<double>(ctx->sz),
                                           P<int>(t.perm), ctx->n, ctx->ngas, P<int>(ctx->f[GHIP_F_TYPE]),
                                           P<double>(ctx->f[GHIP_F_MASS]), P<double>(ctx->f[GHIP_F_VEL]),
                                           P<double>(ctx->f[GHIP_F_DENSITY]), bh_kparams(A.bh),
                                           P<unsigned int>(ctx->bh_swallow), P<double>(ctx->bh_injected));
          HIPCHK(hipGetLastError());
          HIPCHK(ghip_stream_sync(ctx, st));
          D.op = 0;
          return 0;
        }
      // GHIP_DD_BH_SWALLOW: BH masses of this shard's sinks by particle, then every sink's sweep
      GCHK(ghip_ensure(ctx, D.sk_part, (size_t) ns * 72));
      GCHK(ghip_ensure(ctx, D.sk_work, (n > 0 ? n : 1) * 12 + (size_t) (nloc > 0 ? nloc : 1) * 8 + 64));
      double *dpbh = P<double>(D.sk_work);
      int *dvict = reinterpret_cast<int *>(dpbh + n);
      double *dbh = reinterpret_cast<double *>(reinterpret_cast<char *>(D.sk_work.p) + ((n * 12 + 63) & ~(size_t) 63));
      HIPCHK(hipMemsetAsync(dpbh, 0, n * 12, st));
      if(nloc > 0)
        {
          HIPCHK(hipMemcpyAsync(dbh, A.sink_bh_mass, (size_t) nloc * 8, hipMemcpyHostToDevice, st));
          k_bh_scatter<<<cdiv(nloc, 64), 64, 0, st>>>(nloc, P<SinkRec>(D.sk_send), dbh, dpbh);
        }
      k_bh_swallow<<<ns, 64, 0, st>>>(ns, P<SinkRec>(D.sk_all), t.nelem, P<int4>(t.lk), P<double4>(t.cl),
                                      P<double>(ctx->sx), P<double>(ctx->sy), P<double>(ctx->sz),
                                      P<int>(t.perm), ctx->n, P<int>(ctx->f[GHIP_F_TYPE]),
                                      P<double>(ctx->f[GHIP_F_MASS]), P<double>(ctx->f[GHIP_F_VEL]), dpbh,
                                      bh_kparams(A.bh), P<unsigned int>(ctx->bh_swallow), dvict,
                                      P<double>(D.sk_part));
      if(n > 0)
        k_bh_zero_victims<<<cdiv((long long) n, 256), 256, 0, st>>>((int) n, dvict,
                                                                   P<double>(ctx->f[GHIP_F_MASS]));
      HIPCHK(hipGetLastError());
      ghip_dd_set_allgather(D, D.sk_part.p, (size_t) ns * 72, &D.sk_parts);
      D.phase = 2;
      return 1;
    }
  if(D.phase == 2 && op == GHIP_DD_SINK_DENSITY)
    {
      const int ns = D.sk_total;
      std::vector<double> parts, sums;
      GCHK(dd_sink_sum_parts(ctx, 6, parts, sums));
      D.sk_ncur = sink_iterate(A.dens, A.ngb_factor, ns, sums.data(), D.sk_ncur, D.sk_it);
      if(D.sk_ncur > 0)
        {
          const int maxiter = A.dens->MaxIter > 0 ? A.dens->MaxIter : 150;
          if(++D.sk_iter > maxiter)
            return ghip_fail(ctx, GHIP_ENOCONV, "sink density: %d sinks not converged after %d "
                             "h-iterations (reference: endrun(1155))", D.sk_ncur, maxiter);
          return dd_sink_density_pass(ctx);
        }
      const SinkIter &I = D.sk_it;
      for(int a = 0; a < nloc; a++)
        {
          const int g = D.sk_off + a;
          A.hsml[a] = I.h[g];
          A.numngb[a] = I.numngb[g];
          A.bh_density[a] = I.rho[g];
          A.bh_entropy[a] = I.entropy[g];
          for(int k = 0; k < 3; k++)
            A.bh_gasvel[3 * (size_t) a + k] = I.gasvel[3 * (size_t) g + k];
          HIPCHK(hipMemcpyAsync(P<double>(ctx->f[GHIP_F_HSML]) + A.sink_idx[a], &I.h[g], 8,
                                hipMemcpyHostToDevice, st));
        }
      HIPCHK(ghip_stream_sync(ctx, st));
      D.op = 0;
      return 0;
    }
  if(D.phase == 2 && op == GHIP_DD_BH_SWALLOW)
    {
      const int ns = D.sk_total;
      const size_t n = (size_t) ctx->n;
      std::vector<double> parts, sums;
      GCHK(dd_sink_sum_parts(ctx, 9, parts, sums));
      const int *dvict = reinterpret_cast<const int *>(P<double>(D.sk_work) + n);
      std::vector<int> vict(nloc > 0 ? nloc : 1);
      for(int a = 0; a < nloc; a++)
        HIPCHK(hipMemcpyAsync(&vict[a], dvict + A.sink_idx[a], 4, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      for(int a = 0; a < nloc; a++)
        {
          const int g = D.sk_off + a;
          A.acc_mass[a] = sums[g];
          A.acc_bhmass[a] = sums[(size_t) ns + g];
          A.acc_dustmass[a] = sums[2 * (size_t) ns + g];
          for(int k = 0; k < 3; k++)
            A.acc_momentum[3 * (size_t) a + k] = sums[(size_t) (3 + k) * ns + g];
          if(vict[a])
            A.sink_bh_mass[a] = 0;   // P[j].BH_Mass = 0 of a swallowed sink (blackhole.c:1274)
        }
      if(A.counts)   // what THIS shard's particles lost: the planes of this rank's block
        for(int k = 0; k < 3; k++)
          for(int g = 0; g < ns; g++)
            A.counts[k] += (long long) (parts[(size_t) D.rank * 9 * ns + (size_t) (6 + k) * ns + g] + 0.5);
      ctx->gt.built = false;   // masses changed: the trees' moments are stale
      ctx->st.built = false;
      D.op = 0;
      return 0;
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: the sink passes have no phase %d", D.phase);
}
