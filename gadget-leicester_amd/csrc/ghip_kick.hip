// ghip_kick.hip -- "next" row N1: timestep criterion + kick on the resident
// particles, so that P/SphP need not leave HBM between force evaluations.
//
// Replaces the loop of advance_and_find_timesteps() (timestep.c:142-260) with get_timestep()
// (timestep.c:607-1123; flag == 0, TypeOfTimestepCriterion 0) and do_the_kick()
// (timestep.c:364-605) for the minimal periodic flag set -- no PMGRID long-range kick, no
// BLACK_HOLES / DUST / MAGNETIC terms -- and the per-type sums of
// find_dt_displacement_constraint() (timestep.c:1125-1224).  DoDynamicUpdate / force_kick_node
// (forcetree.c:1474-1651) has no counterpart: the device tree is rebuilt every step.  The host
// keeps the linked lists of the time bins (FirstInTimeBin...) -- it rebuilds them from the
// returned TimeBin[] as reconstruct_timebins() does.
// One thread per active particle, SoA planes: a streaming, HBM-bound kernel (per collisionless
// particle 60 B in + 32 B out, per gas particle 156 B in + 96 B out).
#include "ghip_internal.h"

// no a*b+c -> fma contraction in this file: the kick is rounding-for-rounding the host's
// arithmetic (sqrt and division are correctly rounded on the device), so velocities, entropies and
// the integer timeline come out identical to the CPU path; only pow() (MinEgySpec floor) may differ
// in the last bits.  A streaming kernel: the extra instructions are free.
#pragma clang fp contract(off)

#include "ghip_timefac.h"

#define GHIP_TIMEBINS 29                   // allvars.h:39
#define GHIP_TIMEBASE (1 << GHIP_TIMEBINS) // allvars.h:41
#define GAMMA_MINUS1 (GAMMA - 1)

struct KickK
{
  int ti_current;
  double timebase;
  int comoving;
  double time, hubble_a, fac1, fac2, fac3, a3inv, atime;   // timestep.c:52-63
  double errtol, courant, maxdt, mindt, dtdisp;
  double soft[6];
  int adaptive_hsml;
  int pmgrid;            // PMGRID: GravPM enters the criterion and VelPred (timestep.c:648-652, 511-513)
  double dt_gravkickB;   // timestep.c:66-72
  double minegy;
  unsigned int active;
  DriftK tab;   // gravkick / hydrokick tables (comoving)
};

__global__ void k_advance_timesteps(int nact, const int *__restrict__ act, int n, int ngas, KickK k,
                                    const int *__restrict__ type, double *__restrict__ vel,
                                    const double *__restrict__ gravaccel,
                                    const double *__restrict__ gravpm,
                                    const double *__restrict__ hydroaccel,
                                    double *__restrict__ velpred, double *__restrict__ entropy,
                                    double *__restrict__ dtentropy,
                                    const double *__restrict__ density,
                                    const double *__restrict__ hsml,
                                    const double *__restrict__ maxsignalvel,
                                    int *__restrict__ timebin, int *__restrict__ ti_begstep,
                                    int *__restrict__ err, int *errw, double *__restrict__ kick_dv,
                                    int *__restrict__ kick_flag)
{
// (errw: asynchronous mode, the pinned word the next synchronising call checks)
#define D_RAISE(code)                   \
  do                                    \
    {                                   \
      atomicMax(err, (code));           \
      if(errw)                          \
        *(volatile int *) errw = (code); \
    }                                   \
  while(0)
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= nact)
    return;
  const int i = act ? act[a] : a;
  const int ty = type[i];
  const bool gas = (ty == 0) && (i < ngas);
  double g[3], hy[3] = {0, 0, 0};
  for(int j = 0; j < 3; j++)
    g[j] = gravaccel[(size_t) j * n + i];
  if(gas)
    for(int j = 0; j < 3; j++)
      hy[j] = hydroaccel[(size_t) j * ngas + i];

  // ---- get_timestep, timestep.c:636-700, 947-1123 ----
  double ax = k.fac1 * g[0], ay = k.fac1 * g[1], az = k.fac1 * g[2];
  double pm[3] = {0, 0, 0};
  if(k.pmgrid)   // timestep.c:648-652
    {
      for(int j = 0; j < 3; j++)
        pm[j] = gravpm[(size_t) j * n + i];
      ax += k.fac1 * pm[0];
      ay += k.fac1 * pm[1];
      az += k.fac1 * pm[2];
    }
  if(ty == 0)
    {
      ax += k.fac2 * hy[0];
      ay += k.fac2 * hy[1];
      az += k.fac2 * hy[2];
    }
  double ac = sqrt(ax * ax + ay * ay + az * az);
  if(ac == 0)
    ac = 1.0e-30;
  double dt = sqrt(2 * k.errtol * k.atime * k.soft[ty] / ac);
  if(k.adaptive_hsml && ty == 0)   // timestep.c:740-743
    dt = sqrt(2 * k.errtol * k.atime * hsml[i] / 2.8 / ac);
  if(gas)
    {
      double dt_courant;
      if(k.comoving)
        dt_courant = 2 * k.courant * k.time * hsml[i] / (k.fac3 * maxsignalvel[i]);
      else
        dt_courant = 2 * k.courant * hsml[i] / maxsignalvel[i];
      if(dt_courant < dt)
        dt = dt_courant;
    }
  dt *= k.hubble_a;
  if(dt >= k.maxdt)
    dt = k.maxdt;
  if(dt >= k.dtdisp)
    dt = k.dtdisp;
  if(dt < k.mindt)
    {
      D_RAISE(888);   // timestep.c:1082
      return;
    }
  int ti_step = (int) (dt / k.timebase);
  if(!(ti_step > 0 && ti_step < GHIP_TIMEBASE))
    {
      D_RAISE(818);   // timestep.c:1119
      return;
    }
  // ---- advance_and_find_timesteps, timestep.c:146-175: power-of-two step, bin, synchronisation
  int ti_min = GHIP_TIMEBASE;
  while(ti_min > ti_step)
    ti_min >>= 1;
  ti_step = ti_min;
  if(ti_step == 1)
    {
      D_RAISE(112313);   // get_timestep_bin, timestep.c:1233
      return;
    }
  int bin = ti_step ? 31 - __clz(ti_step) : 0;
  const int binold = timebin[i];
  if(bin > binold && ((k.active >> bin) & 1u) == 0)
    {
      bin = binold;   // leave at old step if not synchronised
      ti_step = bin ? (1 << bin) : 0;
    }
  if(k.ti_current >= GHIP_TIMEBASE)
    {
      ti_step = 0;
      bin = 0;
    }
  if((GHIP_TIMEBASE - k.ti_current) < ti_step)
    {
      D_RAISE(888);   // timestep.c:171
      return;
    }
  timebin[i] = bin;
  const int ti_step_old = binold ? (1 << binold) : 0;
  const int tb0 = ti_begstep[i];
  const int tstart = tb0 + ti_step_old / 2;           // midpoint of old step
  const int tend = tb0 + ti_step_old + ti_step / 2;   // midpoint of new step
  const int tcurrent = tb0 + ti_step_old;
  ti_begstep[i] = tcurrent;

  // ---- do_the_kick, timestep.c:378-395 ----
  double dt_entr, dt_gravkick, dt_hydrokick, dt_gravkick2, dt_hydrokick2;
  if(k.comoving)
    {
      dt_entr = (tend - tstart) * k.timebase;
      dt_gravkick = d_table_factor(k.tab.gravkick, tstart, tend, k.tab);
      dt_hydrokick = d_table_factor(k.tab.hydrokick, tstart, tend, k.tab);
      dt_gravkick2 = d_table_factor(k.tab.gravkick, tcurrent, tend, k.tab);
      dt_hydrokick2 = d_table_factor(k.tab.hydrokick, tcurrent, tend, k.tab);
    }
  else
    {
      dt_entr = dt_gravkick = dt_hydrokick = (tend - tstart) * k.timebase;
      dt_gravkick2 = dt_hydrokick2 = (tend - tcurrent) * k.timebase;
    }
  double v[3];
  for(int j = 0; j < 3; j++)
    v[j] = vel[(size_t) j * n + i] + g[j] * dt_gravkick;   // timestep.c:413-424
  if(gas)
    {
      for(int j = 0; j < 3; j++)
        {
          v[j] += hy[j] * dt_hydrokick;   // timestep.c:491-494
          double vp = v[j] - dt_gravkick2 * g[j] - dt_hydrokick2 * hy[j];
          if(k.pmgrid)
            vp += pm[j] * k.dt_gravkickB;   // timestep.c:511-513
          velpred[(size_t) j * ngas + i] = vp;
        }
      double A = entropy[i], dA = dtentropy[i];
      if(dA * dt_entr > -0.5 * A)   // timestep.c:553-557
        A += dA * dt_entr;
      else
        A *= 0.5;
      if(k.minegy != 0)   // timestep.c:574-583
        {
          double minentropy = k.minegy * GAMMA_MINUS1 / pow(density[i] * k.a3inv, GAMMA_MINUS1);
          if(A < minentropy)
            {
              A = minentropy;
              dA = 0;
            }
        }
      dt_entr = (bin ? (1 << bin) : 0) / 2 * k.timebase;   // timestep.c:590-593
      if(A + dA * dt_entr < 0.5 * A)
        dA = -0.5 * A / dt_entr;
      entropy[i] = A;
      dtentropy[i] = dA;
    }
  // a kept tree (ghip_set_dynamic_tree): what force_kick_node(i, dv) receives, timestep.c:584-588
  if(kick_dv)
    {
      for(int j = 0; j < 3; j++)
        kick_dv[(size_t) j * n + i] = g[j] * dt_gravkick + (gas ? hy[j] * dt_hydrokick : 0.0);
      kick_flag[i] = 1;
    }
  for(int j = 0; j < 3; j++)
    vel[(size_t) j * n + i] = v[j];
}

#undef D_RAISE

// TimeBinCount[] / TimeBinCountSph[] (allvars.h:337-338) recounted over all particles
__global__ void k_timebin_histogram(int n, const int *__restrict__ type,
                                    const int *__restrict__ timebin,
                                    unsigned long long *__restrict__ out)
{
  __shared__ unsigned int h[64];
  if(threadIdx.x < 64)
    h[threadIdx.x] = 0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    {
      int b = timebin[i] & 31;
      atomicAdd(&h[b], 1u);
      if(type[i] == 0)
        atomicAdd(&h[32 + b], 1u);
    }
  __syncthreads();
  if(threadIdx.x < 64 && h[threadIdx.x])
    atomicAdd(out + threadIdx.x, (unsigned long long) h[threadIdx.x]);
}

extern "C" int ghip_timebin_counts(ghip_ctx *ctx, long long *TimeBinCount, long long *TimeBinCountSph)
{
  if(!ctx)
    return GHIP_EINVAL;
  if(TimeBinCount)
    memset(TimeBinCount, 0, 32 * sizeof(long long));
  if(TimeBinCountSph)
    memset(TimeBinCountSph, 0, 32 * sizeof(long long));
  const int n = ctx->n;
  if(n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  // (its own 64 words behind the run's accumulated counters)
  unsigned long long *dhist = P<unsigned long long>(ctx->run_acc) + 16;
  HIPCHK(hipMemsetAsync(dhist, 0, 64 * 8, st));
  k_timebin_histogram<<<cdiv(n, 256), 256, 0, st>>>(n, P<int>(ctx->f[GHIP_F_TYPE]),
                                                    P<int>(ctx->f[GHIP_F_TIMEBIN]), dhist);
  HIPCHK(hipGetLastError());
  unsigned long long hist[64];
  HIPCHK(hipMemcpyAsync(hist, dhist, 64 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int b = 0; b < 32; b++)
    {
      if(TimeBinCount)
        TimeBinCount[b] = (long long) hist[b];
      if(TimeBinCountSph)
        TimeBinCountSph[b] = (long long) hist[32 + b];
    }
  return ghip_check_device_errors(ctx);
}

extern "C" int ghip_advance_timesteps(ghip_ctx *ctx, const ghip_kick_params *p,
                                      long long *TimeBinCount, long long *TimeBinCountSph)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p)
    return GHIP_EINVAL;
  if(p->ComovingIntegrationOn && (!p->GravKickTable || !p->HydroKickTable))
    return ghip_fail(ctx, GHIP_EINVAL,
                     "ghip_advance_timesteps: comoving kicks need the factor tables");
  if(!(p->Timebase_interval > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_advance_timesteps: Timebase_interval must be > 0");
  int n = ctx->n, ng = ctx->ngas;
  if(TimeBinCount)
    memset(TimeBinCount, 0, 32 * sizeof(long long));
  if(TimeBinCountSph)
    memset(TimeBinCountSph, 0, 32 * sizeof(long long));
  if(n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  KickK k;
  memset(&k, 0, sizeof(k));
  k.ti_current = p->Ti_Current;
  k.timebase = p->Timebase_interval;
  k.comoving = p->ComovingIntegrationOn;
  k.time = p->Time;
  if(k.comoving)
    {
      // timestep.c:52-60, evaluated on the host exactly as the reference does
      k.fac1 = 1 / (p->Time * p->Time);
      k.fac2 = 1 / pow(p->Time, 3 * GAMMA - 2);
      k.fac3 = pow(p->Time, 3 * (1 - GAMMA) / 2.0);
      k.hubble_a = p->hubble_a;
      k.a3inv = 1 / (p->Time * p->Time * p->Time);
      k.atime = p->Time;
    }
  else
    k.fac1 = k.fac2 = k.fac3 = k.hubble_a = k.a3inv = k.atime = 1;
  k.errtol = p->ErrTolIntAccuracy;
  k.courant = p->CourantFac;
  k.maxdt = p->MaxSizeTimestep;
  k.mindt = p->MinSizeTimestep;
  k.dtdisp = p->dt_displacement;
  for(int t = 0; t < 6; t++)
    k.soft[t] = p->SofteningTable[t];
  k.adaptive_hsml = p->AdaptiveGravsoftForGasHsml;
  k.pmgrid = p->pmgrid;
  k.dt_gravkickB = p->dt_gravkickB;
  k.minegy = p->MinEgySpec;
  k.active = p->TimeBinActive;
  k.tab.timebase = p->Timebase_interval;
  k.tab.logTimeBegin = p->logTimeBegin;
  k.tab.logTimeMax = p->logTimeMax;
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  int *derr = reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 40);
  HIPCHK(hipMemsetAsync(derr, 0, 4, st));
  GCHK(ghip_ensure(ctx, ctx->stage, (2 * DRIFT_TABLE_LENGTH + 64) * 8));
  double *d = P<double>(ctx->stage);
  if(k.comoving)
    {
      HIPCHK(hipMemcpyAsync(d, p->GravKickTable, DRIFT_TABLE_LENGTH * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d + DRIFT_TABLE_LENGTH, p->HydroKickTable, DRIFT_TABLE_LENGTH * 8,
                            hipMemcpyHostToDevice, st));
      k.tab.gravkick = d;
      k.tab.hydrokick = d + DRIFT_TABLE_LENGTH;
    }
  const int nact = ctx->nactive < 0 ? n : ctx->nactive;
  const int *act = ctx->nactive < 0 ? nullptr : P<int>(ctx->act_host_idx);
  HIPCHK(hipEventRecord(ctx->evp[12], st));
  // with a kept tree the kicks are recorded and handed to the nodes (force_kick_node inside
  // do_the_kick, force_finish_kick_nodes at the end of advance_and_find_timesteps, timestep.c:256-263)
  const bool record = ctx->dyn_on && ctx->dyn_valid && ctx->dyn.n == n;
  if(record)
    {
      GCHK(ghip_ensure(ctx, ctx->kick_dv, 4 * (size_t) n * 8));
      GCHK(ghip_ensure(ctx, ctx->kick_flag, (size_t) n * 4));
      HIPCHK(hipMemsetAsync(ctx->kick_flag.p, 0, (size_t) n * 4, st));
    }
  if(nact > 0)
    k_advance_timesteps<<<cdiv(nact, 256), 256, 0, st>>>(
      nact, act, n, ng, k, P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->f[GHIP_F_VEL]),
      P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<double>(ctx->f[GHIP_F_GRAVPM]),
      P<double>(ctx->f[GHIP_F_HYDROACCEL]),
      P<double>(ctx->f[GHIP_F_VELPRED]), P<double>(ctx->f[GHIP_F_ENTROPY]),
      P<double>(ctx->f[GHIP_F_DTENTROPY]), P<double>(ctx->f[GHIP_F_DENSITY]),
      P<double>(ctx->f[GHIP_F_HSML]), P<double>(ctx->f[GHIP_F_MAXSIGNALVEL]),
      P<int>(ctx->f[GHIP_F_TIMEBIN]), P<int>(ctx->f[GHIP_F_TI_BEGSTEP]), derr,
      ctx->async ? ghip_errword(ctx, GHIP_ERRW_TIMESTEP) : nullptr,
      record ? P<double>(ctx->kick_dv) : nullptr, record ? P<int>(ctx->kick_flag) : nullptr);
  if(record && nact > 0)
    GCHK(ghip_dyn_kick_recorded(ctx));
  HIPCHK(hipEventRecord(ctx->evp[13], st));
  HIPCHK(hipGetLastError());
  const bool want_counts = TimeBinCount || TimeBinCountSph;
  if(ctx->async && !want_counts)
    return GHIP_OK;   // (a failed criterion is reported by the next call that synchronises)
  int herr = 0;
  HIPCHK(hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, st));
  if(want_counts)
    {
      int rc = ghip_timebin_counts(ctx, TimeBinCount, TimeBinCountSph);   // (synchronises)
      if(rc != GHIP_OK && rc != GHIP_ETIMESTEP)
        return rc;
    }
  else
    HIPCHK(ghip_stream_sync(ctx, st));
  if(herr)
    {
      ghip_fail(ctx, GHIP_ETIMESTEP,
                "ghip_advance_timesteps: the reference stops here with endrun(%d) "
                "(timestep.c:171/1082: 888, :1119: 818, :1233: 112313)",
                herr);
      ctx->timestep_endrun = herr;
      *ghip_errword(ctx, GHIP_ERRW_TIMESTEP) = 0;
      return GHIP_ETIMESTEP;
    }
  ctx->timestep_endrun = 0;
  return GHIP_OK;
}

extern "C" int ghip_timestep_endrun_code(const ghip_ctx *ctx)
{
  return ctx ? ctx->timestep_endrun : 0;
}

// ---- per-type sums of find_dt_displacement_constraint (timestep.c:1140-1156) -----------------
// stage 1: one partial per block and type (fixed tree order inside the block); stage 2: the
// partials of a type are added in block order by one thread -- deterministic.
__global__ void k_vel_moments(int n, const int *__restrict__ type, const double *__restrict__ vel,
                              const double *__restrict__ mass, double *__restrict__ pv2,
                              double *__restrict__ pmin, unsigned int *__restrict__ pcnt)
{
  __shared__ double sv[256], sm[256];
  __shared__ unsigned int sc[256];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int ty = -1;
  double v2 = 0, m = 1.0e30;
  if(i < n)
    {
      ty = type[i];
      double a = vel[i], b = vel[(size_t) n + i], c = vel[2 * (size_t) n + i];
      v2 = a * a + b * b + c * c;
      double mi = mass[i];
      if(mi > 0)
        m = mi;
    }
  for(int t = 0; t < 6; t++)
    {
      sv[threadIdx.x] = (ty == t) ? v2 : 0.0;
      sm[threadIdx.x] = (ty == t) ? m : 1.0e30;
      sc[threadIdx.x] = (ty == t) ? 1u : 0u;
      __syncthreads();
      for(unsigned int s = 128; s > 0; s >>= 1)
        {
          if(threadIdx.x < s)
            {
              sv[threadIdx.x] += sv[threadIdx.x + s];
              sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + s]);
              sc[threadIdx.x] += sc[threadIdx.x + s];
            }
          __syncthreads();
        }
      if(threadIdx.x == 0)
        {
          pv2[(size_t) t * gridDim.x + blockIdx.x] = sv[0];
          pmin[(size_t) t * gridDim.x + blockIdx.x] = sm[0];
          pcnt[(size_t) t * gridDim.x + blockIdx.x] = sc[0];
        }
      __syncthreads();
    }
}

__global__ void k_vel_moments_final(int nblocks, const double *__restrict__ pv2,
                                    const double *__restrict__ pmin,
                                    const unsigned int *__restrict__ pcnt, double *__restrict__ out)
{
  int t = threadIdx.x;
  if(t >= 6)
    return;
  double v = 0, m = 1.0e30;
  unsigned long long c = 0;
  for(int b = 0; b < nblocks; b++)
    {
      v += pv2[(size_t) t * nblocks + b];
      m = fmin(m, pmin[(size_t) t * nblocks + b]);
      c += pcnt[(size_t) t * nblocks + b];
    }
  out[t] = v;
  out[6 + t] = m;
  out[12 + t] = (double) c;
}

extern "C" int ghip_velocity_moments(ghip_ctx *ctx, double v2sum[6], double min_mass[6],
                                     long long count[6])
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !v2sum || !min_mass || !count)
    return GHIP_EINVAL;
  int n = ctx->n;
  for(int t = 0; t < 6; t++)
    {
      v2sum[t] = 0;
      min_mass[t] = 1.0e30;
      count[t] = 0;
    }
  if(n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  int nb = cdiv(n, 256);
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nb * 6 * (8 + 8 + 4) + 64 * 8));
  double *pv2 = P<double>(ctx->stage), *pmin = pv2 + (size_t) 6 * nb, *out = pmin + (size_t) 6 * nb;
  unsigned int *pcnt = reinterpret_cast<unsigned int *>(out + 24);
  k_vel_moments<<<nb, 256, 0, st>>>(n, P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->f[GHIP_F_VEL]),
                                    P<double>(ctx->f[GHIP_F_MASS]), pv2, pmin, pcnt);
  k_vel_moments_final<<<1, 64, 0, st>>>(nb, pv2, pmin, pcnt, out);
  HIPCHK(hipGetLastError());
  double h[18];
  HIPCHK(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int t = 0; t < 6; t++)
    {
      v2sum[t] = h[t];
      min_mass[t] = h[6 + t];
      count[t] = (long long) h[12 + t];
    }
  return GHIP_OK;
}

// The long-range kick at the end of a PM step (timestep.c:269-345): every particle gets
// Vel += GravPM * dt_gravkick; the predicted velocity of every gas particle is rebuilt from the new
// Vel at the current time (its own step's mid-point kicks + the new PM half-step dt_gravkickB).
__global__ void k_pm_kick(int n, int ngas, int ti_current, DriftK tab, double dt_gravkick,
                          double dt_gravkickB, const int *__restrict__ type,
                          const int *__restrict__ timebin, const int *__restrict__ ti_begstep,
                          double *__restrict__ vel, const double *__restrict__ gravaccel,
                          const double *__restrict__ gravpm, const double *__restrict__ hydroaccel,
                          double *__restrict__ velpred)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  double pm[3], v[3];
  for(int j = 0; j < 3; j++)
    {
      pm[j] = gravpm[(size_t) j * n + i];
      v[j] = vel[(size_t) j * n + i] + pm[j] * dt_gravkick;   // timestep.c:313
      vel[(size_t) j * n + i] = v[j];
    }
  if(type[i] == 0 && i < ngas)
    {
      const int tb = timebin[i];
      const int dt_step = tb ? (1 << tb) : 0;   // timestep.c:317
      const int t0 = ti_begstep[i];
      double dt_gravkickA, dt_hydrokick;
      if(tab.comoving)   // timestep.c:319-325
        {
          dt_gravkickA = d_table_factor(tab.gravkick, t0, ti_current, tab) -
                         d_table_factor(tab.gravkick, t0, t0 + dt_step / 2, tab);
          dt_hydrokick = d_table_factor(tab.hydrokick, t0, ti_current, tab) -
                         d_table_factor(tab.hydrokick, t0, t0 + dt_step / 2, tab);
        }
      else
        dt_gravkickA = dt_hydrokick = (ti_current - (t0 + dt_step / 2)) * tab.timebase;
      for(int j = 0; j < 3; j++)   // timestep.c:330-333
        velpred[(size_t) j * ngas + i] = v[j] + gravaccel[(size_t) j * n + i] * dt_gravkickA +
                                         hydroaccel[(size_t) j * ngas + i] * dt_hydrokick +
                                         pm[j] * dt_gravkickB;
    }
}

extern "C" int ghip_pm_kick(ghip_ctx *ctx, const ghip_pmkick_params *p)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p)
    return GHIP_EINVAL;
  if(p->ComovingIntegrationOn && (!p->GravKickTable || !p->HydroKickTable))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_pm_kick: comoving integration needs the kick tables");
  const int n = ctx->n, ng = ctx->ngas;
  if(n == 0)
    return GHIP_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DriftK tab;
  memset(&tab, 0, sizeof(tab));
  tab.timebase = p->Timebase_interval;
  tab.comoving = p->ComovingIntegrationOn;
  tab.logTimeBegin = p->logTimeBegin;
  tab.logTimeMax = p->logTimeMax;
  if(p->ComovingIntegrationOn)
    {
      GCHK(ghip_ensure(ctx, ctx->stage, 3 * DRIFT_TABLE_LENGTH * 8));
      double *d = P<double>(ctx->stage);
      HIPCHK(hipMemcpyAsync(d + DRIFT_TABLE_LENGTH, p->GravKickTable, DRIFT_TABLE_LENGTH * 8,
                            hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d + 2 * DRIFT_TABLE_LENGTH, p->HydroKickTable, DRIFT_TABLE_LENGTH * 8,
                            hipMemcpyHostToDevice, st));
      tab.gravkick = d + DRIFT_TABLE_LENGTH;
      tab.hydrokick = d + 2 * DRIFT_TABLE_LENGTH;
    }
  k_pm_kick<<<cdiv(n, 256), 256, 0, st>>>(
    n, ng, p->Ti_Current, tab, p->dt_gravkick, p->dt_gravkickB, P<int>(ctx->f[GHIP_F_TYPE]),
    P<int>(ctx->f[GHIP_F_TIMEBIN]), P<int>(ctx->f[GHIP_F_TI_BEGSTEP]), P<double>(ctx->f[GHIP_F_VEL]),
    P<double>(ctx->f[GHIP_F_GRAVACCEL]), P<double>(ctx->f[GHIP_F_GRAVPM]),
    P<double>(ctx->f[GHIP_F_HYDROACCEL]), P<double>(ctx->f[GHIP_F_VELPRED]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}
