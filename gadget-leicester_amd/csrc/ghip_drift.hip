// ghip_drift.hip -- drift of the resident particle set to the current time (pre-condition of
// the force path).
//
// Replaces drift_particle() (predict.c:129-259) and do_box_wrapping() (predict.c:282-310).  The
// reference drifts particles lazily from inside the tree walks and neighbour searches
// (forcetree.c:1911-1912, ngb.c:57-58); the device works on a snapshot, so everything is drifted
// in one streaming pass before the tree is built.  One thread per particle, SoA planes, fully
// coalesced: an HBM-bound kernel (56 B in + 24 B out per collisionless particle, 208 B in + 96 B
// out per gas particle).
#include "ghip_internal.h"

#include "ghip_timefac.h"

__global__ void k_drift(int n, int ngas, DriftK k, double *__restrict__ pos,
                        const double *__restrict__ vel, const int *__restrict__ type,
                        int *__restrict__ ti_current, const int *__restrict__ timebin,
                        const int *__restrict__ ti_begstep, const double *__restrict__ gravaccel,
                        const double *__restrict__ gravpm,
                        double *__restrict__ velpred, const double *__restrict__ hydroaccel,
                        double *__restrict__ density, double *__restrict__ hsml,
                        const double *__restrict__ divvel, const double *__restrict__ entropy,
                        const double *__restrict__ dtentropy, double *__restrict__ pressure,
                        int *__restrict__ err, int *errw)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int time0 = ti_current[i];
  if(k.time1 < time0)
    {
      atomicMax(err, 12);  // predict.c:148-152: endrun(12)
      if(errw)             // asynchronous mode: the pinned word the next synchronising call checks
        *(volatile int *) errw = 12;
      return;
    }
  if(k.time1 != time0)
    {
      double dt_drift, dt_gravkick, dt_hydrokick;
      if(k.comoving)
        {
          dt_drift = d_table_factor(k.drift, time0, k.time1, k);
          dt_gravkick = d_table_factor(k.gravkick, time0, k.time1, k);
          dt_hydrokick = d_table_factor(k.hydrokick, time0, k.time1, k);
        }
      else
        dt_drift = dt_gravkick = dt_hydrokick = (k.time1 - time0) * k.timebase;

      for(int j = 0; j < 3; j++)
        pos[(size_t) j * n + i] += vel[(size_t) j * n + i] * dt_drift;

      if(i < ngas && type[i] == 0)
        {
          for(int j = 0; j < 3; j++)
            {
              double g = gravaccel[(size_t) j * n + i];
              if(gravpm)   // predict.c:181-184 (PMGRID)
                g = g + gravpm[(size_t) j * n + i];
              velpred[(size_t) j * ngas + i] +=
                g * dt_gravkick + hydroaccel[(size_t) j * ngas + i] * dt_hydrokick;
            }
          double dv = divvel[i];
          density[i] *= exp(-dv * dt_drift);
          double h = hsml[i] * exp(0.333333333333 * dv * dt_drift);
          if(h < k.minhsml)
            h = k.minhsml;
          hsml[i] = h;
          int tb = timebin[i];
          int dt_step = (tb ? (1 << tb) : 0);
          double dt_entr = (k.time1 - (ti_begstep[i] + dt_step / 2)) * k.timebase;
          pressure[i] = (entropy[i] + dtentropy[i] * dt_entr) * pow(density[i], GAMMA);
        }
      ti_current[i] = k.time1;
    }
  if(k.wrap)
    {
      // predict.c:299-307
      for(int j = 0; j < 3; j++)
        {
          double x = pos[(size_t) j * n + i];
          while(x < 0)
            x += k.boxsize;
          while(x >= k.boxsize)
            x -= k.boxsize;
          pos[(size_t) j * n + i] = x;
        }
    }
}

extern "C" int ghip_drift(ghip_ctx *ctx, const ghip_drift_params *p)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p)
    return GHIP_EINVAL;
  if(p->ComovingIntegrationOn && (!p->DriftTable || !p->GravKickTable || !p->HydroKickTable))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_drift: comoving drift needs the three factor tables");
  int n = ctx->n, ng = ctx->ngas;
  if(n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  DriftK k;
  memset(&k, 0, sizeof(k));
  k.time1 = p->time1;
  k.timebase = p->Timebase_interval;
  k.comoving = p->ComovingIntegrationOn;
  k.logTimeBegin = p->logTimeBegin;
  k.logTimeMax = p->logTimeMax;
  k.minhsml = p->MinGasHsml;
  k.wrap = p->box_wrap;
  k.boxsize = p->BoxSize;
  GCHK(ghip_ensure(ctx, ctx->counters, 64 * 8));
  int *derr = reinterpret_cast<int *>(P<unsigned long long>(ctx->counters) + 40);
  HIPCHK(hipMemsetAsync(derr, 0, 4, st));
  if(k.comoving)
    {
      GCHK(ghip_ensure(ctx, ctx->stage, 3 * DRIFT_TABLE_LENGTH * 8));
      double *d = P<double>(ctx->stage);
      HIPCHK(hipMemcpyAsync(d, p->DriftTable, DRIFT_TABLE_LENGTH * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d + DRIFT_TABLE_LENGTH, p->GravKickTable, DRIFT_TABLE_LENGTH * 8,
                            hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d + 2 * DRIFT_TABLE_LENGTH, p->HydroKickTable, DRIFT_TABLE_LENGTH * 8,
                            hipMemcpyHostToDevice, st));
      k.drift = d;
      k.gravkick = d + DRIFT_TABLE_LENGTH;
      k.hydrokick = d + 2 * DRIFT_TABLE_LENGTH;
    }
  k_drift<<<cdiv(n, 256), 256, 0, st>>>(
    n, ng, k, P<double>(ctx->f[GHIP_F_POS]), P<double>(ctx->f[GHIP_F_VEL]),
    P<int>(ctx->f[GHIP_F_TYPE]), P<int>(ctx->f[GHIP_F_TI_CURRENT]),
    P<int>(ctx->f[GHIP_F_TIMEBIN]), P<int>(ctx->f[GHIP_F_TI_BEGSTEP]),
    P<double>(ctx->f[GHIP_F_GRAVACCEL]),
    p->pmgrid ? P<double>(ctx->f[GHIP_F_GRAVPM]) : (const double *) nullptr,
    P<double>(ctx->f[GHIP_F_VELPRED]),
    P<double>(ctx->f[GHIP_F_HYDROACCEL]), P<double>(ctx->f[GHIP_F_DENSITY]),
    P<double>(ctx->f[GHIP_F_HSML]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    P<double>(ctx->f[GHIP_F_ENTROPY]), P<double>(ctx->f[GHIP_F_DTENTROPY]),
    P<double>(ctx->f[GHIP_F_PRESSURE]), derr, ctx->async ? ghip_errword(ctx, GHIP_ERRW_DRIFT) : nullptr);
  HIPCHK(hipGetLastError());
  ctx->gt.built = false;  // positions moved: the trees are stale
  ctx->st.built = false;
  if(ctx->async)
    return GHIP_OK;   // (a particle ahead of time1 is reported by the next call that synchronises)
  int herr = 0;
  HIPCHK(hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  if(herr)
    return ghip_fail(ctx, GHIP_EINVAL,
                     "ghip_drift: a particle is ahead of time1 (reference: endrun(12), predict.c:148)");
  return GHIP_OK;
}
