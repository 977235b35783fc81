// ghip_timefac.h -- integer-timeline factors shared by the drift and the kick kernels
// (driftfac.c:123-247: get_drift_factor / get_gravkick_factor / get_hydrokick_factor)
#ifndef GHIP_TIMEFAC_H
#define GHIP_TIMEFAC_H

#define GAMMA (7. / 5.)  // allvars.h:64

struct DriftK
{
  int time1;
  double timebase;
  int comoving;
  double logTimeBegin, logTimeMax;
  const double *drift, *gravkick, *hydrokick;  // DRIFT_TABLE_LENGTH entries each (device)
  double minhsml;
  int wrap;
  double boxsize;
};

#define DRIFT_TABLE_LENGTH 1000  // allvars.h:136

// driftfac.c:123-163 (and the identical :166-205, :208-247 for the kick tables)
__device__ __forceinline__ double d_table_factor(const double *__restrict__ tab, int time0,
                                                 int time1, const DriftK &k)
{
  double a1 = k.logTimeBegin + time0 * k.timebase;
  double a2 = k.logTimeBegin + time1 * k.timebase;
  double u1 = (a1 - k.logTimeBegin) / (k.logTimeMax - k.logTimeBegin) * DRIFT_TABLE_LENGTH;
  int i1 = (int) u1;
  if(i1 >= DRIFT_TABLE_LENGTH)
    i1 = DRIFT_TABLE_LENGTH - 1;
  double df1 = (i1 <= 1) ? u1 * tab[0] : tab[i1 - 1] + (tab[i1] - tab[i1 - 1]) * (u1 - i1);
  double u2 = (a2 - k.logTimeBegin) / (k.logTimeMax - k.logTimeBegin) * DRIFT_TABLE_LENGTH;
  int i2 = (int) u2;
  if(i2 >= DRIFT_TABLE_LENGTH)
    i2 = DRIFT_TABLE_LENGTH - 1;
  double df2 = (i2 <= 1) ? u2 * tab[0] : tab[i2 - 1] + (tab[i2] - tab[i2 - 1]) * (u2 - i2);
  return df2 - df1;
}

#endif
