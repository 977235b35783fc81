// ghip_shard.hip -- multi-GPU target sharding: pack / unpack of per-target results.
// (slice layout: ghip_shard_range in ghip_internal.h -- buckets of 64 curve-consecutive targets
// dealt round-robin to the ranks)
//
// Replaces the reference's MPI export rounds (gravtree.c:175-339, density.c:193-389,
// hydra.c:274-526: "export targets that touched a pseudo-particle, walk remotely, send partial
// sums back").  With 288 GB of HBM per GPU every rank keeps ALL sources and builds the same tree
// (the whole 256^3+256^3 configuration is ~4 GB), so no partial sums ever cross a link: the
// active targets, ordered along the space-filling curve, are cut into `nranks` contiguous
// equal-count slices, each rank evaluates its slice, and the finished per-target results are
// all-gathered (RCCL over xGMI, driven by the caller: torch.distributed or ncclAllGather on
// ghip_stream()).  The slices are padded to a common length `per` so that ONE fixed-size
// all-gather per phase suffices: buffer layout [rank][width][per] doubles.
//   group 0 gravity : ax, ay, az (G-less), ninteractions, -            width 4
//   group 1 density : hsml, numngb, density, dhsmlfac, divvel, curlvel, pressure   width 7
//   group 2 hydro   : ax, ay, az, dtentropy, maxsignalvel             width 5
#include "ghip_internal.h"

static const int kWidth[3] = {4, 7, 5};

static void slice_of(int nt, int nranks, int rank, int *per, int *lo, int *cnt)
{
  ghip_shard_range(nt, nranks, rank, lo, cnt, per);
}

struct ShardLo
{
  int lo[GHIP_MAXRANKS + 1];
};

extern "C" int ghip_shard_count(ghip_ctx *ctx, int gas, int *per, int *mine)
{
  if(!ctx || !per || !mine)
    return GHIP_EINVAL;
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_shard_count: no tree");
  GCHK(ghip_build_target_lists(ctx));
  int lo;
  slice_of(gas ? ctx->nt_gas : ctx->nt_grav, ctx->shard_n, ctx->shard_rank, per, &lo, mine);
  return GHIP_OK;
}

__global__ void k_shard_pack(int group, int cnt, int per, const int *__restrict__ tgt,
                             const int *__restrict__ perm, int n, int ngas,
                             const double *__restrict__ f0, const double *__restrict__ f1,
                             const double *__restrict__ f2, const double *__restrict__ f3,
                             const double *__restrict__ f4, const double *__restrict__ f5,
                             const double *__restrict__ f6, const int *__restrict__ icost,
                             double *__restrict__ buf)
{
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a >= per)
    return;
  bool ok = a < cnt;
  int i = ok ? perm[tgt[a]] : 0;
  if(group == 0)
    {
      buf[a] = ok ? f0[i] : 0;
      buf[(size_t) per + a] = ok ? f0[(size_t) n + i] : 0;
      buf[2 * (size_t) per + a] = ok ? f0[2 * (size_t) n + i] : 0;
      buf[3 * (size_t) per + a] = ok ? (double) icost[i] : 0;
    }
  else if(group == 1)
    {
      const double *src[7] = {f0, f1, f2, f3, f4, f5, f6};
      for(int c = 0; c < 7; c++)
        buf[(size_t) c * per + a] = ok ? src[c][i] : 0;
    }
  else
    {
      buf[a] = ok ? f0[i] : 0;
      buf[(size_t) per + a] = ok ? f0[(size_t) ngas + i] : 0;
      buf[2 * (size_t) per + a] = ok ? f0[2 * (size_t) ngas + i] : 0;
      buf[3 * (size_t) per + a] = ok ? f1[i] : 0;
      buf[4 * (size_t) per + a] = ok ? f2[i] : 0;
    }
}

__global__ void k_shard_unpack(int group, int nt, int per, int width, int skip_rank, int nranks,
                               ShardLo L,
                               const int *__restrict__ tgt, const int *__restrict__ perm, int n,
                               int ngas, const double *__restrict__ buf, double *__restrict__ f0,
                               double *__restrict__ f1, double *__restrict__ f2,
                               double *__restrict__ f3, double *__restrict__ f4,
                               double *__restrict__ f5, double *__restrict__ f6,
                               int *__restrict__ icost, double *__restrict__ gp,
                               double *__restrict__ gq)
{
  int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if(ti >= nt)
    return;
  int r = 0;
  while(r + 1 < nranks && ti >= L.lo[r + 1])
    r++;
  int a = ti - L.lo[r];
  if(r == skip_rank)
    return;  // own slice is already in place
  const double *b = buf + (size_t) r * width * per;
  int s = tgt[ti];
  int i = perm[s];
  if(group == 0)
    {
      f0[i] = b[a];
      f0[(size_t) n + i] = b[(size_t) per + a];
      f0[2 * (size_t) n + i] = b[2 * (size_t) per + a];
      icost[i] = (int) b[3 * (size_t) per + a];
    }
  else if(group == 1)
    {
      double v[7];
      for(int c = 0; c < 7; c++)
        v[c] = b[(size_t) c * per + a];
      f0[i] = v[0];
      f1[i] = v[1];
      f2[i] = v[2];
      f3[i] = v[3];
      f4[i] = v[4];
      f5[i] = v[5];
      f6[i] = v[6];
      // keep the gas records every later walk reads (ghip_sph.hip) in step
      gp[(size_t) 8 * s + 7] = v[0];
      double *q = gq + (size_t) 8 * s;
      q[0] = v[6];
      q[1] = v[2];
      q[2] = v[3];
      q[3] = v[4];
      q[4] = v[5];
    }
  else
    {
      f0[i] = b[a];
      f0[(size_t) ngas + i] = b[(size_t) per + a];
      f0[2 * (size_t) ngas + i] = b[2 * (size_t) per + a];
      f1[i] = b[3 * (size_t) per + a];
      f2[i] = b[4 * (size_t) per + a];
    }
}

// writes this rank's slice: [width][per] doubles at dev_buf (device memory)
extern "C" int ghip_shard_pack(ghip_ctx *ctx, int group, void *dev_buf)
{
  if(!ctx || group < 0 || group > 2 || !dev_buf)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_shard_pack: bad arguments");
  if(group == 0)
    GHIP_JOIN(ctx);   // the SPH groups may be exchanged underneath a gravity pair in flight
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_shard_pack: no tree");
  GCHK(ghip_build_target_lists(ctx));
  bool gas = group != 0;
  int per, lo, cnt;
  slice_of(gas ? ctx->nt_gas : ctx->nt_grav, ctx->shard_n, ctx->shard_rank, &per, &lo, &cnt);
  if(per == 0)
    return GHIP_OK;
  TreeDev &t = gas ? ctx->st : ctx->gt;
  const int *tgt = P<int>(gas ? ctx->tg_gas : ctx->tg_grav) + lo;
  double *f[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if(group == 0)
    f[0] = P<double>(ctx->f[GHIP_F_GRAVACCEL]);
  else if(group == 1)
    {
      int ids[7] = {GHIP_F_HSML, GHIP_F_NUMNGB, GHIP_F_DENSITY, GHIP_F_DHSMLFAC, GHIP_F_DIVVEL,
                    GHIP_F_CURLVEL, GHIP_F_PRESSURE};
      for(int c = 0; c < 7; c++)
        f[c] = P<double>(ctx->f[ids[c]]);
    }
  else
    {
      f[0] = P<double>(ctx->f[GHIP_F_HYDROACCEL]);
      f[1] = P<double>(ctx->f[GHIP_F_DTENTROPY]);
      f[2] = P<double>(ctx->f[GHIP_F_MAXSIGNALVEL]);
    }
  k_shard_pack<<<cdiv(per, 256), 256, 0, ctx->stream>>>(
    group, cnt, per, tgt, P<int>(t.perm), ctx->n, ctx->ngas, f[0], f[1], f[2], f[3], f[4], f[5],
    f[6], P<int>(ctx->f[GHIP_F_GRAVCOST]), (double *) dev_buf);
  HIPCHK(hipGetLastError());
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));  // the caller's collective runs on its own stream
  return GHIP_OK;
}

// consumes the all-gathered buffer [nranks][width][per] and fills in the other ranks' slices
extern "C" int ghip_shard_unpack(ghip_ctx *ctx, int group, const void *dev_buf_all, int nranks)
{
  if(!ctx || group < 0 || group > 2 || !dev_buf_all || nranks != ctx->shard_n)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_shard_unpack: bad arguments");
  if(group == 0)
    GHIP_JOIN(ctx);
  if(!ctx->gt.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_shard_unpack: no tree");
  GCHK(ghip_build_target_lists(ctx));
  bool gas = group != 0;
  int nt = gas ? ctx->nt_gas : ctx->nt_grav;
  int per, lo, cnt;
  slice_of(nt, ctx->shard_n, ctx->shard_rank, &per, &lo, &cnt);
  if(nt == 0)
    return GHIP_OK;
  TreeDev &t = gas ? ctx->st : ctx->gt;
  double *f[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if(group == 0)
    f[0] = P<double>(ctx->f[GHIP_F_GRAVACCEL]);
  else if(group == 1)
    {
      int ids[7] = {GHIP_F_HSML, GHIP_F_NUMNGB, GHIP_F_DENSITY, GHIP_F_DHSMLFAC, GHIP_F_DIVVEL,
                    GHIP_F_CURLVEL, GHIP_F_PRESSURE};
      for(int c = 0; c < 7; c++)
        f[c] = P<double>(ctx->f[ids[c]]);
    }
  else
    {
      f[0] = P<double>(ctx->f[GHIP_F_HYDROACCEL]);
      f[1] = P<double>(ctx->f[GHIP_F_DTENTROPY]);
      f[2] = P<double>(ctx->f[GHIP_F_MAXSIGNALVEL]);
    }
  ShardLo L;
  for(int r = 0; r <= ctx->shard_n && r <= GHIP_MAXRANKS; r++)
    {
      int l, c, p2;
      if(r < ctx->shard_n)
        ghip_shard_range(nt, ctx->shard_n, r, &l, &c, &p2);
      else
        l = nt;
      L.lo[r] = l;
    }
  k_shard_unpack<<<cdiv(nt, 256), 256, 0, ctx->stream>>>(
    group, nt, per, kWidth[group], ctx->shard_rank, ctx->shard_n, L,
    P<int>(gas ? ctx->tg_gas : ctx->tg_grav),
    P<int>(t.perm), ctx->n, ctx->ngas, (const double *) dev_buf_all, f[0], f[1], f[2], f[3], f[4],
    f[5], f[6], P<int>(ctx->f[GHIP_F_GRAVCOST]), P<double>(ctx->gp), P<double>(ctx->gq));
  HIPCHK(hipGetLastError());
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  return GHIP_OK;
}
