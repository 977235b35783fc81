// ghip_comm.hip -- the exchanges of the multi-GPU path: RCCL over xGMI, called from C.
//
// Replaces the MPI calls of the reference's multi-rank force path: MPI_Allgatherv of the top-leaf
// moments (forcetree.c:963), MPI_Allgather of the send counts and the pairwise MPI_Sendrecv rounds
// of gravity_tree / density / hydro_force (gravtree.c:175-339, density.c:193-389,
// hydra.c:274-526).  One process per GPU; every exchange of the state machine in ghip_dd.hip is one
// of two shapes:
//   kind 1  all-gather of equal-sized blocks            -> ncclAllGather
//   kind 2  all-to-all-v of fixed-size records          -> ncclAllGather of the counts, then one
//           group of ncclSend / ncclRecv (xGMI is point-to-point: every pair has its own link, so
//           the 7 transfers of a rank run side by side)
// Several shards living in ONE process (the 8-logical-shards parity test, which runs the shards
// one after the other on one GPU) exchange through ghip_dd_exchange_local instead: the same
// buffers, device-to-device copies in place of the collectives.
//
// RCCL is bound at run time (dlopen), from the directory of the HIP runtime this process already
// uses: a process that has imported PyTorch runs on PyTorch's bundled libamdhip64/librccl, a plain C
// host on /opt/rocm's -- mixing one's RCCL with the other's runtime is what this avoids.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ghip_internal.h"

namespace
{
struct Rccl
{
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string path, err;
};

Rccl g_rccl;

bool rccl_bind(void *h)
{
  Rccl &R = g_rccl;
#define GHIP_SYM(field, name)                                         \
  R.field = reinterpret_cast<decltype(R.field)>(dlsym(h, name));      \
  if(!R.field)                                                        \
    {                                                                 \
      R.err = std::string("librccl lacks ") + name;                   \
      return false;                                                   \
    }
  GHIP_SYM(GetUniqueId, "ncclGetUniqueId");
  GHIP_SYM(CommInitRank, "ncclCommInitRank");
  GHIP_SYM(CommDestroy, "ncclCommDestroy");
  GHIP_SYM(AllGather, "ncclAllGather");
  GHIP_SYM(Send, "ncclSend");
  GHIP_SYM(Recv, "ncclRecv");
  GHIP_SYM(GroupStart, "ncclGroupStart");
  GHIP_SYM(GroupEnd, "ncclGroupEnd");
  GHIP_SYM(GetErrorString, "ncclGetErrorString");
#undef GHIP_SYM
  R.handle = h;
  return true;
}

bool rccl_load()
{
  Rccl &R = g_rccl;
  if(R.handle)
    return true;
  std::vector<std::string> cand;
  if(getenv("GHIP_RCCL_LIB"))
    cand.push_back(getenv("GHIP_RCCL_LIB"));
  // next to the HIP runtime in use
  // (the address through dlsym: &hipMalloc taken here may be this library's own PLT stub)
  Dl_info info;
  void *hipfn = dlsym(RTLD_DEFAULT, "hipMalloc");
  if(hipfn && dladdr(hipfn, &info) && info.dli_fname)
    {
      std::string dir(info.dli_fname);
      size_t slash = dir.rfind('/');
      if(slash != std::string::npos)
        {
          dir.resize(slash);
          cand.push_back(dir + "/librccl.so.1");
          cand.push_back(dir + "/librccl.so");
        }
    }
  cand.push_back("librccl.so.1");
  cand.push_back("/opt/rocm/lib/librccl.so.1");
  for(const std::string &c : cand)
    {
      void *h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
      if(!h)
        continue;
      if(rccl_bind(h))
        {
          R.path = c;
          return true;
        }
      dlclose(h);
    }
  if(R.err.empty())
    R.err = "librccl.so.1 not found next to the HIP runtime, on the loader path or in /opt/rocm/lib";
  return false;
}
}   // namespace

#define NCHK(call)                                                                              \
  do                                                                                            \
    {                                                                                           \
      ncclResult_t r_ = (call);                                                                 \
      if(r_ != ncclSuccess)                                                                     \
        return ghip_fail(ctx, GHIP_ECOMM, "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
                         g_rccl.GetErrorString(r_));                                            \
    }                                                                                           \
  while(0)

extern "C" const char *ghip_dd_rccl_library(void)
{
  return rccl_load() ? g_rccl.path.c_str() : "";
}

extern "C" int ghip_dd_rccl_unique_id(void *id128)
{
  if(!id128)
    return GHIP_EINVAL;
  if(!rccl_load())
    return GHIP_ECOMM;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if(g_rccl.GetUniqueId(&id) != ncclSuccess)
    return GHIP_ECOMM;
  memcpy(id128, &id, sizeof(id));
  return GHIP_OK;
}

extern "C" int ghip_dd_rccl_connect(ghip_ctx *ctx, const void *id128)
{
  if(!ctx || !id128)
    return GHIP_EINVAL;
  if(!ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_rccl_connect: call ghip_dd_init first");
  if(!rccl_load())
    return ghip_fail(ctx, GHIP_ECOMM, "RCCL: %s", g_rccl.err.c_str());
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  NCHK(g_rccl.CommInitRank(&comm, ctx->dd.nranks, id, ctx->dd.rank));
  ctx->dd.nccl = comm;
  return GHIP_OK;
}

void ghip_dd_comm_release(ghip_ctx *ctx)
{
  if(ctx && ctx->dd.nccl && g_rccl.handle)
    (void) g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->dd.nccl));
  if(ctx)
    ctx->dd.nccl = nullptr;
}

// prefix offsets of the receive side once all counts are known; cnt[src][dst] in records
static void recv_layout(DDXchg &x, int rank, int nranks, const int *cnt)
{
  int off = 0;
  for(int src = 0; src < nranks; src++)
    {
      x.rcount[src] = cnt[src * nranks + rank];
      x.roff[src] = off;
      off += x.rcount[src];
    }
  x.rtotal = off;
}

// ---- RCCL -------------------------------------------------------------------------------------
extern "C" int ghip_dd_exchange(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_EINVAL;
  DDState &D = ctx->dd;
  DDXchg &x = D.x;
  if(x.kind == 0)
    return GHIP_OK;
  if(!D.nccl)
    return ghip_fail(ctx, GHIP_ECOMM, "ghip_dd_exchange: no RCCL communicator (ghip_dd_rccl_connect)");
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(D.nccl);
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks, me = D.rank;
  if(x.kind == 1)
    {
      GCHK(ghip_ensure(ctx, *x.recv, x.bytes * P_));
      NCHK(g_rccl.AllGather(x.send, x.recv->p, x.bytes, ncclChar, comm, st));
      D.bytes_sent[D.op] += (long long) x.bytes * (P_ - 1);
      x.kind = 0;
      return GHIP_OK;
    }
  // kind 2: counts first (the host needs them to size the receive buffer and post the receives)
  GCHK(ghip_ensure(ctx, D.xstage, (size_t) (P_ + P_ * P_) * 4));
  int *dmine = P<int>(D.xstage), *dall = dmine + P_;
  HIPCHK(hipMemcpyAsync(dmine, x.scount, (size_t) P_ * 4, hipMemcpyHostToDevice, st));
  NCHK(g_rccl.AllGather(dmine, dall, (size_t) P_ * 4, ncclChar, comm, st));
  std::vector<int> cnt((size_t) P_ * P_);
  HIPCHK(hipMemcpyAsync(cnt.data(), dall, (size_t) P_ * P_ * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  recv_layout(x, me, P_, cnt.data());
  GCHK(ghip_ensure(ctx, *x.recv, (size_t) (x.rtotal > 0 ? x.rtotal : 1) * x.bytes));
  // (a failing send or receive must not leave the group open: the group is always closed, and the
  // first error is reported afterwards)
  NCHK(g_rccl.GroupStart());
  ncclResult_t gr = ncclSuccess;
  for(int peer = 0; peer < P_ && gr == ncclSuccess; peer++)
    {
      if(peer == me)
        continue;
      if(x.scount[peer] > 0)
        gr = g_rccl.Send(reinterpret_cast<const char *>(x.send) + (size_t) x.soff[peer] * x.bytes,
                         (size_t) x.scount[peer] * x.bytes, ncclChar, peer, comm, st);
      if(gr == ncclSuccess && x.rcount[peer] > 0)
        gr = g_rccl.Recv(reinterpret_cast<char *>(x.recv->p) + (size_t) x.roff[peer] * x.bytes,
                         (size_t) x.rcount[peer] * x.bytes, ncclChar, peer, comm, st);
      D.bytes_sent[D.op] += (long long) x.scount[peer] * (long long) x.bytes;
    }
  const ncclResult_t ge = g_rccl.GroupEnd();
  if(gr != ncclSuccess)
    return ghip_fail(ctx, GHIP_ECOMM, "ncclSend / ncclRecv inside the group -> %s", g_rccl.GetErrorString(gr));
  NCHK(ge);
  if(x.scount[me] > 0)   // (a rank never sends to itself in this path; kept for completeness)
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(x.recv->p) + (size_t) x.roff[me] * x.bytes,
                          reinterpret_cast<const char *>(x.send) + (size_t) x.soff[me] * x.bytes,
                          (size_t) x.scount[me] * x.bytes, hipMemcpyDeviceToDevice, st));
  x.kind = 0;
  return GHIP_OK;
}

// ---- through the host ---------------------------------------------------------------------------
// The same exchange staged through host memory and the CALLER's all-gather of equal-sized blocks
// (MPI_Allgather in the reference's world, torch.distributed/gloo in this repo's rehearsals): for a
// host without RCCL between its ranks, and for running several ranks on ONE GPU, which RCCL
// refuses.  Slow path by construction (PCIe both ways, every rank sees every send buffer).
extern "C" int ghip_dd_exchange_host(ghip_ctx *ctx,
                                     int (*allgather)(void *user, const void *send, size_t bytes,
                                                      void *recv),
                                     void *user)
{
  if(!ctx || !allgather)
    return GHIP_EINVAL;
  DDState &D = ctx->dd;
  DDXchg &x = D.x;
  if(x.kind == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  const int P_ = D.nranks, me = D.rank;
  HIPCHK(hipSetDevice(ctx->device));
  if(x.kind == 1)
    {
      std::vector<char> hs(x.bytes), hr(x.bytes * P_);
      HIPCHK(hipMemcpyAsync(hs.data(), x.send, x.bytes, hipMemcpyDeviceToHost, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      if(allgather(user, hs.data(), x.bytes, hr.data()) != 0)
        return ghip_fail(ctx, GHIP_ECOMM, "ghip_dd_exchange_host: the caller's all-gather failed");
      GCHK(ghip_ensure(ctx, *x.recv, x.bytes * P_));
      HIPCHK(hipMemcpyAsync(x.recv->p, hr.data(), x.bytes * P_, hipMemcpyHostToDevice, st));
      HIPCHK(ghip_stream_sync(ctx, st));
      D.bytes_sent[D.op] += (long long) x.bytes * (P_ - 1);
      x.kind = 0;
      return GHIP_OK;
    }
  std::vector<int> row(2 * (size_t) P_), rows(2 * (size_t) P_ * P_);
  long long mytotal = 0;
  for(int r = 0; r < P_; r++)
    {
      row[r] = x.scount[r];
      row[P_ + r] = x.soff[r];
      if(x.soff[r] + x.scount[r] > mytotal)
        mytotal = x.soff[r] + x.scount[r];
    }
  if(allgather(user, row.data(), row.size() * sizeof(int), rows.data()) != 0)
    return ghip_fail(ctx, GHIP_ECOMM, "ghip_dd_exchange_host: the caller's all-gather failed");
  long long maxtotal = 1;
  std::vector<int> cnt((size_t) P_ * P_);
  for(int src = 0; src < P_; src++)
    for(int dst = 0; dst < P_; dst++)
      {
        const int c = rows[(size_t) src * 2 * P_ + dst], o = rows[(size_t) src * 2 * P_ + P_ + dst];
        cnt[(size_t) src * P_ + dst] = c;
        if((long long) o + c > maxtotal)
          maxtotal = (long long) o + c;
      }
  const size_t blk = (size_t) maxtotal * x.bytes;
  std::vector<char> hs(blk), hr(blk * P_);
  if(mytotal > 0)
    HIPCHK(hipMemcpyAsync(hs.data(), x.send, (size_t) mytotal * x.bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  if(allgather(user, hs.data(), blk, hr.data()) != 0)
    return ghip_fail(ctx, GHIP_ECOMM, "ghip_dd_exchange_host: the caller's all-gather failed");
  recv_layout(x, me, P_, cnt.data());
  GCHK(ghip_ensure(ctx, *x.recv, (size_t) (x.rtotal > 0 ? x.rtotal : 1) * x.bytes));
  for(int src = 0; src < P_; src++)
    {
      if(x.rcount[src] == 0)
        continue;
      const int o = rows[(size_t) src * 2 * P_ + P_ + me];
      HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(x.recv->p) + (size_t) x.roff[src] * x.bytes,
                            hr.data() + (size_t) src * blk + (size_t) o * x.bytes,
                            (size_t) x.rcount[src] * x.bytes, hipMemcpyHostToDevice, st));
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  for(int dst = 0; dst < P_; dst++)
    if(dst != me)
      D.bytes_sent[D.op] += (long long) x.scount[dst] * (long long) x.bytes;
  x.kind = 0;
  return GHIP_OK;
}

// ---- shards of one process ------------------------------------------------------------------------
extern "C" int ghip_dd_exchange_local(ghip_ctx **ctxs, int n)
{
  if(!ctxs || n < 1 || n > GHIP_MAXRANKS)
    return GHIP_EINVAL;
  ghip_ctx *ctx = ctxs[0];
  int kind = -1;
  for(int r = 0; r < n; r++)
    {
      if(!ctxs[r] || !ctxs[r]->dd.on || ctxs[r]->dd.nranks != n || ctxs[r]->dd.rank != r)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_exchange_local: contexts must be ranks 0..n-1 of one run");
      if(kind < 0)
        kind = ctxs[r]->dd.x.kind;
      if(ctxs[r]->dd.x.kind != kind ||
         (kind != 0 && ctxs[r]->dd.x.bytes != ctxs[0]->dd.x.bytes))
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_exchange_local: the shards are not at the same exchange");
    }
  if(kind == 0)
    return GHIP_OK;
  // everything the shards enqueued so far must have landed before another shard's stream reads it
  for(int r = 0; r < n; r++)
    {
      ghip_ctx *c = ctxs[r];
      hipError_t e = hipSetDevice(c->device);
      if(e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
      if(e != hipSuccess)
        return ghip_fail(ctx, GHIP_EHIP, "ghip_dd_exchange_local: %s", hipGetErrorString(e));
    }
  std::vector<int> cnt((size_t) n * n, 0);
  if(kind == 2)
    for(int src = 0; src < n; src++)
      for(int dst = 0; dst < n; dst++)
        cnt[(size_t) src * n + dst] = ctxs[src]->dd.x.scount[dst];
  for(int r = 0; r < n; r++)
    {
      ctx = ctxs[r];
      DDState &D = ctx->dd;
      DDXchg &x = D.x;
      HIPCHK(hipSetDevice(ctx->device));
      hipStream_t st = ctx->stream;
      if(kind == 1)
        {
          GCHK(ghip_ensure(ctx, *x.recv, x.bytes * n));
          for(int src = 0; src < n; src++)
            HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(x.recv->p) + (size_t) src * x.bytes,
                                  ctxs[src]->dd.x.send, x.bytes, hipMemcpyDeviceToDevice, st));
          D.bytes_sent[D.op] += (long long) x.bytes * (n - 1);
        }
      else
        {
          recv_layout(x, r, n, cnt.data());
          GCHK(ghip_ensure(ctx, *x.recv, (size_t) (x.rtotal > 0 ? x.rtotal : 1) * x.bytes));
          for(int src = 0; src < n; src++)
            {
              if(x.rcount[src] == 0)
                continue;
              const DDXchg &xs = ctxs[src]->dd.x;
              HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(x.recv->p) + (size_t) x.roff[src] * x.bytes,
                                    reinterpret_cast<const char *>(xs.send) +
                                      (size_t) xs.soff[r] * x.bytes,
                                    (size_t) x.rcount[src] * x.bytes, hipMemcpyDeviceToDevice, st));
            }
          for(int dst = 0; dst < n; dst++)
            if(dst != r)
              D.bytes_sent[D.op] += (long long) x.scount[dst] * (long long) x.bytes;
        }
      HIPCHK(ghip_stream_sync(ctx, st));
    }
  for(int r = 0; r < n; r++)
    ctxs[r]->dd.x.kind = 0;
  return GHIP_OK;
}
