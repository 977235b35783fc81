// mock_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: the nine RCCL entry points
// gadget-leicester_amd/csrc/ghip_comm.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString) for ranks
// that are PROCESSES SHARING ONE GPU -- which the real RCCL refuses (duplicate device).  Selected with
// GHIP_RCCL_LIB=<this library>.  Every call is staged through a POSIX shared-memory segment named
// after the unique id, with a two-phase barrier between the writers and the readers; the point is
// to run ghip_dd_run's all-gather-of-counts -> group of sends and receives branch with more than
// one rank on the builder's one-GPU box, not to be fast.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace
{
const int MAXR = 64;
struct Header
{
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  // per sender: where in its area the block for each receiver starts, and its length in bytes
  size_t off[MAXR][MAXR], len[MAXR][MAXR];
};
struct Comm
{
  int rank, nranks;
  std::string name;
  char *base;
  size_t total, area;   // bytes per rank area
  Header *hdr;
};
struct Op
{
  bool send;
  const void *src;
  void *dst;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t st;
};
std::vector<Op> g_ops;
int g_group = 0;

size_t type_size(ncclDataType_t t)
{
  switch(t)
    {
    case ncclInt8:
    case ncclUint8:
      return 1;
    case ncclFloat16:
#if defined(RCCL_BFLOAT16)
    case ncclBfloat16:
#endif
      return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32:
      return 4;
    default:
      return 8;
    }
}

void barrier(Comm *c)
{
  Header *h = c->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if(h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks)
    {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.fetch_add(1, std::memory_order_acq_rel);
    }
  else
    while(h->generation.load(std::memory_order_acquire) == gen)
      usleep(50);
}

char *area_of(Comm *c, int r) { return c->base + sizeof(Header) + (size_t) r * c->area; }
}   // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/ghip_mock_rccl_%d_%ld", (int) getpid(), (long) random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
  if(nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks)
    return ncclInvalidArgument;
  Comm *c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  c->name = id.internal;
  size_t mb = getenv("GHIP_MOCK_RCCL_MB") ? (size_t) atol(getenv("GHIP_MOCK_RCCL_MB")) : 256;
  c->area = mb << 20;
  c->total = sizeof(Header) + (size_t) nranks * c->area;
  int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if(fd < 0)
    return ncclSystemError;
  if(ftruncate(fd, (off_t) c->total) != 0)   // (every rank sets the same size; new pages are zero)
    return ncclSystemError;
  void *p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if(p == MAP_FAILED)
    return ncclSystemError;
  c->base = (char *) p;
  c->hdr = (Header *) p;
  c->hdr->attached.fetch_add(1);
  while(c->hdr->attached.load() < nranks)   // everybody has mapped the segment
    usleep(100);
  *comm = (ncclComm_t) c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  Comm *c = (Comm *) comm;
  if(!c)
    return ncclSuccess;
  if(c->rank == 0)
    shm_unlink(c->name.c_str());
  munmap(c->base, c->total);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream)
{
  Comm *c = (Comm *) comm;
  const size_t bytes = sendcount * type_size(datatype);
  if(bytes > c->area)
    return ncclInvalidUsage;
  if(hipStreamSynchronize(stream) != hipSuccess)
    return ncclUnhandledCudaError;
  if(hipMemcpy(area_of(c, c->rank), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  barrier(c);
  for(int r = 0; r < c->nranks; r++)
    if(hipMemcpy((char *) recvbuff + (size_t) r * bytes, area_of(c, r), bytes, hipMemcpyHostToDevice) !=
       hipSuccess)
      return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
  g_group++;
  return ncclSuccess;
}

static ncclResult_t flush_group(void)
{
  if(g_ops.empty())
    return ncclSuccess;
  Comm *c = g_ops[0].c;
  hipStream_t st = g_ops[0].st;
  if(hipStreamSynchronize(st) != hipSuccess)
    return ncclUnhandledCudaError;
  // phase 1: my sends into my area, one block per receiver
  size_t used = 0;
  for(int r = 0; r < c->nranks; r++)
    c->hdr->len[c->rank][r] = 0;
  for(const Op &o : g_ops)
    if(o.send)
      {
        if(used + o.bytes > c->area)
          return ncclInvalidUsage;
        if(hipMemcpy(area_of(c, c->rank) + used, o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
          return ncclUnhandledCudaError;
        c->hdr->off[c->rank][o.peer] = used;
        c->hdr->len[c->rank][o.peer] = o.bytes;
        used += o.bytes;
      }
  barrier(c);
  // phase 2: my receives out of the senders' areas
  ncclResult_t res = ncclSuccess;
  for(const Op &o : g_ops)
    if(!o.send)
      {
        if(c->hdr->len[o.peer][c->rank] != o.bytes)
          {
            fprintf(stderr, "mock rccl: rank %d expects %zu bytes from %d, which sent %zu\n", c->rank,
                    o.bytes, o.peer, c->hdr->len[o.peer][c->rank]);
            res = ncclInvalidUsage;
            continue;
          }
        if(hipMemcpy(o.dst, area_of(c, o.peer) + c->hdr->off[o.peer][c->rank], o.bytes,
                     hipMemcpyHostToDevice) != hipSuccess)
          res = ncclUnhandledCudaError;
      }
  barrier(c);
  g_ops.clear();
  return res;
}

ncclResult_t ncclGroupEnd(void)
{
  if(g_group > 0)
    g_group--;
  return g_group == 0 ? flush_group() : ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
  Op o = { true, sendbuff, nullptr, count * type_size(datatype), peer, (Comm *) comm, stream };
  g_ops.push_back(o);
  return g_group == 0 ? flush_group() : ncclSuccess;
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
  Op o = { false, nullptr, recvbuff, count * type_size(datatype), peer, (Comm *) comm, stream };
  g_ops.push_back(o);
  return g_group == 0 ? flush_group() : ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t result)
{
  switch(result)
    {
    case ncclSuccess:
      return "no error (mock rccl)";
    case ncclUnhandledCudaError:
      return "HIP error (mock rccl)";
    case ncclSystemError:
      return "system error (mock rccl: shared memory)";
    case ncclInvalidArgument:
      return "invalid argument (mock rccl)";
    case ncclInvalidUsage:
      return "invalid usage (mock rccl: size mismatch or area too small, GHIP_MOCK_RCCL_MB)";
    default:
      return "error (mock rccl)";
    }
}
}
