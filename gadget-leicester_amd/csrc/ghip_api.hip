// ghip_api.hip -- C-ABI entry points: lifetime, particle data movement, introspection.
// See include/ghip.h for what each entry point replaces in the reference.
#include <cstdarg>

#include <atomic>
#include <dlfcn.h>

#include "ghip_internal.h"

// ---------------------------------------------------------------------------------------------
// Launch counter.  Every `kernel<<<...>>>` of this library -- its own kernels and the rocPRIM /
// hipCUB ones it calls -- ends in hipLaunchKernel.  The library carries a LOCAL definition of that
// symbol (the export map keeps it out of the dynamic symbol table, so nobody else's calls are
// touched) that counts and forwards to the runtime's.  bench.py reports launches per step from it.
// ---------------------------------------------------------------------------------------------
static std::atomic<long long> g_launches{0};
typedef hipError_t (*launch_fn)(const void *, dim3, dim3, void **, size_t, hipStream_t);
static launch_fn real_launch(void)
{
  static launch_fn fn = nullptr;
  if(!fn)
    {
      void *p = dlvsym(RTLD_NEXT, "hipLaunchKernel", "hip_4.2");
      if(!p)
        p = dlsym(RTLD_NEXT, "hipLaunchKernel");
      if(!p)
        {
          // the runtime this library is linked against, found through one of its other symbols
          Dl_info info;
          if(dladdr((void *) &hipMemsetAsync, &info) && info.dli_fname)
            {
              void *h = dlopen(info.dli_fname, RTLD_LAZY | RTLD_NOLOAD);
              if(h)
                p = dlsym(h, "hipLaunchKernel");
            }
        }
      fn = (launch_fn) p;
    }
  return fn;
}
extern "C" hipError_t hipLaunchKernel(const void *function_address, dim3 numBlocks, dim3 dimBlocks,
                                      void **args, size_t sharedMemBytes, hipStream_t stream)
{
  launch_fn fn = real_launch();
  if(!fn)
    return hipErrorNotInitialized;   // loud: every launch fails
  g_launches.fetch_add(1, std::memory_order_relaxed);
  return fn(function_address, numBlocks, dimBlocks, args, sharedMemBytes, stream);
}
long long ghip_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int ghip_fail(ghip_ctx *ctx, int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if(ctx)
    ctx->err = buf;
  return code;
}

int ghip_ensure(ghip_ctx *ctx, DevBuf &b, size_t bytes)
{
  if(bytes == 0)
    bytes = 8;
  if(b.cap >= bytes)
    return GHIP_OK;
  // grow with slack so that per-step size jitter does not reallocate
  size_t want = bytes + bytes / 8 + 256;
  if(b.p)
    {
      // (nothing may still be using the old block: the walks of a gravity pair run on streams of
      // their own, and the host no longer waits for them at the end of every step)
      HIPCHK(ghip_stream_sync(ctx, ctx->stream));
      if(ctx->stream2)
        HIPCHK(ghip_stream_sync(ctx, ctx->stream2));
      if(ctx->stream3)
        HIPCHK(ghip_stream_sync(ctx, ctx->stream3));
      HIPCHK(hipFree(b.p));
      b.p = nullptr;
      b.cap = 0;
    }
  hipError_t e = hipMalloc(&b.p, want);
  if(e != hipSuccess)
    {
      b.p = nullptr;
      return ghip_fail(ctx, GHIP_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
  b.cap = want;
  return GHIP_OK;
}

int ghip_join_pair(ghip_ctx *ctx)
{
  if(ctx && ctx->grav_pending)
    {
      ctx->grav_pending = false;
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->evx[2], 0));
    }
  return GHIP_OK;
}

int ghip_join(ghip_ctx *ctx)
{
  // (first: a tree built without waiting for its sizes is verified -- and, if it turned out bad,
  // rebuilt with the gravity calls made since replayed -- before anything else relies on it)
  if(ctx)
    GCHK(ghip_tree_verify(ctx));
  GCHK(ghip_join_pair(ctx));
  if(!ctx)
    return GHIP_OK;
  GCHK(ghip_finish_gas_tree(ctx));
  return ghip_gas_verify(ctx);
}

int *ghip_errwords(void)
{
  static int *words = nullptr;
  if(!words)
    {
      void *p = nullptr;
      if(hipHostMalloc(&p, 256, hipHostMallocDefault) != hipSuccess)
        {
          static int fallback[64];   // never device-visible: only reached when pinning fails
          return fallback;
        }
      memset(p, 0, 256);
      words = reinterpret_cast<int *>(p);
    }
  return words;
}

// call after a stream synchronisation: has a kernel reported a broken invariant?
int ghip_check_device_errors(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_OK;
  static const char *what[GHIP_ERRW_COUNT] = {
    "the wavefront plan of a gravity walk exceeded its grid (ghip_walk.h, k_plan_fill)",
    "a target had to open a pruned node of an imported (other shard's) tree: the locally "
    "essential tree was incomplete",
    "tree emission outside the element list, a malformed imported element (3), a particle outside its "
    "shard's key range (5: migrate first) or outside the domain cube (6: ghip_dd_set_domain with a fresh extent)",
    "ghost import", "drift", "timestep", "", ""};
  for(int w = 0; w < GHIP_ERRW_COUNT; w++)
    {
      volatile int *e = ghip_errword(ctx, w);
      if(*e != 0)
        {
          int v = *e;
          *e = 0;
          // what an asynchronous ghip_drift / ghip_advance_timesteps would have returned itself
          if(w == GHIP_ERRW_DRIFT)
            return ghip_fail(ctx, GHIP_EINVAL, "ghip_drift: a particle is ahead of time1 (reference: "
                             "endrun(12), predict.c:148)");
          if(w == GHIP_ERRW_TIMESTEP)
            {
              ctx->timestep_endrun = v;
              return ghip_fail(ctx, GHIP_ETIMESTEP, "ghip_advance_timesteps: the reference stops here with "
                               "endrun(%d) (timestep.c:171/1082: 888, :1119: 818, :1233: 112313)", v);
            }
          return ghip_fail(ctx, GHIP_EDEVICE, "device invariant %d broken (%d): %s", w, v, what[w]);
        }
    }
  return GHIP_OK;
}

static void free_buf(DevBuf &b)
{
  if(b.p)
    (void) hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

extern "C" const char *ghip_version(void)
{
  return "ghip 0.3 (gfx950)";
}

extern "C" int ghip_create(int device, ghip_ctx **out)
{
  if(!out)
    return GHIP_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return GHIP_ENODEVICE;
  if(device < 0 || device >= ndev)
    return GHIP_ENODEVICE;
  ghip_ctx *ctx = new(std::nothrow) ghip_ctx();
  if(!ctx)
    return GHIP_ENOMEM;
  ctx->device = device;
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  // the main stream gets the highest priority, the two streams of a gravity pair the lowest: what
  // is enqueued underneath the walks (SPH phases, the deferred gas-tree work) consists of short
  // kernels that must not queue behind two chip-filling ones
  int prio_least = 0, prio_greatest = 0;
  if(hipSetDevice(device) != hipSuccess ||
     hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess ||
     hipStreamCreateWithPriority(&ctx->stream, hipStreamDefault, prio_greatest) != hipSuccess)
    {
      delete ctx;
      return GHIP_ENODEVICE;
    }
  for(int i = 0; i < 16; i++)
    if(hipEventCreate(&ctx->ev[i]) != hipSuccess)
      {
        delete ctx;
        return GHIP_EHIP;
      }
  ctx->ev_ready = true;
  ctx->evp = ctx->ev;
  if(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming) != hipSuccess ||
     hipEventCreateWithFlags(&ctx->ev_sizes, hipEventDisableTiming) != hipSuccess ||
     hipEventCreateWithFlags(&ctx->ev_sizes_gas, hipEventDisableTiming) != hipSuccess)
    {
      delete ctx;
      return GHIP_EHIP;
    }
  ctx->tree_async_ok = !(getenv("GHIP_TREE_SYNC") && atoi(getenv("GHIP_TREE_SYNC")) == 1);
  if(hipStreamCreateWithPriority(&ctx->stream2, hipStreamDefault, prio_least) != hipSuccess ||
     hipStreamCreateWithPriority(&ctx->stream3, hipStreamDefault, prio_least) != hipSuccess)
    {
      delete ctx;
      return GHIP_EHIP;
    }
  for(int i = 0; i < 4; i++)
    if(hipEventCreateWithFlags(&ctx->evx[i], hipEventDisableTiming) != hipSuccess)
      {
        delete ctx;
        return GHIP_EHIP;
      }
  for(int i = 0; i < 2; i++)
    if(hipEventCreateWithFlags(&ctx->evt[i], hipEventDisableTiming) != hipSuccess)
      {
        delete ctx;
        return GHIP_EHIP;
      }
  ctx->evx_ready = true;
  // device counters start at zero (hipMalloc does not clear)
  if(ghip_ensure(ctx, ctx->counters, 64 * 8) != GHIP_OK ||
     hipMemset(ctx->counters.p, 0, ctx->counters.cap) != hipSuccess)
    {
      delete ctx;
      return GHIP_ENOMEM;
    }
  if(ghip_ensure(ctx, ctx->cslots, GHIP_CBUF_BYTES) != GHIP_OK ||
     ghip_ensure(ctx, ctx->rslots, GHIP_CBUF_BYTES) != GHIP_OK ||
     hipMemset(ctx->cslots.p, 0, GHIP_CBUF_BYTES) != hipSuccess ||
     hipMemset(ctx->rslots.p, 0, GHIP_CBUF_BYTES) != hipSuccess)
    {
      delete ctx;
      return GHIP_ENOMEM;
    }
  // pinned, device-visible host words: tree-build read-back [0..7], device error words [32..39]
  if(hipHostMalloc(&ctx->pinned, 1024, hipHostMallocDefault) != hipSuccess)
    {
      ctx->pinned = nullptr;
      delete ctx;
      return GHIP_ENOMEM;
    }
  ctx->pinned_cap = 1024;
  memset(ctx->pinned, 0, 1024);
  // the trees' sizes: on the device and mirrored in pinned memory ([128..] of the block above)
  ctx->gt.hsz = reinterpret_cast<TreeSizes *>(reinterpret_cast<char *>(ctx->pinned) + 128);
  ctx->st.hsz = ctx->gt.hsz + 1;
  if(ghip_ensure(ctx, ctx->gt.dsz, sizeof(TreeSizes)) != GHIP_OK ||
     ghip_ensure(ctx, ctx->st.dsz, sizeof(TreeSizes)) != GHIP_OK ||
     hipMemset(ctx->gt.dsz.p, 0, sizeof(TreeSizes)) != hipSuccess ||
     hipMemset(ctx->st.dsz.p, 0, sizeof(TreeSizes)) != hipSuccess ||
     ghip_ensure(ctx, ctx->run_acc, 80 * 8) != GHIP_OK || hipMemset(ctx->run_acc.p, 0, 80 * 8) != hipSuccess)
    {
      delete ctx;
      return GHIP_ENOMEM;
    }
  // make every event "recorded" so that elapsed-time queries never fault
  for(int i = 0; i < 16; i++)
    (void) hipEventRecord(ctx->ev[i], ctx->stream);
  (void) hipStreamSynchronize(ctx->stream);
  *out = ctx;
  return GHIP_OK;
}

static void free_tree(TreeDev &t)
{
  DevBuf *bs[] = {&t.key, &t.skey, &t.idx, &t.perm, &t.iperm, &t.cpl, &t.cnt, &t.nb,
                  &t.xm,  &t.cl,   &t.lk,  &t.aux, &t.seg_start, &t.seg_nanc, &t.seg_anc, &t.mq, &t.mq2, &t.phkey, &t.phorder, &t.slvl,
                  &t.father, &t.arrived, &t.dsz};
  for(DevBuf *b : bs)
    free_buf(*b);
}

extern "C" void ghip_destroy(ghip_ctx *ctx)
{
  if(!ctx)
    return;
  (void) hipSetDevice(ctx->device);

  if(ctx->stream)
    (void) hipStreamSynchronize(ctx->stream);
  if(ctx->stream2)
    (void) hipStreamSynchronize(ctx->stream2);
  if(ctx->stream3)
    (void) hipStreamSynchronize(ctx->stream3);
  for(int i = 0; i < GHIP_F_COUNT; i++)
    free_buf(ctx->f[i]);
  DevBuf *bs[] = {&ctx->stage,  &ctx->aosP,   &ctx->aosS,    &ctx->sx,      &ctx->sy,
                  &ctx->sz,     &ctx->ssoft,  &ctx->soldacc, &ctx->gp,      &ctx->gq,
                  &ctx->dleft,  &ctx->dright, &ctx->drho,    &ctx->dnumngb, &ctx->ddhsml,
                  &ctx->ddivv,  &ctx->drot,   &ctx->dflags,  &ctx->dtgt_a,  &ctx->dtgt_b,
                  &ctx->act_host_idx, &ctx->tg_grav, &ctx->tg_gas, &ctx->tax, &ctx->tay,
                  &ctx->taz,    &ctx->tcost,  &ctx->ewtab,   &ctx->ewbrick, &ctx->srtab,   &ctx->cubtmp,
                  &ctx->counters, &ctx->cslots, &ctx->rslots, &ctx->dhcur, &ctx->hpart, &ctx->plan_nsub, &ctx->plan_woff,
                  &ctx->plan_wave, &ctx->plan_steps[0][0], &ctx->plan_steps[0][1],
                  &ctx->plan_steps[1][0], &ctx->plan_steps[1][1], &ctx->plan_steps[2][0],
                  &ctx->plan_steps[2][1], &ctx->tax2, &ctx->tay2, &ctx->taz2, &ctx->tcost2,
                  &ctx->plan_nsub2, &ctx->plan_woff2, &ctx->plan_wave2, &ctx->cubtmp2, &ctx->cubtmp3,
                  &ctx->bh_swallow, &ctx->bh_injected};
  for(DevBuf *b : bs)
    free_buf(*b);
  free_tree(ctx->gt);
  free_tree(ctx->st);
  ghip_dyn_release(ctx);
  ghip_dd_release(ctx);
  ghip_pm_release(ctx);
  free_buf(ctx->pm_rho);
  free_buf(ctx->pm_k);
  free_buf(ctx->pm_force);
  if(ctx->pinned)
    (void) hipHostFree(ctx->pinned);
  for(void *p : ctx->host_pins)
    (void) hipHostUnregister(p);
  ctx->host_pins.clear();
  if(ctx->ev_ready)
    for(int i = 0; i < 16; i++)
      (void) hipEventDestroy(ctx->ev[i]);
  if(ctx->pc_ready)
    for(int i = 0; i < 4; i++)
      for(int j = 0; j < 4; j++)
        (void) hipEventDestroy(ctx->pc_ev[i][j]);
  if(ctx->ev_side)
    (void) hipEventDestroy(ctx->ev_side);
  if(ctx->ev_sizes)
    (void) hipEventDestroy(ctx->ev_sizes);
  if(ctx->ev_sizes_gas)
    (void) hipEventDestroy(ctx->ev_sizes_gas);
  for(hipEvent_t e : ctx->ev_ring)
    (void) hipEventDestroy(e);
  free_buf(ctx->run_acc);
  if(ctx->evx_ready)
    {
      for(int i = 0; i < 4; i++)
        (void) hipEventDestroy(ctx->evx[i]);
      for(int i = 0; i < 2; i++)
        (void) hipEventDestroy(ctx->evt[i]);
    }
  if(ctx->stream2)
    (void) hipStreamDestroy(ctx->stream2);
  if(ctx->stream3)
    (void) hipStreamDestroy(ctx->stream3);
  if(ctx->stream)
    (void) hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char *ghip_last_error(const ghip_ctx *ctx)
{
  return ctx ? ctx->err.c_str() : "null context";
}

extern "C" void *ghip_stream(ghip_ctx *ctx)
{
  if(ctx)
    (void) ghip_join(ctx);
  return ctx ? (void *) ctx->stream : nullptr;
}

extern "C" int ghip_sync(ghip_ctx *ctx)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx)
    return GHIP_EINVAL;
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  return ghip_check_device_errors(ctx);
}

// ---------------------------------------------------------------------------------------------
// field table
// ---------------------------------------------------------------------------------------------
struct FieldInfo
{
  int gas;    // 1: sized ngas, 0: sized numpart
  int ncomp;  // 1 or 3
  int isint;
};

static const FieldInfo kField[GHIP_F_COUNT] = {
  /* POS */ {0, 3, 0},        /* VEL */ {0, 3, 0},       /* MASS */ {0, 1, 0},
  /* TYPE */ {0, 1, 1},       /* OLDACC */ {0, 1, 0},    /* HSML */ {0, 1, 0},
  /* TIMEBIN */ {0, 1, 1},    /* TI_BEGSTEP */ {0, 1, 1}, /* VELPRED */ {1, 3, 0},
  /* ENTROPY */ {1, 1, 0},    /* DTENTROPY */ {1, 1, 0}, /* GRAVACCEL */ {0, 3, 0},
  /* GRAVCOST */ {0, 1, 1},   /* NUMNGB */ {1, 1, 0},    /* DENSITY */ {1, 1, 0},
  /* DHSMLFAC */ {1, 1, 0},   /* DIVVEL */ {1, 1, 0},    /* CURLVEL */ {1, 1, 0},
  /* PRESSURE */ {1, 1, 0},   /* HYDROACCEL */ {1, 3, 0}, /* MAXSIGNALVEL */ {1, 1, 0},
  /* TI_CURRENT */ {0, 1, 1}, /* GRAVPM */ {0, 3, 0}, /* ID */ {0, 1, 1}};

// the same table for the migration of ghip_dd.hip
void ghip_field_info(int f, int *gas, int *ncomp, int *isint)
{
  *gas = kField[f].gas;
  *ncomp = kField[f].ncomp;
  *isint = kField[f].isint;
}

static size_t field_count(const ghip_ctx *ctx, int f)
{
  return (size_t) (kField[f].gas ? ctx->ngas : ctx->n);
}

static size_t field_bytes(const ghip_ctx *ctx, int f)
{
  return field_count(ctx, f) * kField[f].ncomp * (kField[f].isint ? 4 : 8);
}

extern "C" int ghip_set_counts(ghip_ctx *ctx, int numpart, int ngas)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || numpart < 0 || ngas < 0 || ngas > numpart)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_counts: need 0 <= ngas <= numpart");
  HIPCHK(hipSetDevice(ctx->device));
  bool changed = (numpart != ctx->n || ngas != ctx->ngas);
  ctx->n = numpart;
  ctx->ngas = ngas;
  for(int f = 0; f < GHIP_F_COUNT; f++)
    {
      size_t before = ctx->f[f].cap;
      GCHK(ghip_ensure(ctx, ctx->f[f], field_bytes(ctx, f)));
      if(ctx->f[f].cap != before)
        HIPCHK(hipMemsetAsync(ctx->f[f].p, 0, ctx->f[f].cap, ctx->stream));
    }
  if(changed)
    {
      ctx->gt.built = false;
      ctx->st.built = false;
      ctx->nactive = -1;
      ctx->lists_dirty = ctx->gas_list_dirty = true;
    }
  return GHIP_OK;
}

// [n][ncomp] (host layout, in `stage`) <-> SoA with pitch n (device layout)
template <class T>
__global__ void k_aos_to_soa(size_t n, int ncomp, const T *__restrict__ src, T *__restrict__ dst)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  for(int c = 0; c < ncomp; c++)
    dst[(size_t) c * n + i] = src[i * ncomp + c];
}

template <class T>
__global__ void k_soa_to_aos(size_t n, int ncomp, const T *__restrict__ src, T *__restrict__ dst)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  for(int c = 0; c < ncomp; c++)
    dst[i * ncomp + c] = src[(size_t) c * n + i];
}

// is every record of the gas block [0, ngas) still gas?  (allvars.h:1384 holds after a
// rearrange_particle_sequence(); a particle converted since keeps its place with another Type)
__global__ void k_check_gas_types(int ngas, const int *__restrict__ type, int *word)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < ngas && type[i] != 0)
    *word = 1;
}

int ghip_check_gas_types(ghip_ctx *ctx)
{
  bool mixed = false;
  if(ctx->ngas > 0)
    {
      int *w = ghip_gas_mixed_word(ctx);
      *w = 0;
      k_check_gas_types<<<cdiv(ctx->ngas, 256), 256, 0, ctx->stream>>>(ctx->ngas, P<int>(ctx->f[GHIP_F_TYPE]), w);
      HIPCHK(hipGetLastError());
      HIPCHK(ghip_stream_sync(ctx, ctx->stream));
      mixed = *reinterpret_cast<volatile int *>(w) != 0;
    }
  if(mixed != ctx->gas_mixed)
    {
      ctx->gas_mixed = mixed;
      ctx->st.built = false;
      ctx->gas_list_dirty = true;
    }
  return GHIP_OK;
}

extern "C" int ghip_set_field(ghip_ctx *ctx, int field, const void *host)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || field < 0 || field >= GHIP_F_COUNT)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_field: bad field %d", field);
  size_t cnt = field_count(ctx, field), bytes = field_bytes(ctx, field);
  if(cnt == 0)
    return GHIP_OK;
  if(!host)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_field: null host pointer");
  hipStream_t st = ctx->stream;
  if(kField[field].ncomp == 1)
    HIPCHK(hipMemcpyAsync(ctx->f[field].p, host, bytes, hipMemcpyHostToDevice, st));
  else
    {
      GCHK(ghip_ensure(ctx, ctx->stage, bytes));
      HIPCHK(hipMemcpyAsync(ctx->stage.p, host, bytes, hipMemcpyHostToDevice, st));
      k_aos_to_soa<double><<<cdiv((long long) cnt, 256), 256, 0, st>>>(
        cnt, 3, P<double>(ctx->stage), P<double>(ctx->f[field]));
      HIPCHK(hipGetLastError());
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  if(field == GHIP_F_POS || field == GHIP_F_MASS || field == GHIP_F_TYPE)
    {
      ctx->gt.built = false;
      ctx->st.built = false;
    }
  if(field == GHIP_F_TYPE)
    GCHK(ghip_check_gas_types(ctx));
  return GHIP_OK;
}

extern "C" int ghip_get_field(ghip_ctx *ctx, int field, void *host)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || field < 0 || field >= GHIP_F_COUNT)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_get_field: bad field %d", field);
  size_t cnt = field_count(ctx, field), bytes = field_bytes(ctx, field);
  if(cnt == 0)
    return GHIP_OK;
  if(!host)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_get_field: null host pointer");
  hipStream_t st = ctx->stream;
  if(kField[field].ncomp == 1)
    HIPCHK(hipMemcpyAsync(host, ctx->f[field].p, bytes, hipMemcpyDeviceToHost, st));
  else
    {
      GCHK(ghip_ensure(ctx, ctx->stage, bytes));
      k_soa_to_aos<double><<<cdiv((long long) cnt, 256), 256, 0, st>>>(
        cnt, 3, P<double>(ctx->f[field]), P<double>(ctx->stage));
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(host, ctx->stage.p, bytes, hipMemcpyDeviceToHost, st));
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  return ghip_check_device_errors(ctx);
}

// ---------------------------------------------------------------------------------------------
// whole-record (AoS) path: the drop-in boundary under accel.c
// ---------------------------------------------------------------------------------------------
__global__ void k_unpack_f64(size_t n, const char *__restrict__ rec, int stride, int off, int ncomp,
                             double *__restrict__ dst)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const double *src = reinterpret_cast<const double *>(rec + i * (size_t) stride + off);
  for(int c = 0; c < ncomp; c++)
    dst[(size_t) c * n + i] = src[c];
}

__global__ void k_unpack_i16(size_t n, const char *__restrict__ rec, int stride, int off,
                             int *__restrict__ dst)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    dst[i] = (int) *reinterpret_cast<const short *>(rec + i * (size_t) stride + off);
}

__global__ void k_unpack_i32(size_t n, const char *__restrict__ rec, int stride, int off,
                             int *__restrict__ dst)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    dst[i] = *reinterpret_cast<const int *>(rec + i * (size_t) stride + off);
}

__global__ void k_pack_f64(size_t n, char *__restrict__ rec, int stride, int off, int ncomp,
                           const double *__restrict__ src)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  double *dst = reinterpret_cast<double *>(rec + i * (size_t) stride + off);
  for(int c = 0; c < ncomp; c++)
    dst[c] = src[(size_t) c * n + i];
}

__global__ void k_pack_cost_f32(size_t n, char *__restrict__ rec, int stride, int off,
                                const int *__restrict__ src)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    *reinterpret_cast<float *>(rec + i * (size_t) stride + off) = (float) src[i];  // P[].GravCost
}

// One pass over a record block for all of its fields (instead of one launch per field: every launch
// re-reads every line of the image).  kind 0: double[ncomp] <-> SoA component arrays of stride n;
// 1: short <-> int; 2: int <-> int; 3 (pack only): int -> float (P[].GravCost).
struct RecField
{
  void *soa;
  int off;
  short ncomp, kind;
};
struct RecMap
{
  size_t n;
  int stride, nf;
  RecField f[16];
};

static void rec_add(RecMap &m, int off, int ncomp, int kind, void *soa)
{
  if(off < 0 || m.nf >= 16)
    return;
  m.f[m.nf].soa = soa;
  m.f[m.nf].off = off;
  m.f[m.nf].ncomp = (short) ncomp;
  m.f[m.nf].kind = (short) kind;
  m.nf++;
}

// type_off >= 0: the records [0, ngas) are supposed to be gas (allvars.h:1384); *mixed is set when
// one of them has another Type
template <bool PACK>
__global__ void k_records(RecMap m, char *__restrict__ rec, int type_off, int ngas, int *mixed)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m.n)
    return;
  char *r = rec + i * (size_t) m.stride;
  for(int k = 0; k < m.nf; k++)
    {
      const RecField f = m.f[k];
      if(f.kind == 0)
        {
          double *a = reinterpret_cast<double *>(r + f.off);
          double *s = reinterpret_cast<double *>(f.soa);
          for(int c = 0; c < f.ncomp; c++)
            {
              if(PACK)
                a[c] = s[(size_t) c * m.n + i];
              else
                s[(size_t) c * m.n + i] = a[c];
            }
        }
      else if(f.kind == 1)
        {
          if(PACK)
            *reinterpret_cast<short *>(r + f.off) = (short) reinterpret_cast<int *>(f.soa)[i];
          else
            reinterpret_cast<int *>(f.soa)[i] = (int) *reinterpret_cast<const short *>(r + f.off);
        }
      else if(f.kind == 2)
        {
          if(PACK)
            *reinterpret_cast<int *>(r + f.off) = reinterpret_cast<int *>(f.soa)[i];
          else
            reinterpret_cast<int *>(f.soa)[i] = *reinterpret_cast<const int *>(r + f.off);
        }
      else if(PACK)
        *reinterpret_cast<float *>(r + f.off) = (float) reinterpret_cast<int *>(f.soa)[i];
    }
  if(!PACK && type_off >= 0 && i < (size_t) ngas && *reinterpret_cast<const short *>(r + type_off) != 0)
    *mixed = 1;
}

template <bool PACK>
static int run_records(ghip_ctx *ctx, const RecMap &m, void *img, int type_off = -1,
                       hipStream_t st = nullptr)
{
  if(m.n == 0 || m.nf == 0)
    return GHIP_OK;
  const int wg = ghip_wg(ctx);
  k_records<PACK><<<cdiv((long long) m.n, wg), wg, 0, st ? st : ctx->stream>>>(
    m, (char *) img, type_off, ctx->ngas, ghip_gas_mixed_word(ctx));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

#define UNPACK64(cnt, img, stride, off, ncomp, fld)                                           \
  do                                                                                          \
    {                                                                                         \
      if((off) >= 0 && (cnt) > 0)                                                             \
        k_unpack_f64<<<cdiv((long long) (cnt), ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(         \
          (size_t) (cnt), (const char *) (img), (stride), (off), (ncomp), P<double>(ctx->f[fld])); \
    }                                                                                         \
  while(0)

#define PACK64(cnt, img, stride, off, ncomp, fld)                                             \
  do                                                                                          \
    {                                                                                         \
      if((off) >= 0 && (cnt) > 0)                                                             \
        k_pack_f64<<<cdiv((long long) (cnt), ghip_wg(ctx)), ghip_wg(ctx), 0, st>>>(           \
          (size_t) (cnt), (char *) (img), (stride), (off), (ncomp), P<double>(ctx->f[fld]));  \
    }                                                                                         \
  while(0)

// the P[] block: copy + unpack of the per-particle fields
static int upload_p_block(ghip_ctx *ctx, const void *Pp, const ghip_layout *lay)
{
  hipStream_t st = ctx->stream;
  const size_t n = (size_t) ctx->n;
  if(n == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->aosP, n * lay->p_stride));
  HIPCHK(hipMemcpyAsync(ctx->aosP.p, Pp, n * lay->p_stride, hipMemcpyHostToDevice, st));
  RecMap m;
  m.n = n;
  m.stride = lay->p_stride;
  m.nf = 0;
  rec_add(m, lay->p_pos, 3, 0, ctx->f[GHIP_F_POS].p);
  rec_add(m, lay->p_vel, 3, 0, ctx->f[GHIP_F_VEL].p);
  rec_add(m, lay->p_mass, 1, 0, ctx->f[GHIP_F_MASS].p);
  rec_add(m, lay->p_oldacc, 1, 0, ctx->f[GHIP_F_OLDACC].p);
  rec_add(m, lay->p_type, 1, 1, ctx->f[GHIP_F_TYPE].p);
  rec_add(m, lay->p_timebin, 1, 1, ctx->f[GHIP_F_TIMEBIN].p);
  rec_add(m, lay->p_ti_begstep, 1, 2, ctx->f[GHIP_F_TI_BEGSTEP].p);
  rec_add(m, lay->p_ti_current, 1, 2, ctx->f[GHIP_F_TI_CURRENT].p);
  rec_add(m, lay->p_gravaccel, 3, 0, ctx->f[GHIP_F_GRAVACCEL].p);
  rec_add(m, lay->p_gravpm, 3, 0, ctx->f[GHIP_F_GRAVPM].p);
  rec_add(m, lay->p_hsml, 1, 0, ctx->f[GHIP_F_HSML].p);
  *ghip_gas_mixed_word(ctx) = 0;
  GCHK(run_records<false>(ctx, m, ctx->aosP.p, lay->p_type));
  return GHIP_OK;
}

// the SphP[] block: copy + unpack of the per-gas-particle fields.  The unpack kernels may run
// underneath a gravity pair in flight: one-wavefront workgroups then (ghip_wg)
static int upload_s_block(ghip_ctx *ctx, const void *Sp, const ghip_layout *lay)
{
  hipStream_t st = ctx->stream;
  const size_t ng = (size_t) ctx->ngas;
  if(ng == 0)
    return GHIP_OK;
  GCHK(ghip_ensure(ctx, ctx->aosS, ng * lay->s_stride));
  HIPCHK(hipMemcpyAsync(ctx->aosS.p, Sp, ng * lay->s_stride, hipMemcpyHostToDevice, st));
  RecMap m;
  m.n = ng;
  m.stride = lay->s_stride;
  m.nf = 0;
  rec_add(m, lay->s_hsml, 1, 0, ctx->f[GHIP_F_HSML].p);   // first ngas entries
  rec_add(m, lay->s_velpred, 3, 0, ctx->f[GHIP_F_VELPRED].p);
  rec_add(m, lay->s_entropy, 1, 0, ctx->f[GHIP_F_ENTROPY].p);
  rec_add(m, lay->s_dtentropy, 1, 0, ctx->f[GHIP_F_DTENTROPY].p);
  rec_add(m, lay->s_density, 1, 0, ctx->f[GHIP_F_DENSITY].p);
  rec_add(m, lay->s_dhsmlfac, 1, 0, ctx->f[GHIP_F_DHSMLFAC].p);
  rec_add(m, lay->s_divvel, 1, 0, ctx->f[GHIP_F_DIVVEL].p);
  rec_add(m, lay->s_curlvel, 1, 0, ctx->f[GHIP_F_CURLVEL].p);
  rec_add(m, lay->s_pressure, 1, 0, ctx->f[GHIP_F_PRESSURE].p);
  rec_add(m, lay->s_hydroaccel, 3, 0, ctx->f[GHIP_F_HYDROACCEL].p);
  rec_add(m, lay->s_maxsignalvel, 1, 0, ctx->f[GHIP_F_MAXSIGNALVEL].p);
  GCHK(run_records<false>(ctx, m, ctx->aosS.p));
  return GHIP_OK;
}

static int check_upload_args(ghip_ctx *ctx, const void *Pp, const void *Sp, const ghip_layout *lay,
                             int numpart, int ngas, bool need_s)
{
  if(!ctx || !lay || numpart < 0 || ngas < 0 || ngas > numpart || (numpart > 0 && !Pp) ||
     (need_s && ngas > 0 && !Sp))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_upload_aos: bad arguments");
  if(lay->p_stride <= 0 || (ngas > 0 && lay->s_stride <= 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_upload_aos: bad strides");
  if((lay->p_hsml >= 0) == (lay->s_hsml >= 0) && ngas > 0)
    return ghip_fail(ctx, GHIP_EINVAL,
                     "ghip_upload_aos: exactly one of p_hsml / s_hsml must be set (PPP macro)");
  return GHIP_OK;
}

extern "C" int ghip_upload_aos(ghip_ctx *ctx, const void *Pp, const void *Sp, const ghip_layout *lay,
                               int numpart, int ngas)
{
  if(ctx)
    GHIP_JOIN(ctx);
  GCHK(check_upload_args(ctx, Pp, Sp, lay, numpart, ngas, true));
  GCHK(ghip_set_counts(ctx, numpart, ngas));
  if(numpart == 0)
    return GHIP_OK;
  GCHK(upload_p_block(ctx, Pp, lay));
  GCHK(upload_s_block(ctx, Sp, lay));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  ctx->gt.built = false;
  ctx->st.built = false;
  ctx->gas_wait_upload = false;
  ctx->gas_mixed = lay->p_type >= 0 && *reinterpret_cast<volatile int *>(ghip_gas_mixed_word(ctx)) != 0;
  ctx->gas_list_dirty = true;
  return GHIP_OK;
}

// The same in two calls, for a host that starts gravity before the gas data are across: the P[]
// block (everything the gravity tree and its walks read -- unless the smoothing lengths, which
// ADAPTIVE_GRAVSOFT_FORGAS makes softenings, live in SphP[]) ...
extern "C" int ghip_upload_aos_particles(ghip_ctx *ctx, const void *Pp, const ghip_layout *lay,
                                         int numpart, int ngas)
{
  if(ctx)
    GHIP_JOIN(ctx);
  GCHK(check_upload_args(ctx, Pp, nullptr, lay, numpart, ngas, false));
  if(ctx->adaptive_gravsoft && lay->p_hsml < 0 && ngas > 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_upload_aos_particles: with adaptive gravitational softening "
                     "the smoothing lengths of SphP[] are needed first: use ghip_upload_aos");
  GCHK(ghip_set_counts(ctx, numpart, ngas));
  if(numpart == 0)
    return GHIP_OK;
  GCHK(upload_p_block(ctx, Pp, lay));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  ctx->gt.built = false;
  ctx->st.built = false;
  ctx->gas_wait_upload = ngas > 0;   // (whatever joins in between must leave the gas tree deferred)
  ctx->gas_mixed = lay->p_type >= 0 && *reinterpret_cast<volatile int *>(ghip_gas_mixed_word(ctx)) != 0;
  ctx->gas_list_dirty = true;
  return GHIP_OK;
}

// ... and the SphP[] block, which may follow while a gravity pair is in flight (it is not waited
// for): before the first SPH call of the step.
extern "C" int ghip_upload_aos_gas(ghip_ctx *ctx, const void *Sp, const ghip_layout *lay)
{
  if(!ctx || !lay || (ctx->ngas > 0 && (!Sp || lay->s_stride <= 0)))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_upload_aos_gas: bad arguments");
  GCHK(upload_s_block(ctx, Sp, lay));
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  ctx->gas_wait_upload = false;
  return GHIP_OK;
}

static int download_aos_impl(ghip_ctx *ctx, void *Pp, void *Sp, const ghip_layout *lay,
                             int want_gravity, int want_density, int want_hydro, bool sync)
{
  // the SPH results do not depend on a gravity pair still in flight underneath them
  if(ctx && want_gravity)
    GHIP_JOIN(ctx);
  if(!ctx || !lay)
    return GHIP_EINVAL;
  size_t n = (size_t) ctx->n, ng = (size_t) ctx->ngas;
  if(n == 0)
    return GHIP_OK;
  if(!Pp || (ng > 0 && !Sp))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_download_aos: null record pointers");
  if(ctx->aosP.cap < n * lay->p_stride || (ng > 0 && ctx->aosS.cap < ng * lay->s_stride))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_download_aos: no device image (call ghip_upload_aos)");
  hipStream_t st = ctx->stream;
  void *ip = ctx->aosP.p, *is = ctx->aosS.p;
  bool touchP = false, touchS = false;
  RecMap mp, mg, ms;   // P fields of all particles, P fields of the gas block, SphP fields
  mp.n = n, mp.stride = lay->p_stride, mp.nf = 0;
  mg.n = ng, mg.stride = lay->p_stride, mg.nf = 0;
  ms.n = ng, ms.stride = lay->s_stride, ms.nf = 0;
  if(want_gravity)
    {
      rec_add(mp, lay->p_gravaccel, 3, 0, ctx->f[GHIP_F_GRAVACCEL].p);
      rec_add(mp, lay->p_oldacc, 1, 0, ctx->f[GHIP_F_OLDACC].p);
      rec_add(mp, lay->p_gravpm, 3, 0, ctx->f[GHIP_F_GRAVPM].p);
      rec_add(mp, lay->p_gravcost, 1, 3, ctx->f[GHIP_F_GRAVCOST].p);
      touchP = true;
    }
  if(want_density && ng > 0)
    {
      if(lay->p_hsml >= 0)
        {
          rec_add(mg, lay->p_hsml, 1, 0, ctx->f[GHIP_F_HSML].p);
          rec_add(mg, lay->p_numngb, 1, 0, ctx->f[GHIP_F_NUMNGB].p);
          touchP = true;
        }
      else
        {
          rec_add(ms, lay->s_hsml, 1, 0, ctx->f[GHIP_F_HSML].p);
          rec_add(ms, lay->s_numngb, 1, 0, ctx->f[GHIP_F_NUMNGB].p);
        }
      rec_add(ms, lay->s_density, 1, 0, ctx->f[GHIP_F_DENSITY].p);
      rec_add(ms, lay->s_dhsmlfac, 1, 0, ctx->f[GHIP_F_DHSMLFAC].p);
      rec_add(ms, lay->s_divvel, 1, 0, ctx->f[GHIP_F_DIVVEL].p);
      rec_add(ms, lay->s_curlvel, 1, 0, ctx->f[GHIP_F_CURLVEL].p);
      rec_add(ms, lay->s_pressure, 1, 0, ctx->f[GHIP_F_PRESSURE].p);
      touchS = true;
    }
  if(want_hydro && ng > 0)
    {
      rec_add(ms, lay->s_hydroaccel, 3, 0, ctx->f[GHIP_F_HYDROACCEL].p);
      rec_add(ms, lay->s_dtentropy, 1, 0, ctx->f[GHIP_F_DTENTROPY].p);
      rec_add(ms, lay->s_maxsignalvel, 1, 0, ctx->f[GHIP_F_MAXSIGNALVEL].p);
      touchS = true;
    }
  GCHK(run_records<true>(ctx, mp, ip));
  GCHK(run_records<true>(ctx, mg, ip));
  GCHK(run_records<true>(ctx, ms, is));
  HIPCHK(hipGetLastError());
  if(touchP)
    HIPCHK(hipMemcpyAsync(Pp, ip, n * lay->p_stride, hipMemcpyDeviceToHost, st));
  if(touchS)
    HIPCHK(hipMemcpyAsync(Sp, is, ng * lay->s_stride, hipMemcpyDeviceToHost, st));
  if(!sync)
    return GHIP_OK;
  HIPCHK(ghip_stream_sync(ctx, st));
  return ghip_check_device_errors(ctx);
}

extern "C" int ghip_download_aos(ghip_ctx *ctx, void *Pp, void *Sp, const ghip_layout *lay,
                                 int want_gravity, int want_density, int want_hydro)
{
  return download_aos_impl(ctx, Pp, Sp, lay, want_gravity, want_density, want_hydro, true);
}

extern "C" int ghip_download_aos_async(ghip_ctx *ctx, void *Pp, void *Sp, const ghip_layout *lay,
                                       int want_gravity, int want_density, int want_hydro)
{
  return download_aos_impl(ctx, Pp, Sp, lay, want_gravity, want_density, want_hydro, false);
}

// gravity_tree()'s post-pass and the gravity fields of P[] for a host that called the SPH drivers
// underneath a pair in flight: ordered after the pair on the pair's own stream, not behind the SPH
// kernels queued on the main stream -- the P[] block crosses the link while hydro's tail still runs.
extern "C" int ghip_gravity_to_records(ghip_ctx *ctx, double G, int pmgrid, double comoving_fac,
                                       void *Pp, const ghip_layout *lay)
{
  if(!ctx || !lay)
    return GHIP_EINVAL;
  GCHK(ghip_tree_verify(ctx));
  const size_t n = (size_t) ctx->n;
  if(n == 0)
    return GHIP_OK;
  if(!Pp)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_to_records: null record pointer");
  if(ctx->aosP.cap < n * lay->p_stride)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_gravity_to_records: no device image (call ghip_upload_aos)");
  // (records with Hsml in P[] carry SPH results in the same block: a download queued on the main
  // stream could then copy the block with stale gravity fields over this one -- everything in order on
  // the main stream for such layouts)
  const bool side = ctx->grav_pending && lay->p_hsml < 0;
  hipStream_t st = ctx->stream;
  if(side)
    st = ctx->stream2;   // (the pair's last kernel, the Ewald sums' combine, is on this stream)
  else
    GHIP_JOIN(ctx);
  GCHK(ghip_gravity_finish_on(ctx, G, pmgrid, comoving_fac, 0, st));
  RecMap mp;
  mp.n = n, mp.stride = lay->p_stride, mp.nf = 0;
  rec_add(mp, lay->p_gravaccel, 3, 0, ctx->f[GHIP_F_GRAVACCEL].p);
  rec_add(mp, lay->p_oldacc, 1, 0, ctx->f[GHIP_F_OLDACC].p);
  rec_add(mp, lay->p_gravpm, 3, 0, ctx->f[GHIP_F_GRAVPM].p);
  rec_add(mp, lay->p_gravcost, 1, 3, ctx->f[GHIP_F_GRAVCOST].p);
  GCHK(run_records<true>(ctx, mp, ctx->aosP.p, -1, st));
  HIPCHK(hipMemcpyAsync(Pp, ctx->aosP.p, n * lay->p_stride, hipMemcpyDeviceToHost, st));
  if(side)
    {
      // whatever the main stream does next comes after the pair and after this
      HIPCHK(hipEventRecord(ctx->ev_side, st));
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
      ctx->grav_pending = false;
    }
  HIPCHK(ghip_stream_sync(ctx, st));
  return ghip_check_device_errors(ctx);
}

extern "C" int ghip_gas_block_mixed(ghip_ctx *ctx)
{
  if(!ctx)
    return GHIP_EINVAL;
  return *reinterpret_cast<volatile int *>(ghip_gas_mixed_word(ctx)) != 0;
}

extern "C" int ghip_pin_host(ghip_ctx *ctx, void *ptr, size_t bytes)
{
  if(!ctx || !ptr || bytes == 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_pin_host: bad arguments");
  (void) hipSetDevice(ctx->device);
  for(void *p : ctx->host_pins)
    if(p == ptr)
      return ghip_fail(ctx, GHIP_EINVAL, "ghip_pin_host: this array is already locked (ghip_unpin_host first)");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if(e != hipSuccess)
    {
      (void) hipGetLastError();   // not an error of the path: the copies stay staged
      ctx->err = std::string("ghip_pin_host: the range could not be page-locked (") + hipGetErrorString(e) +
                 "); record copies go through the runtime's staging buffers";
      return GHIP_OK;
    }
  ctx->host_pins.push_back(ptr);
  return GHIP_OK;
}

extern "C" int ghip_unpin_host(ghip_ctx *ctx, void *ptr)
{
  if(!ctx)
    return GHIP_EINVAL;
  for(size_t i = 0; i < ctx->host_pins.size(); i++)
    if(ctx->host_pins[i] == ptr)
      {
        // a copy in flight may still read the range
        GHIP_JOIN(ctx);
        (void) ghip_stream_sync(ctx, ctx->stream);
        (void) hipHostUnregister(ptr);
        (void) hipGetLastError();
        ctx->host_pins.erase(ctx->host_pins.begin() + (long) i);
        return GHIP_OK;
      }
  return GHIP_OK;   // never locked (or the lock had failed): nothing to do
}

__global__ void k_pack_i16(size_t n, char *__restrict__ rec, int stride, int off,
                           const int *__restrict__ src)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    *reinterpret_cast<short *>(rec + i * (size_t) stride + off) = (short) src[i];
}

__global__ void k_pack_i32(size_t n, char *__restrict__ rec, int stride, int off,
                           const int *__restrict__ src)
{
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    *reinterpret_cast<int *>(rec + i * (size_t) stride + off) = src[i];
}

// results of ghip_advance_timesteps into the record images (timestep.c: P[].Vel, TimeBin,
// Ti_begstep; SphP[].VelPred, Entropy, e.DtEntropy)
extern "C" int ghip_download_aos_kick(ghip_ctx *ctx, void *Pp, void *Sp, const ghip_layout *lay)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !lay)
    return GHIP_EINVAL;
  size_t n = (size_t) ctx->n, ng = (size_t) ctx->ngas;
  if(n == 0)
    return GHIP_OK;
  if(!Pp || (ng > 0 && !Sp))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_download_aos_kick: null record pointers");
  if(ctx->aosP.cap < n * lay->p_stride || (ng > 0 && ctx->aosS.cap < ng * lay->s_stride))
    return ghip_fail(ctx, GHIP_EINVAL,
                     "ghip_download_aos_kick: no device image (call ghip_upload_aos)");
  hipStream_t st = ctx->stream;
  void *ip = ctx->aosP.p, *is = ctx->aosS.p;
  PACK64(n, ip, lay->p_stride, lay->p_vel, 3, GHIP_F_VEL);
  if(lay->p_timebin >= 0)
    k_pack_i16<<<cdiv((long long) n, 256), 256, 0, st>>>(n, (char *) ip, lay->p_stride,
                                                         lay->p_timebin,
                                                         P<int>(ctx->f[GHIP_F_TIMEBIN]));
  if(lay->p_ti_begstep >= 0)
    k_pack_i32<<<cdiv((long long) n, 256), 256, 0, st>>>(n, (char *) ip, lay->p_stride,
                                                         lay->p_ti_begstep,
                                                         P<int>(ctx->f[GHIP_F_TI_BEGSTEP]));
  if(ng > 0)
    {
      PACK64(ng, is, lay->s_stride, lay->s_velpred, 3, GHIP_F_VELPRED);
      PACK64(ng, is, lay->s_stride, lay->s_entropy, 1, GHIP_F_ENTROPY);
      PACK64(ng, is, lay->s_stride, lay->s_dtentropy, 1, GHIP_F_DTENTROPY);
    }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(Pp, ip, n * lay->p_stride, hipMemcpyDeviceToHost, st));
  if(ng > 0)
    HIPCHK(hipMemcpyAsync(Sp, is, ng * lay->s_stride, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return ghip_check_device_errors(ctx);
}

// ---------------------------------------------------------------------------------------------
// active list / shard / tree / stats
// ---------------------------------------------------------------------------------------------
extern "C" int ghip_set_active(ghip_ctx *ctx, const int *idx, int nactive)
{
  // "everybody" while everybody already is: nothing changes, and a gravity pair in flight (whose
  // target lists this call would otherwise rebuild) need not be waited for
  if(ctx && !idx && ctx->nactive < 0 && (nactive == 0 || nactive == ctx->n))
    return GHIP_OK;
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || nactive < 0)
    return GHIP_EINVAL;
  if(!idx)
    {
      if(nactive != 0 && nactive != ctx->n)
        return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_active: NULL list means all particles");
      ctx->nactive = (nactive == 0 && ctx->n != 0) ? -1 : -1;
      ctx->lists_dirty = ctx->gas_list_dirty = true;
      return GHIP_OK;
    }
  for(int a = 0; a < nactive; a++)
    if(idx[a] < 0 || idx[a] >= ctx->n)
      return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_active: index %d out of range", idx[a]);
  GCHK(ghip_ensure(ctx, ctx->act_host_idx, (size_t) nactive * 4));
  if(nactive > 0)
    {
      HIPCHK(hipMemcpyAsync(ctx->act_host_idx.p, idx, (size_t) nactive * 4, hipMemcpyHostToDevice,
                            ctx->stream));
      HIPCHK(ghip_stream_sync(ctx, ctx->stream));
    }
  ctx->nactive = nactive;
  ctx->lists_dirty = ctx->gas_list_dirty = true;
  return GHIP_OK;
}

extern "C" int ghip_set_shard(ghip_ctx *ctx, int rank, int nranks)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || nranks < 1 || nranks > GHIP_MAXRANKS || rank < 0 || rank >= nranks)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_shard: need 0 <= rank < nranks <= %d",
                     GHIP_MAXRANKS);
  if(nranks != ctx->shard_n)
    ctx->lists_dirty = ctx->gas_list_dirty = true;   // the stored list order is rank-major
  ctx->shard_rank = rank;
  ctx->shard_n = nranks;
  return GHIP_OK;
}

extern "C" int ghip_tree_build(ghip_ctx *ctx, const double corner[3], const double center[3],
                               double len, const double soft[6])
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !corner || !center || !soft || !(len > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_build: bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  for(int j = 0; j < 3; j++)
    {
      ctx->corner[j] = corner[j];
      ctx->center[j] = center[j];
    }
  ctx->dlen = len;
  for(int j = 0; j < 6; j++)
    ctx->soft[j] = soft[j];
  ctx->dyn_use = false;
  GCHK(ghip_tree_build_impl(ctx));
  // a full build (TreeReconstructFlag): with ghip_set_dynamic_tree the tree sub-steps will drift
  if(ctx->dyn_on)
    GCHK(ghip_dyn_capture(ctx));
  return GHIP_OK;
}

extern "C" int ghip_set_massless_gas_rule(ghip_ctx *ctx, int on)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  const int rule = on == 1 ? 1 : (on ? 3 : 0);
  if(ctx->skip_massless != rule)
    ctx->st.built = false;   // the rule is baked into the gas records
  ctx->skip_massless = rule;
  return GHIP_OK;
}

extern "C" int ghip_set_adaptive_gravsoft(ghip_ctx *ctx, int on)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  if(ctx->adaptive_gravsoft != (on != 0))
    ctx->gt.built = false;   // particle and node softenings are baked into the element records
  ctx->adaptive_gravsoft = (on != 0);
  return GHIP_OK;
}

int ghip_read_slots(ghip_ctx *ctx, DevBuf &buf, unsigned long long out[GHIP_CK_COUNT][2])
{
  std::vector<unsigned long long> h((size_t) GHIP_CK_COUNT * GHIP_CKIND_U64);
  HIPCHK(hipMemcpy(h.data(), buf.p, GHIP_CBUF_BYTES, hipMemcpyDeviceToHost));
  for(int k = 0; k < GHIP_CK_COUNT; k++)
    for(int w = 0; w < 2; w++)
      {
        unsigned long long t = 0;
        for(int s = 0; s < GHIP_CSLOTS; s++)
          t += h[(size_t) k * GHIP_CKIND_U64 + (size_t) s * GHIP_CSLOT_U64 + w];
        out[k][w] = t;
      }
  return GHIP_OK;
}

extern "C" int ghip_get_stats(const ghip_ctx *cctx, ghip_stats *out)
{
  ghip_ctx *ctx = const_cast<ghip_ctx *>(cctx);
  if(!ctx || !out)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  HIPCHK(ghip_stream_sync(ctx, ctx->stream));
  ghip_stats &S = ctx->stats;
  if(ctx->cslots.p)
    {
      unsigned long long c[GHIP_CK_COUNT][2];
      GCHK(ghip_read_slots(ctx, ctx->cslots, c));
      S.grav_wave_steps = (long long) c[GHIP_CK_NEWTON][1];
      S.ewald_wave_steps = (long long) c[GHIP_CK_EWALD][1];
      S.grav_interactions = (long long) c[GHIP_CK_NEWTON][0];
      S.ewald_interactions = (long long) c[GHIP_CK_EWALD][0];
      S.dens_neighbours = (long long) c[GHIP_CK_DENS][0];
      S.hydro_pairs = (long long) c[GHIP_CK_HYDRO][0];
    }
  auto el = [&](int a, int b) {
    float ms = 0;
    if(hipEventElapsedTime(&ms, ctx->evp[a], ctx->evp[b]) != hipSuccess)
      ms = 0;
    return ms;
  };
  S.ms_tree = el(0, 1);
  S.ms_grav = el(2, 3);
  S.ms_ewald = el(4, 5);
  S.ms_dens = el(6, 7);
  S.ms_hmax = el(8, 9);
  S.ms_hydro = el(10, 11);
  S.ms_kick = el(12, 13);
  S.ms_pm = el(14, 15);
  *out = S;
  return ghip_check_device_errors(ctx);
}

extern "C" int ghip_set_async(ghip_ctx *ctx, int on)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  if(!on && ctx->async)
    {
      // leaving asynchronous mode: whatever was deferred is reported now
      HIPCHK(ghip_stream_sync(ctx, ctx->stream));
      ctx->async = false;
      return ghip_check_device_errors(ctx);
    }
  ctx->async = (on != 0);
  return GHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// statistics of a run of steps without a host synchronisation per step
// ---------------------------------------------------------------------------------------------
#define RUN_EV (GHIP_NEV + 2)   // events per ring slot: the phase events + step begin / end marks

extern "C" int ghip_run_begin(ghip_ctx *ctx, int max_steps)
{
  if(!ctx || max_steps < 1)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t want = (size_t) max_steps * RUN_EV;
  while(ctx->ev_ring.size() < want)
    {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      ctx->ev_ring.push_back(e);
    }
  ctx->ring_slots = max_steps;
  ctx->ring_cur = -1;
  ctx->run_steps = 0;
  ctx->run_dens_iter = 0;
  ctx->run_syncs0 = ctx->n_syncs;
  ctx->run_launches0 = ghip_launch_count();
  // (the walks and the SPH kernels add to the run's counter slots themselves)
  HIPCHK(hipMemsetAsync(ctx->rslots.p, 0, GHIP_CBUF_BYTES, ctx->stream));
  return GHIP_OK;
}

extern "C" int ghip_step_begin(ghip_ctx *ctx)
{
  if(!ctx || ctx->ring_slots <= 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_step_begin: call ghip_run_begin first");
  ctx->ring_cur++;
  ctx->evp = &ctx->ev_ring[(size_t) (ctx->ring_cur % ctx->ring_slots) * RUN_EV];
  // (on the main stream, which is in order: after the previous step's end mark)
  HIPCHK(hipEventRecord(ctx->evp[GHIP_NEV], ctx->stream));
  return GHIP_OK;
}

extern "C" int ghip_step_end(ghip_ctx *ctx)
{
  if(!ctx || ctx->ring_slots <= 0 || ctx->ring_cur < 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_step_end: no step in progress");
  GCHK(ghip_join_pair(ctx));   // (a gravity pair still in flight belongs to this step)
  HIPCHK(hipEventRecord(ctx->evp[GHIP_NEV + 1], ctx->stream));
  ctx->run_steps++;
  ctx->run_dens_iter += ctx->stats.dens_iterations;
  return GHIP_OK;
}

extern "C" int ghip_run_end(ghip_ctx *ctx, ghip_run_stats *out)
{
  if(!ctx || !out || ctx->ring_slots <= 0)
    return GHIP_EINVAL;
  memset(out, 0, sizeof(*out));
  out->launches = ghip_launch_count() - ctx->run_launches0;
  out->blocking_syncs = ctx->n_syncs - ctx->run_syncs0;
  GHIP_JOIN(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  unsigned long long acc[GHIP_CK_COUNT][2];
  GCHK(ghip_read_slots(ctx, ctx->rslots, acc));
  out->steps = ctx->run_steps;
  out->grav_interactions = (long long) acc[GHIP_CK_NEWTON][0];
  out->ewald_interactions = (long long) acc[GHIP_CK_EWALD][0];
  out->dens_neighbours = (long long) acc[GHIP_CK_DENS][0];
  out->hydro_pairs = (long long) acc[GHIP_CK_HYDRO][0];
  out->grav_wave_steps = (long long) acc[GHIP_CK_NEWTON][1];
  out->ewald_wave_steps = (long long) acc[GHIP_CK_EWALD][1];
  out->dens_extra_iterations = ctx->run_dens_iter;
  auto el = [](hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if(hipEventElapsedTime(&ms, a, b) != hipSuccess)
      ms = 0;   // (a phase that did not run in that step)
    return (double) ms;
  };
  const long long timed = ctx->run_steps < ctx->ring_slots ? ctx->run_steps : ctx->ring_slots;
  out->steps_timed = timed;
  const long long first = ctx->run_steps - timed;
  hipEvent_t *prev = nullptr, *e0 = nullptr;
  for(long long k = first; k < ctx->run_steps; k++)
    {
      hipEvent_t *e = &ctx->ev_ring[(size_t) (k % ctx->ring_slots) * RUN_EV];
      out->ms_tree += el(e[0], e[1]);
      out->ms_grav += el(e[2], e[3]);
      out->ms_ewald += el(e[4], e[5]);
      out->ms_dens += el(e[6], e[7]);
      out->ms_hmax += el(e[8], e[9]);
      out->ms_hydro += el(e[10], e[11]);
      out->ms_kick += el(e[12], e[13]);
      out->ms_steps_device += el(e[GHIP_NEV], e[GHIP_NEV + 1]);
      if(prev)
        out->ms_between_steps += el(prev[GHIP_NEV + 1], e[GHIP_NEV]);
      else
        e0 = e;
      prev = e;
    }
  if(e0 && prev)
    out->ms_first_to_last = el(e0[GHIP_NEV], prev[GHIP_NEV + 1]);
  ctx->evp = ctx->ev;   // back to the fixed event set of ghip_get_stats
  ctx->ring_slots = 0;
  ctx->ring_cur = -1;
  return ghip_check_device_errors(ctx);
}

extern "C" int ghip_tree_dump(ghip_ctx *ctx, int which, int *nelem, double *xm4, double *cl4,
                              int *lk4, double *aux, int *perm)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !nelem || which < 0 || which > 1)
    return GHIP_EINVAL;
  TreeDev &t = which ? ctx->st : ctx->gt;
  if(!t.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_dump: tree not built");
  *nelem = t.nelem;
  if(t.n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  size_t ne = (size_t) t.nelem;
  if(xm4)
    HIPCHK(hipMemcpyAsync(xm4, t.xm.p, ne * 32, hipMemcpyDeviceToHost, st));
  if(cl4)
    HIPCHK(hipMemcpyAsync(cl4, t.cl.p, ne * 32, hipMemcpyDeviceToHost, st));
  if(lk4)
    HIPCHK(hipMemcpyAsync(lk4, t.lk.p, ne * 16, hipMemcpyDeviceToHost, st));
  if(aux)
    HIPCHK(hipMemcpyAsync(aux, t.aux.p, ne * 8, hipMemcpyDeviceToHost, st));
  if(perm)
    HIPCHK(hipMemcpyAsync(perm, t.perm.p, (size_t) t.n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}
