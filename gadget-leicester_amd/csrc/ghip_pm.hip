// ghip_pm.hip -- "next" row N3: the periodic particle-mesh long-range force of the
// TreePM configurations on the device.
//
// Replaces pmforce_periodic() (pm_periodic.c:199-800): CIC mass assignment onto a PMGRID^3 mesh
// (:226-330), forward FFT, multiplication with the Green's function -exp(-k^2 asmth^2)/k^2 and the
// CIC deconvolution sinc^-4 (:430-486), inverse FFT, 4-point finite differences of the potential
// in each dimension (:489-560) and CIC interpolation of the mesh force to the particles (:640-690)
// into P[].GravPM.  The reference distributes the mesh in slabs over MPI ranks and sorts the
// particles' mesh points to exchange them; on one GPU the whole 128^3 (16 MB) mesh is resident
// and none of that exists.  FFTs: hipFFT (rocFFT), unnormalised in both directions like FFTW --
// the reference's prefactor G/(pi L) * PMGRID/(2L) already assumes that.
//
// Deliberate difference: pm_periodic.c:263-264 clamps `slab_y` where `slab_z` is meant (a typo that
// only matters for a coordinate exactly equal to BoxSize); the clamp is applied to slab_z here.
// The mass assignment uses fp64 atomic adds: the summation order on a mesh point is not fixed,
// so the long-range force is reproducible to rounding (1e-16 relative per mesh point), not bitwise.
#include <hipfft/hipfft.h>

#include "ghip_internal.h"

#define FFTCHK(call)                                                                          \
  do                                                                                          \
    {                                                                                         \
      hipfftResult r_ = (call);                                                               \
      if(r_ != HIPFFT_SUCCESS)                                                                \
        return ghip_fail(ctx, GHIP_EHIP, "%s failed with hipfftResult %d", #call, (int) r_);  \
    }                                                                                         \
  while(0)

// pm_periodic.c:226-330: cloud-in-cell assignment of the particle masses
__global__ void k_pm_deposit(int n, int N, double to_slab_fac, const double *__restrict__ pos,
                             const double *__restrict__ mass, double *__restrict__ rho)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  double px = to_slab_fac * pos[i], py = to_slab_fac * pos[(size_t) n + i],
         pz = to_slab_fac * pos[2 * (size_t) n + i];
  int sx = (int) px, sy = (int) py, sz = (int) pz;
  double dx = px - sx, dy = py - sy, dz = pz - sz;   // (before the clamp, as the reference: :318-325)
  if(sx >= N)
    sx = N - 1;
  if(sy >= N)
    sy = N - 1;
  if(sz >= N)
    sz = N - 1;
  const double m = mass[i];
  for(int xx = 0; xx < 2; xx++)
    for(int yy = 0; yy < 2; yy++)
      for(int zz = 0; zz < 2; zz++)
        {
          int gx = sx + xx, gy = sy + yy, gz = sz + zz;
          if(gx >= N)
            gx -= N;
          if(gy >= N)
            gy -= N;
          if(gz >= N)
            gz -= N;
          double w = m * (xx ? dx : 1.0 - dx) * (yy ? dy : 1.0 - dy) * (zz ? dz : 1.0 - dz);
          atomicAdd(&rho[((size_t) gx * N + gy) * N + gz], w);
        }
}

// pm_periodic.c:430-486 on hipFFT's layout [x][y][z = 0..N/2]
__global__ void k_pm_green(int N, double asmth2, double2 *__restrict__ fk)
{
  const int nz = N / 2 + 1;
  size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(idx >= (size_t) N * N * nz)
    return;
  int z = (int) (idx % nz), y = (int) ((idx / nz) % N), x = (int) (idx / ((size_t) nz * N));
  double kx = x > N / 2 ? x - N : x, ky = y > N / 2 ? y - N : y, kz = z > N / 2 ? z - N : z;
  double k2 = kx * kx + ky * ky + kz * kz;
  double2 v = fk[idx];
  if(k2 > 0)
    {
      double smth = -exp(-k2 * asmth2) / k2;
      double fx = 1, fy = 1, fz = 1;
      if(kx != 0)
        {
          fx = (M_PI * kx) / N;
          fx = sin(fx) / fx;
        }
      if(ky != 0)
        {
          fy = (M_PI * ky) / N;
          fy = sin(fy) / fy;
        }
      if(kz != 0)
        {
          fz = (M_PI * kz) / N;
          fz = sin(fz) / fz;
        }
      double ff = 1 / (fx * fy * fz);
      smth *= ff * ff * ff * ff;
      v.x *= smth;
      v.y *= smth;
    }
  else
    v.x = v.y = 0;   // :485
  fk[idx] = v;
}

// pm_periodic.c:489-560: force_dim = fac * (4/3 (phi[-1] - phi[+1]) - 1/6 (phi[-2] - phi[+2]))
// along dimension dim; the three components as planes of `force`
__global__ void k_pm_gradient(int N, double fac, const double *__restrict__ phi,
                              double *__restrict__ force)
{
  size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n3 = (size_t) N * N * N;
  if(idx >= n3)
    return;
  int z = (int) (idx % N), y = (int) ((idx / N) % N), x = (int) (idx / ((size_t) N * N));
  for(int dim = 0; dim < 3; dim++)
    {
      int c = dim == 0 ? x : dim == 1 ? y : z;
      int l = c - 1, r = c + 1, ll = c - 2, rr = c + 2;
      if(r >= N)
        r -= N;
      if(rr >= N)
        rr -= N;
      if(l < 0)
        l += N;
      if(ll < 0)
        ll += N;
      size_t stride = dim == 0 ? (size_t) N * N : dim == 1 ? (size_t) N : 1;
      size_t base = idx - (size_t) c * stride;
      force[(size_t) dim * n3 + idx] =
        fac * ((4.0 / 3) * (phi[base + l * stride] - phi[base + r * stride]) -
               (1.0 / 6) * (phi[base + ll * stride] - phi[base + rr * stride]));
    }
}

// pm_periodic.c:640-690: CIC interpolation of the mesh force, GravPM[dim] += acc_dim
__global__ void k_pm_interpolate(int n, int N, double to_slab_fac, const double *__restrict__ pos,
                                 const double *__restrict__ force, double *__restrict__ gravpm)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const size_t n3 = (size_t) N * N * N;
  double px = to_slab_fac * pos[i], py = to_slab_fac * pos[(size_t) n + i],
         pz = to_slab_fac * pos[2 * (size_t) n + i];
  int sx = (int) px, sy = (int) py, sz = (int) pz;
  double dx = px - sx, dy = py - sy, dz = pz - sz;
  if(sx >= N)
    sx = N - 1;
  if(sy >= N)
    sy = N - 1;
  if(sz >= N)
    sz = N - 1;
  double acc[3] = {0, 0, 0};
  for(int xx = 0; xx < 2; xx++)
    for(int yy = 0; yy < 2; yy++)
      for(int zz = 0; zz < 2; zz++)
        {
          int gx = sx + xx, gy = sy + yy, gz = sz + zz;
          if(gx >= N)
            gx -= N;
          if(gy >= N)
            gy -= N;
          if(gz >= N)
            gz -= N;
          double w = (xx ? dx : 1.0 - dx) * (yy ? dy : 1.0 - dy) * (zz ? dz : 1.0 - dz);
          size_t g = ((size_t) gx * N + gy) * N + gz;
          for(int dim = 0; dim < 3; dim++)
            acc[dim] += force[(size_t) dim * n3 + g] * w;   // same corner order as the reference
        }
  for(int dim = 0; dim < 3; dim++)
    gravpm[(size_t) dim * n + i] += acc[dim];
}

static int pm_check(ghip_ctx *ctx, const ghip_pm_params *p)
{
  const int N = p->pmgrid;
  if(N < 4 || N > 2048 || (N & 1) || !(p->BoxSize > 0) || !(p->Asmth > 0))
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_pm_periodic: need an even PMGRID >= 4, BoxSize > 0, Asmth > 0");
  return GHIP_OK;
}

// plans and mesh buffers for PMGRID = N
static int pm_prepare(ghip_ctx *ctx, int N)
{
  hipStream_t st = ctx->stream;
  const size_t n3 = (size_t) N * N * N, nk = (size_t) N * N * (N / 2 + 1);
  if(ctx->pm_n != N)
    {
      if(ctx->pm_n)
        {
          (void) hipfftDestroy((hipfftHandle) ctx->pm_fwd);
          (void) hipfftDestroy((hipfftHandle) ctx->pm_inv);
          ctx->pm_n = 0;
        }
      hipfftHandle f, b;
      FFTCHK(hipfftPlan3d(&f, N, N, N, HIPFFT_D2Z));
      FFTCHK(hipfftPlan3d(&b, N, N, N, HIPFFT_Z2D));
      FFTCHK(hipfftSetStream(f, st));
      FFTCHK(hipfftSetStream(b, st));
      ctx->pm_fwd = (void *) f;
      ctx->pm_inv = (void *) b;
      ctx->pm_n = N;
    }
  GCHK(ghip_ensure(ctx, ctx->pm_rho, n3 * sizeof(double)));
  GCHK(ghip_ensure(ctx, ctx->pm_k, nk * sizeof(double2)));
  GCHK(ghip_ensure(ctx, ctx->pm_force, 3 * n3 * sizeof(double)));
  return GHIP_OK;
}

// mass of this context's particles onto the (zeroed) mesh; GRAVPM zeroed as long_range_force() does
static int pm_deposit(ghip_ctx *ctx, const ghip_pm_params *p)
{
  hipStream_t st = ctx->stream;
  const int N = p->pmgrid, n = ctx->n;
  const size_t n3 = (size_t) N * N * N;
  const double to_slab_fac = N / p->BoxSize;                   // pm_periodic.c:102
  double *rho = P<double>(ctx->pm_rho);
  HIPCHK(hipMemsetAsync(rho, 0, n3 * sizeof(double), st));
  if(n > 0)
    {
      HIPCHK(hipMemsetAsync(ctx->f[GHIP_F_GRAVPM].p, 0, (size_t) n * 3 * sizeof(double), st));
      k_pm_deposit<<<cdiv(n, 256), 256, 0, st>>>(n, N, to_slab_fac, P<double>(ctx->f[GHIP_F_POS]),
                                                 P<double>(ctx->f[GHIP_F_MASS]), rho);
      HIPCHK(hipGetLastError());
    }
  return GHIP_OK;
}

// mesh density -> potential -> force field -> GRAVPM of this context's particles
static int pm_solve_and_interpolate(ghip_ctx *ctx, const ghip_pm_params *p)
{
  hipStream_t st = ctx->stream;
  const int N = p->pmgrid, n = ctx->n;
  const size_t n3 = (size_t) N * N * N, nk = (size_t) N * N * (N / 2 + 1);
  double *rho = P<double>(ctx->pm_rho), *force = P<double>(ctx->pm_force);
  double2 *fk = P<double2>(ctx->pm_k);
  const double to_slab_fac = N / p->BoxSize;
  double asmth2 = (2 * M_PI) * p->Asmth / p->BoxSize;          // :221-222
  asmth2 *= asmth2;
  double fac = p->G / (M_PI * p->BoxSize);                     // :224-225
  fac *= 1 / (2 * p->BoxSize / N);
  FFTCHK(hipfftExecD2Z((hipfftHandle) ctx->pm_fwd, rho, reinterpret_cast<hipfftDoubleComplex *>(fk)));
  k_pm_green<<<cdiv((long long) nk, 256), 256, 0, st>>>(N, asmth2, fk);
  HIPCHK(hipGetLastError());
  FFTCHK(hipfftExecZ2D((hipfftHandle) ctx->pm_inv, reinterpret_cast<hipfftDoubleComplex *>(fk), rho));
  k_pm_gradient<<<cdiv((long long) n3, 256), 256, 0, st>>>(N, fac, rho, force);
  if(n > 0)
    k_pm_interpolate<<<cdiv(n, 256), 256, 0, st>>>(n, N, to_slab_fac, P<double>(ctx->f[GHIP_F_POS]),
                                                   force, P<double>(ctx->f[GHIP_F_GRAVPM]));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

extern "C" int ghip_pm_periodic(ghip_ctx *ctx, const ghip_pm_params *p)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !p)
    return GHIP_EINVAL;
  GCHK(pm_check(ctx, p));
  if(ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_pm_periodic: on a multi-GPU shard use GHIP_DD_PM");
  if(ctx->n == 0)
    return GHIP_OK;
  hipStream_t st = ctx->stream;
  GCHK(pm_prepare(ctx, p->pmgrid));
  HIPCHK(hipEventRecord(ctx->evp[14], st));
  GCHK(pm_deposit(ctx, p));
  GCHK(pm_solve_and_interpolate(ctx, p));
  HIPCHK(hipEventRecord(ctx->evp[15], st));
  return GHIP_OK;
}

// rank-ordered sum of the all-gathered meshes (identical on every shard, whatever the transport)
__global__ void k_pm_sum_meshes(size_t n3, int nranks, const double *__restrict__ all,
                                double *__restrict__ rho)
{
  size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= n3)
    return;
  double s = 0;
  for(int r = 0; r < nranks; r++)
    s += all[(size_t) r * n3 + g];
  rho[g] = s;
}

// GHIP_DD_PM (ghip_dd_begin / ghip_dd_step): phase 0 deposits and posts the all-gather of the meshes,
// phase 1 adds them and solves
int ghip_dd_pm_begin(ghip_ctx *ctx)
{
  GHIP_JOIN(ctx);
  return pm_check(ctx, &ctx->dd.pm);
}

int ghip_dd_pm_step(ghip_ctx *ctx)
{
  DDState &D = ctx->dd;
  const ghip_pm_params *p = &D.pm;
  hipStream_t st = ctx->stream;
  const size_t n3 = (size_t) p->pmgrid * p->pmgrid * p->pmgrid;
  if(D.phase == 0)
    {
      GCHK(pm_prepare(ctx, p->pmgrid));
      HIPCHK(hipEventRecord(ctx->evp[14], st));
      GCHK(pm_deposit(ctx, p));
      ghip_dd_set_allgather(D, ctx->pm_rho.p, n3 * sizeof(double), &D.pm_all);
      D.phase = 1;
      return 1;
    }
  if(D.phase == 1)
    {
      k_pm_sum_meshes<<<cdiv((long long) n3, 256), 256, 0, st>>>(n3, D.nranks, P<double>(D.pm_all),
                                                                 P<double>(ctx->pm_rho));
      HIPCHK(hipGetLastError());
      GCHK(pm_solve_and_interpolate(ctx, p));
      HIPCHK(hipEventRecord(ctx->evp[15], st));
      D.op = 0;
      return 0;
    }
  return ghip_fail(ctx, GHIP_EINVAL, "ghip_dd_step: the mesh force has no phase %d", D.phase);
}

void ghip_pm_release(ghip_ctx *ctx)
{
  if(ctx && ctx->pm_n)
    {
      (void) hipfftDestroy((hipfftHandle) ctx->pm_fwd);
      (void) hipfftDestroy((hipfftHandle) ctx->pm_inv);
      ctx->pm_n = 0;
    }
}
