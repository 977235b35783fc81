// ghip_export.hip -- "next" row N2: hand the device-built tree back to the host in
// the reference's own representation, so that host walks that are not part of the GPU path
// (potential.c, the black-hole / dust neighbour loops) keep working on Nodes[] / Extnodes[] /
// Nextnode[] / Father[] without force_treebuild() (forcetree.c:67-872) running on the CPU.
//
// The device tree has exactly the reference's cells, children in the same (Morton digit) order,
// so the exported links thread the same walk: node index = MaxPart + pre-order rank (the root is
// Nodes[MaxPart], forcetree.c:134), particles keep their host indices, "nextnode" is the next
// element of the pre-order list, "sibling" the skip link (-1 past the end), "father" the
// enclosing node.  On top of the monopoles the gravity walk needs, the moments of
// force_update_node_recursive (forcetree.c:468-872) that only host code reads are computed here,
// level by level like the other moments: vs (mass-weighted velocity), vmax, hmax and divVmax
// (gas only, floored at 0 as in the reference) and the MULTIPLEPARTICLES count.
// Struct layouts differ with the -D flags (NODE is 88 or 96 bytes, ...), so records are written
// through a byte-offset table like the particle records.
#include "ghip_internal.h"

// forcetree.h:13-20
#define BITFLAG_MAX_SOFTENING_TYPE 2
#define BITFLAG_MIXED_SOFTENINGS_IN_NODE 5
#define BITFLAG_MULTIPLEPARTICLES 7

// per-element extension moments: ev = (vs or v, vmax), eh = (hmax, divVmax), ecnt = particles
// counted the reference's way (saturating at 2), efather = enclosing node element
__global__ void k_ext_particles(int nelem, int n, int ngas, const int4 *__restrict__ lk,
                                const int *__restrict__ perm, const double *__restrict__ vel,
                                const int *__restrict__ type, const double *__restrict__ hsml,
                                const double *__restrict__ divvel, double4 *__restrict__ ev,
                                double2 *__restrict__ eh, int *__restrict__ ecnt,
                                int *__restrict__ efather)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(e == 0)
    efather[0] = -1;
  if(me.y < 0)
    return;
  int i = perm[me.y];
  double vx = vel[i], vy = vel[(size_t) n + i], vz = vel[2 * (size_t) n + i];
  double vmax = fmax(fabs(vx), fmax(fabs(vy), fabs(vz)));   // forcetree.c:608-610
  ev[e] = make_double4(vx, vy, vz, vmax);
  double hm = 0, dv = 0;
  if(type[i] == 0 && i < ngas)   // forcetree.c:600-606
    {
      hm = fmax(hsml[i], 0.0);
      dv = fmax(divvel[i], 0.0);
    }
  eh[e] = make_double2(hm, dv);
  ecnt[e] = 1;
}

__global__ void k_ext_level(int nelem, int level, const int4 *__restrict__ lk,
                            const double4 *__restrict__ xm, double4 *__restrict__ ev,
                            double2 *__restrict__ eh, int *__restrict__ ecnt,
                            int *__restrict__ efather)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y != -(level + 1))
    return;
  double mass = 0, vx = 0, vy = 0, vz = 0, vmax = 0, hmax = 0, dvmax = 0;
  int count = 0;
  for(int c = e + 1; c < me.x;)
    {
      int4 ck = lk[c];
      double m = xm[c].w;
      double4 v = ev[c];
      double2 h = eh[c];
      mass += m;
      vx += m * v.x;
      vy += m * v.y;
      vz += m * v.z;
      vmax = fmax(vmax, v.w);
      hmax = fmax(hmax, h.x);
      dvmax = fmax(dvmax, h.y);
      if(ck.y >= 0)
        count++;            // a particle counts whatever its mass (forcetree.c:589)
      else if(m > 0)
        count += ecnt[c];   // forcetree.c:570-576: 2 for a MULTIPLEPARTICLES node, else 1
      efather[c] = e;
      c = ck.x;
    }
  if(mass != 0)
    {
      vx /= mass;
      vy /= mass;
      vz /= mass;
    }
  else
    vx = vy = vz = 0;
  ev[e] = make_double4(vx, vy, vz, vmax);
  eh[e] = make_double2(hmax, dvmax);
  ecnt[e] = count > 1 ? 2 : 1;
}

struct ExportK
{
  ghip_node_layout lay;
  int maxpart, ti_current, unequal, adaptive;
  double soft[6];
};

__device__ __forceinline__ int d_export_index(int e, int nelem, int maxpart,
                                              const int4 *__restrict__ lk,
                                              const int *__restrict__ perm)
{
  if(e < 0 || e >= nelem)
    return -1;
  int4 k = lk[e];
  return k.y >= 0 ? perm[k.y] : maxpart + (e - k.z);   // k.z = particles before a node element
}

template <class T> __device__ __forceinline__ void d_put(char *rec, int off, T v)
{
  if(off >= 0)
    *reinterpret_cast<T *>(rec + off) = v;
}

__global__ void k_export(int nelem, ExportK K, const int4 *__restrict__ lk,
                         const double4 *__restrict__ cl, const double4 *__restrict__ xm,
                         const double *__restrict__ aux, const double4 *__restrict__ ev,
                         const double2 *__restrict__ eh, const int *__restrict__ ecnt,
                         const int *__restrict__ efather, const int *__restrict__ perm,
                         char *__restrict__ nodes, char *__restrict__ ext,
                         int *__restrict__ nextnode_p, int *__restrict__ father_p)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  int father = d_export_index(efather[e], nelem, K.maxpart, lk, perm);
  int next = d_export_index(e + 1, nelem, K.maxpart, lk, perm);
  if(me.y >= 0)
    {
      int i = perm[me.y];
      nextnode_p[i] = next;   // forcetree.c:858-866
      father_p[i] = father;
      return;
    }
  const size_t rank = (size_t) (e - me.z);
  char *nd = nodes + rank * K.lay.node_stride;
  char *ex = ext + rank * K.lay.ext_stride;
  const double4 c = cl[e], m = xm[e], v = ev[e];
  const double2 h = eh[e];
  d_put<double>(nd, K.lay.n_len, c.w);
  for(int j = 0; j < 3; j++)
    {
      d_put<double>(nd, K.lay.n_center >= 0 ? K.lay.n_center + 8 * j : -1, j == 0 ? c.x : j == 1 ? c.y : c.z);
      d_put<double>(nd, K.lay.n_s >= 0 ? K.lay.n_s + 8 * j : -1, j == 0 ? m.x : j == 1 ? m.y : m.z);
      d_put<double>(ex, K.lay.e_vs >= 0 ? K.lay.e_vs + 8 * j : -1, j == 0 ? v.x : j == 1 ? v.y : v.z);
      d_put<double>(ex, K.lay.e_dp >= 0 ? K.lay.e_dp + 8 * j : -1, 0.0);
    }
  d_put<double>(nd, K.lay.n_mass, m.w);
  unsigned int flags = ecnt[e] > 1 ? (1u << BITFLAG_MULTIPLEPARTICLES) : 0u;
  if(K.adaptive)
    d_put<double>(nd, K.lay.n_maxsoft, fabs(aux[e]));   // forcetree.c:845-846
  else if(K.unequal)
    {
      // forcetree.c:612-700 (UNEQUALSOFTENINGS): type of the largest softening below, mixed flag
      double a = aux[e];
      double amax = fabs(a);
      int t = 0;
      for(int q = 0; q < 6; q++)
        if(K.soft[q] == amax)
          {
            t = q;
            break;
          }
      flags |= (unsigned int) t << BITFLAG_MAX_SOFTENING_TYPE;
      if(a < 0)
        flags |= 1u << BITFLAG_MIXED_SOFTENINGS_IN_NODE;
    }
  d_put<unsigned int>(nd, K.lay.n_bitflags, flags);
  d_put<int>(nd, K.lay.n_sibling, d_export_index(me.x, nelem, K.maxpart, lk, perm));
  d_put<int>(nd, K.lay.n_nextnode, next);
  d_put<int>(nd, K.lay.n_father, father);
  d_put<int>(nd, K.lay.n_ti_current, K.ti_current);
  d_put<double>(ex, K.lay.e_vmax, v.w);
  d_put<double>(ex, K.lay.e_hmax, h.x);
  d_put<double>(ex, K.lay.e_divvmax, h.y);
  d_put<int>(ex, K.lay.e_ti_lastkicked, K.ti_current);
  d_put<int>(ex, K.lay.e_flag, 0);
}

extern "C" int ghip_tree_export(ghip_ctx *ctx, const ghip_node_layout *lay, int MaxPart,
                                int Ti_Current, int unequal_softenings, void *Nodes_base,
                                void *Extnodes_base, int *Nextnode, int *Father, int max_nodes,
                                int *numnodes)
{
  if(ctx)
    GHIP_JOIN(ctx);
  if(!ctx || !lay || !numnodes)
    return GHIP_EINVAL;
  TreeDev &t = ctx->gt;
  if(!t.built)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_export: no tree (call ghip_tree_build)");
  *numnodes = t.nnodes;
  if(t.n == 0)
    return GHIP_OK;
  if(MaxPart < ctx->n || !Nodes_base || !Extnodes_base || !Nextnode || !Father)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_export: bad arguments");
  if(t.nnodes > max_nodes)
    return ghip_fail(ctx, GHIP_ENOMEM,
                     "ghip_tree_export: %d nodes do not fit MaxNodes = %d (reference: "
                     "endrun(1), forcetree.c:287)", t.nnodes, max_nodes);
  if(lay->node_stride <= 0 || lay->ext_stride <= 0)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_export: bad layout");
  hipStream_t st = ctx->stream;
  const int nelem = t.nelem, n = ctx->n;
  const size_t nb = (size_t) t.nnodes * lay->node_stride, eb = (size_t) t.nnodes * lay->ext_stride;
  // scratch: [ev double4][eh double2][ecnt int][efather int][nodes][ext][nextnode int n][father int n]
  size_t off_ev = 0, off_eh = off_ev + (size_t) nelem * sizeof(double4),
         off_cnt = off_eh + (size_t) nelem * sizeof(double2), off_fa = off_cnt + (size_t) nelem * 4,
         off_nd = (off_fa + (size_t) nelem * 4 + 15) & ~(size_t) 15, off_ex = (off_nd + nb + 15) & ~(size_t) 15,
         off_nn = (off_ex + eb + 15) & ~(size_t) 15, off_fp = off_nn + (size_t) n * 4,
         total = off_fp + (size_t) n * 4;
  GCHK(ghip_ensure(ctx, ctx->stage, total));
  char *base = reinterpret_cast<char *>(ctx->stage.p);
  double4 *ev = reinterpret_cast<double4 *>(base + off_ev);
  double2 *eh = reinterpret_cast<double2 *>(base + off_eh);
  int *ecnt = reinterpret_cast<int *>(base + off_cnt), *efather = reinterpret_cast<int *>(base + off_fa);
  char *dnodes = base + off_nd, *dext = base + off_ex;
  int *dnn = reinterpret_cast<int *>(base + off_nn), *dfp = reinterpret_cast<int *>(base + off_fp);
  HIPCHK(hipMemsetAsync(dnodes, 0, nb, st));
  HIPCHK(hipMemsetAsync(dext, 0, eb, st));
  k_ext_particles<<<cdiv(nelem, 256), 256, 0, st>>>(
    nelem, n, ctx->ngas, P<int4>(t.lk), P<int>(t.perm), P<double>(ctx->f[GHIP_F_VEL]),
    P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->f[GHIP_F_HSML]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    ev, eh, ecnt, efather);
  for(int L = t.maxlevel; L >= 0; L--)
    k_ext_level<<<cdiv(nelem, 256), 256, 0, st>>>(nelem, L, P<int4>(t.lk), P<double4>(t.xm), ev, eh,
                                                  ecnt, efather);
  ExportK K;
  K.lay = *lay;
  K.maxpart = MaxPart;
  K.ti_current = Ti_Current;
  K.unequal = unequal_softenings;
  K.adaptive = ctx->adaptive_gravsoft;
  for(int q = 0; q < 6; q++)
    K.soft[q] = ctx->soft[q];
  k_export<<<cdiv(nelem, 256), 256, 0, st>>>(nelem, K, P<int4>(t.lk), P<double4>(t.cl),
                                             P<double4>(t.xm), P<double>(t.aux), ev, eh, ecnt,
                                             efather, P<int>(t.perm), dnodes, dext, dnn, dfp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(Nodes_base, dnodes, nb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(Extnodes_base, dext, eb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(Nextnode, dnn, (size_t) n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(Father, dfp, (size_t) n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}


// ---------------------------------------------------------------------------------------------
// The gravity tree between two full builds ("dynamic tree update", forcetree.c:1356-1520).
//
// On a sub-step the reference does not rebuild its tree: nodes keep the cells of the last build,
// their centres of mass move with the mass-weighted velocity vs, their side length grows by
// 2 vmax dt so that the cell still covers its particles, and the momentum the kicked particles gained
// is folded into vs at the next drift.  Interaction sets on such a tree differ from those on a tree
// of the current positions, so a host that wants the reference's sub-step forces needs THIS tree.
// Kept here as a second element list (ctx->dyn): a copy of the gravity tree of the last
// ghip_tree_build with (vs, vmax) and the pending kicks per node.  ghip_tree_substep rebuilds the
// tree of the current positions as ever (target order, the gas tree and everything SPH derive from
// it; neighbour sets are geometric and do not depend on which tree finds them) and brings the kept
// tree to the current time; the gravity walks then read the kept tree's records.
// The reference drifts a node when a walk or a kick first meets it; here all nodes move at once, which
// gives the same state up to the rounding of s += vs dt in one piece or in several.
// ---------------------------------------------------------------------------------------------
static int dyn_copy(ghip_ctx *ctx, DevBuf &dst, const DevBuf &src, size_t bytes)
{
  GCHK(ghip_ensure(ctx, dst, bytes > 0 ? bytes : 16));
  if(bytes > 0)
    HIPCHK(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return GHIP_OK;
}

int ghip_dyn_capture(ghip_ctx *ctx)
{
  TreeDev &g = ctx->gt, &d = ctx->dyn;
  ctx->dyn_valid = false;
  ctx->dyn_use = false;
  if(!g.built || g.n == 0)
    return GHIP_OK;
  if(ctx->dd.on)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_dynamic_tree: not on a multi-GPU shard (the merged tree of "
                     "a domain-decomposed run is rebuilt every call)");
  if(ctx->adaptive_gravsoft)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_set_dynamic_tree: not with ADAPTIVE_GRAVSOFT_FORGAS (the nodes' "
                     "maxsoft follows the smoothing lengths)");
  GCHK(ghip_tree_verify(ctx));   // exact sizes on the host
  hipStream_t st = ctx->stream;
  const size_t ne = (size_t) g.nelem, n = (size_t) g.n;
  d.n = g.n;
  d.nnodes = g.nnodes;
  d.nelem = g.nelem;
  d.maxlevel = g.maxlevel;
  d.cap_nodes = g.nnodes;
  GCHK(dyn_copy(ctx, d.xm, g.xm, ne * sizeof(double4)));
  GCHK(dyn_copy(ctx, d.cl, g.cl, ne * sizeof(double4)));
  GCHK(dyn_copy(ctx, d.lk, g.lk, ne * sizeof(int4)));
  GCHK(dyn_copy(ctx, d.aux, g.aux, (ne + 1) * sizeof(double)));
  GCHK(dyn_copy(ctx, d.perm, g.perm, n * sizeof(int)));
  GCHK(dyn_copy(ctx, d.dsz, g.dsz, sizeof(TreeSizes)));
  d.hsz = g.hsz;
  GCHK(ghip_ensure(ctx, ctx->dyn_ev, ne * sizeof(double4)));
  GCHK(ghip_ensure(ctx, ctx->dyn_dp, ne * sizeof(double4)));
  GCHK(ghip_ensure(ctx, ctx->dyn_kick, ne * sizeof(double4)));
  GCHK(ghip_ensure(ctx, ctx->dyn_eh, ne * sizeof(double2)));
  GCHK(ghip_ensure(ctx, ctx->dyn_cnt, ne * 4));
  GCHK(ghip_ensure(ctx, ctx->dyn_fa, ne * 4));
  HIPCHK(hipMemsetAsync(ctx->dyn_dp.p, 0, ne * sizeof(double4), st));
  const int nelem = g.nelem;
  // vs, vmax per element (force_update_node_recursive, forcetree.c:560-611, 830-846)
  k_ext_particles<<<cdiv(nelem, 256), 256, 0, st>>>(
    nelem, ctx->n, ctx->ngas, P<int4>(d.lk), P<int>(d.perm), P<double>(ctx->f[GHIP_F_VEL]),
    P<int>(ctx->f[GHIP_F_TYPE]), P<double>(ctx->f[GHIP_F_HSML]), P<double>(ctx->f[GHIP_F_DIVVEL]),
    P<double4>(ctx->dyn_ev), P<double2>(ctx->dyn_eh), P<int>(ctx->dyn_cnt), P<int>(ctx->dyn_fa));
  for(int L = d.maxlevel; L >= 0; L--)
    k_ext_level<<<cdiv(nelem, 256), 256, 0, st>>>(nelem, L, P<int4>(d.lk), P<double4>(d.xm),
                                                  P<double4>(ctx->dyn_ev), P<double2>(ctx->dyn_eh),
                                                  P<int>(ctx->dyn_cnt), P<int>(ctx->dyn_fa));
  HIPCHK(hipGetLastError());
  d.built = true;
  GCHK(ghip_build_segments(ctx, d, true));
  ctx->dyn_valid = true;
  return GHIP_OK;
}

void ghip_dyn_release(ghip_ctx *ctx)
{
  DevBuf *bs[] = {&ctx->dyn.xm, &ctx->dyn.cl, &ctx->dyn.lk, &ctx->dyn.aux, &ctx->dyn.perm, &ctx->dyn.dsz,
                  &ctx->dyn.seg_start, &ctx->dyn.seg_nanc, &ctx->dyn.seg_anc, &ctx->dyn.mq, &ctx->dyn.mq2,
                  &ctx->dyn_ev, &ctx->dyn_dp, &ctx->dyn_eh, &ctx->dyn_cnt, &ctx->dyn_fa, &ctx->dyn_kick,
                  &ctx->kick_dv, &ctx->kick_flag};
  for(DevBuf *b : bs)
    {
      if(b->p)
        (void) hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
  ctx->dyn_valid = ctx->dyn_use = false;
}

// force_kick_node (forcetree.c:1455-1520), first half: what a kicked particle hands to its ancestors --
// (Mass dv, max_j |Vel_j| of the NEW velocity); a particle that was not kicked hands up (0, -1)
__global__ void k_dyn_kick_particles(int nelem, int n, const int4 *__restrict__ lk,
                                     const int *__restrict__ perm, const double *__restrict__ mass,
                                     const double *__restrict__ vel, const double *__restrict__ dv,
                                     const double *__restrict__ vmaxk, const int *__restrict__ flag,
                                     double4 *__restrict__ up)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y < 0)
    return;
  int i = perm[me.y];
  if(!flag[i])
    {
      up[e] = make_double4(0, 0, 0, -1);
      return;
    }
  double m = mass[i];
  // (vmaxk: the caller's max_j |Vel_j| where the resident VEL is not the kicked one yet)
  double vmax = vmaxk ? vmaxk[i]
                      : fmax(fabs(vel[i]), fmax(fabs(vel[(size_t) n + i]), fabs(vel[2 * (size_t) n + i])));
  up[e] = make_double4(m * dv[i], m * dv[(size_t) n + i], m * dv[2 * (size_t) n + i], vmax);
}

// second half, level by level from the deepest: a node sums its children's contributions (list order),
// records them in its pending dp, raises its vmax and the KICKED flag (dp.w) -- only nodes with a kicked
// particle below are touched, like the ancestor loop of the reference
__global__ void k_dyn_kick_level(int nelem, int level, const int4 *__restrict__ lk,
                                 double4 *__restrict__ up, double4 *__restrict__ dp,
                                 double4 *__restrict__ ev)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y != -(level + 1))
    return;
  double sx = 0, sy = 0, sz = 0, vmax = -1;
  for(int c = e + 1; c < me.x;)
    {
      double4 u = up[c];
      if(u.w >= 0)
        {
          sx += u.x;
          sy += u.y;
          sz += u.z;
          vmax = fmax(vmax, u.w);
        }
      c = lk[c].x;
    }
  up[e] = make_double4(sx, sy, sz, vmax);
  if(vmax >= 0)
    {
      double4 p = dp[e];
      dp[e] = make_double4(p.x + sx, p.y + sy, p.z + sz, 1.0);
      double4 v = ev[e];
      if(v.w < vmax)
        ev[e] = make_double4(v.x, v.y, v.z, vmax);
    }
}

static int dyn_kick_pass(ghip_ctx *ctx, const double *dv, const int *flag, const double *vmaxk = nullptr)
{
  TreeDev &d = ctx->dyn;
  hipStream_t st = ctx->stream;
  const int nelem = d.nelem;
  k_dyn_kick_particles<<<cdiv(nelem, 256), 256, 0, st>>>(
    nelem, ctx->n, P<int4>(d.lk), P<int>(d.perm), P<double>(ctx->f[GHIP_F_MASS]),
    P<double>(ctx->f[GHIP_F_VEL]), dv, vmaxk, flag, P<double4>(ctx->dyn_kick));
  for(int L = d.maxlevel; L >= 0; L--)
    k_dyn_kick_level<<<cdiv(nelem, 256), 256, 0, st>>>(nelem, L, P<int4>(d.lk), P<double4>(ctx->dyn_kick),
                                                       P<double4>(ctx->dyn_dp), P<double4>(ctx->dyn_ev));
  HIPCHK(hipGetLastError());
  return GHIP_OK;
}

// what the last ghip_advance_timesteps recorded (kick_dv, kick_flag)
int ghip_dyn_kick_recorded(ghip_ctx *ctx)
{
  if(!ctx->dyn_on || !ctx->dyn_valid || ctx->dyn.n != ctx->n)
    return GHIP_OK;
  return dyn_kick_pass(ctx, P<double>(ctx->kick_dv), P<int>(ctx->kick_flag));
}

// (a particle kicked twice at one sync point -- gravity, then a feedback kick -- hands up the sum of
// its kicks and the largest of its |Vel|: the rows were cleared, every entry adds)
__global__ void k_dyn_scatter_kicks(int nk, int n, const int *__restrict__ idx, const double *__restrict__ dv3,
                                    const double *__restrict__ vmaxk, double *__restrict__ dv,
                                    int *__restrict__ flag)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= nk)
    return;
  int i = idx[k];
  if(i < 0 || i >= n)
    return;
  flag[i] = 1;
  for(int j = 0; j < 3; j++)
    atomicAdd(&dv[(size_t) j * n + i], dv3[3 * (size_t) k + j]);
  if(vmaxk)   // (non-negative doubles order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned long long *>(&dv[3 * (size_t) n + i]),
              (unsigned long long) __double_as_longlong(vmaxk[k]));
}

static int kick_nodes_host(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3, const double *vmaxk);

extern "C" int ghip_tree_kick_nodes(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3)
{
  return kick_nodes_host(ctx, nkicked, idx, dv3, nullptr);
}

extern "C" int ghip_tree_kick_nodes_vmax(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3,
                                         const double *vmaxk)
{
  if(!vmaxk && nkicked > 0)
    return GHIP_EINVAL;
  return kick_nodes_host(ctx, nkicked, idx, dv3, vmaxk);
}

static int kick_nodes_host(ghip_ctx *ctx, int nkicked, const int *idx, const double *dv3, const double *vmaxk)
{
  if(!ctx || nkicked < 0 || (nkicked > 0 && (!idx || !dv3)))
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  if(!ctx->dyn_on || !ctx->dyn_valid)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_kick_nodes: no kept tree (ghip_set_dynamic_tree, then "
                     "ghip_tree_build)");
  if(nkicked == 0)
    return GHIP_OK;
  const size_t n = (size_t) ctx->n;
  hipStream_t st = ctx->stream;
  GCHK(ghip_ensure(ctx, ctx->kick_dv, 4 * n * 8));
  GCHK(ghip_ensure(ctx, ctx->kick_flag, n * 4));
  GCHK(ghip_ensure(ctx, ctx->stage, (size_t) nkicked * 36 + 64));
  int *didx = P<int>(ctx->stage);
  double *ddv = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->stage.p) + (((size_t) nkicked * 4 + 15) & ~(size_t) 15));
  double *dvm = ddv + 3 * (size_t) nkicked;
  HIPCHK(hipMemcpyAsync(didx, idx, (size_t) nkicked * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ddv, dv3, (size_t) nkicked * 24, hipMemcpyHostToDevice, st));
  if(vmaxk)
    HIPCHK(hipMemcpyAsync(dvm, vmaxk, (size_t) nkicked * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(ctx->kick_flag.p, 0, n * 4, st));
  HIPCHK(hipMemsetAsync(ctx->kick_dv.p, 0, 4 * n * 8, st));
  k_dyn_scatter_kicks<<<cdiv(nkicked, 256), 256, 0, st>>>(nkicked, ctx->n, didx, ddv, vmaxk ? dvm : nullptr,
                                                          P<double>(ctx->kick_dv), P<int>(ctx->kick_flag));
  GCHK(dyn_kick_pass(ctx, P<double>(ctx->kick_dv), P<int>(ctx->kick_flag),
                     vmaxk ? P<double>(ctx->kick_dv) + 3 * n : nullptr));
  HIPCHK(ghip_stream_sync(ctx, st));   // (idx / dv3 are the caller's)
  return GHIP_OK;
}

// force_drift_node (forcetree.c:1356-1452) for every node, and the particles' current positions and
// masses into the kept element list (the reference reads P[] live in its walks)
__global__ void k_dyn_drift(int nelem, int n, const int4 *__restrict__ lk, const int *__restrict__ perm,
                            const double *__restrict__ pos, const double *__restrict__ mass,
                            double dt_drift, double4 *__restrict__ xm, double4 *__restrict__ cl,
                            double4 *__restrict__ ev, double4 *__restrict__ dp)
{
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if(e >= nelem)
    return;
  int4 me = lk[e];
  if(me.y >= 0)
    {
      int i = perm[me.y];
      xm[e] = make_double4(pos[i], pos[(size_t) n + i], pos[2 * (size_t) n + i], mass[i]);
      return;
    }
  double4 x = xm[e], v = ev[e], p = dp[e];
  if(p.w != 0)   // BITFLAG_NODEHASBEENKICKED: :1368-1400
    {
      double fac = x.w != 0 ? 1 / x.w : 0;
      v.x += fac * p.x;
      v.y += fac * p.y;
      v.z += fac * p.z;
      ev[e] = v;
      dp[e] = make_double4(0, 0, 0, 0);
    }
  x.x += v.x * dt_drift;   // :1440-1441
  x.y += v.y * dt_drift;
  x.z += v.z * dt_drift;
  xm[e] = x;
  double4 c = cl[e];
  c.w += 2 * v.w * dt_drift;   // :1442
  cl[e] = c;
}

extern "C" int ghip_set_dynamic_tree(ghip_ctx *ctx, int on)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  ctx->dyn_on = on != 0;
  if(!ctx->dyn_on)
    ghip_dyn_release(ctx);
  return GHIP_OK;
}

extern "C" int ghip_tree_substep(ghip_ctx *ctx, double dt_drift)
{
  if(!ctx)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  if(!ctx->dyn_on || !ctx->dyn_valid)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_substep: no kept tree (ghip_set_dynamic_tree, then a full "
                     "ghip_tree_build)");
  if(ctx->dyn.n != ctx->n)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_substep: the particle number changed since the full build "
                     "(%d -> %d): rebuild", ctx->dyn.n, ctx->n);
  // the tree of the current positions: target order, the gas tree, everything SPH
  ctx->dyn_use = false;
  GCHK(ghip_tree_build_impl(ctx));
  TreeDev &d = ctx->dyn;
  k_dyn_drift<<<cdiv(d.nelem, 256), 256, 0, ctx->stream>>>(
    d.nelem, ctx->n, P<int4>(d.lk), P<int>(d.perm), P<double>(ctx->f[GHIP_F_POS]),
    P<double>(ctx->f[GHIP_F_MASS]), dt_drift, P<double4>(d.xm), P<double4>(d.cl), P<double4>(ctx->dyn_ev),
    P<double4>(ctx->dyn_dp));
  HIPCHK(hipGetLastError());
  GCHK(ghip_fill_walk_records(ctx, d));
  ctx->dyn_use = true;
  return GHIP_OK;
}

// the kept tree's nodes in pre-order: s, mass, len, vs, vmax (tests)
extern "C" int ghip_tree_dump_dynamic(ghip_ctx *ctx, int *nelem, double *xm4, double *cl4, double *ev4, int *lk4)
{
  if(!ctx || !nelem)
    return GHIP_EINVAL;
  GHIP_JOIN(ctx);
  if(!ctx->dyn_valid)
    return ghip_fail(ctx, GHIP_EINVAL, "ghip_tree_dump_dynamic: no kept tree");
  TreeDev &d = ctx->dyn;
  *nelem = d.nelem;
  hipStream_t st = ctx->stream;
  const size_t ne = (size_t) d.nelem;
  if(xm4)
    HIPCHK(hipMemcpyAsync(xm4, d.xm.p, ne * 32, hipMemcpyDeviceToHost, st));
  if(cl4)
    HIPCHK(hipMemcpyAsync(cl4, d.cl.p, ne * 32, hipMemcpyDeviceToHost, st));
  if(ev4)
    HIPCHK(hipMemcpyAsync(ev4, ctx->dyn_ev.p, ne * 32, hipMemcpyDeviceToHost, st));
  if(lk4)
    HIPCHK(hipMemcpyAsync(lk4, d.lk.p, ne * 16, hipMemcpyDeviceToHost, st));
  HIPCHK(ghip_stream_sync(ctx, st));
  return GHIP_OK;
}
