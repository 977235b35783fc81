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
