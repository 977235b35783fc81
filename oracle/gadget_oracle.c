/*
 * gadget_oracle.c -- CPU restatement of the GADGET-3 (Leicester fork) per-step force path.
 *
 * TEST INFRASTRUCTURE ONLY (see gadget_oracle.h).  PARITY UNPINNED BY UPSTREAM: no reference
 * fixtures exist and the reference cannot be built here (GSL absent, stand-ins forbidden).
 * Every function cites the reference file:line whose algorithm it restates.  The code is
 * written from the algorithm, on SoA arrays, and is not a copy of the reference source.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC)
 */
#include "gadget_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * keys
 * ---------------------------------------------------------------------------------------- */

/* peano.c:320-333: Morton key, x is the lowest bit of every triplet */
orc_key orc_morton_key(int x, int y, int z, int bits)
{
  orc_key m = 0;
  for(int b = bits - 1; b >= 0; b--)
    {
      m <<= 3;
      m |= (orc_key) (((z >> b) & 1) << 2 | ((y >> b) & 1) << 1 | ((x >> b) & 1));
    }
  return m;
}

/* peano.c:300-316 (table-driven state machine, tables peano.c:195-295).
 *
 * Restated generatively instead of with the 48x8 tables: the reference curve is the 3-D
 * Hilbert curve whose base cell visits the octants pix = 4*xbit + 2*ybit + zbit in the order
 * given by ph_base[] and whose 8 sub-cells carry the cube symmetries ph_child[] (a signed axis
 * permutation each: out[i] = in[perm[i]] ^ flip[i]).  Walking one level down with orientation g
 * (initially the identity): local = g(pix); digit = ph_base[local]; g <- ph_child[local] o g.
 * The 48 states of the reference tables are exactly the 48 signed permutations reachable this
 * way; state numbering does not enter the key. */
static const unsigned char ph_base[8] = { 0, 7, 1, 6, 3, 4, 2, 5 };
static const unsigned char ph_child_perm[8][3] = {
  {0, 2, 1}, {0, 2, 1}, {2, 1, 0}, {2, 1, 0}, {0, 1, 2}, {0, 1, 2}, {2, 1, 0}, {2, 1, 0}
};
static const unsigned char ph_child_flip[8][3] = {
  {0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 0}, {0, 0, 0}, {1, 0, 1}
};

orc_key orc_peano_hilbert_key(int x, int y, int z, int bits)
{
  unsigned char perm[3] = { 0, 1, 2 }, flip[3] = { 0, 0, 0 };
  orc_key key = 0;
  for(int b = bits - 1; b >= 0; b--)
    {
      unsigned char v[3] = { (unsigned char) ((x >> b) & 1), (unsigned char) ((y >> b) & 1),
        (unsigned char) ((z >> b) & 1) };
      unsigned char w[3];
      for(int i = 0; i < 3; i++)
        w[i] = v[perm[i]] ^ flip[i];
      int local = w[0] * 4 + w[1] * 2 + w[2];
      key = (key << 3) | ph_base[local];
      /* g <- child o g : (a o b).perm[i] = b.perm[a.perm[i]], flip[i] = b.flip[a.perm[i]] ^ a.flip[i] */
      unsigned char np[3], nf[3];
      for(int i = 0; i < 3; i++)
        {
          np[i] = perm[ph_child_perm[local][i]];
          nf[i] = flip[ph_child_perm[local][i]] ^ ph_child_flip[local][i];
        }
      memcpy(perm, np, 3);
      memcpy(flip, nf, 3);
    }
  return key;
}

/* domain.c:1972-2014 domain_findExtent */
void orc_domain_extent(int n, const double *pos, double corner[3], double center[3], double *len)
{
  double xmin[3] = { 1e300, 1e300, 1e300 }, xmax[3] = { -1e300, -1e300, -1e300 };
  for(int i = 0; i < n; i++)
    for(int j = 0; j < 3; j++)
      {
        if(xmin[j] > pos[3 * i + j])
          xmin[j] = pos[3 * i + j];
        if(xmax[j] < pos[3 * i + j])
          xmax[j] = pos[3 * i + j];
      }
  double l = 0;
  for(int j = 0; j < 3; j++)
    if(xmax[j] - xmin[j] > l)
      l = xmax[j] - xmin[j];
  l *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      center[j] = 0.5 * (xmin[j] + xmax[j]);
      corner[j] = 0.5 * (xmin[j] + xmax[j]) - 0.5 * l;
    }
  *len = l;
}

/* ------------------------------------------------------------------------------------------
 * tree
 * ---------------------------------------------------------------------------------------- */

typedef struct
{
  double len, center[3];
  int suns[8];
  /* filled by the recursive moment pass */
  double s[3], mass;
  int multi;          /* BITFLAG_MULTIPLEPARTICLES */
  int sibling, nextnode, father;
  double maxsoft;     /* largest ForceSoftening[type] of any particle below (UNEQUALSOFTENINGS) */
  int mixedsoft;      /* BITFLAG_MIXED_SOFTENINGS_IN_NODE */
  int softset;        /* maxsofttype != 7 */
  /* extNODE */
  double hmax, vmax, divVmax, vs[3];
  double dp[3];       /* Extnodes[].dp: momentum kicks not yet folded into vs (forcetree.c:1483-1490) */
  int kicked;         /* BITFLAG_NODEHASBEENKICKED */
} onode;

struct orc_tree
{
  int n;                /* = All.MaxPart: particle indices [0,n), node indices [n, n+numnodes) */
  int maxnodes, numnodes;
  onode *nodes;         /* nodes[k] is node index n+k */
  int *nextnode;        /* Nextnode[] for particles */
  int *father;          /* Father[] for particles */
  const double *pos, *vel, *mass, *hsml, *divvel;
  const int *type;
  double soft[6];
  int last;
  int adaptive;         /* ADAPTIVE_GRAVSOFT_FORGAS: a gas particle's softening is its Hsml */
};

#define NODE(t, no) ((t)->nodes[(no) - (t)->n])

static int new_node(orc_tree *t, int parent, int subnode)
{
  if(t->numnodes >= t->maxnodes)
    {
      int nm = t->maxnodes * 2;
      onode *nn = (onode *) realloc(t->nodes, (size_t) nm * sizeof(onode));
      if(!nn)
        return -1;
      t->nodes = nn;
      t->maxnodes = nm;
    }
  int no = t->n + t->numnodes++;
  onode *c = &NODE(t, no);
  onode *p = &NODE(t, parent);
  /* forcetree.c:263-279 geometry of a daughter cell */
  double lenhalf = 0.25 * p->len;
  c->len = 0.5 * p->len;
  c->center[0] = (subnode & 1) ? p->center[0] + lenhalf : p->center[0] - lenhalf;
  c->center[1] = (subnode & 2) ? p->center[1] + lenhalf : p->center[1] - lenhalf;
  c->center[2] = (subnode & 4) ? p->center[2] + lenhalf : p->center[2] - lenhalf;
  for(int j = 0; j < 8; j++)
    c->suns[j] = -1;
  return no;
}

/* forcetree.c:384-429 force_create_empty_nodes: complete grid of empty nodes down to `depth` */
static int create_empty_nodes(orc_tree *t, int no, int depth)
{
  if(depth <= 0)
    return 0;
  for(int sub = 0; sub < 8; sub++)
    {
      int c = new_node(t, no, sub);
      if(c < 0)
        return -1;
      NODE(t, no).suns[sub] = c;
      if(create_empty_nodes(t, c, depth - 1) < 0)
        return -1;
    }
  return 0;
}

/* deterministic stand-in for get_random_number() in the NOTREERND branch (forcetree.c:219-232);
 * only reached for (near-)coincident particles, which the tests avoid */
static double tiny_rng(unsigned int id)
{
  id = id * 1664525u + 1013904223u;
  id ^= id >> 15;
  id *= 2246822519u;
  id ^= id >> 13;
  return (id & 0xFFFFFF) / (double) 0x1000000;
}

/* forcetree.c:468-872 force_update_node_recursive */
static void update_node_recursive(orc_tree *t, int no, int sib, int father)
{
  if(no >= t->n)
    {
      onode *nd = &NODE(t, no);
      int suns[8];
      memcpy(suns, nd->suns, sizeof(suns));

      if(t->last >= 0)
        {
          if(t->last >= t->n)
            NODE(t, t->last).nextnode = no;
          else
            t->nextnode[t->last] = no;
        }
      t->last = no;

      double mass = 0, s[3] = { 0, 0, 0 }, vs[3] = { 0, 0, 0 };
      double hmax = 0, vmax = 0, divVmax = 0, maxsoft = 0;
      int count_particles = 0, softset = 0, mixed = 0;

      for(int j = 0; j < 8; j++)
        {
          int p = suns[j];
          if(p < 0)
            continue;
          int jj, pp = -1;
          for(jj = j + 1; jj < 8; jj++)
            if((pp = suns[jj]) >= 0)
              break;
          int nextsib = (jj < 8) ? pp : sib;

          update_node_recursive(t, p, nextsib, no);
          nd = &NODE(t, no); /* realloc-safe: no allocation happens here, but keep it simple */

          if(p >= t->n)
            {
              onode *c = &NODE(t, p);
              mass += c->mass;
              s[0] += c->mass * c->s[0];
              s[1] += c->mass * c->s[1];
              s[2] += c->mass * c->s[2];
              vs[0] += c->mass * c->vs[0];
              vs[1] += c->mass * c->vs[1];
              vs[2] += c->mass * c->vs[2];
              if(c->mass > 0)
                count_particles += c->multi ? 2 : 1;
              if(c->hmax > hmax)
                hmax = c->hmax;
              if(c->vmax > vmax)
                vmax = c->vmax;
              if(c->divVmax > divVmax)
                divVmax = c->divVmax;
              /* forcetree.c:612-636 softening bookkeeping, by value instead of by type */
              mixed |= c->mixedsoft;
              if(c->softset)
                {
                  if(!softset)
                    {
                      maxsoft = c->maxsoft;
                      softset = 1;
                    }
                  else if(c->maxsoft > maxsoft)
                    {
                      maxsoft = c->maxsoft;
                      mixed = 1;
                    }
                  else if(c->maxsoft < maxsoft)
                    mixed = 1;
                }
            }
          else
            {
              count_particles++;
              double m = t->mass[p];
              mass += m;
              s[0] += m * t->pos[3 * p + 0];
              s[1] += m * t->pos[3 * p + 1];
              s[2] += m * t->pos[3 * p + 2];
              vs[0] += m * t->vel[3 * p + 0];
              vs[1] += m * t->vel[3 * p + 1];
              vs[2] += m * t->vel[3 * p + 2];
              if(t->type[p] == 0)
                {
                  if(t->hsml && t->hsml[p] > hmax)
                    hmax = t->hsml[p];
                  if(t->divvel && t->divvel[p] > divVmax)
                    divVmax = t->divvel[p];
                }
              for(int k = 0; k < 3; k++)
                {
                  double v = fabs(t->vel[3 * p + k]);
                  if(v > vmax)
                    vmax = v;
                }
              double sp = t->soft[t->type[p]];
              if(!softset)
                {
                  maxsoft = sp;
                  softset = 1;
                }
              else if(sp > maxsoft)
                {
                  maxsoft = sp;
                  mixed = 1;
                }
              else if(sp < maxsoft)
                mixed = 1;
            }
        }

      if(mass)
        {
          for(int k = 0; k < 3; k++)
            {
              s[k] /= mass;
              vs[k] /= mass;
            }
        }
      else
        {
          for(int k = 0; k < 3; k++)
            {
              s[k] = nd->center[k];
              vs[k] = 0;
            }
        }
      nd->mass = mass;
      for(int k = 0; k < 3; k++)
        {
          nd->s[k] = s[k];
          nd->vs[k] = vs[k];
        }
      nd->hmax = hmax;
      nd->vmax = vmax;
      nd->divVmax = divVmax;
      nd->dp[0] = nd->dp[1] = nd->dp[2] = 0;
      nd->kicked = 0;
      nd->multi = (count_particles > 1);
      nd->maxsoft = maxsoft;
      nd->softset = softset;
      nd->mixedsoft = mixed;
      nd->sibling = sib;
      nd->father = father;
    }
  else
    {
      if(t->last >= 0)
        {
          if(t->last >= t->n)
            NODE(t, t->last).nextnode = no;
          else
            t->nextnode[t->last] = no;
        }
      t->last = no;
      t->father[no] = father;
    }
}

/* forcetree.c:125-361 force_treebuild_single (single rank: no pseudo particles) */
orc_tree *orc_tree_build(int n, const double *pos, const double *vel, const double *mass,
                         const int *type, const double *hsml, const double *divvel,
                         const double soft[6], const double corner[3], const double center[3],
                         double len, int toplevels)
{
  orc_tree *t = (orc_tree *) calloc(1, sizeof(orc_tree));
  if(!t)
    return NULL;
  t->n = n;
  t->maxnodes = (int) (2.0 * n) + 4096;
  t->nodes = (onode *) malloc((size_t) t->maxnodes * sizeof(onode));
  t->nextnode = (int *) malloc((size_t) (n > 0 ? n : 1) * sizeof(int));
  t->father = (int *) malloc((size_t) (n > 0 ? n : 1) * sizeof(int));
  orc_key *morton_list = (orc_key *) malloc((size_t) (n > 0 ? n : 1) * sizeof(orc_key));
  if(!t->nodes || !t->nextnode || !t->father || !morton_list)
    {
      free(morton_list);
      orc_tree_free(t);
      return NULL;
    }
  t->pos = pos;
  t->vel = vel;
  t->mass = mass;
  t->type = type;
  t->hsml = hsml;
  t->divvel = divvel;
  memcpy(t->soft, soft, 6 * sizeof(double));

  double domainfac = 1.0 / len * (double) (((orc_key) 1) << ORC_BITS_PER_DIMENSION);

  /* root, forcetree.c:137-151 */
  int root = t->n;
  t->numnodes = 1;
  t->nodes[0].len = len;
  for(int j = 0; j < 3; j++)
    t->nodes[0].center[j] = center[j];
  for(int j = 0; j < 8; j++)
    t->nodes[0].suns[j] = -1;
  if(create_empty_nodes(t, root, toplevels) < 0)
    {
      free(morton_list);
      orc_tree_free(t);
      return NULL;
    }

  for(int i = 0; i < n; i++)
    {
      int rep = 0;
      orc_key morton = orc_morton_key((int) ((pos[3 * i + 0] - corner[0]) * domainfac),
                                      (int) ((pos[3 * i + 1] - corner[1]) * domainfac),
                                      (int) ((pos[3 * i + 2] - corner[2]) * domainfac),
                                      ORC_BITS_PER_DIMENSION);
      morton_list[i] = morton;
      int shift = 3 * (ORC_BITS_PER_DIMENSION - 1);
      int th = root, parent = -1, subnode = 0;

      while(1)
        {
          if(th >= t->n)
            {
              onode *nd = &NODE(t, th);
              if(shift >= 0)
                subnode = (int) ((morton >> shift) & 7);
              else
                {
                  subnode = 0;
                  if(pos[3 * i + 0] > nd->center[0])
                    subnode += 1;
                  if(pos[3 * i + 1] > nd->center[1])
                    subnode += 2;
                  if(pos[3 * i + 2] > nd->center[2])
                    subnode += 4;
                }
              if(nd->len < 1.0e-3 * soft[type[i]])
                {
                  subnode = (int) (8.0 * tiny_rng((unsigned) (i + rep)));
                  if(subnode >= 8)
                    subnode = 7;
                }
              int nn = nd->suns[subnode];
              shift -= 3;
              if(nn >= 0)
                {
                  parent = th;
                  th = nn;
                  rep++;
                }
              else
                {
                  nd->suns[subnode] = i;
                  break;
                }
            }
          else
            {
              /* leaf holding particle th: make a new internal node (forcetree.c:253-346) */
              int nf = new_node(t, parent, subnode);
              if(nf < 0)
                {
                  free(morton_list);
                  orc_tree_free(t);
                  return NULL;
                }
              NODE(t, parent).suns[subnode] = nf;
              onode *nd = &NODE(t, nf);
              int sub2;
              if(shift >= 0)
                sub2 = (int) ((morton_list[th] >> shift) & 7);
              else
                {
                  sub2 = 0;
                  if(pos[3 * th + 0] > nd->center[0])
                    sub2 += 1;
                  if(pos[3 * th + 1] > nd->center[1])
                    sub2 += 2;
                  if(pos[3 * th + 2] > nd->center[2])
                    sub2 += 4;
                }
              if(nd->len < 1.0e-3 * soft[type[th]])
                {
                  sub2 = (int) (8.0 * tiny_rng((unsigned) (th + rep)));
                  if(sub2 >= 8)
                    sub2 = 7;
                }
              nd->suns[sub2] = th;
              th = nf;
            }
        }
    }
  free(morton_list);

  t->last = -1;
  update_node_recursive(t, root, -1, -1);
  if(t->last >= t->n)
    NODE(t, t->last).nextnode = -1;
  else if(t->last >= 0)
    t->nextnode[t->last] = -1;
  return t;
}

void orc_tree_free(orc_tree *t)
{
  if(!t)
    return;
  free(t->nodes);
  free(t->nextnode);
  free(t->father);
  free(t);
}

int orc_tree_numnodes(const orc_tree *t)
{
  return t->numnodes;
}

void orc_tree_dump(const orc_tree *t, double *len, double *center3, double *s3, double *mass,
                   int *sibling, int *nextnode, int *father, int *multi, double *hmax)
{
  for(int k = 0; k < t->numnodes; k++)
    {
      const onode *nd = &t->nodes[k];
      len[k] = nd->len;
      mass[k] = nd->mass;
      for(int j = 0; j < 3; j++)
        {
          center3[3 * k + j] = nd->center[j];
          s3[3 * k + j] = nd->s[j];
        }
      sibling[k] = nd->sibling;
      nextnode[k] = nd->nextnode;
      father[k] = nd->father;
      multi[k] = nd->multi;
      hmax[k] = nd->hmax;
    }
}

/* the extNODE members and the softening flags of the nodes (forcetree.c:612-700, 830-846) */
void orc_tree_dump_ext(const orc_tree *t, double *vs3, double *vmax, double *divvmax,
                       double *maxsoft, int *mixedsoft)
{
  for(int k = 0; k < t->numnodes; k++)
    {
      const onode *nd = &t->nodes[k];
      for(int j = 0; j < 3; j++)
        vs3[3 * k + j] = nd->vs[j];
      vmax[k] = nd->vmax;
      divvmax[k] = nd->divVmax;
      maxsoft[k] = nd->maxsoft;
      mixedsoft[k] = nd->mixedsoft;
    }
}

void orc_tree_dump_particles(const orc_tree *t, int *nextnode, int *father)
{
  memcpy(nextnode, t->nextnode, (size_t) t->n * sizeof(int));
  memcpy(father, t->father, (size_t) t->n * sizeof(int));
}

/* forcetree.c:1661-1786 force_update_hmax (single rank part) */
void orc_update_hmax(orc_tree *t, int nactive, const int *active, const double *hsml,
                     const double *divvel)
{
  for(int a = 0; a < nactive; a++)
    {
      int i = active[a];
      if(t->type[i] != 0)
        continue;
      int no = t->father[i];
      while(no >= 0)
        {
          onode *nd = &NODE(t, no);
          double dv = divvel ? divvel[i] : 0;
          if(hsml[i] > nd->hmax || dv > nd->divVmax)
            {
              if(hsml[i] > nd->hmax)
                nd->hmax = hsml[i];
              if(dv > nd->divVmax)
                nd->divVmax = dv;
            }
          else
            break;
          no = nd->father;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * the tree between two builds (sub-steps): forcetree.c:1356-1520
 *
 * The reference drifts a node when a walk or a kick first meets it at the current time
 * (force_drift_node, called from forcetree.c:1759, 2007, 2568, 3372, 3914, ngb.c:134, 272, 607 and
 * from force_kick_node).  A node's state at time T does not depend on WHEN it was brought there --
 * only on the kicks recorded up to T -- except for the rounding of s += vs * dt in one piece or in
 * several, so the restatement updates all nodes at once:
 *   at a sync point:  orc_tree_drift_nodes(dt)   every node from the previous sync point to this one
 *                     (walks)
 *                     orc_tree_kick_nodes(...)   the particles kicked at this sync point
 * ---------------------------------------------------------------------------------------- */

/* force_drift_node (forcetree.c:1356-1452) for every node: a node kicked at the previous sync point
 * first folds the recorded momentum into vs (:1368-1400), then s += vs * dt_drift, the side length
 * grows by 2 vmax dt_drift (:1437-1438) and hmax by exp(divVmax dt_drift_hmax / 3) (:1449).
 * dt_drift = dt_drift_hmax = (time1 - time0) * Timebase_interval, or get_drift_factor(time0, time1)
 * in comoving runs (:1403-1425; FLTROUNDOFFREDUCTION off). */
void orc_tree_drift_nodes(orc_tree *t, double dt_drift, double dt_drift_hmax)
{
  for(int k = 0; k < t->numnodes; k++)
    {
      onode *nd = &t->nodes[k];
      if(nd->kicked)
        {
          double fac = nd->mass ? 1 / nd->mass : 0;
          for(int j = 0; j < 3; j++)
            {
              nd->vs[j] += fac * nd->dp[j];
              nd->dp[j] = 0;
            }
          nd->kicked = 0;
        }
      for(int j = 0; j < 3; j++)
        nd->s[j] += nd->vs[j] * dt_drift;
      nd->len += 2 * nd->vmax * dt_drift;
      nd->hmax *= exp(0.333333333333 * nd->divVmax * dt_drift_hmax);
    }
}

/* force_kick_node (forcetree.c:1455-1520) for the particles idx[0, n) whose velocities just changed by
 * dv (the tree's vel[] already holds the NEW velocities, as P[i].Vel does at timestep.c:588): every
 * ancestor records dp += Mass * dv, vmax = max(vmax, max_j |Vel[j]|) and the KICKED flag. */
void orc_tree_kick_nodes(orc_tree *t, int n, const int *idx, const double *dv3)
{
  for(int k = 0; k < n; k++)
    {
      int i = idx[k];
      double dp[3], vmax = 0;
      for(int j = 0; j < 3; j++)
        {
          dp[j] = t->mass[i] * dv3[3 * (size_t) k + j];
          double v = fabs(t->vel[3 * (size_t) i + j]);
          if(v > vmax)
            vmax = v;
        }
      int no = t->father[i];
      while(no >= 0)
        {
          onode *nd = &NODE(t, no);
          for(int j = 0; j < 3; j++)
            nd->dp[j] += dp[j];
          if(nd->vmax < vmax)
            nd->vmax = vmax;
          nd->kicked = 1;
          no = nd->father;
        }
    }
}

/* s, len, vs, vmax of every node (dump order of orc_tree_dump_nodes) */
void orc_tree_dump_dynamic(const orc_tree *t, double *s3, double *len, double *vs3, double *vmax)
{
  for(int k = 0; k < t->numnodes; k++)
    {
      const onode *nd = &t->nodes[k];
      for(int j = 0; j < 3; j++)
        {
          s3[3 * k + j] = nd->s[j];
          vs3[3 * k + j] = nd->vs[j];
        }
      len[k] = nd->len;
      vmax[k] = nd->vmax;
    }
}

/* ------------------------------------------------------------------------------------------
 * gravity walks
 * ---------------------------------------------------------------------------------------- */

static inline double nearest(double x, double boxsize, double boxhalf)
{
  /* forcetree.c:49 NEAREST */
  return (x > boxhalf) ? (x - boxsize) : ((x < -boxhalf) ? (x + boxsize) : x);
}

/* the softened monopole kernel, forcetree.c:2143-2171 */
static inline double grav_fac(double mass, double r2, double r, double h)
{
  if(r >= h)
    return mass / (r2 * r);
  double h_inv = 1.0 / h;
  double h3_inv = h_inv * h_inv * h_inv;
  double u = r * h_inv;
  if(u < 0.5)
    return mass * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
  return mass * h3_inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u -
                          0.066666666667 / (u * u * u));
}

/* softening of particle i: All.ForceSoftening[type], or its Hsml for gas under
 * ADAPTIVE_GRAVSOFT_FORGAS (forcetree.c:1851-1856, 2038-2058) */
static inline double particle_soft(const orc_tree *t, int i)
{
  if(t->adaptive && t->type[i] == 0)
    return t->hsml ? t->hsml[i] : 0;
  return t->soft[t->type[i]];
}

/* ADAPTIVE_GRAVSOFT_FORGAS (forcetree.c:535-541, 634-635, 705-726, 845-846): NODE.maxsoft = the
 * largest softening below, gas particles counting with Hsml; the walk then opens a node whenever
 * the target lies inside that softening (forcetree.c:2125-2139), so mixedsoft is set throughout.
 * Children are created after their parents, so one reverse sweep sees every child first. */
void orc_tree_adaptive_gravsoft(orc_tree *t)
{
  t->adaptive = 1;
  for(int k = t->numnodes - 1; k >= 0; k--)
    {
      onode *nd = &t->nodes[k];
      double maxsoft = 0;
      for(int j = 0; j < 8; j++)
        {
          int p = nd->suns[j];
          if(p < 0)
            continue;
          double sp = (p >= t->n) ? NODE(t, p).maxsoft : particle_soft(t, p);
          if(sp > maxsoft)
            maxsoft = sp;
        }
      nd->maxsoft = maxsoft;
      nd->mixedsoft = 1;
    }
}

static float shortrange_table[ORC_NTAB];
static int shortrange_ready = 0;

static void shortrange_init(void)
{
  /* forcetree.c:4195-4202 */
  if(shortrange_ready)
    return;
  for(int i = 0; i < ORC_NTAB; i++)
    {
      double u = 3.0 / ORC_NTAB * (i + 0.5);
      shortrange_table[i] = (float) (erfc(u) + 2.0 * u / sqrt(M_PI) * exp(-u * u));
    }
  shortrange_ready = 1;
}

/* forcetree.c:1797-2317 (shortrange == 0) and forcetree.c:2330-2845 (shortrange == 1) */
static int treeevaluate(const orc_tree *t, const orc_grav_params *p, int shortrange,
                        const double tpos[3], int ptype, double tsoft, double oldacc, double acc[3])
{
  /* tsoft: the target's own softening (All.ForceSoftening[ptype]; Hsml of a gas target under
   * ADAPTIVE_GRAVSOFT_FORGAS, forcetree.c:1851-1856 / gravdata_in.Soft :1875-1878) */
  const int unequal = p->unequal_softenings || t->adaptive;
  double pos_x = tpos[0], pos_y = tpos[1], pos_z = tpos[2];
  double aold = p->ErrTolForceAcc * oldacc;
  double boxsize = p->BoxSize, boxhalf = 0.5 * p->BoxSize;
  double acc_x = 0, acc_y = 0, acc_z = 0;
  int ninteractions = 0;
  double h = tsoft;
  double rcut = p->rcut, rcut2 = rcut * rcut;
  double asmthfac = shortrange ? 0.5 / p->asmth * (ORC_NTAB / 3.0) : 0;

  int no = t->n;
  while(no >= 0)
    {
      double dx, dy, dz, mass, r2;
      if(no < t->n)
        {
          dx = t->pos[3 * no + 0] - pos_x;
          dy = t->pos[3 * no + 1] - pos_y;
          dz = t->pos[3 * no + 2] - pos_z;
          mass = t->mass[no];
          if(p->periodic)
            {
              dx = nearest(dx, boxsize, boxhalf);
              dy = nearest(dy, boxsize, boxhalf);
              dz = nearest(dz, boxsize, boxhalf);
            }
          r2 = dx * dx + dy * dy + dz * dz;
          if(unequal)
            {
              h = tsoft;
              if(h < particle_soft(t, no))
                h = particle_soft(t, no);
            }
          no = t->nextnode[no];
        }
      else
        {
          const onode *nop = &NODE(t, no);
          mass = nop->mass;
          if(!nop->multi)
            {
              /* forcetree.c:1996-2004: open if it has mass; forcetree.c:2560-2565: always */
              if(shortrange || mass)
                {
                  no = nop->nextnode;
                  continue;
                }
            }
          dx = nop->s[0] - pos_x;
          dy = nop->s[1] - pos_y;
          dz = nop->s[2] - pos_z;
          if(p->periodic)
            {
              dx = nearest(dx, boxsize, boxhalf);
              dy = nearest(dy, boxsize, boxhalf);
              dz = nearest(dz, boxsize, boxhalf);
            }
          r2 = dx * dx + dy * dy + dz * dz;

          if(shortrange && r2 > rcut2)
            {
              /* forcetree.c:2598-2632 */
              double eff_dist = rcut + 0.5 * nop->len;
              double dist = nop->center[0] - pos_x;
              if(p->periodic)
                dist = nearest(dist, boxsize, boxhalf);
              if(dist < -eff_dist || dist > eff_dist)
                {
                  no = nop->sibling;
                  continue;
                }
              dist = nop->center[1] - pos_y;
              if(p->periodic)
                dist = nearest(dist, boxsize, boxhalf);
              if(dist < -eff_dist || dist > eff_dist)
                {
                  no = nop->sibling;
                  continue;
                }
              dist = nop->center[2] - pos_z;
              if(p->periodic)
                dist = nearest(dist, boxsize, boxhalf);
              if(dist < -eff_dist || dist > eff_dist)
                {
                  no = nop->sibling;
                  continue;
                }
            }

          if(p->ErrTolTheta)
            {
              if(nop->len * nop->len > r2 * p->ErrTolTheta * p->ErrTolTheta)
                {
                  no = nop->nextnode;
                  continue;
                }
            }
          else
            {
              if(mass * nop->len * nop->len > r2 * r2 * aold)
                {
                  no = nop->nextnode;
                  continue;
                }
              if(fabs(nop->center[0] - pos_x) < 0.60 * nop->len)
                if(fabs(nop->center[1] - pos_y) < 0.60 * nop->len)
                  if(fabs(nop->center[2] - pos_z) < 0.60 * nop->len)
                    {
                      no = nop->nextnode;
                      continue;
                    }
            }

          if(unequal)
            {
              /* forcetree.c:2108-2139 */
              h = tsoft;
              if(h < nop->maxsoft)
                {
                  h = nop->maxsoft;
                  if(r2 < h * h)
                    if(nop->mixedsoft)
                      {
                        no = nop->nextnode;
                        continue;
                      }
                }
            }
          no = nop->sibling;
        }

      double r = sqrt(r2);
      double fac = grav_fac(mass, r2, r, h);
      if(shortrange)
        {
          /* forcetree.c:2739-2752 */
          int tabindex = (int) (asmthfac * r);
          if(tabindex < ORC_NTAB)
            {
              fac *= shortrange_table[tabindex];
              acc_x += dx * fac;
              acc_y += dy * fac;
              acc_z += dz * fac;
              ninteractions++;
            }
        }
      else
        {
          acc_x += dx * fac;
          acc_y += dy * fac;
          acc_z += dz * fac;
          if(mass > 0)
            ninteractions++;
        }
    }
  acc[0] = acc_x;
  acc[1] = acc_y;
  acc[2] = acc_z;
  return ninteractions;
}

void orc_gravity(const orc_tree *t, const orc_grav_params *p, int nt, const int *targets,
                 const double *oldacc, double *acc, int *cost)
{
#pragma omp parallel for schedule(dynamic, 64)
  for(int a = 0; a < nt; a++)
    {
      int i = targets[a];
      cost[a] = treeevaluate(t, p, 0, &t->pos[3 * i], t->type[i], particle_soft(t, i), oldacc[i],
                             &acc[3 * a]);
    }
}

void orc_gravity_ext_soft(const orc_tree *t, const orc_grav_params *p, int nt, const double *tpos,
                          const int *ttype, const double *tsoft, const double *toldacc, double *acc,
                          int *cost)
{
#pragma omp parallel for schedule(dynamic, 64)
  for(int a = 0; a < nt; a++)
    {
      double h = (t->adaptive && ttype[a] == 0 && tsoft) ? tsoft[a] : t->soft[ttype[a]];
      cost[a] = treeevaluate(t, p, 0, &tpos[3 * a], ttype[a], h, toldacc[a], &acc[3 * a]);
    }
}

void orc_gravity_ext(const orc_tree *t, const orc_grav_params *p, int nt, const double *tpos,
                     const int *ttype, const double *toldacc, double *acc, int *cost)
{
  orc_gravity_ext_soft(t, p, nt, tpos, ttype, NULL, toldacc, acc, cost);
}

void orc_gravity_shortrange(const orc_tree *t, const orc_grav_params *p, int nt,
                            const int *targets, const double *oldacc, double *acc, int *cost)
{
  shortrange_init();
#pragma omp parallel for schedule(dynamic, 64)
  for(int a = 0; a < nt; a++)
    {
      int i = targets[a];
      cost[a] = treeevaluate(t, p, 1, &t->pos[3 * i], t->type[i], particle_soft(t, i), oldacc[i],
                             &acc[3 * a]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Ewald
 * ---------------------------------------------------------------------------------------- */

/* forcetree.c:4727-4778 ewald_force (alpha = 2, |n|,|h| <= 4) */
void orc_ewald_force(int iii, int jjj, int kkk, const double x[3], double force[3])
{
  const double alpha = 2.0;
  force[0] = force[1] = force[2] = 0;
  if(iii == 0 && jjj == 0 && kkk == 0)
    return;
  double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  for(int i = 0; i < 3; i++)
    force[i] += x[i] / (r2 * sqrt(r2));
  for(int n0 = -4; n0 <= 4; n0++)
    for(int n1 = -4; n1 <= 4; n1++)
      for(int n2 = -4; n2 <= 4; n2++)
        {
          double dx[3] = { x[0] - n0, x[1] - n1, x[2] - n2 };
          double r = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
          double val = erfc(alpha * r) + 2 * alpha * r / sqrt(M_PI) * exp(-alpha * alpha * r * r);
          for(int i = 0; i < 3; i++)
            force[i] -= dx[i] / (r * r * r) * val;
        }
  for(int h0 = -4; h0 <= 4; h0++)
    for(int h1 = -4; h1 <= 4; h1++)
      for(int h2_ = -4; h2_ <= 4; h2_++)
        {
          int hv[3] = { h0, h1, h2_ };
          double hdotx = x[0] * h0 + x[1] * h1 + x[2] * h2_;
          int h2 = h0 * h0 + h1 * h1 + h2_ * h2_;
          if(h2 > 0)
            {
              double val = 2.0 / ((double) h2) * exp(-M_PI * M_PI * h2 / (alpha * alpha)) *
                sin(2 * M_PI * hdotx);
              for(int i = 0; i < 3; i++)
                force[i] -= hv[i] * val;
            }
        }
}

/* forcetree.c:4402-4527 ewald_init, without the file cache */
void orc_ewald_init(double *tab, double boxsize)
{
  const int E1 = ORC_EN + 1;
  const long long ntab = (long long) E1 * E1 * E1;
#pragma omp parallel for schedule(dynamic, 256)
  for(long long n = 0; n < ntab; n++)
    {
      int i = (int) (n / ((long long) E1 * E1)), j = (int) ((n / E1) % E1), k = (int) (n % E1);
      double x[3] = { 0.5 * ((double) i) / ORC_EN, 0.5 * ((double) j) / ORC_EN,
        0.5 * ((double) k) / ORC_EN };
      double f[3];
      orc_ewald_force(i, j, k, x, f);
      tab[0 * ntab + n] = f[0] / (boxsize * boxsize);
      tab[1 * ntab + n] = f[1] / (boxsize * boxsize);
      tab[2 * ntab + n] = f[2] / (boxsize * boxsize);
    }
}

static inline void ewald_interp(const double *tab, double fac_intp, double dx, double dy, double dz,
                                double out[3])
{
  /* forcetree.c:3097-3170: sign convention and trilinear weights */
  const int E1 = ORC_EN + 1;
  const long long ntab = (long long) E1 * E1 * E1;
  int signx, signy, signz;
  if(dx < 0)
    {
      dx = -dx;
      signx = +1;
    }
  else
    signx = -1;
  if(dy < 0)
    {
      dy = -dy;
      signy = +1;
    }
  else
    signy = -1;
  if(dz < 0)
    {
      dz = -dz;
      signz = +1;
    }
  else
    signz = -1;
  double u = dx * fac_intp;
  int i = (int) u;
  if(i >= ORC_EN)
    i = ORC_EN - 1;
  u -= i;
  double v = dy * fac_intp;
  int j = (int) v;
  if(j >= ORC_EN)
    j = ORC_EN - 1;
  v -= j;
  double w = dz * fac_intp;
  int k = (int) w;
  if(k >= ORC_EN)
    k = ORC_EN - 1;
  w -= k;
  double f1 = (1 - u) * (1 - v) * (1 - w), f2 = (1 - u) * (1 - v) * (w);
  double f3 = (1 - u) * (v) * (1 - w), f4 = (1 - u) * (v) * (w);
  double f5 = (u) * (1 - v) * (1 - w), f6 = (u) * (1 - v) * (w);
  double f7 = (u) * (v) * (1 - w), f8 = (u) * (v) * (w);
#define TAB(c, a, b, d) tab[(c) * ntab + (((long long) (a)) * E1 + (b)) * E1 + (d)]
  int sg[3] = { signx, signy, signz };
  for(int c = 0; c < 3; c++)
    out[c] = sg[c] * (TAB(c, i, j, k) * f1 + TAB(c, i, j, k + 1) * f2 + TAB(c, i, j + 1, k) * f3 +
                      TAB(c, i, j + 1, k + 1) * f4 + TAB(c, i + 1, j, k) * f5 +
                      TAB(c, i + 1, j, k + 1) * f6 + TAB(c, i + 1, j + 1, k) * f7 +
                      TAB(c, i + 1, j + 1, k + 1) * f8);
#undef TAB
}

/* forcetree.c:2873-3204 */
static int treeevaluate_ewald(const orc_tree *t, const orc_grav_params *p, const double *tab,
                              const double tpos[3], double oldacc, double acc[3])
{
  double boxsize = p->BoxSize, boxhalf = 0.5 * p->BoxSize;
  double fac_intp = 2 * ORC_EN / boxsize;
  double pos_x = tpos[0], pos_y = tpos[1], pos_z = tpos[2];
  double aold = p->ErrTolForceAcc * oldacc;
  double acc_x = 0, acc_y = 0, acc_z = 0;
  int cost = 0;
  int no = t->n;
  while(no >= 0)
    {
      double dx, dy, dz, mass;
      const onode *nop = NULL;
      if(no < t->n)
        {
          dx = t->pos[3 * no + 0] - pos_x;
          dy = t->pos[3 * no + 1] - pos_y;
          dz = t->pos[3 * no + 2] - pos_z;
          mass = t->mass[no];
        }
      else
        {
          nop = &NODE(t, no);
          mass = nop->mass;
          dx = nop->s[0] - pos_x;
          dy = nop->s[1] - pos_y;
          dz = nop->s[2] - pos_z;
        }
      dx = nearest(dx, boxsize, boxhalf);
      dy = nearest(dy, boxsize, boxhalf);
      dz = nearest(dz, boxsize, boxhalf);

      if(no < t->n)
        no = t->nextnode[no];
      else
        {
          int openflag = 0;
          double r2 = dx * dx + dy * dy + dz * dz;
          if(p->ErrTolTheta)
            {
              if(nop->len * nop->len > r2 * p->ErrTolTheta * p->ErrTolTheta)
                openflag = 1;
            }
          else
            {
              if(mass * nop->len * nop->len > r2 * r2 * aold)
                openflag = 1;
              else if(fabs(nop->center[0] - pos_x) < 0.60 * nop->len &&
                      fabs(nop->center[1] - pos_y) < 0.60 * nop->len &&
                      fabs(nop->center[2] - pos_z) < 0.60 * nop->len)
                openflag = 1;
            }
          if(openflag)
            {
              /* forcetree.c:3039-3088: can we avoid opening? */
              int must_open = 0;
              const double tp[3] = { pos_x, pos_y, pos_z };
              for(int k = 0; k < 3 && !must_open; k++)
                {
                  double u = nop->center[k] - tp[k];
                  if(u > boxhalf)
                    u -= boxsize;
                  if(u < -boxhalf)
                    u += boxsize;
                  if(fabs(u) > 0.5 * (boxsize - nop->len))
                    must_open = 1;
                }
              if(!must_open && nop->len > 0.20 * boxsize)
                must_open = 1;
              if(must_open)
                {
                  no = nop->nextnode;
                  continue;
                }
            }
          no = nop->sibling;
        }
      double f[3];
      ewald_interp(tab, fac_intp, dx, dy, dz, f);
      acc_x += mass * f[0];
      acc_y += mass * f[1];
      acc_z += mass * f[2];
      cost++;
    }
  acc[0] = acc_x;
  acc[1] = acc_y;
  acc[2] = acc_z;
  return cost;
}

void orc_gravity_ewald(const orc_tree *t, const orc_grav_params *p, const double *tab, int nt,
                       const int *targets, const double *oldacc, double *acc, int *cost)
{
#pragma omp parallel for schedule(dynamic, 64)
  for(int a = 0; a < nt; a++)
    {
      int i = targets[a];
      double d[3];
      int c = treeevaluate_ewald(t, p, tab, &t->pos[3 * i], oldacc[i], d);
      acc[3 * a + 0] += d[0];
      acc[3 * a + 1] += d[1];
      acc[3 * a + 2] += d[2];
      cost[a] += c;
    }
}

/* independent check: softened direct summation (formula of forcetree.c:4273-4336) */
static void gravity_direct_impl(int n, const double *pos, const double *mass, const int *type,
                                const double soft[6], const double *psoft, int unequal, int periodic,
                                double boxsize, const double *ewald_tab, int nt, const int *targets,
                                double *acc)
{
  double boxhalf = 0.5 * boxsize;
  double fac_intp = periodic ? 2 * ORC_EN / boxsize : 0;
#pragma omp parallel for schedule(dynamic, 16)
  for(int a = 0; a < nt; a++)
    {
      int i = targets[a];
      double ax = 0, ay = 0, az = 0;
      for(int j = 0; j < n; j++)
        {
          double dx = pos[3 * j + 0] - pos[3 * i + 0];
          double dy = pos[3 * j + 1] - pos[3 * i + 1];
          double dz = pos[3 * j + 2] - pos[3 * i + 2];
          if(periodic)
            {
              dx = nearest(dx, boxsize, boxhalf);
              dy = nearest(dy, boxsize, boxhalf);
              dz = nearest(dz, boxsize, boxhalf);
            }
          double r2 = dx * dx + dy * dy + dz * dz;
          double h = psoft ? psoft[i] : soft[type[i]];
          double hj = psoft ? psoft[j] : soft[type[j]];
          if(unequal && h < hj)
            h = hj;
          double r = sqrt(r2);
          double fac = grav_fac(mass[j], r2, r, h);
          ax += dx * fac;
          ay += dy * fac;
          az += dz * fac;
          if(periodic && ewald_tab)
            {
              double f[3];
              ewald_interp(ewald_tab, fac_intp, dx, dy, dz, f);
              ax += mass[j] * f[0];
              ay += mass[j] * f[1];
              az += mass[j] * f[2];
            }
        }
      acc[3 * a + 0] = ax;
      acc[3 * a + 1] = ay;
      acc[3 * a + 2] = az;
    }
}

void orc_gravity_direct(int n, const double *pos, const double *mass, const int *type,
                        const double soft[6], int unequal, int periodic, double boxsize,
                        const double *ewald_tab, int nt, const int *targets, double *acc)
{
  gravity_direct_impl(n, pos, mass, type, soft, NULL, unequal, periodic, boxsize, ewald_tab, nt,
                      targets, acc);
}

/* the same with one softening per particle (ADAPTIVE_GRAVSOFT_FORGAS: Hsml for gas,
 * ForceSoftening[type] otherwise), pairs softened with the larger of the two */
void orc_gravity_direct_psoft(int n, const double *pos, const double *mass, const double *psoft,
                              int periodic, double boxsize, const double *ewald_tab, int nt,
                              const int *targets, double *acc)
{
  gravity_direct_impl(n, pos, mass, NULL, NULL, psoft, 1, periodic, boxsize, ewald_tab, nt, targets,
                      acc);
}

/* ------------------------------------------------------------------------------------------
 * neighbour search
 * ---------------------------------------------------------------------------------------- */

#define FACT1 0.366025403785 /* allvars.h:310 */

static inline double ngb_periodic(double x, int periodic, double boxsize, double boxhalf)
{
  /* allvars.h:300-308 NGB_PERIODIC_LONG_* */
  double xtmp = fabs(x);
  if(periodic && xtmp > boxhalf)
    return boxsize - xtmp;
  return xtmp;
}

/* ngb.c:169-297 (pairs == 0) and ngb.c:32-160 (pairs == 1), mode 0 */
static int ngb_treefind(const orc_tree *t, const double c[3], double hsml, const double *hs,
                        int pairs, int periodic, double boxsize, int *ngblist)
{
  double boxhalf = 0.5 * boxsize;
  int numngb = 0;
  int no = t->n;
  while(no >= 0)
    {
      if(no < t->n)
        {
          int p = no;
          no = t->nextnode[no];
          if(t->type[p] > 0)
            continue;
          double dist = hsml;
          if(pairs && hs[p] > dist)
            dist = hs[p];
          double dx = ngb_periodic(t->pos[3 * p + 0] - c[0], periodic, boxsize, boxhalf);
          if(dx > dist)
            continue;
          double dy = ngb_periodic(t->pos[3 * p + 1] - c[1], periodic, boxsize, boxhalf);
          if(dy > dist)
            continue;
          double dz = ngb_periodic(t->pos[3 * p + 2] - c[2], periodic, boxsize, boxhalf);
          if(dz > dist)
            continue;
          if(dx * dx + dy * dy + dz * dz > dist * dist)
            continue;
          ngblist[numngb++] = p;
        }
      else
        {
          const onode *cur = &NODE(t, no);
          double dist = hsml;
          if(pairs && cur->hmax > dist)
            dist = cur->hmax;
          dist += 0.5 * cur->len;
          no = cur->sibling;
          double dx = ngb_periodic(cur->center[0] - c[0], periodic, boxsize, boxhalf);
          if(dx > dist)
            continue;
          double dy = ngb_periodic(cur->center[1] - c[1], periodic, boxsize, boxhalf);
          if(dy > dist)
            continue;
          double dz = ngb_periodic(cur->center[2] - c[2], periodic, boxsize, boxhalf);
          if(dz > dist)
            continue;
          dist += FACT1 * cur->len;
          if(dx * dx + dy * dy + dz * dz > dist * dist)
            continue;
          no = cur->nextnode;
        }
    }
  return numngb;
}

int orc_ngb_treefind_variable(const orc_tree *t, const double c[3], double h, int periodic,
                              double boxsize, int *ngblist)
{
  return ngb_treefind(t, c, h, NULL, 0, periodic, boxsize, ngblist);
}

int orc_ngb_treefind_pairs(const orc_tree *t, const double c[3], double h, const double *hsml,
                           int periodic, double boxsize, int *ngblist)
{
  return ngb_treefind(t, c, h, hsml, 1, periodic, boxsize, ngblist);
}

/* ------------------------------------------------------------------------------------------
 * SPH density
 * ---------------------------------------------------------------------------------------- */

/* allvars.h:247-253 */
#define KERNEL_COEFF_1 2.546479089470
#define KERNEL_COEFF_2 15.278874536822
#define KERNEL_COEFF_3 45.836623610466
#define KERNEL_COEFF_4 30.557749073644
#define KERNEL_COEFF_5 5.092958178941
#define KERNEL_COEFF_6 (-15.278874536822)
#define NORM_COEFF 4.188790204786
#define NUMDIMS 3
#define GAMMA (7. / 5.) /* allvars.h:64: this fork uses 7/5 */
#define GAMMA_MINUS1 (GAMMA - 1)

/* density.c:711-1029 density_evaluate, mode 0, plain-SPH members only */
/* the neighbour loops skip gas of mass 0: bit 0 density() under -DBLACK_HOLES || -DDUST (density.c:831-834),
 * bit 1 hydro_force() under -DBLACK_HOLES (hydra.c:1235-1238) */
static int SkipMassless = 0;
void orc_set_massless_gas_rule(int on)
{
  SkipMassless = on;
}

static int density_eval(const orc_tree *t, const orc_dens_params *p, const double pos[3],
                        const double vel[3], double h, int *ngblist, double out7[7])
{
  double boxsize = p->BoxSize, boxhalf = 0.5 * p->BoxSize;
  double h2 = h * h, hinv = 1.0 / h;
  double hinv3 = hinv * hinv * hinv, hinv4 = hinv3 * hinv;
  double rho = 0, weighted_numngb = 0, dhsmlrho = 0, divv = 0, rotv[3] = { 0, 0, 0 };
  int numngb = 0;

  int numngb_inbox = ngb_treefind(t, pos, h, NULL, 0, p->periodic, boxsize, ngblist);
  for(int n = 0; n < numngb_inbox; n++)
    {
      int j = ngblist[n];
      if((SkipMassless & 1) && t->mass[j] == 0)
        continue;
      double dx = pos[0] - t->pos[3 * j + 0];
      double dy = pos[1] - t->pos[3 * j + 1];
      double dz = pos[2] - t->pos[3 * j + 2];
      if(p->periodic)
        {
          if(dx > boxhalf)
            dx -= boxsize;
          if(dx < -boxhalf)
            dx += boxsize;
          if(dy > boxhalf)
            dy -= boxsize;
          if(dy < -boxhalf)
            dy += boxsize;
          if(dz > boxhalf)
            dz -= boxsize;
          if(dz < -boxhalf)
            dz += boxsize;
        }
      double r2 = dx * dx + dy * dy + dz * dz;
      if(r2 < h2)
        {
          numngb++;
          double r = sqrt(r2);
          double u = r * hinv, wk, dwk;
          if(u < 0.5)
            {
              wk = hinv3 * (KERNEL_COEFF_1 + KERNEL_COEFF_2 * (u - 1) * u * u);
              dwk = hinv4 * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4);
            }
          else
            {
              wk = hinv3 * KERNEL_COEFF_5 * (1.0 - u) * (1.0 - u) * (1.0 - u);
              dwk = hinv4 * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
            }
          double mass_j = t->mass[j];
          rho += mass_j * wk;
          weighted_numngb += NORM_COEFF * wk / hinv3;
          dhsmlrho += -mass_j * (NUMDIMS * hinv * wk + u * dwk);
          if(r > 0)
            {
              double fac = mass_j * dwk / r;
              /* velpred of gas neighbour j: vel argument array is the caller's; neighbours use
               * the tree's stored predicted velocities (t->vel is VelPred for gas in tests) */
              double dvx = vel[0] - t->vel[3 * j + 0];
              double dvy = vel[1] - t->vel[3 * j + 1];
              double dvz = vel[2] - t->vel[3 * j + 2];
              divv += -fac * (dx * dvx + dy * dvy + dz * dvz);
              rotv[0] += fac * (dz * dvy - dy * dvz);
              rotv[1] += fac * (dx * dvz - dz * dvx);
              rotv[2] += fac * (dy * dvx - dx * dvy);
            }
        }
    }
  out7[0] = rho;
  out7[1] = weighted_numngb;
  out7[2] = dhsmlrho;
  out7[3] = divv;
  out7[4] = rotv[0];
  out7[5] = rotv[1];
  out7[6] = rotv[2];
  return numngb;
}

/* NOTE on velocities: the reference reads SphP[j].VelPred for neighbours (density.c:904-906)
 * and SphP[target].VelPred for the target.  The oracle takes one `velpred` array [ngas][3]
 * for both; the tree's own vel[] (P[].Vel) is only used for node vs/vmax. */
void orc_density_evaluate(const orc_tree *t, const orc_dens_params *p, int target, double h,
                          const double *velpred, double out7[7])
{
  orc_tree tmp = *t;
  tmp.vel = velpred; /* neighbours' VelPred (gas indices are [0,ngas)) */
  int *ngblist = (int *) malloc((size_t) t->n * sizeof(int));
  double zero[3] = { 0, 0, 0 };
  const double *v = (t->type[target] == 0) ? &velpred[3 * target] : zero;
  density_eval(&tmp, p, &t->pos[3 * target], v, h, ngblist, out7);
  free(ngblist);
}

/* density.c:89-704 density(): h iteration + finalisation */
int orc_density(const orc_tree *t, const orc_dens_params *p, int nactive, const int *active,
                const double *velpred, const double *entropy, const double *dtentropy_in,
                const int *timebin, const int *ti_begstep, double *hsml, double *numngb,
                double *density, double *dhsmlfac, double *divvel, double *curlvel,
                double *pressure, long long *nngb_visits)
{
  orc_tree tmp = *t;
  tmp.vel = velpred;
  int n = t->n;
  double *Left = (double *) calloc((size_t) n, sizeof(double));
  double *Right = (double *) calloc((size_t) n, sizeof(double));
  char *done = (char *) calloc((size_t) n, 1); /* TimeBin negation trick, density.c:571,650 */
  double *rot = (double *) calloc((size_t) n * 3, sizeof(double));
  int maxiter = p->maxiter > 0 ? p->maxiter : 150;
  int iter = 0;
  long long npleft;
  long long visits = 0;
  double desnumngb = p->DesNumNgb;

  do
    {
#pragma omp parallel
      {
        int *ngblist = (int *) malloc((size_t) n * sizeof(int));
        long long myvisits = 0;
#pragma omp for schedule(dynamic, 64)
        for(int a = 0; a < nactive; a++)
          {
            int i = active[a];
            if(done[i] || t->type[i] != 0)
              continue;
            double out7[7];
            myvisits += density_eval(&tmp, p, &t->pos[3 * i], &velpred[3 * i], hsml[i], ngblist,
                                     out7);
            density[i] = out7[0];
            numngb[i] = out7[1];
            dhsmlfac[i] = out7[2];
            divvel[i] = out7[3];
            rot[3 * i + 0] = out7[4];
            rot[3 * i + 1] = out7[5];
            rot[3 * i + 2] = out7[6];
          }
#pragma omp atomic
        visits += myvisits;
        free(ngblist);
      }

      npleft = 0;
      for(int a = 0; a < nactive; a++)
        {
          int i = active[a];
          if(done[i] || t->type[i] != 0)
            continue;
          /* density.c:436-452 */
          if(density[i] > 0)
            {
              dhsmlfac[i] *= hsml[i] / (NUMDIMS * density[i]);
              if(dhsmlfac[i] > -0.9)
                dhsmlfac[i] = 1 / (1 + dhsmlfac[i]);
              else
                dhsmlfac[i] = 1;
              curlvel[i] = sqrt(rot[3 * i] * rot[3 * i] + rot[3 * i + 1] * rot[3 * i + 1] +
                                rot[3 * i + 2] * rot[3 * i + 2]) / density[i];
              divvel[i] /= density[i];
            }
          /* density.c:487-496 */
          int dt_step = (timebin[i] ? (1 << timebin[i]) : 0);
          double dt_entr = (p->Ti_Current - (ti_begstep[i] + dt_step / 2)) * p->Timebase_interval;
          pressure[i] = (entropy[i] + dtentropy_in[i] * dt_entr) * pow(density[i], GAMMA);

          /* density.c:559-652 */
          if(numngb[i] < (desnumngb - p->MaxNumNgbDeviation) ||
             (numngb[i] > (desnumngb + p->MaxNumNgbDeviation) && hsml[i] > (1.01 * p->MinGasHsml)))
            {
              npleft++;
              if(Left[i] > 0 && Right[i] > 0)
                if((Right[i] - Left[i]) < 1.0e-3 * Left[i])
                  {
                    npleft--;
                    done[i] = 1;
                    continue;
                  }
              if(numngb[i] < (desnumngb - p->MaxNumNgbDeviation))
                Left[i] = (hsml[i] > Left[i]) ? hsml[i] : Left[i];
              else
                {
                  if(Right[i] != 0)
                    {
                      if(hsml[i] < Right[i])
                        Right[i] = hsml[i];
                    }
                  else
                    Right[i] = hsml[i];
                }
              if(Right[i] > 0 && Left[i] > 0)
                hsml[i] = pow(0.5 * (pow(Left[i], 3) + pow(Right[i], 3)), 1.0 / 3);
              else
                {
                  if(Right[i] == 0 && Left[i] > 0)
                    {
                      if(fabs(numngb[i] - desnumngb) < 0.5 * desnumngb)
                        {
                          double fac = 1 - (numngb[i] - desnumngb) / (NUMDIMS * numngb[i]) *
                            dhsmlfac[i];
                          if(fac < 1.26)
                            hsml[i] *= fac;
                          else
                            hsml[i] *= 1.26;
                        }
                      else
                        hsml[i] *= 1.26;
                    }
                  if(Right[i] > 0 && Left[i] == 0)
                    {
                      if(fabs(numngb[i] - desnumngb) < 0.5 * desnumngb)
                        {
                          double fac = 1 - (numngb[i] - desnumngb) / (NUMDIMS * numngb[i]) *
                            dhsmlfac[i];
                          if(fac > 1 / 1.26)
                            hsml[i] *= fac;
                          else
                            hsml[i] /= 1.26;
                        }
                      else
                        hsml[i] /= 1.26;
                    }
                }
              if(hsml[i] < p->MinGasHsml)
                hsml[i] = p->MinGasHsml;
            }
          else
            done[i] = 1;
        }
      if(npleft > 0)
        {
          iter++;
          if(iter > maxiter)
            {
              iter = -1;
              break;
            }
        }
    }
  while(npleft > 0);

  free(Left);
  free(Right);
  free(done);
  free(rot);
  if(nngb_visits)
    *nngb_visits = visits;
  return iter;
}

/* ------------------------------------------------------------------------------------------
 * SPH hydro
 * ---------------------------------------------------------------------------------------- */

/* hydra.c:822-1995 hydro_evaluate (live lines with optional physics off), mode 0, and the
 * entropy-rate conversion of hydro_force (hydra.c:583) */
void orc_hydro(const orc_tree *t, const orc_hydro_params *p, int nactive, const int *active,
               const double *velpred, const double *hsml, const double *density,
               const double *pressure, const double *dhsmlfac, const double *divvel,
               const double *curlvel, const int *timebin, double *hydroaccel, double *dtentropy,
               double *maxsignalvel, long long *npairs)
{
  double boxsize = p->BoxSize, boxhalf = 0.5 * p->BoxSize;
  double hubble_a2 = p->hubble_a2, fac_mu = p->fac_mu, fac_vsic_fix = p->fac_vsic_fix;
  long long pairs = 0;
  int n = t->n;
#pragma omp parallel
  {
    int *ngblist = (int *) malloc((size_t) n * sizeof(int));
    long long mypairs = 0;
#pragma omp for schedule(dynamic, 64)
    for(int a = 0; a < nactive; a++)
      {
        int i = active[a];
        if(t->type[i] != 0)
          continue;
        const double *pos = &t->pos[3 * i];
        const double *vel = &velpred[3 * i];
        double h_i = hsml[i], mass = t->mass[i], rho = density[i], press = pressure[i];
        int timestep = (timebin[i] ? (1 << timebin[i]) : 0);
        double soundspeed_i = sqrt(GAMMA * press / rho);
        double f1 = fabs(divvel[i]) / (fabs(divvel[i]) + curlvel[i] +
                                       0.0001 * soundspeed_i / hsml[i] / fac_mu);
        double p_over_rho2_i = press / (rho * rho);
        p_over_rho2_i *= dhsmlfac[i];
        double h_i2 = h_i * h_i;
        double acc[3] = { 0, 0, 0 }, dtEntropy = 0, maxSignalVel = 0;

        int numngb = ngb_treefind(t, pos, h_i, hsml, 1, p->periodic, boxsize, ngblist);
        for(int nn = 0; nn < numngb; nn++)
          {
            int j = ngblist[nn];
            if((SkipMassless & 2) && t->mass[j] == 0) /* hydra.c:1235-1238: -DBLACK_HOLES only */
              continue;
            double dx = pos[0] - t->pos[3 * j + 0];
            double dy = pos[1] - t->pos[3 * j + 1];
            double dz = pos[2] - t->pos[3 * j + 2];
            if(p->periodic)
              {
                if(dx > boxhalf)
                  dx -= boxsize;
                if(dx < -boxhalf)
                  dx += boxsize;
                if(dy > boxhalf)
                  dy -= boxsize;
                if(dy < -boxhalf)
                  dy += boxsize;
                if(dz > boxhalf)
                  dz -= boxsize;
                if(dz < -boxhalf)
                  dz += boxsize;
              }
            double r2 = dx * dx + dy * dy + dz * dz;
            double h_j = hsml[j];
            if(r2 < h_i2 || r2 < h_j * h_j)
              {
                double r = sqrt(r2);
                if(r > 0)
                  {
                    mypairs++;
                    double p_over_rho2_j = pressure[j] / (density[j] * density[j]);
                    double soundspeed_j = sqrt(GAMMA * p_over_rho2_j * density[j]);
                    double dvx = vel[0] - velpred[3 * j + 0];
                    double dvy = vel[1] - velpred[3 * j + 1];
                    double dvz = vel[2] - velpred[3 * j + 2];
                    double vdotr = dx * dvx + dy * dvy + dz * dvz;
                    double vdotr2 = p->ComovingIntegrationOn ? vdotr + hubble_a2 * r2 : vdotr;
                    double dwk_i, dwk_j;
                    if(r2 < h_i2)
                      {
                        double hinv = 1.0 / h_i;
                        double hinv4 = hinv * hinv * hinv * hinv;
                        double u = r * hinv;
                        if(u < 0.5)
                          dwk_i = hinv4 * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4);
                        else
                          dwk_i = hinv4 * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
                      }
                    else
                      dwk_i = 0;
                    if(r2 < h_j * h_j)
                      {
                        double hinv = 1.0 / h_j;
                        double hinv4 = hinv * hinv * hinv * hinv;
                        double u = r * hinv;
                        if(u < 0.5)
                          dwk_j = hinv4 * u * (KERNEL_COEFF_3 * u - KERNEL_COEFF_4);
                        else
                          dwk_j = hinv4 * KERNEL_COEFF_6 * (1.0 - u) * (1.0 - u);
                      }
                    else
                      dwk_j = 0;

                    double vsig = soundspeed_i + soundspeed_j;
                    if(vsig > maxSignalVel)
                      maxSignalVel = vsig;
                    double visc;
                    if(vdotr2 < 0)
                      {
                        double mu_ij = fac_mu * vdotr2 / r;
                        vsig -= 3 * mu_ij;
                        if(vsig > maxSignalVel)
                          maxSignalVel = vsig;
                        double rho_ij = 0.5 * (rho + density[j]);
                        double f2 = fabs(divvel[j]) / (fabs(divvel[j]) + curlvel[j] +
                                                       0.0001 * soundspeed_j / fac_mu / hsml[j]);
                        visc = 0.25 * p->ArtBulkViscConst * vsig * (-mu_ij) / rho_ij * (f1 + f2);
                        /* hydra.c:1584-1594 viscosity limiter */
                        int tj = (timebin[j] ? (1 << timebin[j]) : 0);
                        double dt = 2 * (timestep > tj ? timestep : tj) * p->Timebase_interval;
                        if(dt > 0 && (dwk_i + dwk_j) < 0)
                          {
                            double lim = 0.5 * fac_vsic_fix * vdotr2 /
                              (0.5 * (mass + t->mass[j]) * (dwk_i + dwk_j) * r * dt);
                            if(lim < visc)
                              visc = lim;
                          }
                      }
                    else
                      visc = 0;
                    p_over_rho2_j *= dhsmlfac[j];
                    double hfc_visc = 0.5 * t->mass[j] * visc * (dwk_i + dwk_j) / r;
                    double hfc = hfc_visc +
                      t->mass[j] * (p_over_rho2_i * dwk_i + p_over_rho2_j * dwk_j) / r;
                    acc[0] += -hfc * dx;
                    acc[1] += -hfc * dy;
                    acc[2] += -hfc * dz;
                    dtEntropy += 0.5 * hfc_visc * vdotr2;
                  }
              }
          }
        hydroaccel[3 * i + 0] = acc[0];
        hydroaccel[3 * i + 1] = acc[1];
        hydroaccel[3 * i + 2] = acc[2];
        /* hydra.c:583 */
        dtentropy[i] = dtEntropy * (GAMMA_MINUS1 / (hubble_a2 * pow(density[i], GAMMA_MINUS1)));
        maxsignalvel[i] = maxSignalVel;
      }
#pragma omp atomic
    pairs += mypairs;
    free(ngblist);
  }
  if(npairs)
    *npairs = pairs;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int nthreads)
{
#ifdef _OPENMP
  omp_set_num_threads(nthreads);
#else
  (void) nthreads;
#endif
}

/* ------------------------------------------------------------------------------------------
 * drift (pre-condition of the path)
 * ---------------------------------------------------------------------------------------- */

/* driftfac.c:123-163 get_drift_factor (same form for the two kick tables, :166-247) */
static double table_factor(const double *tab, int time0, int time1, double timebase,
                           double logTimeBegin, double logTimeMax)
{
  const int NT = 1000; /* DRIFT_TABLE_LENGTH, allvars.h:136 */
  double a1 = logTimeBegin + time0 * timebase;
  double a2 = logTimeBegin + time1 * timebase;
  double u1 = (a1 - logTimeBegin) / (logTimeMax - logTimeBegin) * NT;
  int i1 = (int) u1;
  if(i1 >= NT)
    i1 = NT - 1;
  double df1 = (i1 <= 1) ? u1 * tab[0] : tab[i1 - 1] + (tab[i1] - tab[i1 - 1]) * (u1 - i1);
  double u2 = (a2 - logTimeBegin) / (logTimeMax - logTimeBegin) * NT;
  int i2 = (int) u2;
  if(i2 >= NT)
    i2 = NT - 1;
  double df2 = (i2 <= 1) ? u2 * tab[0] : tab[i2 - 1] + (tab[i2] - tab[i2 - 1]) * (u2 - i2);
  return df2 - df1;
}

/* predict.c:129-259 drift_particle for all particles, then predict.c:282-310 do_box_wrapping.
 * tables = NULL: non-comoving.  Arrays: pos/vel/gravaccel [n][3]; gas arrays [ngas](,3). */
int orc_drift(int n, int ngas, int time1, double timebase, const double *tables,
              double logTimeBegin, double logTimeMax, double minhsml, int wrap, double boxsize,
              double *pos, const double *vel, const int *type, int *ti_current, const int *timebin,
              const int *ti_begstep, const double *gravaccel, double *velpred,
              const double *hydroaccel, double *density, double *hsml, const double *divvel,
              const double *entropy, const double *dtentropy, double *pressure)
{
  return orc_drift_pm(n, ngas, time1, timebase, tables, logTimeBegin, logTimeMax, minhsml, wrap,
                      boxsize, pos, vel, type, ti_current, timebin, ti_begstep, gravaccel, NULL,
                      velpred, hydroaccel, density, hsml, divvel, entropy, dtentropy, pressure);
}

/* gravpm != NULL: PMGRID, VelPred += (GravAccel + GravPM) * dt_gravkick (predict.c:181-184) */
int orc_drift_pm(int n, int ngas, int time1, double timebase, const double *tables,
                 double logTimeBegin, double logTimeMax, double minhsml, int wrap, double boxsize,
                 double *pos, const double *vel, const int *type, int *ti_current,
                 const int *timebin, const int *ti_begstep, const double *gravaccel,
                 const double *gravpm, double *velpred, const double *hydroaccel, double *density,
                 double *hsml, const double *divvel, const double *entropy, const double *dtentropy,
                 double *pressure)
{
  for(int i = 0; i < n; i++)
    {
      int time0 = ti_current[i];
      if(time1 < time0)
        return 12;
      if(time1 != time0)
        {
          double dt_drift, dt_gravkick, dt_hydrokick;
          if(tables)
            {
              dt_drift = table_factor(tables, time0, time1, timebase, logTimeBegin, logTimeMax);
              dt_gravkick = table_factor(tables + 1000, time0, time1, timebase, logTimeBegin,
                                         logTimeMax);
              dt_hydrokick = table_factor(tables + 2000, time0, time1, timebase, logTimeBegin,
                                          logTimeMax);
            }
          else
            dt_drift = dt_gravkick = dt_hydrokick = (time1 - time0) * timebase;
          for(int j = 0; j < 3; j++)
            pos[3 * i + j] += vel[3 * i + j] * dt_drift;
          if(i < ngas && type[i] == 0)
            {
              for(int j = 0; j < 3; j++)
                {
                  double g = gravaccel[3 * i + j];
                  if(gravpm)
                    g = g + gravpm[3 * i + j];
                  velpred[3 * i + j] += g * dt_gravkick + hydroaccel[3 * i + j] * dt_hydrokick;
                }
              density[i] *= exp(-divvel[i] * dt_drift);
              hsml[i] *= exp(0.333333333333 * divvel[i] * dt_drift);
              if(hsml[i] < minhsml)
                hsml[i] = minhsml;
              int dt_step = (timebin[i] ? (1 << timebin[i]) : 0);
              double dt_entr = (time1 - (ti_begstep[i] + dt_step / 2)) * timebase;
              pressure[i] = (entropy[i] + dtentropy[i] * dt_entr) * pow(density[i], GAMMA);
            }
          ti_current[i] = time1;
        }
      if(wrap)
        for(int j = 0; j < 3; j++)
          {
            while(pos[3 * i + j] < 0)
              pos[3 * i + j] += boxsize;
            while(pos[3 * i + j] >= boxsize)
              pos[3 * i + j] -= boxsize;
          }
    }
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * "Next" row N1: timestep criterion + kick (timestep.c).  Live code of the minimal periodic flag
 * set: no PMGRID, BLACK_HOLES, DUST, MAGNETIC, CONDUCTION, ... ; DoDynamicUpdate (node kicks,
 * forcetree.c:1474-1651) is not part of it -- the device rebuilds its tree every step.
 * ------------------------------------------------------------------------------------------- */
#define ORC_TIMEBINS 29              /* allvars.h:39 */
#define ORC_TIMEBASE (1 << ORC_TIMEBINS) /* allvars.h:41 */
#define ORC_GAMMA (7.0 / 5.0)        /* allvars.h:64 (this fork) */
#define ORC_GAMMA_MINUS1 (ORC_GAMMA - 1)

/* find_dt_displacement_constraint, timestep.c:1125-1224: the per-type sums (the reference adds
 * them up serially per task, then MPI_Allreduce) */
void orc_velocity_moments(int n, const double *vel, const double *mass, const int *type,
                          double v2[6], double minmass[6], long long count[6])
{
  for(int t = 0; t < 6; t++)
    {
      v2[t] = 0;
      minmass[t] = 1.0e30;
      count[t] = 0;
    }
  for(int i = 0; i < n; i++)
    {
      int t = type[i];
      v2[t] += vel[3 * i] * vel[3 * i] + vel[3 * i + 1] * vel[3 * i + 1] + vel[3 * i + 2] * vel[3 * i + 2];
      if(mass[i] > 0 && minmass[t] > mass[i])
        minmass[t] = mass[i];
      count[t]++;
    }
}

/* the rest of find_dt_displacement_constraint: dt_displacement from the moments */
double orc_dt_displacement(const double v2[6], const double minmass[6], const long long count[6],
                           int comoving, double hfac, double MaxSizeTimestep,
                           double MaxRMSDisplacementFac, double Omega0, double OmegaBaryon,
                           double Hubble, double G, int StarformationOn)
{
  double dtd = MaxSizeTimestep;
  if(!comoving)
    return dtd;
  for(int t = 0; t < 6; t++)
    if(count[t] > 0)
      {
        double dmean;
        if(t == 0 || (t == 4 && StarformationOn))
          dmean = pow(minmass[t] / (OmegaBaryon * 3 * Hubble * Hubble / (8 * M_PI * G)), 1.0 / 3);
        else
          dmean = pow(minmass[t] / ((Omega0 - OmegaBaryon) * 3 * Hubble * Hubble / (8 * M_PI * G)),
                      1.0 / 3);
        double dt = MaxRMSDisplacementFac * hfac * dmean / sqrt(v2[t] / count[t]);
        if(dt < dtd)
          dtd = dt;
      }
  return dtd;
}

/* get_timestep_bin, timestep.c:1226-1246; -1 flags the endrun(112313) case */
static int timestep_bin(int ti_step)
{
  int bin = -1;
  if(ti_step == 0)
    return 0;
  if(ti_step == 1)
    return -1;
  while(ti_step)
    {
      bin++;
      ti_step >>= 1;
    }
  return bin;
}

/* advance_and_find_timesteps (timestep.c:29-362) with get_timestep (:607-1123, flag == 0,
 * TypeOfTimestepCriterion 0) and do_the_kick (:364-605) for the active particles.  Arrays:
 * vel/gravaccel [n][3]; gas arrays [ngas](,3).  bincount/bincount_sph: TimeBinCount[] /
 * TimeBinCountSph[] after the update, recounted over all particles.
 * Returns 0 or the reference's endrun code (888, 818, 112313). */
int orc_advance_timesteps(int n, int ngas, const orc_kick_params *p, int nactive, const int *active,
                          const int *type, double *vel, const double *gravaccel,
                          const double *hydroaccel, double *velpred, double *entropy,
                          double *dtentropy, const double *density, const double *pressure,
                          const double *hsml, const double *maxsignalvel, int *timebin,
                          int *ti_begstep, long long bincount[32], long long bincount_sph[32])
{
  double fac1, fac2, fac3, hubble_a, atime, a3inv;
  int err = 0;
  if(p->ComovingIntegrationOn)
    {
      /* timestep.c:52-60 */
      fac1 = 1 / (p->Time * p->Time);
      fac2 = 1 / pow(p->Time, 3 * ORC_GAMMA - 2);
      fac3 = pow(p->Time, 3 * (1 - ORC_GAMMA) / 2.0);
      hubble_a = p->hubble_a;
      a3inv = 1 / (p->Time * p->Time * p->Time);
      atime = p->Time;
    }
  else
    fac1 = fac2 = fac3 = hubble_a = a3inv = atime = 1;

  for(int a = 0; a < (active ? nactive : n); a++)
    {
      int i = active ? active[a] : a;
      /* ---- get_timestep ---- */
      double ax = fac1 * gravaccel[3 * i], ay = fac1 * gravaccel[3 * i + 1],
             az = fac1 * gravaccel[3 * i + 2];
      if(p->pmgrid) /* timestep.c:648-652 */
        {
          ax += fac1 * p->gravpm[3 * i];
          ay += fac1 * p->gravpm[3 * i + 1];
          az += fac1 * p->gravpm[3 * i + 2];
        }
      if(type[i] == 0)
        {
          ax += fac2 * hydroaccel[3 * i];
          ay += fac2 * hydroaccel[3 * i + 1];
          az += fac2 * hydroaccel[3 * i + 2];
        }
      double ac = sqrt(ax * ax + ay * ay + az * az);
      if(ac == 0)
        ac = 1.0e-30;
      double dt = sqrt(2 * p->ErrTolIntAccuracy * atime * p->SofteningTable[type[i]] / ac);
      if(p->AdaptiveGravsoftForGasHsml && type[i] == 0) /* timestep.c:740-743 */
        dt = sqrt(2 * p->ErrTolIntAccuracy * atime * hsml[i] / 2.8 / ac);
      if(type[i] == 0)
        {
          double dt_courant;
          if(p->ComovingIntegrationOn)
            dt_courant = 2 * p->CourantFac * p->Time * hsml[i] / (fac3 * maxsignalvel[i]);
          else
            dt_courant = 2 * p->CourantFac * hsml[i] / maxsignalvel[i];
          if(dt_courant < dt)
            dt = dt_courant;
        }
      dt *= hubble_a;
      if(dt >= p->MaxSizeTimestep)
        dt = p->MaxSizeTimestep;
      if(dt >= p->dt_displacement)
        dt = p->dt_displacement;
      if(dt < p->MinSizeTimestep)
        {
          err = err > 888 ? err : 888; /* timestep.c:1082 */
          continue;
        }
      int ti_step = (int) (dt / p->Timebase_interval);
      if(!(ti_step > 0 && ti_step < ORC_TIMEBASE))
        {
          err = err > 818 ? err : 818; /* timestep.c:1119 */
          continue;
        }
      /* ---- advance_and_find_timesteps loop body, timestep.c:146-260 ---- */
      int ti_min = ORC_TIMEBASE;
      while(ti_min > ti_step)
        ti_min >>= 1;
      ti_step = ti_min;
      int bin = timestep_bin(ti_step);
      if(bin < 0)
        {
          err = err > 112313 ? err : 112313;
          continue;
        }
      int binold = timebin[i];
      if(bin > binold)
        if(((p->TimeBinActive >> bin) & 1u) == 0)
          {
            bin = binold;
            ti_step = bin ? (1 << bin) : 0;
          }
      if(p->Ti_Current >= ORC_TIMEBASE)
        {
          ti_step = 0;
          bin = 0;
        }
      if((ORC_TIMEBASE - p->Ti_Current) < ti_step)
        {
          err = err > 888 ? err : 888; /* timestep.c:171 */
          continue;
        }
      timebin[i] = bin;
      int ti_step_old = binold ? (1 << binold) : 0;
      int tstart = ti_begstep[i] + ti_step_old / 2;
      int tend = ti_begstep[i] + ti_step_old + ti_step / 2;
      ti_begstep[i] += ti_step_old;
      int tcurrent = ti_begstep[i];
      /* ---- do_the_kick ---- */
      double dt_entr, dt_gravkick, dt_hydrokick, dt_gravkick2, dt_hydrokick2;
      if(p->ComovingIntegrationOn)
        {
          dt_entr = (tend - tstart) * p->Timebase_interval;
          dt_gravkick = table_factor(p->tables + 1000, tstart, tend, p->Timebase_interval,
                                     p->logTimeBegin, p->logTimeMax);
          dt_hydrokick = table_factor(p->tables + 2000, tstart, tend, p->Timebase_interval,
                                      p->logTimeBegin, p->logTimeMax);
          dt_gravkick2 = table_factor(p->tables + 1000, tcurrent, tend, p->Timebase_interval,
                                      p->logTimeBegin, p->logTimeMax);
          dt_hydrokick2 = table_factor(p->tables + 2000, tcurrent, tend, p->Timebase_interval,
                                       p->logTimeBegin, p->logTimeMax);
        }
      else
        {
          dt_entr = dt_gravkick = dt_hydrokick = (tend - tstart) * p->Timebase_interval;
          dt_gravkick2 = dt_hydrokick2 = (tend - tcurrent) * p->Timebase_interval;
        }
      for(int j = 0; j < 3; j++)
        vel[3 * i + j] += gravaccel[3 * i + j] * dt_gravkick;
      if(type[i] == 0)
        {
          for(int j = 0; j < 3; j++)
            {
              vel[3 * i + j] += hydroaccel[3 * i + j] * dt_hydrokick;
              velpred[3 * i + j] = vel[3 * i + j] - dt_gravkick2 * gravaccel[3 * i + j] -
                                   dt_hydrokick2 * hydroaccel[3 * i + j];
              if(p->pmgrid) /* timestep.c:511-513 */
                velpred[3 * i + j] += p->gravpm[3 * i + j] * p->dt_gravkickB;
            }
          /* timestep.c:553-557 (DO_NOT_PROTECT off) */
          if(dtentropy[i] * dt_entr > -0.5 * entropy[i])
            entropy[i] += dtentropy[i] * dt_entr;
          else
            entropy[i] *= 0.5;
          if(p->MinEgySpec)
            {
              double minentropy =
                p->MinEgySpec * ORC_GAMMA_MINUS1 / pow(density[i] * a3inv, ORC_GAMMA_MINUS1);
              if(entropy[i] < minentropy)
                {
                  entropy[i] = minentropy;
                  dtentropy[i] = 0;
                }
            }
          /* timestep.c:590-593 */
          dt_entr = (timebin[i] ? (1 << timebin[i]) : 0) / 2 * p->Timebase_interval;
          if(entropy[i] + dtentropy[i] * dt_entr < 0.5 * entropy[i])
            dtentropy[i] = -0.5 * entropy[i] / dt_entr;
        }
    }
  (void) pressure;
  for(int b = 0; b < 32; b++)
    bincount[b] = bincount_sph[b] = 0;
  for(int i = 0; i < n; i++)
    {
      bincount[timebin[i]]++;
      if(type[i] == 0)
        bincount_sph[timebin[i]]++;
    }
  (void) ngas;
  return err;
}

/* the long-range kick ending a PM step (timestep.c:301-345): all particles, not the active list.
 * tables as in orc_drift (NULL when not comoving); dt_gravkick / dt_gravkickB from the caller's
 * PM-step bookkeeping (timestep.c:273-300). */
void orc_pm_kick(int n, int ngas, int ti_current, double timebase, const double *tables,
                 double logTimeBegin, double logTimeMax, double dt_gravkick, double dt_gravkickB,
                 const int *type, const int *timebin, const int *ti_begstep, double *vel,
                 const double *gravaccel, const double *gravpm, const double *hydroaccel,
                 double *velpred)
{
  for(int i = 0; i < n; i++)
    {
      for(int j = 0; j < 3; j++)
        vel[3 * i + j] += gravpm[3 * i + j] * dt_gravkick;
      if(type[i] == 0 && i < ngas)
        {
          int dt_step = (timebin[i] ? (1 << timebin[i]) : 0);
          double dt_gravkickA, dt_hydrokick;
          if(tables)
            {
              dt_gravkickA =
                table_factor(tables + 1000, ti_begstep[i], ti_current, timebase, logTimeBegin,
                             logTimeMax) -
                table_factor(tables + 1000, ti_begstep[i], ti_begstep[i] + dt_step / 2, timebase,
                             logTimeBegin, logTimeMax);
              dt_hydrokick =
                table_factor(tables + 2000, ti_begstep[i], ti_current, timebase, logTimeBegin,
                             logTimeMax) -
                table_factor(tables + 2000, ti_begstep[i], ti_begstep[i] + dt_step / 2, timebase,
                             logTimeBegin, logTimeMax);
            }
          else
            dt_gravkickA = dt_hydrokick = (ti_current - (ti_begstep[i] + dt_step / 2)) * timebase;
          for(int j = 0; j < 3; j++)
            velpred[3 * i + j] = vel[3 * i + j] + gravaccel[3 * i + j] * dt_gravkickA +
                                 hydroaccel[3 * i + j] * dt_hydrokick +
                                 gravpm[3 * i + j] * dt_gravkickB;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * "next" row N4: sink (black-hole) neighbour passes of the shipped flag bundle
 * (BLACK_HOLES, SWALLOWGAS, ACCRETION_RADIUS, ACCRETION_DENSITY, ACCRETION_OF_DUST_ONLY,
 *  BH_MERGERS_WITHIN_H, BH_THERMALFEEDBACK + TMP_FEEDBACK, DUST; Makefile:67-72, 96, 124, 199-204)
 * ---------------------------------------------------------------------------------------- */

/* ngb_treefind_blackhole (blackhole.c:1351-1474), mode 0: like ngb_treefind_variable, but gas,
 * sinks and (DUST) dust grains are candidates */
static int ngb_treefind_blackhole(const orc_tree *t, const double c[3], double hsml, int dust,
                                  int periodic, double boxsize, int *ngblist)
{
  double boxhalf = 0.5 * boxsize;
  int numngb = 0;
  int no = t->n;
  while(no >= 0)
    {
      if(no < t->n)
        {
          int p = no;
          no = t->nextnode[no];
          int ty = t->type[p];
          if(ty != 0 && ty != 5 && !(dust && ty == 2)) /* :1372-1380 */
            continue;
          double dist = hsml;
          double dx = ngb_periodic(t->pos[3 * p + 0] - c[0], periodic, boxsize, boxhalf);
          if(dx > dist)
            continue;
          double dy = ngb_periodic(t->pos[3 * p + 1] - c[1], periodic, boxsize, boxhalf);
          if(dy > dist)
            continue;
          double dz = ngb_periodic(t->pos[3 * p + 2] - c[2], periodic, boxsize, boxhalf);
          if(dz > dist)
            continue;
          if(dx * dx + dy * dy + dz * dz > dist * dist)
            continue;
          ngblist[numngb++] = p;
        }
      else
        {
          const onode *cur = &NODE(t, no);
          double dist = hsml + 0.5 * cur->len; /* :1453 */
          no = cur->sibling;
          double dx = ngb_periodic(cur->center[0] - c[0], periodic, boxsize, boxhalf);
          if(dx > dist)
            continue;
          double dy = ngb_periodic(cur->center[1] - c[1], periodic, boxsize, boxhalf);
          if(dy > dist)
            continue;
          double dz = ngb_periodic(cur->center[2] - c[2], periodic, boxsize, boxhalf);
          if(dz > dist)
            continue;
          dist += FACT1 * cur->len;
          if(dx * dx + dy * dy + dz * dz > dist * dist)
            continue;
          no = cur->nextnode;
        }
    }
  return numngb;
}

static inline void nearest_image(double *dx, double *dy, double *dz, int periodic, double boxsize)
{
  double boxhalf = 0.5 * boxsize;
  if(!periodic)
    return;
  if(*dx > boxhalf)
    *dx -= boxsize;
  if(*dx < -boxhalf)
    *dx += boxsize;
  if(*dy > boxhalf)
    *dy -= boxsize;
  if(*dy < -boxhalf)
    *dy += boxsize;
  if(*dz > boxhalf)
    *dz -= boxsize;
  if(*dz < -boxhalf)
    *dz += boxsize;
}

/* density() for Type-5 targets (density.c:125-704 with the BLACK_HOLES branches: 406-413,
 * 521-532, 548-551, 613, 629; density_evaluate :763-765, 831-834, 879-886, 972-978): the h
 * iteration against the gas, wanting DesNumNgb * BlackHoleNgbFactor neighbours, without the
 * Newton step (that needs SphP[i].h.DhsmlDensityFactor: gas only, :613, 629); the sink's
 * BH_Density, BH_Entropy (kernel-weighted mean entropy) and BH_SurroundingGasVel.
 * hsml [n] in/out at the sinks' indices; outputs [nsink].  Returns iterations or -1. */
int orc_sink_density(const orc_tree *t, const orc_dens_params *p, double ngbfactor, int nsink,
                     const int *sink, const double *velpred, const double *entropy, double *hsml,
                     double *numngb, double *bh_density, double *bh_entropy, double *bh_gasvel)
{
  int n = t->n;
  double boxsize = p->BoxSize;
  double *Left = (double *) calloc((size_t) (nsink > 0 ? nsink : 1), sizeof(double));
  double *Right = (double *) calloc((size_t) (nsink > 0 ? nsink : 1), sizeof(double));
  char *done = (char *) calloc((size_t) (nsink > 0 ? nsink : 1), 1);
  int *ngblist = (int *) malloc((size_t) n * sizeof(int));
  double desnumngb = p->DesNumNgb * ngbfactor; /* :548-551 */
  int iter = 0, npleft;
  do
    {
      npleft = 0;
      for(int a = 0; a < nsink; a++)
        {
          if(done[a])
            continue;
          int i = sink[a];
          const double *pos = t->pos + 3 * (size_t) i;
          double h = hsml[i], h2 = h * h, hinv = 1.0 / h, hinv3 = hinv * hinv * hinv;
          double rho = 0, wn = 0, smoothentr = 0, gasvel[3] = { 0, 0, 0 };
          int nn = ngb_treefind(t, pos, h, NULL, 0, p->periodic, boxsize, ngblist);
          for(int q = 0; q < nn; q++)
            {
              int j = ngblist[q];
              if(t->mass[j] == 0) /* :831-834 */
                continue;
              double dx = pos[0] - t->pos[3 * j], dy = pos[1] - t->pos[3 * j + 1],
                     dz = pos[2] - t->pos[3 * j + 2];
              nearest_image(&dx, &dy, &dz, p->periodic, boxsize);
              double r2 = dx * dx + dy * dy + dz * dz;
              if(r2 < h2)
                {
                  double r = sqrt(r2), u = r * hinv, wk;
                  if(u < 0.5)
                    wk = hinv3 * (KERNEL_COEFF_1 + KERNEL_COEFF_2 * (u - 1) * u * u);
                  else
                    wk = hinv3 * KERNEL_COEFF_5 * (1.0 - u) * (1.0 - u) * (1.0 - u);
                  double mass_j = t->mass[j];
                  rho += mass_j * wk;
                  wn += NORM_COEFF * wk / hinv3;
                  gasvel[0] += mass_j * wk * velpred[3 * j];
                  gasvel[1] += mass_j * wk * velpred[3 * j + 1];
                  gasvel[2] += mass_j * wk * velpred[3 * j + 2];
                  smoothentr += mass_j * wk * entropy[j];
                }
            }
          numngb[a] = wn;
          bh_density[a] = rho;
          if(rho > 0) /* :521-532 */
            {
              bh_entropy[a] = smoothentr / rho;
              for(int k = 0; k < 3; k++)
                bh_gasvel[3 * a + k] = gasvel[k] / rho;
            }
          else
            {
              bh_entropy[a] = smoothentr;
              for(int k = 0; k < 3; k++)
                bh_gasvel[3 * a + k] = gasvel[k];
            }
          /* :559-652 */
          if(wn < (desnumngb - p->MaxNumNgbDeviation) ||
             (wn > (desnumngb + p->MaxNumNgbDeviation) && h > (1.01 * p->MinGasHsml)))
            {
              npleft++;
              if(Left[a] > 0 && Right[a] > 0)
                if((Right[a] - Left[a]) < 1.0e-3 * Left[a])
                  {
                    npleft--;
                    done[a] = 1;
                    continue;
                  }
              if(wn < (desnumngb - p->MaxNumNgbDeviation))
                Left[a] = h > Left[a] ? h : Left[a];
              else
                {
                  if(Right[a] != 0)
                    {
                      if(h < Right[a])
                        Right[a] = h;
                    }
                  else
                    Right[a] = h;
                }
              if(Right[a] > 0 && Left[a] > 0)
                h = pow(0.5 * (pow(Left[a], 3) + pow(Right[a], 3)), 1.0 / 3);
              else
                {
                  if(Right[a] == 0 && Left[a] > 0)
                    h *= 1.26; /* no Newton step for non-gas targets, :613 */
                  if(Right[a] > 0 && Left[a] == 0)
                    h /= 1.26;
                }
              if(h < p->MinGasHsml)
                h = p->MinGasHsml;
              hsml[i] = h;
            }
          else
            done[a] = 1;
        }
      if(npleft > 0)
        {
          iter++;
          if(iter > p->maxiter)
            {
              iter = -1;
              break;
            }
        }
    }
  while(npleft > 0);
  free(Left);
  free(Right);
  free(done);
  free(ngblist);
  return iter;
}

/* blackhole_evaluate (blackhole.c:794-1190), mode 0, for the flag bundle above.
 * Per neighbour j within Hsml of sink i (both with mass > 0):
 *   sink   (:935-1001)  closer than the accretion boundary (InnerBoundary for the central object,
 *                       mass > 0.95 SMBHmass, else SofteningBndry) and not heavier: marked
 *   dust   (:1004-1036) closer than InnerBoundary / SinkBoundary and bound (e_tot <= 0): marked
 *   gas    (:1040-1160) kernel weight; the central object marks gas inside InnerBoundary; other
 *                       sinks mark gas inside SinkBoundary only without ACCRETION_OF_DUST_ONLY
 *                       (bound, or denser than CritDensity with ACCRETION_DENSITY); thermal
 *                       feedback of the smaller sinks is spread with the kernel
 * "marked" = SwallowID[j] = ID of the sink.  Two deliberate deviations from the reference text:
 *   (1) a victim claimed by several sinks goes to the LARGEST ID, for all three kinds -- the
 *       reference says so for gas (:1068, 1088, 1106) and lets the last sink of the active list win
 *       for dust and sinks (:984-988, 1028: a result that depends on the list order);
 *   (2) the dust branch computes e_tot with r = sqrt(r2) of the grain -- the reference reads `r`
 *       there (:1023) before any assignment in that iteration (it is set in the gas branch only).
 * dt_fac = Timebase_interval / hubble_a (:822); ascale = All.Time or 1. */
void orc_blackhole_evaluate(const orc_tree *t, const orc_bh_params *p, int nsink, const int *sink,
                            const unsigned int *id, const double *hsml, const int *timebin,
                            const double *bh_mdot, const double *bh_density,
                            const double *gas_density, unsigned int *swallowid,
                            double *injected_energy)
{
  int n = t->n;
  int *ngblist = (int *) malloc((size_t) n * sizeof(int));
  for(int a = 0; a < nsink; a++)
    {
      int i = sink[a];
      const double *pos = t->pos + 3 * (size_t) i, *velocity = t->vel + 3 * (size_t) i;
      double mass = t->mass[i], h_i = hsml[i], h_i2 = h_i * h_i;
      double rho = bh_density[a], mdot = bh_mdot[a];
      double dt = (timebin[i] ? (1 << timebin[i]) : 0) * p->dt_fac;
      unsigned int myid = id[i];
      int central = mass > 0.95 * p->SMBHmass;
      int nn = ngb_treefind_blackhole(t, pos, h_i, p->dust, p->periodic, p->BoxSize, ngblist);
      for(int q = 0; q < nn; q++)
        {
          int j = ngblist[q];
          if(!(t->mass[j] > 0) || !(mass > 0))
            continue;
          double dx = pos[0] - t->pos[3 * j], dy = pos[1] - t->pos[3 * j + 1],
                 dz = pos[2] - t->pos[3 * j + 2];
          nearest_image(&dx, &dy, &dz, p->periodic, p->BoxSize);
          double r2 = dx * dx + dy * dy + dz * dz;
          if(!(r2 < h_i2))
            continue;
          double vrel = 0;
          for(int k = 0; k < 3; k++)
            vrel += (t->vel[3 * j + k] - velocity[k]) * (t->vel[3 * j + k] - velocity[k]);
          vrel = sqrt(vrel) / p->ascale;
          if(t->type[j] == 5 && r2 > 0)
            {
              double acc_boundary = central ? p->InnerBoundary : p->SofteningBndry;
              if(!(pow(r2, 0.5) > acc_boundary))
                if(t->mass[j] <= mass && swallowid[j] < myid)
                  swallowid[j] = myid;
            }
          if(p->dust && t->type[j] == 2 && r2 > 0)
            {
              double acc_boundary = central ? p->InnerBoundary : p->SinkBoundary;
              if(pow(r2, 0.5) < acc_boundary)
                {
                  double etotal = vrel * vrel / 2. - mass / (sqrt(r2) + 1.e-20);
                  if(etotal <= 0. && swallowid[j] < myid)
                    swallowid[j] = myid;
                }
            }
          if(t->type[j] == 0)
            {
              double r = sqrt(r2), hinv = 1 / h_i, hinv3 = hinv * hinv * hinv, u = r * hinv, wk;
              if(u < 0.5)
                wk = hinv3 * (KERNEL_COEFF_1 + KERNEL_COEFF_2 * (u - 1) * u * u);
              else
                wk = hinv3 * KERNEL_COEFF_5 * (1.0 - u) * (1.0 - u) * (1.0 - u);
              double etotal = vrel * vrel / 2. - mass / (r + 1.e-20);
              if(central)
                {
                  if(r < p->InnerBoundary && swallowid[j] < myid)
                    swallowid[j] = myid;
                }
              else if(!p->accretion_of_dust_only)
                {
                  if(r < p->SinkBoundary)
                    {
                      int ok = p->accretion_density ? (gas_density[j] >= p->CritDensity) : (etotal < 0.);
                      if(ok && swallowid[j] < myid)
                        swallowid[j] = myid;
                    }
                }
              /* :1114-1150 TMP_FEEDBACK without FRACTION_OF_LSOLAR_FB: only the smaller sinks */
              double energy = 0.;
              if(mass < 0.95 * p->SMBHmass)
                energy = p->FeedbackCoeff * pow(mass * p->UnitMass_in_g, 0.6667) * mdot *
                         p->UnitMass_in_g * dt;
              injected_energy[j] += energy * t->mass[j] * wk / rho;
            }
        }
    }
  free(ngblist);
}

/* blackhole_evaluate_swallow (blackhole.c:1201-1346), mode 0: every neighbour marked with the
 * sink's ID is swallowed -- its mass and momentum are summed, its own mass (and BH_Mass) set to
 * zero.  mass [n] and particle_bh_mass [n] (BH_Mass of sinks, else ignored) are modified.
 * Outputs [nsink]: accreted mass, accreted BH mass, accreted dust mass, momentum [nsink][3];
 * counts[3] = gas, sinks, dust swallowed. */
void orc_blackhole_swallow(const orc_tree *t, const orc_bh_params *p, int nsink, const int *sink,
                           const unsigned int *id, const double *hsml,
                           const unsigned int *swallowid, double *mass, double *particle_bh_mass,
                           double *acc_mass, double *acc_bhmass, double *acc_dustmass,
                           double *acc_momentum, long long counts[3])
{
  int n = t->n;
  int *ngblist = (int *) malloc((size_t) n * sizeof(int));
  counts[0] = counts[1] = counts[2] = 0;
  for(int a = 0; a < nsink; a++)
    {
      int i = sink[a];
      const double *pos = t->pos + 3 * (size_t) i;
      unsigned int myid = id[i];
      double am = 0, ab = 0, ad = 0, mom[3] = { 0, 0, 0 };
      int nn = ngb_treefind_blackhole(t, pos, hsml[i], p->dust, p->periodic, p->BoxSize, ngblist);
      for(int q = 0; q < nn; q++)
        {
          int j = ngblist[q];
          if(swallowid[j] != myid)
            continue;
          if(t->type[j] == 5)
            {
              am += mass[j];
              ab += particle_bh_mass[j];
              for(int k = 0; k < 3; k++)
                mom[k] += mass[j] * t->vel[3 * j + k];
              mass[j] = 0;
              particle_bh_mass[j] = 0;
              counts[1]++;
            }
          else if(t->type[j] == 2)
            {
              am += mass[j];
              ad += mass[j];
              for(int k = 0; k < 3; k++)
                mom[k] += mass[j] * t->vel[3 * j + k];
              mass[j] = 0;
              counts[2]++;
            }
          else if(t->type[j] == 0)
            {
              am += mass[j];
              for(int k = 0; k < 3; k++)
                mom[k] += mass[j] * t->vel[3 * j + k];
              mass[j] = 0.;
              counts[0]++;
            }
        }
      acc_mass[a] = am;
      acc_bhmass[a] = ab;
      acc_dustmass[a] = ad;
      for(int k = 0; k < 3; k++)
        acc_momentum[3 * a + k] = mom[k];
    }
  free(ngblist);
}

/* cooling_and_starformation (sfr_eff.c:82-947), the deterministic per-particle part that is active
 * for the shipped bundle (COOLING, SFR, BH_FORM without JEANS_MASS_SF, BLACK_HOLES,
 * BH_THERMALFEEDBACK) with the cooling function itself (DoCooling, cooling.c) left to the caller
 * as unew -> unew (identity here):
 *   flag   (:226-229, 459-462)  0 = the gas particle qualifies for conversion into a sink
 *                               (Density >= CritPhysDensity in code units) unless its mass is 0
 *   unew   (:486-488)           max(MinEgySpec, (A + dA/dt dt) / (gamma-1) rho^(gamma-1))
 *          (:509-531)           + Injected_BH_Energy / Mass, capped at 5e9 K; the injection is consumed
 *   dA/dt  (:582-594)           (unew (gamma-1) / rho^(gamma-1) - A) / dt, floor -0.5 A / dt
 * Non-comoving (a3inv = 1).  The conversion itself (Type = 5, the random mass factor of :618) is
 * the host's: it needs the GSL stream. */
void orc_cooling_and_starformation(int nactive, const int *active, int ngas, const int *type,
                                   const double *mass, const int *timebin, double timebase,
                                   double CritPhysDensity_code, double MinEgySpec,
                                   double u_to_temp_fac, const double *density,
                                   const double *entropy, double *dtentropy,
                                   double *injected_energy, int *flag_sink)
{
  for(int a = 0; a < nactive; a++)
    {
      int i = active[a];
      if(i >= ngas || type[i] != 0)
        continue;
      double dt = (timebin[i] ? (1 << timebin[i]) : 0) * timebase;
      int flag = 1;
      if(density[i] >= CritPhysDensity_code)
        flag = 0;
      if(mass[i] == 0)
        flag = 1;
      flag_sink[i] = !flag;
      if(flag == 1)
        {
          double unew = (entropy[i] + dtentropy[i] * dt) / GAMMA_MINUS1 * pow(density[i], GAMMA_MINUS1);
          if(unew < MinEgySpec)
            unew = MinEgySpec;
          if(injected_energy[i])
            {
              if(mass[i] == 0)
                injected_energy[i] = 0;
              else
                unew += injected_energy[i] / mass[i];
              double temp = u_to_temp_fac * unew;
              if(temp > 5.0e9)
                unew = 5.0e9 / u_to_temp_fac;
              injected_energy[i] = 0;
            }
          if(timebin[i] && dt > 0)
            {
              dtentropy[i] = (unew * GAMMA_MINUS1 / pow(density[i], GAMMA_MINUS1) - entropy[i]) / dt;
              if(dtentropy[i] < -0.5 * entropy[i] / dt)
                dtentropy[i] = -0.5 * entropy[i] / dt;
            }
        }
    }
}
