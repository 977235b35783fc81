// Work counters without a hot address.  Every wavefront of the walks and of the SPH kernels adds its
// totals to device counters; tens of thousands of atomics on ONE address serialise at the memory side
// (measured on a c2 shard's walks, 1 024 buckets = 49 000 wavefronts: Newtonian 1.62 -> 1.05 ms, Ewald
// 1.12 -> 0.63 ms alone, the pair 2.33 -> 1.29 ms with the counters switched off).  A counter is
// therefore 64 slots, one 128-byte line each, chosen by the workgroup index; readers add the slots.
#ifndef GHIP_COUNT_H
#define GHIP_COUNT_H

#define GHIP_CSLOTS 64
#define GHIP_CSLOT_U64 16   // u64 per slot (one 128-byte line): [0] interactions / neighbours / pairs, [1] element visits
#define GHIP_CK_NEWTON 0    // counter kinds (Newtonian and short-range walk / Ewald walk / external
#define GHIP_CK_EWALD 1     // targets / density passes / single-target density / hydro)
#define GHIP_CK_EXT 2
#define GHIP_CK_DENS 3
#define GHIP_CK_DENS1 4
#define GHIP_CK_HYDRO 5
#define GHIP_CK_COUNT 8
#define GHIP_CKIND_U64 (GHIP_CSLOTS * GHIP_CSLOT_U64)
#define GHIP_CBUF_BYTES ((size_t) GHIP_CK_COUNT * GHIP_CKIND_U64 * 8)

#ifdef __HIPCC__
__device__ __forceinline__ void d_count(unsigned long long *__restrict__ kind_base, int which,
                                        unsigned long long v)
{
  atomicAdd(kind_base + (blockIdx.x & (GHIP_CSLOTS - 1)) * GHIP_CSLOT_U64 + which, v);
}
#endif
#endif
