// ssx_lanestat.h -- profiling build only (-DSSX_LANESTAT, tools/lanestat.py): how many lanes are active
// where.  SSX_STAT(region): the first active lane of the wave adds popcount(exec) and 1 to the region's two
// global counters (atomics: slow, which does not matter for a count).  The product library is built
// without the macro: SSX_STAT expands to nothing.
#pragma once
#ifdef SSX_LANESTAT
#define SSX_NSTAT 20
__device__ unsigned long long g_lanestat[2 * SSX_NSTAT];
#define SSX_STAT(r) do { const unsigned long long m_ = __ballot(1); if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) { \
	atomicAdd(&g_lanestat[2 * (r)], (unsigned long long)__popcll(m_)); atomicAdd(&g_lanestat[2 * (r) + 1], 1ull); } } while (0)
#else
#define SSX_STAT(r) do {} while (0)
#endif




