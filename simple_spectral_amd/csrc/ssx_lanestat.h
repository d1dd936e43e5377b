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

// Profiling build of its own (-DSSX_REGTIME, tools/regtime.py; NOT together with SSX_LANESTAT, whose global atomics slow the kernel 40-fold).
// SSX_TIME(tm, r): where the waves' TIME goes, measured (tools/regtime.py; VERDICT r04 item 1 asked for a cycle-weighted view and
// the static census of tools/isa_census.py prices issue slots only): the shader clock (s_memtime) since the wave's previous mark is
// added to region r -- the region that ends here -- in the wave's own LDS accumulators (behind the log counters: 16 x 8 bytes per wave),
// flushed to g_regtime when the wave leaves the kernel.  A wave's time in a region includes the cycles it waited for the SIMD's other
// three waves, for LDS / HBM round trips and for the instruction cache: regions whose share of the time exceeds their share of the
// issue cycles are the latency-bound ones.
#ifdef SSX_REGTIME
#define SSX_NTIME 16
__device__ unsigned long long g_regtime[SSX_NTIME];
struct SsxTimer { unsigned long long last; unsigned long long* acc; };
#define SSX_TIME(tm, r) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0u) atomicAdd((tm).acc + (r), t_ - (tm).last); (tm).last = t_; } while (0)
// the same, with the clock read pinned BEHIND the computation of `dep` (a VGPR value the region produces): the scheduler otherwise moves
// a clock read over pure arithmetic (pass 1 of the intersection is nothing else)
#define SSX_TIME_AFTER(tm, r, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); \
	if ((threadIdx.x & 63u) == 0u) atomicAdd((tm).acc + (r), t_ - (tm).last); (tm).last = t_; } while (0)
#else
struct SsxTimer {};
#define SSX_TIME(tm, r) do { (void)(tm); } while (0)
#define SSX_TIME_AFTER(tm, r, dep) do { (void)(tm); } while (0)
#endif




