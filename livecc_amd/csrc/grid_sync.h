// Device-wide synchronisation inside ONE launch on gfx950 (8 XCDs with private, mutually non-coherent L2s; per-CU L1s that are
// never refreshed by other CUs' stores): the XCD-hierarchical grid barrier of MI355X_MICROARCH.md ("barrier-xcd": 4.1 us at 256
// workgroups against 7.4-10.4 us for one agent-scope counter polled by every block) and its census.
//
// Protocol (placement-independent: groups are formed by the XCC id the hardware reports, never by blockIdx):
//   census (once per launch) : every block adds itself to members[xcc]; one flat counter barrier makes the counts final.
//   barrier k = 1, 2, ...    : every thread drains its stores (vmcnt(0)) -> __syncthreads -> lane 0 arrives at its XCD's counter
//                              (relaxed agent atomic).  The LAST arriver of an XCD is that XCD's leader for this barrier: ONE
//                              agent-scope release fence (buffer_wbl2: writes back the XCD L2's dirty lines, i.e. the stores of
//                              every block of the XCD, which reached the shared L2 before they arrived) -> asm vmcnt(0) -> top
//                              counter -> polls the top counter (relaxed, s_sleep) until all XCDs arrived -> publishes the XCD's
//                              generation word.  Everyone else polls its XCD's generation word (one L2-local line per XCD instead
//                              of 256 pollers on one line across the fabric).  Every block ends with ONE agent-scope acquire
//                              (buffer_inv sc1: drops its CU's stale L1 lines) -> __syncthreads -> plain loads.
// Every spin is bounded: a block that gives up sets *fail and returns false (the caller leaves the kernel; nothing hangs).
// State: zero `GridSyncState` before every launch (hipMemsetAsync in the launch function); counters are monotonic within a launch.
#pragma once
#include <hip/hip_runtime.h>

namespace lcc {

struct GridSyncState {
  unsigned members[8][32];   // [xcc][0] = blocks of this launch running on that XCD (one 128-byte line per XCD)
  unsigned arrive[8][32];    // [xcc][0] = monotonic arrival counter of the XCD
  unsigned gen[8][32];       // [xcc][0] = last completed barrier index of the XCD
  unsigned top[32];          // [0] = monotonic arrival counter of the XCD leaders
  unsigned census[32];       // [0] = flat arrival counter of the census barrier
  unsigned fail[32];         // [0] = blocks that gave up
};

#define LCC_GS_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned gs_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

// bounded relaxed poll of one word until it reaches `target` (monotonic counters: >=)
__device__ __forceinline__ bool gs_wait_ge(unsigned* w, unsigned target, unsigned* fail, int max_spins) {
  int spins = 0;
  while (__hip_atomic_load(w, LCC_GS_RLX) < target) {
    if (++spins > max_spins) { __hip_atomic_fetch_add(fail, 1u, LCC_GS_RLX); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

struct GridSync {
  GridSyncState* s;
  unsigned xcc, n_mine, n_xcc, k;   // my XCD, its member count, number of populated XCDs, barriers passed
  bool ok;
};

// once per launch, by every block (all threads call it; returns the same value in every thread of the block)
__device__ __forceinline__ GridSync gs_begin(GridSyncState* s, int max_spins = 200000) {
  __shared__ __attribute__((aligned(16))) unsigned sh[4];
  GridSync g; g.s = s; g.k = 0;
  if (threadIdx.x == 0) {
    const unsigned xcc = gs_xcc_id();
    __hip_atomic_fetch_add(&s->members[xcc][0], 1u, LCC_GS_RLX);
    __hip_atomic_fetch_add(&s->census[0], 1u, LCC_GS_RLX);
    const bool ok = gs_wait_ge(&s->census[0], gridDim.x * gridDim.y * gridDim.z, &s->fail[0], max_spins);
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&s->members[i][0], LCC_GS_RLX) != 0u;
    sh[0] = xcc; sh[1] = __hip_atomic_load(&s->members[xcc][0], LCC_GS_RLX); sh[2] = nx; sh[3] = ok ? 1u : 0u;
  }
  __syncthreads();
  g.xcc = sh[0]; g.n_mine = sh[1]; g.n_xcc = sh[2]; g.ok = sh[3] != 0u;
  __syncthreads();
  return g;
}

// the barrier; all threads of every block call it.  false = this block (or the launch) gave up: leave the kernel.
__device__ __forceinline__ bool gs_barrier(GridSync& g, int max_spins = 200000) {
  __shared__ __attribute__((aligned(16))) unsigned sh_okv[4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its stores have reached the XCD's L2
  __syncthreads();
  g.k += 1;
  if (threadIdx.x == 0) {
    GridSyncState* s = g.s;
    bool ok = g.ok;
    const unsigned old = __hip_atomic_fetch_add(&s->arrive[g.xcc][0], 1u, LCC_GS_RLX);
    if (old + 1 == g.k * g.n_mine) {                    // last arriver of this XCD: its leader for barrier k
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the wait behind buffer_wbl2 (guide: compiler hazard)
      __hip_atomic_fetch_add(&s->top[0], 1u, LCC_GS_RLX);
      ok = gs_wait_ge(&s->top[0], g.k * g.n_xcc, &s->fail[0], max_spins) && ok;
      __hip_atomic_store(&s->gen[g.xcc][0], g.k, LCC_GS_RLX);
    } else {
      ok = gs_wait_ge(&s->gen[g.xcc][0], g.k, &s->fail[0], max_spins) && ok;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    sh_okv[0] = ok ? 1u : 0u;
  }
  __syncthreads();
  g.ok = sh_okv[0] != 0u;
  return g.ok;
}

}  // namespace lcc
