// Shared host/device helpers for the nlbac HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/nlbac_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

extern thread_local char nlbac_err_buf[512];

#define NLBAC_FAIL(...)                                              \
    do {                                                             \
        snprintf(nlbac_err_buf, sizeof(nlbac_err_buf), __VA_ARGS__); \
        return -1;                                                   \
    } while (0)

#define NLBAC_CHECK_LAUNCH(name)                                                           \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) NLBAC_FAIL("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define NLBAC_REQUIRE(cond, ...) \
    do {                         \
        if (!(cond)) NLBAC_FAIL(__VA_ARGS__); \
    } while (0)

static inline int nlbac_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Block-wide deterministic sum of NV values per thread (256-thread blocks):
// wave shuffle tree, then a fixed-order combine of the 4 wave results.
template <int NV>
__device__ __forceinline__ void block_sum_256(float (&v)[NV], float* lds /* >= 4*NV floats */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) lds[wave * NV + i] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
    __syncthreads();
}

// Last-workgroup election for "partials -> final value in the same launch", WITHOUT an agent-scope fence: on gfx950 a
// release at agent scope writes the XCD's whole L2 back (node_kernels.hip measured +15..30 us for it behind a kernel's
// own stores).  Wave 0 publishes thread 0's N partial results with device-scope atomic exchanges (performed at the
// level every XCD sees), waits for them to return, then thread 0 takes a ticket; the block that draws the last ticket reads the
// others' results with ``coherent_load``.  Returns the verdict in every thread of the block (contains a barrier).
//
// CONTRACT — gfx950 only.  The HIP / LLVM memory model gives no happens-before between a relaxed exchange, a relaxed
// ticket and a relaxed load; what orders them here is the hardware and one compiler barrier:
//   * an agent-scope atomic that RETURNS a value has been performed at the memory-side coherence point (beyond the
//     XCDs' L2s) when its result arrives, and ``asm volatile("" :: "v"(old) : "memory")`` makes the wave wait for that
//     result (s_waitcnt vmcnt(0)) and keeps the compiler from moving the ticket above it;
//   * the ticket is an atomic at the same point, so the workgroup that draws the last one does so after every other
//     workgroup's exchanges were performed there;
//   * ``coherent_load`` (agent-scope atomic load: sc1) does not hit a stale line of the reading XCD's L2.
// Any other target gets fences instead (NLBAC_ELECT_FENCED: release before the ticket, acquire in the elected block).
// tests/test_elect_gpu.py drives nlbac_elect_selftest — thousands of workgroups over all XCDs, salted values, hundreds
// of back-to-back launches — against host sums.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#define NLBAC_ELECT_FENCED 1
#endif
__device__ __forceinline__ void elect_release_() {
#ifdef NLBAC_ELECT_FENCED
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}
__device__ __forceinline__ void elect_acquire_() {
#ifdef NLBAC_ELECT_FENCED
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
template <int N>
__device__ __forceinline__ bool publish_and_elect(float* dst, const float (&vals)[N], unsigned* ticket, unsigned n_blocks) {
    __shared__ unsigned s_elect_;
    if (threadIdx.x < 64) {
        // thread 0's N values go out as ONE wave instruction (lane k sends value k): N dependent round trips to the
        // memory side otherwise, ~0.6 us each
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float bk = __shfl(vals[k], 0, 64);
            if ((int)threadIdx.x == k) mine = bk;
        }
        float old = 0.f;
        if ((int)threadIdx.x < N) old = __hip_atomic_exchange(dst + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");        // every exchange has returned before the ticket is taken
        elect_release_();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_elect_ = (t == n_blocks - 1u) ? 1u : 0u;
            if (s_elect_) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_elect_ != 0u) elect_acquire_();
    return s_elect_ != 0u;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global loads AND stores
// (s_waitcnt vmcnt(0)): behind a burst of stores that is the burst's whole trip to memory.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The same election for launches of many workgroups: ONE ticket word takes ~88 atomics per microsecond, so the 768
// workgroups of a three-net quarter-panel launch that all finish together queued ~9 us at it (mlp_rrq_bwd with the TD
// head: 31 us in the update against 21 with plain dy).  Two levels: workgroup b takes a ticket of its group's word
// (ELECT_GROUP consecutive indices share one), the last of a group takes one of the top word, the last there is elected.
// tickets: 1 + ceil(n_blocks / ELECT_GROUP) zeroed words (left zero again).  Ordering: as above, level by level — a
// group's last ticket is drawn after every member's exchanges were performed, the top word's last after every group's.
#define ELECT_GROUP 16
template <int N>
__device__ __forceinline__ bool publish_and_elect_grouped(float* dst, const float (&vals)[N], unsigned* tickets, unsigned index,
                                                          unsigned n_blocks) {
    __shared__ unsigned s_elect_g_;
    if (threadIdx.x < 64) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float bk = __shfl(vals[k], 0, 64);
            if ((int)threadIdx.x == k) mine = bk;
        }
        float old = 0.f;
        if ((int)threadIdx.x < N) old = __hip_atomic_exchange(dst + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");        // every exchange has returned before the ticket is taken
        elect_release_();
        if (threadIdx.x == 0) {
            const unsigned grp = index / ELECT_GROUP, n_groups = (n_blocks + ELECT_GROUP - 1) / ELECT_GROUP;
            const unsigned in_group = min((unsigned)ELECT_GROUP, n_blocks - grp * ELECT_GROUP);
            unsigned elected = 0u;
            const unsigned t1 = __hip_atomic_fetch_add(tickets + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t1 == in_group - 1u) {
                __hip_atomic_store(tickets + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned t2 = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t2 == n_groups - 1u) {
                    __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    elected = 1u;
                }
            }
            s_elect_g_ = elected;
        }
    }
    __syncthreads();
    if (s_elect_g_ != 0u) elect_acquire_();
    return s_elect_g_ != 0u;
}
// The same with the n (<= 64) values already spread over the lanes of wave 0 (lane k < n holds value k in `mine`).
__device__ __forceinline__ bool publish_and_elect_grouped_lanes(float* dst, float mine, int n, unsigned* tickets, unsigned index,
                                                                unsigned n_blocks) {
    __shared__ unsigned s_elect_gl_;
    if (threadIdx.x < 64) {
        float old = 0.f;
        if ((int)threadIdx.x < n) old = __hip_atomic_exchange(dst + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");        // every exchange has returned before the ticket is taken
        elect_release_();
        if (threadIdx.x == 0) {
            const unsigned grp = index / ELECT_GROUP, n_groups = (n_blocks + ELECT_GROUP - 1) / ELECT_GROUP;
            const unsigned in_group = min((unsigned)ELECT_GROUP, n_blocks - grp * ELECT_GROUP);
            unsigned elected = 0u;
            const unsigned t1 = __hip_atomic_fetch_add(tickets + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t1 == in_group - 1u) {
                __hip_atomic_store(tickets + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned t2 = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t2 == n_groups - 1u) {
                    __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    elected = 1u;
                }
            }
            s_elect_gl_ = elected;
        }
    }
    __syncthreads();
    if (s_elect_gl_ != 0u) elect_acquire_();
    return s_elect_gl_ != 0u;
}
__device__ __forceinline__ float coherent_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
