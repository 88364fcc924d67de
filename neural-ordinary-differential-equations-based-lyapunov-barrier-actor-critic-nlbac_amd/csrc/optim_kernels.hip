// Adam (torch.optim.Adam single-tensor arithmetic), gradient-slab reduction and
// soft target update.  Replaces torch.optim.Adam.step at
// U/sac_cbf_clf/sac_cbf_clf.py:249-255,284-308, U/sac_cbf_clf/model.py:258 and
// soft_update at U/sac_cbf_clf/utils.py:75-79.
//
// HBM-bound streaming kernels: one float4 per lane, grid capped at 2048 blocks.
#include <cstdint>
#include "common.h"

thread_local char nlbac_err_buf[512] = "";

extern "C" int nlbac_abi_version(void) { return NLBAC_ABI_VERSION; }
extern "C" const char* nlbac_last_error(void) { return nlbac_err_buf; }

struct AdamState {
    int step;
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    unsigned ticket;   // nlbac_adam_fused: workgroups that have finished (zero between launches)
};

__global__ void adam_prepare_kernel(AdamState* st, double lr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = st->step + 1;
        st->step = t;
        const double bc1 = 1.0 - pow(0.9, (double)t);
        const double bc2 = 1.0 - pow(0.999, (double)t);
        st->step_size = (float)(lr / bc1);
        st->bc2_sqrt = (float)sqrt(bc2);
    }
}

__device__ __forceinline__ float adam_one(float& p, float& m, float& v, float g, float step_size, float bc2_sqrt) {
    // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    // (the scalars are python doubles cast to fp32 at use, as ATen does)
    m = m + (g - m) * (float)(1.0 - 0.9);
    v = v * 0.999f + ((float)(1.0 - 0.999) * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    p = p + ((-step_size) * m) / denom;   // addcdiv_: self + value * t1 / t2
    return p;
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ grad,
                                                        int n_slabs, long slab_stride, long n,
                                                        const AdamState* __restrict__ st,
                                                        float* __restrict__ target, float tau) {
    const float step_size = st->step_size, bc2_sqrt = st->bc2_sqrt;
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        for (int s = 1; s < n_slabs; ++s) {
            const float4 gs = reinterpret_cast<const float4*>(grad + s * slab_stride)[i];
            g.x += gs.x; g.y += gs.y; g.z += gs.z; g.w += gs.w;
        }
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, mm.x, vv.x, g.x, step_size, bc2_sqrt);
        adam_one(pp.y, mm.y, vv.y, g.y, step_size, bc2_sqrt);
        adam_one(pp.z, mm.z, vv.z, g.z, step_size, bc2_sqrt);
        adam_one(pp.w, mm.w, vv.w, g.w, step_size, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (target) {
            float4 t = reinterpret_cast<float4*>(target)[i];
            t.x = t.x * (1.0f - tau) + pp.x * tau; t.y = t.y * (1.0f - tau) + pp.y * tau;
            t.z = t.z * (1.0f - tau) + pp.z * tau; t.w = t.w * (1.0f - tau) + pp.w * tau;
            reinterpret_cast<float4*>(target)[i] = t;
        }
    }
    // tail (n not a multiple of 4)
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grad[i];
        for (int s = 1; s < n_slabs; ++s) g += grad[s * slab_stride + i];
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, mm, vv, g, step_size, bc2_sqrt);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (target) target[i] = target[i] * (1.0f - tau) + pp * tau;
    }
}

// One launch per optimiser step: ++step and its bias corrections (adam_prepare), the slab sum + Adam + Polyak update
// (adam_step), and the refresh of the MFMA-fragment copies of the weights (mlp_pack) through a scatter table: for
// parameter i, scat[SLOTS i + k] are the device addresses of its fragment slots (forward / backward pack, and the RR
// packs of nets that have them: SLOTS = 2 or 4; 0 = none;
// biases and skinny layers are read from the flat parameters).  Every workgroup derives the step constants from the
// still un-incremented counter; the workgroup that finishes last publishes the new counter.
template <int SLOTS>
__device__ __forceinline__ void scatter_n(const unsigned long long* __restrict__ scat, long i, float val) {
#pragma unroll
    for (int k = 0; k < SLOTS / 2; ++k) {
        const ulonglong2 ab = reinterpret_cast<const ulonglong2*>(scat)[i * (SLOTS / 2) + k];   // two slots per load
        if (ab.x) *reinterpret_cast<float*>(ab.x) = val;
        if (ab.y) *reinterpret_cast<float*>(ab.y) = val;
    }
}
#define scatter2 scatter_n<SLOTS>

// sum of the gradient slabs at float4 index i, in slab order (the additions are sequential as before; the loads of
// four slabs are issued together instead of one per loop trip - the trips were a chain of exposed memory latencies)
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ grad, int n_slabs, long slab_stride, long i) {
    float4 g = reinterpret_cast<const float4*>(grad)[i];
    int s = 1;
    for (; s + 3 < n_slabs; s += 4) {
        const float4 a = reinterpret_cast<const float4*>(grad + (long)s * slab_stride)[i];
        const float4 b = reinterpret_cast<const float4*>(grad + (long)(s + 1) * slab_stride)[i];
        const float4 c = reinterpret_cast<const float4*>(grad + (long)(s + 2) * slab_stride)[i];
        const float4 d = reinterpret_cast<const float4*>(grad + (long)(s + 3) * slab_stride)[i];
        g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
        g.x += b.x; g.y += b.y; g.z += b.z; g.w += b.w;
        g.x += c.x; g.y += c.y; g.z += c.z; g.w += c.w;
        g.x += d.x; g.y += d.y; g.z += d.z; g.w += d.w;
    }
    for (; s < n_slabs; ++s) {
        const float4 a = reinterpret_cast<const float4*>(grad + (long)s * slab_stride)[i];
        g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
    }
    return g;
}

// temperatures refreshed by the step itself: alpha = exp(log_alpha) for up to two log_alpha entries of this arena
// (sac_cbf_clf.py:297, 308), written by the thread that has just stepped the entry
// mirror: the last workgroup of the step also sends `n_mirror` floats at `mirror_src` (the agent's scalars block: the
// losses this update returns) to `mirror_dst`, a pinned HOST buffer the device writes directly — the update's last
// launch delivers its results itself instead of a copy launch behind it, on which the host would wait.  Entries the
// step refreshes (the temperatures) go out from the thread that computes them.
struct AlphaRefresh { long off[2]; float* dst[2]; const float* mirror_src; float* mirror_dst; int n_mirror; };
__device__ __forceinline__ void alpha_refresh(const AlphaRefresh& AR, long e, float p_new) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (e == AR.off[k]) {
            const float a = expf(p_new);
            *AR.dst[k] = a;
            if (AR.mirror_dst) {
                const long i = AR.dst[k] - AR.mirror_src;
                if (i >= 0 && i < AR.n_mirror) AR.mirror_dst[i] = a;
            }
        }
}

template <int SLOTS>
__global__ __launch_bounds__(256) void adam_fused_kernel(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ grad,
                                                         int n_slabs, long slab_stride, long n, AdamState* st, double lr,
                                                         float* __restrict__ target, float tau,
                                                         const unsigned long long* __restrict__ scat,
                                                         const unsigned long long* __restrict__ scat_t,
                                                         const AlphaRefresh AR) {
    __shared__ float s_const[2];
    __shared__ int s_step;
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const double bc1 = 1.0 - pow(0.9, (double)t);
        const double bc2 = 1.0 - pow(0.999, (double)t);
        s_const[0] = (float)(lr / bc1);
        s_const[1] = (float)sqrt(bc2);
        s_step = t;
    }
    __syncthreads();
    const float step_size = s_const[0], bc2_sqrt = s_const[1];
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = slab_sum4(grad, n_slabs, slab_stride, i);
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, mm.x, vv.x, g.x, step_size, bc2_sqrt);
        adam_one(pp.y, mm.y, vv.y, g.y, step_size, bc2_sqrt);
        adam_one(pp.z, mm.z, vv.z, g.z, step_size, bc2_sqrt);
        adam_one(pp.w, mm.w, vv.w, g.w, step_size, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (AR.off[0] >= 0 || AR.off[1] >= 0) {
            alpha_refresh(AR, 4 * i + 0, pp.x); alpha_refresh(AR, 4 * i + 1, pp.y);
            alpha_refresh(AR, 4 * i + 2, pp.z); alpha_refresh(AR, 4 * i + 3, pp.w);
        }
        if (scat) {
            scatter2(scat, 4 * i + 0, pp.x); scatter2(scat, 4 * i + 1, pp.y);
            scatter2(scat, 4 * i + 2, pp.z); scatter2(scat, 4 * i + 3, pp.w);
        }
        if (target) {
            float4 t = reinterpret_cast<float4*>(target)[i];
            t.x = t.x * (1.0f - tau) + pp.x * tau; t.y = t.y * (1.0f - tau) + pp.y * tau;
            t.z = t.z * (1.0f - tau) + pp.z * tau; t.w = t.w * (1.0f - tau) + pp.w * tau;
            reinterpret_cast<float4*>(target)[i] = t;
            if (scat_t) {
                scatter2(scat_t, 4 * i + 0, t.x); scatter2(scat_t, 4 * i + 1, t.y);
                scatter2(scat_t, 4 * i + 2, t.z); scatter2(scat_t, 4 * i + 3, t.w);
            }
        }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grad[i];
        for (int s = 1; s < n_slabs; ++s) g += grad[s * slab_stride + i];
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, mm, vv, g, step_size, bc2_sqrt);
        p[i] = pp; m[i] = mm; v[i] = vv;
        alpha_refresh(AR, i, pp);
        if (scat) scatter2(scat, i, pp);
        if (target) {
            const float t = target[i] * (1.0f - tau) + pp * tau;
            target[i] = t;
            if (scat_t) scatter2(scat_t, i, t);
        }
    }
    __syncthreads();                                   // every wave of this block has read the constants
    __shared__ int s_is_last;
    if (threadIdx.x == 0) {
        s_is_last = 0;
        // relaxed: every workgroup consumed st->step (it computed its constants from it) before the barrier above, so
        // the counter may move once the last ticket is drawn; no agent-scope release per workgroup (that would write the
        // XCD's L2 back behind every block's parameter stores)
        const unsigned ticket = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {                 // all workgroups have read st->step by now
            __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st->step_size = step_size;
            st->bc2_sqrt = bc2_sqrt;
            __hip_atomic_store(&st->step, s_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_is_last = 1;
        }
    }
    if (!AR.mirror_dst) return;
    __syncthreads();
    if (s_is_last)                                     // (written by earlier launches, except the refreshed entries)
        for (int i = threadIdx.x; i < AR.n_mirror; i += blockDim.x) {
            const bool refreshed = (AR.off[0] >= 0 && AR.dst[0] == AR.mirror_src + i) ||
                                   (AR.off[1] >= 0 && AR.dst[1] == AR.mirror_src + i);
            if (!refreshed) AR.mirror_dst[i] = AR.mirror_src[i];
        }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* __restrict__ out, const float* __restrict__ grad,
                                                           int n_slabs, long slab_stride, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grad[i];
        for (int s = 1; s < n_slabs; ++s) g += grad[s * slab_stride + i];
        out[i] = g;
    }
}

__global__ __launch_bounds__(256) void soft_update_kernel(float* __restrict__ target, const float* __restrict__ src,
                                                          long n, float tau) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        target[i] = target[i] * (1.0f - tau) + src[i] * tau;
}

__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* x, float b, const float* y, long n, float* out) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// out[c] = mul * sum_b partials[b][c]   (fixed order; one thread per column)
__global__ void sum_partials_kernel(const float* partials, int n_blk, int n_cols, float mul, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    float s = 0.f;
    for (int b = 0; b < n_blk; ++b) s += partials[(long)b * n_cols + c];
    out[c] = s * mul;
}

static inline int stream_grid(long n, int per_thread = 1) {
    long b = (n / per_thread + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

extern "C" int nlbac_adam_prepare(void* state, double lr, nlbac_stream_t s) {
    NLBAC_REQUIRE(state, "nlbac_adam_prepare: null state");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (AdamState*)state, lr);
    NLBAC_CHECK_LAUNCH("nlbac_adam_prepare");
    return 0;
}

extern "C" int nlbac_adam_step(float* p, float* m, float* v, const float* grad, int n_slabs, long slab_stride,
                               long n, const void* state, float* target, float tau, nlbac_stream_t s) {
    NLBAC_REQUIRE(p && m && v && grad && state, "nlbac_adam_step: null pointer");
    NLBAC_REQUIRE(n_slabs >= 1 && n >= 1, "nlbac_adam_step: bad sizes");
    NLBAC_REQUIRE(((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad | (uintptr_t)target) % 16 == 0 &&
                      slab_stride % 4 == 0,
                  "nlbac_adam_step: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_step_kernel, dim3(stream_grid(n, 4)), dim3(256), 0, (hipStream_t)s, p, m, v, grad,
                       n_slabs, slab_stride, n, (const AdamState*)state, (tau >= 0.f) ? target : nullptr, tau);
    NLBAC_CHECK_LAUNCH("nlbac_adam_step");
    return 0;
}

extern "C" int nlbac_adam_fused(float* p, float* m, float* v, const float* grad, int n_slabs, long slab_stride, long n,
                                void* state, double lr, float* target, float tau, const void* scatter,
                                const void* scatter_target, int scatter_slots, int n_alpha, const long* alpha_off,
                                float* const* alpha_dst, const float* mirror_src, float* mirror_dst, int n_mirror,
                                nlbac_stream_t s) {
    NLBAC_REQUIRE(p && m && v && grad && state, "nlbac_adam_fused: null pointer");
    NLBAC_REQUIRE(scatter_slots == 2 || scatter_slots == 4 || (!scatter && !scatter_target),
                  "nlbac_adam_fused: scatter tables hold 2 or 4 slots per parameter");
    NLBAC_REQUIRE(n_alpha >= 0 && n_alpha <= 2 && (n_alpha == 0 || (alpha_off && alpha_dst)), "nlbac_adam_fused: bad alpha refresh");
    AlphaRefresh AR;
    AR.off[0] = AR.off[1] = -1; AR.dst[0] = AR.dst[1] = nullptr;
    NLBAC_REQUIRE((mirror_dst == nullptr) || (mirror_src && n_mirror > 0 && n_mirror <= 1024), "nlbac_adam_fused: bad mirror");
    AR.mirror_src = mirror_src; AR.mirror_dst = mirror_dst; AR.n_mirror = mirror_dst ? n_mirror : 0;
    for (int k = 0; k < n_alpha; ++k) {
        NLBAC_REQUIRE(alpha_off[k] >= 0 && alpha_off[k] < n && alpha_dst[k], "nlbac_adam_fused: alpha entry out of range");
        AR.off[k] = alpha_off[k]; AR.dst[k] = alpha_dst[k];
    }
    NLBAC_REQUIRE(n_slabs >= 1 && n >= 1 && lr > 0.0, "nlbac_adam_fused: bad sizes");
    NLBAC_REQUIRE(((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad | (uintptr_t)target) % 16 == 0 &&
                      slab_stride % 4 == 0 && ((uintptr_t)scatter | (uintptr_t)scatter_target) % 8 == 0,
                  "nlbac_adam_fused: buffers must be 16-byte aligned");
    NLBAC_REQUIRE(!scatter_target || (target && tau >= 0.f), "nlbac_adam_fused: scatter_target without a target");
    hipLaunchKernelGGL((scatter_slots == 4 ? adam_fused_kernel<4> : adam_fused_kernel<2>), dim3(stream_grid(n, 4)), dim3(256),
                       0, (hipStream_t)s, p, m, v, grad, n_slabs, slab_stride, n, (AdamState*)state, lr,
                       (tau >= 0.f) ? target : nullptr, tau, (const unsigned long long*)scatter,
                       (const unsigned long long*)scatter_target, AR);
    NLBAC_CHECK_LAUNCH("nlbac_adam_fused");
    return 0;
}

extern "C" int nlbac_reduce_slabs(float* out, const float* grad, int n_slabs, long slab_stride, long n, nlbac_stream_t s) {
    NLBAC_REQUIRE(out && grad && n_slabs >= 1, "nlbac_reduce_slabs: bad arguments");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, out, grad, n_slabs, slab_stride, n);
    NLBAC_CHECK_LAUNCH("nlbac_reduce_slabs");
    return 0;
}

extern "C" int nlbac_soft_update(float* target, const float* src, long n, float tau, nlbac_stream_t s) {
    NLBAC_REQUIRE(target && src, "nlbac_soft_update: null pointer");
    hipLaunchKernelGGL(soft_update_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, target, src, n, tau);
    NLBAC_CHECK_LAUNCH("nlbac_soft_update");
    return 0;
}

extern "C" int nlbac_axpby(float a, const float* x, float b, const float* y, long n, float* out, nlbac_stream_t s) {
    NLBAC_REQUIRE(x && out, "nlbac_axpby: null pointer");
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, a, x, b, y, n, out);
    NLBAC_CHECK_LAUNCH("nlbac_axpby");
    return 0;
}

extern "C" int nlbac_fill(float* p, float v, long n, nlbac_stream_t s) {
    NLBAC_REQUIRE(p, "nlbac_fill: null pointer");
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)s, p, v, n);
    NLBAC_CHECK_LAUNCH("nlbac_fill");
    return 0;
}

extern "C" int nlbac_sum_partials(const float* partials, int n_blk, int n_cols, float mul, float* out, nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && out && n_blk >= 1 && n_cols >= 1, "nlbac_sum_partials: bad arguments");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(nlbac_ceil_div(n_cols, 64)), dim3(64), 0, (hipStream_t)s, partials, n_blk, n_cols, mul, out);
    NLBAC_CHECK_LAUNCH("nlbac_sum_partials");
    return 0;
}


// ---------------------------------------------------------------------------
// Replay minibatch gather (replay_memory.py:21-25 on the device): dst[r] = src[idx[r]] for rows of ld floats
// (ld % 4 == 0, 16-byte aligned rows): one lane per float4, consecutive lanes on consecutive 16-byte pieces of a row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* __restrict__ src, int ld4,
                                                          const long* __restrict__ idx, long n_rows, long src_rows,
                                                          float4* __restrict__ dst) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * ld4) return;
    const long r = t / ld4;
    const int c = (int)(t - r * ld4);
    long j = idx[r];
    j = j < 0 ? 0 : (j >= src_rows ? src_rows - 1 : j);       // never read outside the buffer
    dst[r * ld4 + c] = src[j * ld4 + c];
}

// ---------------------------------------------------------------------------
// Device-side draw of a minibatch (device_rng mode): row indices uniform on [0, src_rows) with
// replacement, the gather into minibatch layout and the N(0,1) policy noise of the update, in ONE launch (the update
// boundary is where the host is the bottleneck).  Philox4x32-10, counter = (draw number, row | lane); Box-Muller.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
        ctr = make_uint4((unsigned)(p1 >> 32) ^ ctr.y ^ key.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ key.y,
                         (unsigned)p0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}

__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 0x1p-24f + 0x1p-25f; }   // (0, 1)

__global__ __launch_bounds__(256) void sample_rows_kernel(const float4* __restrict__ src, int ld4, long n_rows,
                                                          long src_rows, float4* __restrict__ dst,
                                                          float* __restrict__ eps, long n_eps,
                                                          unsigned long long seed, unsigned long long draw) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    if (t < n_rows * ld4) {
        const long r = t / ld4;
        const int c = (int)(t - r * ld4);
        const uint4 x = philox4x32_10(make_uint4((unsigned)draw, (unsigned)(draw >> 32), (unsigned)r, 0u), key);
        const long j = (long)(((unsigned long long)x.x * (unsigned long long)src_rows) >> 32);
        dst[r * ld4 + c] = src[j * ld4 + c];
    }
    const long q = (n_eps + 3) >> 2;
    if (eps && t < q) {
        const uint4 x = philox4x32_10(make_uint4((unsigned)draw, (unsigned)(draw >> 32), (unsigned)t, 1u), key);
        const float r0 = sqrtf(-2.f * logf(u01(x.x))), r1 = sqrtf(-2.f * logf(u01(x.z)));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u01(x.y), &s0, &c0);
        sincosf(6.283185307179586f * u01(x.w), &s1, &c1);
        const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
        for (int k = 0; k < 4; ++k)
            if (4 * t + k < n_eps) eps[4 * t + k] = v[k];
    }
}

extern "C" int nlbac_sample_rows(const float* src, long src_rows, int ld, long n_rows, float* dst, float* eps,
                                 long n_eps, unsigned long long seed, unsigned long long draw, nlbac_stream_t s) {
    NLBAC_REQUIRE(src && dst && src_rows >= 1 && src_rows < (1L << 32) && n_rows >= 1 && n_eps >= 0,
                  "nlbac_sample_rows: bad arguments");
    NLBAC_REQUIRE(ld >= 4 && ld % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                  "nlbac_sample_rows: rows must be whole float4s (ld %d)", ld);
    NLBAC_REQUIRE(eps || n_eps == 0, "nlbac_sample_rows: n_eps without a buffer");
    const long total = n_rows * (ld / 4), q = (n_eps + 3) / 4;
    const long threads = total > q ? total : q;
    hipLaunchKernelGGL(sample_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       (const float4*)src, ld / 4, n_rows, src_rows, (float4*)dst, eps, n_eps, seed, draw);
    NLBAC_CHECK_LAUNCH("nlbac_sample_rows");
    return 0;
}

// n_blocks blocks of block_len contiguous dwords each, block b at src + b * src_stride -> dst + b * dst_stride
// (a row range of a stage-major buffer: one block per stage).  Moves bits: float and int32 buffers alike.
__global__ __launch_bounds__(256) void copy_blocks_kernel(const float* __restrict__ src, long src_stride,
                                                          float* __restrict__ dst, long dst_stride, long block_len,
                                                          long n_blocks) {
    const long total = block_len * n_blocks, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long b = i / block_len, j = i - b * block_len;
        dst[b * dst_stride + j] = src[b * src_stride + j];
    }
}

extern "C" int nlbac_copy_blocks(const void* src, long src_stride, void* dst, long dst_stride, long block_len,
                                 long n_blocks, nlbac_stream_t s) {
    NLBAC_REQUIRE(src && dst && block_len >= 1 && n_blocks >= 1 && src_stride >= block_len && dst_stride >= block_len,
                  "nlbac_copy_blocks: bad arguments");
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(stream_grid(block_len * n_blocks)), dim3(256), 0, (hipStream_t)s,
                       (const float*)src, src_stride, (float*)dst, dst_stride, block_len, n_blocks);
    NLBAC_CHECK_LAUNCH("nlbac_copy_blocks");
    return 0;
}

extern "C" int nlbac_gather_rows(const float* src, long src_rows, int ld, const long* idx, long n_rows, float* dst,
                                 nlbac_stream_t s) {
    NLBAC_REQUIRE(src && idx && dst && src_rows >= 1 && n_rows >= 1, "nlbac_gather_rows: bad arguments");
    NLBAC_REQUIRE(ld >= 4 && ld % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                  "nlbac_gather_rows: rows must be whole float4s (ld %d)", ld);
    const long total = n_rows * (ld / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       (const float4*)src, ld / 4, idx, n_rows, src_rows, (float4*)dst);
    NLBAC_CHECK_LAUNCH("nlbac_gather_rows");
    return 0;
}
