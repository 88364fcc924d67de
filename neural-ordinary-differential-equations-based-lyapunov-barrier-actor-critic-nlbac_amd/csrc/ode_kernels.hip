// Explicit Runge-Kutta machinery of odeint on t=[t0,t1] for the control-affine
// NODE field  k = f(x) + g(x) u  (U/sac_cbf_clf/model.py:208-217), replacing
// the torchdiffeq.odeint call sites U/sac_cbf_clf/sac_cbf_clf.py:453,577 and
// U/sac_cbf_clf/model.py:252.  euler / rk4 (3/8 rule) / dopri5 (FSAL, RMS
// error norm over the whole batch tensor incl. the carried action columns,
// one shared step per problem, 4th-order interpolant at t1).
//
// The heavy part of every stage (f_net / g_net) runs in mlp_kernels.hip; the
// kernels here are the per-row stage algebra: rows are P problems x B rows,
// state dimension n_s <= 8, action dimension n_u <= 4.  Stage derivatives are
// kept stage-major: K[stage][row][n_s].
#include "common.h"
#include "ode_control.h"

#define MAX_NS 8
#define MAX_NUA 4
#define MAX_STAGES 8
#define MAX_PROBLEMS 8

struct RkCoef { float c[MAX_STAGES]; float h[MAX_PROBLEMS]; };

__device__ __forceinline__ float step_of(const RkCoef& rc, const double* h_dev, int h_stride, int p) {
    return h_dev ? (float)h_dev[(long)p * h_stride] : rc.h[p];
}

// k[r] = f[r] + sum_c g[r][c] u[c]
__global__ __launch_bounds__(256) void affine_fwd_kernel(const float* f, const float* g, const float* u, int n_s,
                                                         int n_u, int n, float* k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int r = 0; r < n_s; ++r) {
        float a = f[(long)i * n_s + r];
        for (int c = 0; c < n_u; ++c) a += g[(long)i * n_s * n_u + r * n_u + c] * u[(long)i * n_u + c];
        k[(long)i * n_s + r] = a;
    }
}

// dg[r][c] = dk[r] u[c] ; du[c] (+)= mul * sum_r g[r][c] dk[r]   (df == dk, no copy)
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* dk, const float* g, const float* u, int n_s,
                                                         int n_u, int n, float mul, float* dg, float* du, int acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < n_u; ++c) {
        float a = 0.f;
        const float uc = u[(long)i * n_u + c];
        for (int r = 0; r < n_s; ++r) {
            const float d = dk[(long)i * n_s + r];
            a += g[(long)i * n_s * n_u + r * n_u + c] * d;
            if (dg) dg[(long)i * n_s * n_u + r * n_u + c] = d * uc;
        }
        if (du) du[(long)i * n_u + c] = (acc ? du[(long)i * n_u + c] : 0.f) + mul * a;
    }
}

// out = (y0 ? y0 : 0) + sum_j (coef[j]*h_p) * K[j]
__global__ __launch_bounds__(256) void rk_combine_kernel(const float* y0, const float* K, int n_k, const RkCoef rc,
                                                         const double* h_dev, int h_stride, int rpp, int n_s, int n,
                                                         float* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float h = step_of(rc, h_dev, h_stride, i / rpp);
    for (int r = 0; r < n_s; ++r) {
        float a = y0 ? y0[(long)i * n_s + r] : 0.f;
        for (int j = 0; j < n_k; ++j)
            if (rc.c[j] != 0.f) a = a + K[((long)j * n + i) * n_s + r] * (rc.c[j] * h);
        out[(long)i * n_s + r] = a;
    }
}

// Stage backward:  dY = (dYup?) + (dXf?) + (dXg?) ;  dy0 (+)= dY ; dK[j] += coef[j]*h*dY (j<n_k)
// (dXf/dXg are the input grads of f_net/g_net at this stage, ld = dx_ld)
__global__ __launch_bounds__(256) void rk_stage_bwd_kernel(const float* dYup, const float* dXf, const float* dXg,
                                                           int dx_ld, int n_k, const RkCoef rc, const double* h_dev,
                                                           int h_stride, int rpp, int n_s, int n, float* dK,
                                                           float* dy0, int acc_dy0, float* du_ext, int n_ext) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float h = step_of(rc, h_dev, h_stride, i / rpp);
    // non-affine field: columns n_s.. of the net's input gradient belong to the carried inputs (u, t)
    if (du_ext && dXf)
        for (int c = 0; c < n_ext; ++c) du_ext[(long)i * n_ext + c] += dXf[(long)i * dx_ld + n_s + c];
    for (int r = 0; r < n_s; ++r) {
        float d = dYup ? dYup[(long)i * n_s + r] : 0.f;
        if (dXf) d += dXf[(long)i * dx_ld + r];
        if (dXg) d += dXg[(long)i * dx_ld + r];
        if (dy0) dy0[(long)i * n_s + r] = (acc_dy0 ? dy0[(long)i * n_s + r] : 0.f) + d;
        for (int j = 0; j < n_k; ++j)
            if (rc.c[j] != 0.f) dK[((long)j * n + i) * n_s + r] += (rc.c[j] * h) * d;
    }
}

// partial sums of squared scaled quantities; partials [P][nblk][2]
//  mode 0: col0 = sum (y0/scale)^2 (+ (u/scale_u)^2), col1 = sum (a/scale)^2          a = f0
//  mode 1: col0 = sum ((a-b)/scale)^2                                                  a = f1, b = f0
//  mode 2: col0 = sum (a/tol)^2, tol = atol + rtol*max(|y0|,|y1|)                      a = err
__device__ __forceinline__ void dopri_norm_block(const float* a, const float* b, const float* y0, const float* y1,
                                                 const float* u, int mode, float rtol, float atol, int n_s, int n_u,
                                                 int rpp, float* partials, const double* slot_ctl = nullptr,
                                                 long slot_floats = 0, bool publish = false) {
    __shared__ float red[8];
    const int p = blockIdx.y;
    if (slot_ctl) {      // device-driven chain (mode 2): the attempt's buffers are those of step slot C_NACC
        const int slot = (int)slot_ctl[(long)p * NLBAC_DOPRI_CTL + C_NACC];
        a += (long)slot * slot_floats;
        y1 += (long)slot * slot_floats;
        if (slot > 0) y0 = y1 - slot_floats;          // the step starts from its predecessor's y1
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (i < rpp) {
        const long row = (long)p * rpp + i;
        for (int r = 0; r < n_s; ++r) {
            const float y = y0[row * n_s + r];
            if (mode == 2) {
                const float tol = atol + rtol * fmaxf(fabsf(y), fabsf(y1[row * n_s + r]));
                const float q = a[row * n_s + r] / tol;
                v[0] += q * q;
            } else {
                const float sc = atol + fabsf(y) * rtol;
                if (mode == 0) {
                    const float q0 = y / sc, q1 = a[row * n_s + r] / sc;
                    v[0] += q0 * q0; v[1] += q1 * q1;
                } else {
                    const float q = (a[row * n_s + r] - b[row * n_s + r]) / sc;
                    v[0] += q * q;
                }
            }
        }
        if (mode == 0)
            for (int c = 0; c < n_u; ++c) {
                const float y = u[row * n_u + c];
                const float q = y / (atol + fabsf(y) * rtol);
                v[0] += q * q;
            }
    }
    block_sum_256<2>(v, red);
    float* q = partials + ((long)p * gridDim.x + blockIdx.x) * 2;
    if (!publish) {
        if (threadIdx.x == 0) { q[0] = v[0]; q[1] = v[1]; }
        return;
    }
    // for a reader in another workgroup of THIS launch (dopri_norm_control_kernel): both sums leave as device-scope
    // atomic exchanges, one wave instruction, and have returned when this function does — no agent-scope fence, which on
    // gfx950 writes the XCD's L2 back (common.h::publish_and_elect)
    if (threadIdx.x < 64) {
        float old = 0.f;
        if (threadIdx.x < 2) old = __hip_atomic_exchange(q + threadIdx.x, threadIdx.x ? v[1] : v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");
    }
}

__global__ __launch_bounds__(256) void dopri_norm_kernel(const float* a, const float* b, const float* y0,
                                                         const float* y1, const float* u, int mode, float rtol,
                                                         float atol, int n_s, int n_u, int rpp, float* partials,
                                                         const double* slot_ctl, long slot_floats) {
    if (slot_ctl && mode == 2 && slot_ctl[(long)blockIdx.y * NLBAC_DOPRI_CTL + C_DONE] > 0.0) return;
    dopri_norm_block(a, b, y0, y1, u, mode, rtol, atol, n_s, n_u, rpp, partials, slot_ctl, slot_floats);
}

// the controller of problem p on its finished squared-norm sums (plain RMS norm over the problem's rows)
__device__ __forceinline__ void dopri_control_one(double s0, double s1, int p, int mode, int n_s, int n_u,
                                                  int rpp, double t_end, double* ctl) {
    const double cnt = (double)rpp * (double)(n_s + n_u);
    dopri_control_vals(sqrt(s0 / cnt), sqrt(s1 / cnt), p, mode, t_end, ctl);
}

__global__ void dopri_control_kernel(const float* partials, int nblk, int mode, int n_s, int n_u, int rpp,
                                     double t_end, double* ctl, int n_slots, double* hslots, double* alog,
                                     int alog_cap) {
    if (threadIdx.x != 0) return;
    const int p = blockIdx.x;
    double* c = ctl + (long)p * NLBAC_DOPRI_CTL;
    if (n_slots > 0 && mode == 2 && c[C_DONE] > 0.0) return;       // chained: a finished solve stays as it is
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += (double)partials[((long)p * nblk + b) * 2 + 0];
        s1 += (double)partials[((long)p * nblk + b) * 2 + 1];
    }
    const int slot_before = (int)c[C_NACC];
    const double h_try = c[C_H];
    const double cnt = (double)rpp * (double)(n_s + n_u);
    dopri_control_vals(sqrt(s0 / cnt), sqrt(s1 / cnt), p, mode, t_end, ctl, n_slots > 0 ? n_slots : (1 << 30));
    if (hslots && mode == 2 && c[C_ACCEPT] > 0.0) hslots[(long)p * n_slots + slot_before] = h_try;
    if (alog && mode == 2) {
        const int k = (int)c[C_NSTEPS] - 1;
        if (k >= 0 && k < alog_cap) {
            double* a = alog + ((long)p * alog_cap + k) * 3;
            a[0] = h_try; a[1] = c[C_RATIO]; a[2] = c[C_ACCEPT];
        }
    }
}

// The controller of an attempted step on the tile partials its RK launch left (nlbac_rk_chain::norm_defer with norm
// mode 2): one wave per problem, the sums in the order the fused form's elected workgroup takes them (same bits), then
// everything dopri_norm_control_kernel's last thread does — step slots, attempt log, the host's copy.
__global__ __launch_bounds__(64) void dopri_control_tiles_kernel(const float* partials, int nblk, int n_s, int n_u, int rpp,
                                                                 double t_end, double* ctl, int n_slots, double* hslots,
                                                                 double* alog, int alog_cap, double* ctl_host,
                                                                 double host_seq) {
    const int p = blockIdx.x, tid = threadIdx.x;
    double* c = ctl + (long)p * NLBAC_DOPRI_CTL;
    if (c[C_DONE] > 0.0) return;                  // (a finished solve is left alone: its tiles wrote nothing)
    double d0 = 0.0, d1 = 0.0;
    for (int b = tid; b < nblk; b += 64) {
        const float* q = partials + ((long)p * nblk + b) * 2;
        d0 += (double)q[0];
        d1 += (double)q[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); }
    if (tid != 0) return;
    const int slot_before = (int)c[C_NACC];
    const double h_try = c[C_H];
    const double cnt = (double)rpp * (double)(n_s + n_u);
    dopri_control_vals(sqrt(d0 / cnt), sqrt(d1 / cnt), p, 2, t_end, ctl, n_slots > 0 ? n_slots : (1 << 30));
    if (hslots && c[C_ACCEPT] > 0.0) hslots[(long)p * n_slots + slot_before] = h_try;
    if (alog) {
        const int k = (int)c[C_NSTEPS] - 1;
        if (k >= 0 && k < alog_cap) {
            double* al = alog + ((long)p * alog_cap + k) * 3;
            al[0] = h_try; al[1] = c[C_RATIO]; al[2] = c[C_ACCEPT];
        }
    }
    if (ctl_host) ctl_host_post(ctl_host + (long)p * NLBAC_DOPRI_CTL, c, host_seq);
}

// norm + controller in one launch: the workgroup that finishes a problem's sums last (a ticket counter per problem,
// left at zero again for the next launch) runs that problem's controller
__global__ __launch_bounds__(256) void dopri_norm_control_kernel(const float* a, const float* b, const float* y0,
                                                                 const float* y1, const float* u, int mode, float rtol,
                                                                 float atol, int n_s, int n_u, int rpp, double t_end,
                                                                 float* partials, unsigned* tickets, double* ctl,
                                                                 const double* slot_ctl, long slot_floats, int n_slots,
                                                                 double* hslots, double* alog, int alog_cap,
                                                                 double* ctl_host, double host_seq) {
    // device-driven chain (slot_ctl): a finished problem is left alone by all of its blocks
    if (slot_ctl && mode == 2 && slot_ctl[(long)blockIdx.y * NLBAC_DOPRI_CTL + C_DONE] > 0.0) return;
    dopri_norm_block(a, b, y0, y1, u, mode, rtol, atol, n_s, n_u, rpp, partials, slot_ctl, slot_floats, true);
    __shared__ unsigned s_last;
    __shared__ double s_red[2][256];
    const int p = blockIdx.y, nblk = (int)gridDim.x;
    if (threadIdx.x == 0) {                            // (this block's sums have been performed device-wide: see above)
        const unsigned ticket = __hip_atomic_fetch_add(tickets + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x - 1) ? 1u : 0u;
        if (s_last) __hip_atomic_store(tickets + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;                               // (uniform per block)
    // the sums were written by other workgroups of this launch: read them past the non-coherent caches; fixed
    // assignment of blocks to lanes and a fixed tree, so the result does not depend on which block came last
    double v0 = 0.0, v1 = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float* q = partials + ((long)p * nblk + b) * 2;
        v0 += (double)__hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 += (double)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_red[0][threadIdx.x] = v0;
    s_red[1][threadIdx.x] = v1;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_red[0][threadIdx.x] += s_red[0][threadIdx.x + w];
            s_red[1][threadIdx.x] += s_red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* c = ctl + (long)p * NLBAC_DOPRI_CTL;
        const int slot_before = (int)c[C_NACC];
        const double h_try = c[C_H];
        const double cnt = (double)rpp * (double)(n_s + n_u);
        dopri_control_vals(sqrt(s_red[0][0] / cnt), sqrt(s_red[1][0] / cnt), p, mode, t_end, ctl,
                           n_slots > 0 ? n_slots : (1 << 30));
        if (hslots && mode == 2 && c[C_ACCEPT] > 0.0) hslots[(long)p * n_slots + slot_before] = h_try;
        if (alog && mode == 2) {
            const int k = (int)c[C_NSTEPS] - 1;
            if (k >= 0 && k < alog_cap) {
                double* al = alog + ((long)p * alog_cap + k) * 3;
                al[0] = h_try; al[1] = c[C_RATIO]; al[2] = c[C_ACCEPT];
            }
        }
        // the host's copy of this problem's block (pinned memory the device writes directly): what a copy launch on a
        // side stream did before; the host waits on an event behind this launch, or (host_seq) polls the block's stamp
        if (ctl_host) ctl_host_post(ctl_host + (long)p * NLBAC_DOPRI_CTL, c, host_seq);
    }
}

struct InterpArg { float h[MAX_PROBLEMS]; float x[MAX_PROBLEMS]; const double* ctl; long slot_floats; };

// device-driven chain: the last accepted step of problem p lives in step slot C_NACC (0 without slots)
__device__ __forceinline__ int interp_slot(const InterpArg& ia, int p) {
    return (ia.ctl && ia.slot_floats) ? (int)ia.ctl[(long)p * NLBAC_DOPRI_CTL + C_NACC] : 0;
}

// step size / interpolation abscissa of problem p: from the device control block when given
// (so a captured hipGraph replays with the current values), else by value
__device__ __forceinline__ void interp_hx(const InterpArg& ia, int p, float& h, float& x) {
    if (ia.ctl) {
        h = (float)ia.ctl[(long)p * NLBAC_DOPRI_CTL + C_HUSED];
        x = (float)ia.ctl[(long)p * NLBAC_DOPRI_CTL + C_X];
    } else {
        h = ia.h[p]; x = ia.x[p];
    }
}

// y(t_end) = y0 + x(d + x(c + x(b + x a)))   (K: [7][n][n_s])
// OutMap (nlbac_out_map): a per-row map of the solve's output evaluated by the interpolation launches themselves —
// kind 1, the planar look-ahead point p = (x0 + l cos x2, x1 + l sin x2) of the Unicycle tasks (sac_cbf_clf.py:439-447:
// `next_p_x = x_next[0] + l_p cos(theta)`): forward writes p next to x(t_end), backward takes d loss / d p (two
// addends) instead of d loss / d x(t_end).  Same arithmetic as nlbac_unicycle_lookahead / _lookahead_bwd.
struct OutMap { int kind; float l; float* p; const float* dp; const float* dp2; const float* x; };

__global__ __launch_bounds__(256) void dopri_interp_fwd_kernel(const float* y0, const float* y1, const float* K,
                                                               const InterpArg ia, int rpp, int n_s, int n,
                                                               float* out, const OutMap om) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float h, x;
    interp_hx(ia, i / rpp, h, x);
    {
        const int slot = interp_slot(ia, i / rpp);
        y1 += (long)slot * ia.slot_floats;
        K += (long)slot * ia.slot_floats;
        if (slot > 0) y0 = y1 - ia.slot_floats;       // (y1 points at the slot's last stage input: the predecessor's is its y0)
    }
    for (int r = 0; r < n_s; ++r) {
        float k[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = K[((long)j * n + i) * n_s + r];
        out[(long)i * n_s + r] = dopri_interp_value(y0[(long)i * n_s + r], y1[(long)i * n_s + r], k, h, x);
    }
    if (om.kind == 1) {
        const float th = out[(long)i * n_s + 2];
        om.p[i * 2 + 0] = out[(long)i * n_s + 0] + om.l * cosf(th);
        om.p[i * 2 + 1] = out[(long)i * n_s + 1] + om.l * sinf(th);
    }
}

// backward of the interpolant: writes dy0, dy1 and dK[0..6]
__global__ __launch_bounds__(256) void dopri_interp_bwd_kernel(const float* dout, const InterpArg ia, int rpp,
                                                               int n_s, int n, float* dy0, float* dy1, float* dK,
                                                               const OutMap om) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float h, x;
    interp_hx(ia, i / rpp, h, x);
    {
        const long off = (long)interp_slot(ia, i / rpp) * ia.slot_floats;
        dy0 += off; dy1 += off; dK += off;
    }
    float gm[3] = {0.f, 0.f, 0.f};
    if (om.kind == 1) {
        float d0 = om.dp[i * 2 + 0], d1 = om.dp[i * 2 + 1];
        if (om.dp2) { d0 += om.dp2[i * 2 + 0]; d1 += om.dp2[i * 2 + 1]; }
        const float th = om.x[(long)i * n_s + 2];
        gm[0] = d0; gm[1] = d1; gm[2] = om.l * (-sinf(th) * d0 + cosf(th) * d1);
    }
    for (int r = 0; r < n_s; ++r) {
        const float g = (om.kind == 1) ? gm[r] : dout[(long)i * n_s + r];
        float d0v, d1v, dk[7];
        dopri_interp_grad(g, h, x, d0v, d1v, dk);
        dy0[(long)i * n_s + r] = d0v;
        dy1[(long)i * n_s + r] = d1v;
#pragma unroll
        for (int j = 0; j < 7; ++j) dK[((long)j * n + i) * n_s + r] = dk[j];
    }
}

// ---------------------------------------------------------------------------
#define GRID1(n) dim3(nlbac_ceil_div((n), 256)), dim3(256), 0, (hipStream_t)s

static int fill_rc(RkCoef& rc, const float* coef, int n_k, const float* h_host, int P, const char* who) {
    NLBAC_REQUIRE(n_k >= 0 && n_k <= MAX_STAGES, "%s: n_k %d out of range", who, n_k);
    NLBAC_REQUIRE(P >= 1 && P <= MAX_PROBLEMS, "%s: P %d out of range", who, P);
    memset(&rc, 0, sizeof(rc));
    for (int j = 0; j < n_k; ++j) rc.c[j] = coef[j];
    for (int p = 0; p < P; ++p) rc.h[p] = h_host ? h_host[p] : 1.f;
    return 0;
}

extern "C" int nlbac_affine_combine_fwd(const float* f, const float* g, const float* u, int n_s, int n_u, int n,
                                        float* k, nlbac_stream_t s) {
    NLBAC_REQUIRE(f && g && u && k, "nlbac_affine_combine_fwd: null pointer");
    NLBAC_REQUIRE(n_s <= MAX_NS && n_u <= MAX_NUA, "nlbac_affine_combine_fwd: dims too large");
    hipLaunchKernelGGL(affine_fwd_kernel, GRID1(n), f, g, u, n_s, n_u, n, k);
    NLBAC_CHECK_LAUNCH("nlbac_affine_combine_fwd");
    return 0;
}

extern "C" int nlbac_affine_combine_bwd(const float* dk, const float* g, const float* u, int n_s, int n_u, int n,
                                        float mul, float* dg, float* du, int accumulate_du, nlbac_stream_t s) {
    NLBAC_REQUIRE(dk && g && u, "nlbac_affine_combine_bwd: null pointer");
    hipLaunchKernelGGL(affine_bwd_kernel, GRID1(n), dk, g, u, n_s, n_u, n, mul, dg, du, accumulate_du);
    NLBAC_CHECK_LAUNCH("nlbac_affine_combine_bwd");
    return 0;
}

extern "C" int nlbac_rk_combine(const float* y0, const float* K, int n_k, const float* coef, const float* h_host,
                                const double* h_dev, int h_dev_stride, int P, int rows_per_problem, int n_s,
                                float* out, nlbac_stream_t s) {
    RkCoef rc;
    NLBAC_REQUIRE(K && out && coef, "nlbac_rk_combine: null pointer");
    if (fill_rc(rc, coef, n_k, h_host, P, "nlbac_rk_combine")) return -1;
    const int n = P * rows_per_problem;
    hipLaunchKernelGGL(rk_combine_kernel, GRID1(n), y0, K, n_k, rc, h_dev, h_dev_stride, rows_per_problem, n_s, n, out);
    NLBAC_CHECK_LAUNCH("nlbac_rk_combine");
    return 0;
}

extern "C" int nlbac_rk_stage_bwd(const float* dYup, const float* dXf, const float* dXg, int dx_ld, int n_k,
                                  const float* coef, const float* h_host, const double* h_dev, int h_dev_stride,
                                  int P, int rows_per_problem, int n_s, float* dK, float* dy0, int accumulate_dy0,
                                  float* du_ext, int n_ext, nlbac_stream_t s) {
    RkCoef rc;
    NLBAC_REQUIRE(n_k == 0 || (dK && coef), "nlbac_rk_stage_bwd: null pointer");
    if (fill_rc(rc, coef, n_k, h_host, P, "nlbac_rk_stage_bwd")) return -1;
    const int n = P * rows_per_problem;
    hipLaunchKernelGGL(rk_stage_bwd_kernel, GRID1(n), dYup, dXf, dXg, dx_ld, n_k, rc, h_dev, h_dev_stride,
                       rows_per_problem, n_s, n, dK, dy0, accumulate_dy0, du_ext, n_ext);
    NLBAC_CHECK_LAUNCH("nlbac_rk_stage_bwd");
    return 0;
}

extern "C" int nlbac_dopri_norm_partials(const float* a, const float* b, const float* y0, const float* y1,
                                         const float* u, int mode, float rtol, float atol, int n_s, int n_u,
                                         int rows_per_problem, int P, float* partials, const double* slot_ctl,
                                         long slot_floats, nlbac_stream_t s) {
    NLBAC_REQUIRE(a && y0 && partials && mode >= 0 && mode <= 2, "nlbac_dopri_norm_partials: bad arguments");
    NLBAC_REQUIRE((mode != 0 || u) && (mode != 1 || b) && (mode != 2 || y1), "nlbac_dopri_norm_partials: missing operand");
    NLBAC_REQUIRE(!slot_ctl || mode == 2, "nlbac_dopri_norm_partials: step slots apply to the error norm (mode 2)");
    hipLaunchKernelGGL(dopri_norm_kernel, dim3(nlbac_ceil_div(rows_per_problem, 256), P), dim3(256), 0,
                       (hipStream_t)s, a, b, y0, y1, u, mode, rtol, atol, n_s, n_u, rows_per_problem, partials,
                       slot_ctl, slot_floats);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_norm_partials");
    return 0;
}

extern "C" int nlbac_dopri_norm_control(const float* a, const float* b, const float* y0, const float* y1, const float* u,
                                        int mode, float rtol, float atol, int n_s, int n_u, int rows_per_problem, int P,
                                        double t_end, float* partials, unsigned* tickets, double* ctl,
                                        const nlbac_rk_chain* chain, nlbac_stream_t s) {
    NLBAC_REQUIRE(a && y0 && partials && tickets && ctl && mode >= 0 && mode <= 2,
                  "nlbac_dopri_norm_control: bad arguments");
    NLBAC_REQUIRE((mode != 0 || u) && (mode != 1 || b) && (mode != 2 || y1), "nlbac_dopri_norm_control: missing operand");
    NLBAC_REQUIRE(P >= 1 && P <= MAX_PROBLEMS, "nlbac_dopri_norm_control: P %d out of range", P);
    hipLaunchKernelGGL(dopri_norm_control_kernel, dim3(nlbac_ceil_div(rows_per_problem, 256), P), dim3(256), 0,
                       (hipStream_t)s, a, b, y0, y1, u, mode, rtol, atol, n_s, n_u, rows_per_problem, t_end, partials,
                       tickets, ctl, (chain && mode == 2) ? chain->ctl : nullptr, chain ? chain->slot_floats : 0,
                       chain ? chain->n_slots : 0, chain ? chain->hslots : nullptr, chain ? chain->alog : nullptr,
                       chain ? chain->alog_cap : 0, chain ? chain->ctl_host : nullptr, chain ? chain->ctl_seq : 0.0);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_norm_control");
    return 0;
}

extern "C" int nlbac_dopri_control(const float* partials, int n_blk_per_problem, int mode, int n_s, int n_u,
                                   int rows_per_problem, int P, double t_end, double* ctl, int n_slots, double* hslots,
                                   double* alog, int alog_cap, nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && ctl && mode >= 0 && mode <= 2, "nlbac_dopri_control: bad arguments");
    NLBAC_REQUIRE(!hslots || n_slots >= 1, "nlbac_dopri_control: hslots needs n_slots");
    hipLaunchKernelGGL(dopri_control_kernel, dim3(P), dim3(64), 0, (hipStream_t)s, partials, n_blk_per_problem, mode,
                       n_s, n_u, rows_per_problem, t_end, ctl, n_slots, hslots, alog, alog_cap);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_control");
    return 0;
}

extern "C" int nlbac_dopri_control_tiles(const nlbac_rk_chain* chain, int n_s, int n_u, int rows_per_problem, int P,
                                         nlbac_stream_t s) {
    NLBAC_REQUIRE(chain && chain->partials && chain->ctl_w && chain->norm_mode == 2 && chain->norm_defer,
                  "nlbac_dopri_control_tiles: needs the chain of an attempt launch with norm mode 2 and norm_defer");
    NLBAC_REQUIRE(P >= 1 && P <= MAX_PROBLEMS && (P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0),
                  "nlbac_dopri_control_tiles: bad problem sizes");
    NLBAC_REQUIRE(!chain->hslots || chain->n_slots >= 1, "nlbac_dopri_control_tiles: hslots needs n_slots");
    hipLaunchKernelGGL(dopri_control_tiles_kernel, dim3(P), dim3(64), 0, (hipStream_t)s, chain->partials,
                       nlbac_ceil_div(rows_per_problem, NLBAC_MLP_TILE), n_s, n_u, rows_per_problem, chain->t_end,
                       chain->ctl_w, chain->n_slots, chain->hslots, chain->alog, chain->alog_cap, chain->ctl_host,
                       chain->ctl_seq);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_control_tiles");
    return 0;
}

static int fill_ia(InterpArg& ia, const float* h_host, const float* x_host, const double* ctl, int P,
                   const char* who, long slot_floats = 0) {
    NLBAC_REQUIRE(((h_host && x_host) || ctl) && P >= 1 && P <= MAX_PROBLEMS, "%s: bad arguments", who);
    NLBAC_REQUIRE(slot_floats == 0 || ctl, "%s: step slots need the control block", who);
    memset(&ia, 0, sizeof(ia));
    ia.ctl = ctl;
    ia.slot_floats = slot_floats;
    if (!ctl)
        for (int p = 0; p < P; ++p) { ia.h[p] = h_host[p]; ia.x[p] = x_host[p]; }
    return 0;
}

extern "C" int nlbac_dopri_interp_fwd(const float* y0, const float* y1, const float* K, const float* h_host,
                                      const float* x_host, const double* ctl, int P, int rows_per_problem, int n_s,
                                      float* out, long slot_floats,
                                      const nlbac_out_map* map, nlbac_stream_t s) {
    InterpArg ia;
    NLBAC_REQUIRE(y0 && y1 && K && out, "nlbac_dopri_interp_fwd: null pointer");
    if (fill_ia(ia, h_host, x_host, ctl, P, "nlbac_dopri_interp_fwd", slot_floats)) return -1;
    const int n = P * rows_per_problem;
    OutMap om{0, 0.f, nullptr, nullptr, nullptr, nullptr};
    if (map && map->kind) {
        NLBAC_REQUIRE(map->kind == 1 && n_s == 3 && map->p, "nlbac_dopri_interp_fwd: out map 1 needs n_s == 3 and p");
        om.kind = 1; om.l = map->l; om.p = map->p;
    }
    hipLaunchKernelGGL(dopri_interp_fwd_kernel, GRID1(n), y0, y1, K, ia, rows_per_problem, n_s, n, out, om);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_interp_fwd");
    return 0;
}

extern "C" int nlbac_dopri_interp_bwd(const float* dout, const float* h_host, const float* x_host,
                                      const double* ctl, int P, int rows_per_problem, int n_s, float* dy0,
                                      float* dy1, float* dK, long slot_floats, const nlbac_out_map* map,
                                      nlbac_stream_t s) {
    InterpArg ia;
    NLBAC_REQUIRE((dout || (map && map->kind)) && dy0 && dy1 && dK, "nlbac_dopri_interp_bwd: null pointer");
    if (fill_ia(ia, h_host, x_host, ctl, P, "nlbac_dopri_interp_bwd", slot_floats)) return -1;
    const int n = P * rows_per_problem;
    OutMap om{0, 0.f, nullptr, nullptr, nullptr, nullptr};
    if (map && map->kind) {
        NLBAC_REQUIRE(map->kind == 1 && n_s == 3 && map->dp && map->x, "nlbac_dopri_interp_bwd: out map 1 needs n_s == 3, dp and x");
        om.kind = 1; om.l = map->l; om.dp = map->dp; om.dp2 = map->dp2; om.x = map->x;
    }
    hipLaunchKernelGGL(dopri_interp_bwd_kernel, GRID1(n), dout, ia, rows_per_problem, n_s, n, dy0, dy1, dK, om);
    NLBAC_CHECK_LAUNCH("nlbac_dopri_interp_bwd");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// The adjoint of the single-net NODE  dx/dt = out_mu + out_sig * net(([x | c] - in_mu) * in_isig)  (c: carried inputs;
// C/sac_cbf_clf/model.py:179-205; no normaliser: mu = 0, sig = 1), stage by stage on the MLP kernels: per stage of the
// augmented system z = [y | a_y | a_c] (W = 2 n_s + n_c floats per row, integrated in s = t1 - t)
//     nlbac_concat_adj_in   the net's input rows and the cotangent of its output from the stage point,
//     nlbac_mlp_fwd / nlbac_mlp_bwd_data  (f and the vector-Jacobian product (d net / d input)^T a),
//     nlbac_concat_adj_out  the stage derivative  dz/ds = [ -f | +J_y^T a_y | +J_c^T a_y ].
// (torchdiffeq 0.2.3 OdeintAdjointMethod's augmented dynamics on a field whose carried columns have zero derivative.)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_adj_in_kernel(const float* __restrict__ ZS, int W, const float* __restrict__ c,
                                                            int ns, int nc, const float* __restrict__ norm, int n,
                                                            float* __restrict__ Xin, float* __restrict__ Ay) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int idim = ns + nc;
    const float* z = ZS + (long)i * W;
    for (int k = 0; k < idim; ++k) {
        float v = (k < ns) ? z[k] : c[(long)i * nc + (k - ns)];
        if (norm) v = (v - norm[k]) * norm[idim + k];
        Xin[(long)i * idim + k] = v;
    }
    for (int r = 0; r < ns; ++r) Ay[(long)i * ns + r] = norm ? z[ns + r] * norm[2 * idim + ns + r] : z[ns + r];
}

__global__ __launch_bounds__(256) void concat_adj_out_kernel(const float* __restrict__ fnet, const float* __restrict__ dX,
                                                             int ns, int nc, const float* __restrict__ norm, int n, int W,
                                                             float* __restrict__ KZ) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int idim = ns + nc;
    float* k = KZ + (long)i * W;
    for (int r = 0; r < ns; ++r) {
        const float f = fnet[(long)i * ns + r];
        k[r] = -(norm ? norm[2 * idim + r] + norm[2 * idim + ns + r] * f : f);
    }
    for (int j = 0; j < idim; ++j) {
        const float d = dX[(long)i * idim + j];
        k[ns + j] = norm ? d * norm[idim + j] : d;
    }
}

extern "C" int nlbac_concat_adj_in(const float* ZS, int w, const float* c, int n_s, int n_c, const float* norm, int n,
                                   float* Xin, float* Ay, nlbac_stream_t s) {
    NLBAC_REQUIRE(ZS && c && Xin && Ay && n >= 1 && n_s >= 1 && n_c >= 0 && w >= 2 * n_s + n_c, "nlbac_concat_adj_in: bad arguments");
    hipLaunchKernelGGL(concat_adj_in_kernel, GRID1(n), ZS, w, c, n_s, n_c, norm, n, Xin, Ay);
    NLBAC_CHECK_LAUNCH("nlbac_concat_adj_in");
    return 0;
}

extern "C" int nlbac_concat_adj_out(const float* fnet, const float* dX, int n_s, int n_c, const float* norm, int n, int w,
                                    float* KZ, nlbac_stream_t s) {
    NLBAC_REQUIRE(fnet && dX && KZ && n >= 1 && w >= 2 * n_s + n_c, "nlbac_concat_adj_out: bad arguments");
    hipLaunchKernelGGL(concat_adj_out_kernel, GRID1(n), fnet, dX, n_s, n_c, norm, n, w, KZ);
    NLBAC_CHECK_LAUNCH("nlbac_concat_adj_out");
    return 0;
}
