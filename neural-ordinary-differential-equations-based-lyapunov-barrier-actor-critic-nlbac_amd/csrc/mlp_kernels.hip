// Fused ReLU-MLP kernels for gfx950 (MI355X): forward, backward-data and
// backward-weights, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact
// fp32 FMA chain, so results stay within fp32 rounding of the reference's
// ATen GEMMs).
//
// Replaces the op sequences of QNetwork / LyaNetwork / GaussianPolicy.forward
// (U/sac_cbf_clf/model.py:53-64, 77-83, 108-114), f_net / g_net of
// NeuralODEModel (model.py:186-206) and autograd's backward through them.
//
// Work decomposition (CDNA4: 64-lane waves, 4 SIMDs per CU, one MFMA pipe per
// SIMD):
//   * one workgroup = 4 waves = one tile of 32 samples of ONE net; grid.y
//     batches independent nets (twin Q, Lyapunov, policies, f/g nets) so a
//     4096-sample minibatch still fills the 256 CUs;
//   * activations of the tile ping-pong between two LDS buffers [32][hid+4]
//     (row stride = 4 mod 8 dwords -> conflict-free ds_read_b128 A fragments);
//   * weights are read as the MFMA B operand straight from an L2-resident,
//     fragment-ordered packed copy (nlbac_mlp_pack): one fully coalesced
//     1 KiB global_load_dwordx4 per wave per 8-deep K chunk, two chunks in
//     flight, no LDS round trip (each wave owns its own output columns);
//   * the skinny first-input / last-output contractions (K or N <= 16) stay
//     on the VALU.
#include <cstdlib>
#include "common.h"
#include "mlp_device.h"
#include "rr_device.h"
#include "dy_heads.h"
#include "mlp_launch.h"


// ---------------------------------------------------------------------------
// weight packing
//   forward pack of W[N][K]  : P[((tile*KC + kc)*64 + lane)*4 + j] = W[32*tile + (lane&31)][8*kc + 4*(lane>>5) + j]
//   backward pack            : same formula applied to W^T (tile over K, chunks over N)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_pack_kernel(const MlpLaunch L) {
    const nlbac_mlp& net = L.net[blockIdx.y];
    const int nwide = net.n_layers - 1;
    const int hid = net.hid;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (int l = 0; l < nwide; ++l) {
        const int K = (l == 0) ? net.in_dim : hid;
        const float* W = net.params + net.w_off[l];
        if (net.pf_off[l] >= 0) {   // forward pack: tiles over N=hid, chunks over K
            const int KC = pad8(K) >> 3, NT = pad32(hid) >> 5;
            float* P = net.packed + net.pf_off[l];
            const long total = (long)NT * KC * 256;
            for (long i = gid; i < total; i += stride) {
                const int j = i & 3, lane = (i >> 2) & 63;
                const long tc = i >> 8;
                const int kc = tc % KC, tile = tc / KC;
                const int n = 32 * tile + (lane & 31), k = 8 * kc + 4 * (lane >> 5) + j;
                P[i] = (n < hid && k < K) ? W[(long)n * K + k] : 0.f;
            }
        }
        if (l >= 1 && net.pb_off[l] >= 0) {   // backward pack: tiles over K(=hid), chunks over N(=hid)
            const int NC = pad8(hid) >> 3, KT = pad32(K) >> 5;
            float* P = net.packed + net.pb_off[l];
            const long total = (long)KT * NC * 256;
            for (long i = gid; i < total; i += stride) {
                const int j = i & 3, lane = (i >> 2) & 63;
                const long tc = i >> 8;
                const int nc = tc % NC, tile = tc / NC;
                const int k = 32 * tile + (lane & 31), n = 8 * nc + 4 * (lane >> 5) + j;
                P[i] = (n < hid && k < K) ? W[(long)n * K + k] : 0.f;
            }
        }
    }
    // RR packs (rr_device.h) of the hid x hid layers 1 .. nwide-1: float4 v of lane `lane` = the A-fragment values of
    // MFMAs 4v .. 4v+3 in the chains' issue order; forward W[unit out][unit in], backward its transpose
    if (net.rr_kind == RR_KIND_PANEL) {
        // the one hid x hid layer as two panels (RRPanel): panel ch holds output blocks ch*NBH .. ch*NBH + NBH-1 in groups
        // of two, k-steps inner; float4 v of lane `lane` = the A-fragment values of MFMAs 4v .. 4v+3
        const int NBH = hid >> 5, KS = hid >> 2;
        const float* W = net.params + net.w_off[1];
        float* Pf = net.packed + net.rr_fwd_off;
        float* Pb = net.packed + net.rr_bwd_off;
        const long total = (long)hid * hid, per_panel = total >> 1;
        for (long i = gid; i < total; i += stride) {
            const int ch = (int)(i / per_panel);
            const long ip = i - (long)ch * per_panel;
            const int c = ip & 3, lane = (ip >> 2) & 63, v = (int)(ip >> 8), m = 4 * v + c;
            const int g = m / (2 * KS), rem = m - g * 2 * KS, ks = rem >> 1, jj = rem & 1;
            const int uo = 16 * (NBH * ch + 2 * g + jj) + (lane & 15), ui = 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
            Pf[i] = W[(long)uo * hid + ui];
            Pb[i] = W[(long)ui * hid + uo];
        }
        // layer 0 as A fragments of [W_0 | b_0] over k-steps of [x | 1]: float4 ((k0 * NBA/4 + jo/4) * 64 + lane) holds
        // the values of output blocks jo .. jo+3 for k-step k0 (row = lane & 15 of the block, column 4 k0 + (lane >> 4))
        {
            const int NBA = hid >> 4, idim = net.in_dim;
            const float* W0 = net.params + net.w_off[0];
            const float* b0 = net.params + net.b_off[0];
            float* P0 = net.packed + net.rr_bwd_off + (long)hid * hid;
            const long n0 = rr_panel_l0_floats(hid);
            for (long i = gid; i < n0; i += stride) {
                const int c = i & 3, lane = (i >> 2) & 63;
                const long t = i >> 8;
                const int j4 = (int)(t % (NBA >> 2)), k0 = (int)(t / (NBA >> 2));
                const int uo = 16 * (4 * j4 + c) + (lane & 15), col = 4 * k0 + (lane >> 4);
                P0[i] = (col < idim) ? W0[(long)uo * idim + col] : (col == idim ? b0[uo] : 0.f);
            }
            // what the register-resident data backward (mlp_rr_bwd_kernel) reads, behind layer 0's fragments:
            // the last layer as A fragments of W_L^T over k-steps of dy — float4 ((k0 * NBA/4 + j4) * 64 + lane), component
            // c = W_L[4 k0 + (lane >> 4)][16 (4 j4 + c) + (lane & 15)] —
            const int nw = net.n_layers - 1, odim = net.out_dim;
            const float* WL = net.params + net.w_off[nw];
            float* PT = P0 + n0;
            for (long i = gid; i < n0; i += stride) {
                const int c = i & 3, lane = (i >> 2) & 63;
                const long t = i >> 8;
                const int j4 = (int)(t % (NBA >> 2)), k0 = (int)(t / (NBA >> 2));
                const int o = 4 * k0 + (lane >> 4), u = 16 * (4 * j4 + c) + (lane & 15);
                PT[i] = (o < odim) ? WL[(long)o * hid + u] : 0.f;
            }
            // and layer 0 as A fragments of W_0^T (dx = dz0 W_0) — float4 (jb * 64 + lane), component
            // r = W_0[16 jb + 4 (lane >> 4) + r][lane & 15]
            float* PX = PT + n0;
            for (long i = gid; i < n0; i += stride) {
                const int r = i & 3, lane = (i >> 2) & 63, jb = (int)(i >> 8);
                const int u = 16 * jb + 4 * (lane >> 4) + r, ii = lane & 15;
                PX[i] = (ii < idim) ? W0[(long)u * idim + ii] : 0.f;
            }
        }
    }
    if (net.rr_kind == RR_KIND_CHAIN) {
        const int NB = (hid + 15) >> 4, R = (hid - 16 * (NB - 1)) >> 2, KS = hid >> 2, NM = NB * KS;
        const long per_layer = rr_layer_floats(hid);
        for (int l = 1; l < nwide; ++l) {
            const float* W = net.params + net.w_off[l];
            float* Pf = net.packed + net.rr_fwd_off + (long)(l - 1) * per_layer;
            float* Pb = net.packed + net.rr_bwd_off + (long)(l - 1) * per_layer;
            for (long i = gid; i < per_layer; i += stride) {
                const int c = i & 3, lane = (i >> 2) & 63, v = (int)(i >> 8), m = 4 * v + c;
                float vf = 0.f, vb = 0.f;
                if (m < NM) {
                    int jo, ks;
                    rr_mfma_of(NB, KS, m, jo, ks);
                    const int uo = rr_unit_out(NB, R, jo, lane & 15), ui = rr_unit_in(NB, R, ks, lane >> 4);
                    if (uo >= 0) { vf = W[(long)uo * hid + ui]; vb = W[(long)ui * hid + uo]; }
                }
                Pf[i] = vf; Pb[i] = vb;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// MODE 1: every net has <= 4 column tiles (one per wave); MODE 2: every net has 8 (two per wave);
// MODE 0: mixed / other widths (both register sets live).
// MODE 3: every net has 8 column tiles and the workgroup has 8 waves, one tile each (same code path as MODE 1):
// half the per-tile latency of MODE 2 and twice the waves per CU.
// OCC (MODE 3): cap the kernel at 128 VGPRs so that two workgroups share a CU (a few spills): pays once the grid has
// more tiles than CUs (6 nets x 128 tiles: 45 -> 43 us), costs ~1 us on the one- and two-net launches.
template <int MODE, int OCC = 0>
__global__ __launch_bounds__(MODE == 3 ? 512 : 256, OCC ? 4 : 2) void mlp_fwd_kernel(const MlpLaunch L,
                                                                                        const nlbac_gauss_head G) {
    constexpr int NTHR = (MODE == 3) ? 512 : 256;
    constexpr bool ONE = (MODE == 1 || MODE == 3);      // one column tile per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B, LD = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const int hid = net.hid, NT = pad32(hid) >> 5;
    const int nwide = net.n_layers - 1;
    const int inp = pad8(net.in_dim);
    float* in = smem;
    float* out = smem + NLBAC_MLP_TILE * LD;

    const bool active = wave < NT, two = (MODE == 2) || (MODE == 0 && (wave + 4) < NT);
    WaveGemm<ONE ? 1 : 2> wg2;              // MODE 1/3 never touch wg2 / MODE 2 never touches wg1:
    WaveGemm<1> wg1;                        // the unused one is dead code
    if (active) {          // weights do not depend on the input: start streaming them before staging it
        if constexpr (!ONE) { if (two) fwd_prime<2>(wg2, net, inp, false, wave, lane); }
        if constexpr (MODE != 2) { if (!two) fwd_prime<1>(wg1, net, inp, false, wave, lane); }
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * inp; idx += NTHR) {
        const int r = idx / inp, c = idx - r * inp, row = row0 + r;
        float v = 0.f;
        if (row < B) {
            if (c < io.x0_dim) v = io.x0[(long)row * io.x0_ld + c];
            else if (c < net.in_dim) v = io.x1[(long)row * io.x1_ld + (c - io.x0_dim)];
        }
        in[r * LD + c] = v;
    }
    __syncthreads();

    {
        const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
        float* acts_tile = io.acts ? io.acts + (long)row0 * hid : nullptr;
        const int n_rows = min(NLBAC_MLP_TILE, B - row0);
        if constexpr (MODE == 2) fwd_wide_layers<2>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, ls, n_rows, nwide, false);
        else if constexpr (ONE) fwd_wide_layers<1>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, ls, n_rows, nwide, false);
        else {
            if (two) fwd_wide_layers<2>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, ls, n_rows, nwide, false);
            else fwd_wide_layers<1>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, ls, n_rows, nwide, false);
        }
    }

    // skinny output layer on the VALU: 8 lanes per sample row + shuffle reduction.  (With 1-4 outputs and
    // hid = 256 this keeps all 256 threads busy; the one-dot-per-thread form used by the fused RK kernel
    // measured 20 % slower here.)
    if (tid < 256) {       // (waves 4-7 of MODE 3 are done)
        const float* W = net.params + net.w_off[nwide];
        const float* bias = net.params + net.b_off[nwide];
        const int m = tid >> 3, part = tid & 7, row = row0 + m;
        for (int o0 = 0; o0 < net.out_dim; o0 += 4) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            const int no = min(4, net.out_dim - o0);
            const float* Wo = W + (long)o0 * hid;
            const float* hrow = in + m * LD;
            if (no == 1) skinny_dot<1>(hrow, Wo, hid, part, a);
            else if (no == 2) skinny_dot<2>(hrow, Wo, hid, part, a);
            else if (no == 3) skinny_dot<3>(hrow, Wo, hid, part, a);
            else skinny_dot<4>(hrow, Wo, hid, part, a);
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                a[0] += __shfl_xor(a[0], off, 64); a[1] += __shfl_xor(a[1], off, 64);
                a[2] += __shfl_xor(a[2], off, 64); a[3] += __shfl_xor(a[3], off, 64);
            }
            if (part == 0 && row < B) {
                float* y = io.y + (long)row * io.y_ld + o0;
                y[0] = a[0] + bias[o0];
                if (no > 1) y[1] = a[1] + bias[o0 + 1];
                if (no > 2) y[2] = a[2] + bias[o0 + 2];
                if (no > 3) y[3] = a[3] + bias[o0 + 3];
            }
        }
        // squashed-Gaussian head (nlbac_gauss_head): the thread that has just written a row's (mean | log_std) draws
        // the row's action and log-probability from it — gauss_fwd_kernel's arithmetic, no launch of its own
        if (G.eps && part == 0 && row < B)
            gauss_fwd_row(io.y + (long)row * io.y_ld, G.eps, G.scale, G.bias, G.n_u, (long)blockIdx.y * B + row, G.action,
                          G.action_ld, G.logp);
    }
}

// ---------------------------------------------------------------------------
// backward, data path:  dz[j] for every wide layer j and (optionally) dx
//   dz[nwide-1] = (dy W_last) * [acts[nwide-1] > 0]
//   dz[j-1]     = (dz[j] W_j) * [acts[j-1] > 0]            (MFMA, backward pack)
//   dx          =  dz[0] W_0                                (VALU)
// ---------------------------------------------------------------------------
template <int NI>
__device__ __forceinline__ void dx_dot(const float* __restrict__ lds_row, const float* __restrict__ W, int hid,
                                       int idim, int part, float (&acc)[4]) {
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int n = part * 4 + 32 * it;
        const bool ok = n < hid;
        const int nn = ok ? n : 0;
        float4 d = *reinterpret_cast<const float4*>(lds_row + nn);
        if (!ok) d = make_float4(0, 0, 0, 0);
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] += dd[q] * W[(long)(nn + q) * idim + i];
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void mlp_bwd_data_kernel(const MlpLaunch L, const nlbac_dy_head H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B, LD = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const int hid = net.hid, NT = pad32(hid) >> 5, hidp32 = NT * 32;
    const int nwide = net.n_layers - 1;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    float* in = smem;
    float* out = smem + NLBAC_MLP_TILE * LD;
    float* sdy = smem + 2 * NLBAC_MLP_TILE * LD;   // [32][16]
    float* sx = sdy + NLBAC_MLP_TILE * 16;         // [32][16] input rows (only for the skinny-gradient partials)
    // skinny-gradient partials of this 32-row tile (= one chunk of mlp_bwd_skinny_partial_kernel, same sums in the same
    // order): thread = hidden column, quantity q at w[q * 256]
    // (one set per NLBAC_SK_CHUNK = 16 rows: this tile leaves two; w = the first, the second wstep floats behind it)
    const bool sk = io.skinny_ws != nullptr && io.dz != nullptr;
    const long wstep = (long)((net.n_layers - 1) + net.in_dim + net.out_dim + 1) * 256;
    float* w = sk ? io.skinny_ws + (long)blockIdx.x * (NLBAC_MLP_TILE / NLBAC_SK_CHUNK) * wstep + tid : nullptr;

    const bool active = wave < NT, two = (MODE == 2) || (MODE == 0 && (wave + 4) < NT);
    WaveGemm<(MODE == 1) ? 1 : 2> wg2;      // MODE 1 never touches wg2 / MODE 2 never touches wg1:
    WaveGemm<1> wg1;                        // the unused one is dead code
    if (active) {
        if constexpr (MODE != 1) { if (two) bwd_prime<2>(wg2, net, wave, lane); }
        if constexpr (MODE != 2) { if (!two) bwd_prime<1>(wg1, net, wave, lane); }
    }
    // (the tile's dy and input rows: every global load issued before the first LDS store — clamped addresses, selects
    // afterwards; as guarded loads in two-iteration loops they were up to six dependent round trips at the head of every
    // tile's latency chain)
    // (kind 3: nets behind the Q pairs read io.dy — but for the first of them when the head evaluates the constraint backward: cb_kind)
    const bool plain_dy = H.kind == 0 || (H.kind == 3 && (int)blockIdx.y >= 2 * H.n_prob && !(H.cb_kind && (int)blockIdx.y == 2 * H.n_prob));
    {
        float vdy[2] = {0.f, 0.f}, vx0[2] = {0.f, 0.f}, vx1[2] = {0.f, 0.f};
        const bool has_x1 = io.x1 != nullptr && io.x1_dim > 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, r = idx >> 4, c = idx & 15;
            const long row = min(row0 + r, B - 1);
            if (plain_dy) vdy[it] = io.dy[row * io.dy_ld + min(c, net.out_dim - 1)];
            if (sk) {
                vx0[it] = io.x0[row * io.x0_ld + min(c, io.x0_dim - 1)];
                if (has_x1) vx1[it] = io.x1[row * io.x1_ld + min(max(c - io.x0_dim, 0), io.x1_dim - 1)];
            }
        }
        if (!plain_dy) dy_head_fill(H, blockIdx.y, row0, B, gridDim.x, sdy, sx, gridDim.y);     // (dy heads: dL/dy is produced here, dy_heads.h)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, r = idx >> 4, c = idx & 15, row = row0 + r;
            if (plain_dy) sdy[idx] = (row < B && c < net.out_dim) ? vdy[it] : 0.f;
            if (sk) sx[idx] = (row < B && c < net.in_dim) ? (c < io.x0_dim ? vx0[it] : vx1[it]) : 0.f;
        }
    }

    {   // top (skinny) layer: thread = hidden column; ReLU masks are fetched up front (clamped, unconditional)
        const int k = tid, kc = min(k, hid - 1);
        const float* acts = io.acts + (long)(nwide - 1) * ls;
        float av[NLBAC_MLP_TILE];
#pragma unroll
        for (int m = 0; m < NLBAC_MLP_TILE; ++m) av[m] = acts[(long)min(row0 + m, B - 1) * hid + kc];
        __syncthreads();
        const float* W = net.params + net.w_off[nwide] + kc;
        float s[NLBAC_MLP_TILE];
#pragma unroll
        for (int m = 0; m < NLBAC_MLP_TILE; ++m) s[m] = 0.f;
        for (int o0 = 0; o0 < net.out_dim; o0 += 4) {
            const int no = min(4, net.out_dim - o0);
            const float* Wo = W + (long)o0 * hid;
            if (no == 1) top_layer_bwd<1>(sdy + o0, Wo, hid, 0, s);
            else if (no == 2) top_layer_bwd<2>(sdy + o0, Wo, hid, 0, s);
            else if (no == 3) top_layer_bwd<3>(sdy + o0, Wo, hid, 0, s);
            else top_layer_bwd<4>(sdy + o0, Wo, hid, 0, s);
        }
        if (k < hidp32) {
            const bool colok = k < hid;
#pragma unroll
            for (int m = 0; m < NLBAC_MLP_TILE; ++m) {
                const bool ok = colok && (row0 + m < B);
                in[m * LD + k] = (ok && av[m] > 0.f) ? s[m] : 0.f;
            }
        }
        if (sk) {      // last layer: dW_L[o][k] = sum_m dy[m][o] a_L[m][k] (av holds the activations), db_L[o] = sum_m dy[m][o]
            const int idim = net.in_dim, odim = net.out_dim;
            const bool live = k < hid;
#pragma unroll
            for (int hh = 0; hh < NLBAC_MLP_TILE / NLBAC_SK_CHUNK; ++hh) {
                const int m0 = hh * NLBAC_SK_CHUNK;
                for (int o = 0; o < odim; ++o) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < NLBAC_SK_CHUNK; ++m) a = __builtin_fmaf(sdy[(m0 + m) * 16 + o], av[m0 + m], a);   // (fused, as
                                                                   // mlp_bwd_skinny_partial_kernel's contracted multiply-adds)
                    w[hh * wstep + (long)(nwide + idim + o) * 256] = live ? a : 0.f;
                }
                float bl = 0.f;
                if (k < 16)
                    for (int m = 0; m < NLBAC_SK_CHUNK; ++m) bl += sdy[(m0 + m) * 16 + k];
                w[hh * wstep + (long)(nwide + idim + odim) * 256] = bl;
            }
        }
    }
    __syncthreads();
    if (sk) {          // bias gradient of the top hidden layer: column sum of the finished dz tile
#pragma unroll
        for (int hh = 0; hh < NLBAC_MLP_TILE / NLBAC_SK_CHUNK; ++hh) {
            float a = 0.f;
            if (tid < hid)
                for (int m = 0; m < NLBAC_SK_CHUNK; ++m) a += in[(hh * NLBAC_SK_CHUNK + m) * LD + tid];
            w[hh * wstep + (long)(nwide - 1) * 256] = a;
        }
    }
    if (io.dz)       // the top layer's dz leaves from the finished LDS tile (coalesced, overlaps the first GEMM)
        tile_to_global(in, LD, io.dz + (long)(nwide - 1) * ls + (long)row0 * hid, hid, min(NLBAC_MLP_TILE, B - row0), tid, 256);

    {
        const int n_rows = min(NLBAC_MLP_TILE, B - row0);
        const float* acts_tile = io.acts + (long)row0 * hid;
        float* dz_tile = io.dz ? io.dz + (long)row0 * hid : nullptr;
        if constexpr (MODE == 2) bwd_wide_layers<2>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, ls, n_rows, n_rows - 1, -1, false, 256, nullptr, w, wstep);
        else if constexpr (MODE == 1) bwd_wide_layers<1>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, ls, n_rows, n_rows - 1, -1, false, 256, nullptr, w, wstep);
        else {
            if (two) bwd_wide_layers<2>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, ls, n_rows, n_rows - 1, -1, false, 256, nullptr, w, wstep);
            else bwd_wide_layers<1>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, ls, n_rows, n_rows - 1, -1, false, 256, nullptr, w, wstep);
        }
    }

    if (sk) {      // first layer: dW_0[k][i] = sum_m dz0[m][k] x[m][i]
        const int idim = net.in_dim;
        const bool live = tid < hid;
#pragma unroll
        for (int hh = 0; hh < NLBAC_MLP_TILE / NLBAC_SK_CHUNK; ++hh) {
            const int m0 = hh * NLBAC_SK_CHUNK;
            for (int i = 0; i < idim; ++i) {
                float a = 0.f;
                if (live)
                    for (int m = 0; m < NLBAC_SK_CHUNK; ++m) a = __builtin_fmaf(in[(m0 + m) * LD + tid], sx[(m0 + m) * 16 + i], a);
                w[hh * wstep + (long)(nwide + i) * 256] = a;
            }
        }
    }
    if (io.dx) {   // dx[m][i] = sum_n dz0[m][n] W0[n][i]
        const float* W = net.params + net.w_off[0];
        const int idim = net.in_dim;
        const int m = tid >> 3, part = tid & 7, row = row0 + m;
        for (int i0 = io.dx_first; i0 < idim; i0 += 4) {      // (columns below dx_first are not wanted: nlbac_mlp_io)
            const int ni = min(4, idim - i0);
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            const float* drow = in + m * LD;
            if (ni == 1) dx_dot<1>(drow, W + i0, hid, idim, part, a);
            else if (ni == 2) dx_dot<2>(drow, W + i0, hid, idim, part, a);
            else if (ni == 3) dx_dot<3>(drow, W + i0, hid, idim, part, a);
            else dx_dot<4>(drow, W + i0, hid, idim, part, a);
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                a[0] += __shfl_xor(a[0], off, 64); a[1] += __shfl_xor(a[1], off, 64);
                a[2] += __shfl_xor(a[2], off, 64); a[3] += __shfl_xor(a[3], off, 64);
            }
            if (part == 0 && row < B) {
                float* dx = io.dx + (long)row * io.dx_ld + i0;
                dx[0] = a[0];
                if (ni > 1) dx[1] = a[1];
                if (ni > 2) dx[2] = a[2];
                if (ni > 3) dx[3] = a[3];
            }
        }
    }
    // ---- batch sums an earlier launch's head left to this one (nlbac_dy_head::finish)
    dy_head_jobs(H, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, sx);
}

// ---------------------------------------------------------------------------
// backward, weights of the hidden->hidden layers:
//   dW_j[n][k] = sum_b dz[j][b][n] * acts[j-1][b][k]      (j = 1..nwide-1)
// grid.x = (layer, 64x64 output tile), grid.y = row slab, grid.z = net.
// Both operands are staged through LDS in 16-row chunks; each of the 4 waves
// owns one 32x32 MFMA tile.
// ---------------------------------------------------------------------------
#define DW_CHUNK 32
#define DW_LDS_LD 68
__device__ __forceinline__ float4 sel4(bool c, const float4 v) {   // component selects stay in registers
    return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}
struct SkinnyLaunch {
    int rows_per_chunk, n_chunks;
    long net_stride;      // floats between nets in ws
};

__device__ __forceinline__ int skinny_nq(const nlbac_mlp& net) {
    return (net.n_layers - 1) + net.in_dim + net.out_dim + 1;
}

__device__ __forceinline__ void skinny_reduce_block(const MlpLaunch& L, const SkinnyLaunch& S, const float* __restrict__ ws,
                                                    int blk, int inet, float (*part)[64]);

// red_planes > 0: the first red_planes z-planes of the grid are not GEMM tiles but the blocks of the skinny-gradient
// reduction (mlp_bwd_skinny_reduce_kernel's work: independent of the GEMM, both only read what the data backward left):
// they are dispatched first and finish under the GEMM tiles instead of in a launch of their own.  Block index within the
// planes -> (net, reduce block) with red_per_net blocks per net.
__global__ __launch_bounds__(256) void mlp_bwd_wide_kernel(const MlpLaunch L, const SkinnyLaunch S,
                                                           const float* __restrict__ ws, int red_planes, int red_per_net,
                                                           int n_nets) {
    __shared__ __attribute__((aligned(16))) float sA[2][DW_CHUNK][DW_LDS_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][DW_CHUNK][DW_LDS_LD];
    if ((int)blockIdx.z < red_planes) {
        const int idx = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int inet = idx / red_per_net;
        if (inet < n_nets)
            skinny_reduce_block(L, S, ws, idx - inet * red_per_net, inet, reinterpret_cast<float (*)[64]>(&sA[0][0][0]));
        return;
    }
    const int znet = (int)blockIdx.z - red_planes;
    const nlbac_mlp& net = L.net[znet];
    const nlbac_mlp_io& io = L.io[znet];
    const int B = L.B, hid = net.hid, nwide = net.n_layers - 1;
    const int T = (hid + 63) >> 6;
    const int per_layer = T * T;
    if ((int)blockIdx.x >= (nwide - 1) * per_layer) return;
    const int j = 1 + blockIdx.x / per_layer;
    const int tt = blockIdx.x % per_layer;
    const int n0 = (tt / T) * 64, k0 = (tt % T) * 64;
    const int slab = blockIdx.y;
    const int rb = slab * L.rows_per_slab, re = min(B, rb + L.rows_per_slab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    const float* dz = io.dz + (long)j * ls;
    const float* at = io.acts + (long)(j - 1) * ls;
    const int lr = tid >> 4, lc = (tid & 15) * 4;        // this thread stages rows lr and lr+16 of a chunk
    const bool a_ok = n0 + lc < hid, b_ok = k0 + lc < hid;
    const float* pa = dz + n0 + (a_ok ? lc : 0);
    const float* pb = at + k0 + (b_ok ? lc : 0);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // operand rows are fetched two chunks ahead (registers) and written to LDS one chunk ahead, so a
    // load has ~2 chunks (2 x 16 MFMAs per wave) to land; out-of-range rows/columns are clamped then zeroed
#define DW_LOAD(r0_, A0, A1, B0, B1)                                                        \
    {                                                                                       \
        const int ra_ = min((r0_) + lr, re - 1), rb_ = min((r0_) + lr + 16, re - 1);        \
        A0 = *reinterpret_cast<const float4*>(pa + (long)ra_ * hid);                        \
        A1 = *reinterpret_cast<const float4*>(pa + (long)rb_ * hid);                        \
        B0 = *reinterpret_cast<const float4*>(pb + (long)ra_ * hid);                        \
        B1 = *reinterpret_cast<const float4*>(pb + (long)rb_ * hid);                        \
    }
#define DW_STAGE(r0_, A0, A1, B0, B1, buf_)                                                 \
    {                                                                                       \
        const bool lo_ = (r0_) + lr < re, hi_ = (r0_) + lr + 16 < re;                       \
        *reinterpret_cast<float4*>(&sA[buf_][lr][lc]) = sel4(a_ok && lo_, A0);              \
        *reinterpret_cast<float4*>(&sA[buf_][lr + 16][lc]) = sel4(a_ok && hi_, A1);         \
        *reinterpret_cast<float4*>(&sB[buf_][lr][lc]) = sel4(b_ok && lo_, B0);              \
        *reinterpret_cast<float4*>(&sB[buf_][lr + 16][lc]) = sel4(b_ok && hi_, B1);         \
    }
    // Register double buffering without copies: the loop is unrolled by two and the two operand register sets swap
    // roles (X: loaded last step, staged this step; Y: in flight) - a "c = n" copy at the end of each step would make
    // the compiler wait for the in-flight loads there and serialise the stream.
#define DW_MFMA(buf_)                                                                       \
    {                                                                                       \
        float av[DW_CHUNK / 2], bv[DW_CHUNK / 2];                                           \
        _Pragma("unroll") for (int bb = 0; bb < DW_CHUNK / 2; ++bb) {                       \
            av[bb] = sA[buf_][2 * bb + half][wn * 32 + (lane & 31)];                        \
            bv[bb] = sB[buf_][2 * bb + half][wk * 32 + (lane & 31)];                        \
        }                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        _Pragma("unroll") for (int bb = 0; bb < DW_CHUNK / 2; ++bb)                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bb], bv[bb], acc, 0, 0, 0);       \
    }
    if (rb < re) {
        float4 x_a0, x_a1, x_b0, x_b1, y_a0, y_a1, y_b0, y_b1;
        DW_LOAD(rb, x_a0, x_a1, x_b0, x_b1)
        DW_STAGE(rb, x_a0, x_a1, x_b0, x_b1, 0)
        DW_LOAD(rb + DW_CHUNK, x_a0, x_a1, x_b0, x_b1)
        __syncthreads();
        for (int r0 = rb; r0 < re; r0 += 2 * DW_CHUNK) {
            if (r0 + 2 * DW_CHUNK < re) DW_LOAD(r0 + 2 * DW_CHUNK, y_a0, y_a1, y_b0, y_b1)
            __builtin_amdgcn_sched_barrier(0);     // the loads are issued HERE (the scheduler would sink them to their use)
            DW_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            if (r0 + DW_CHUNK < re) DW_STAGE(r0 + DW_CHUNK, x_a0, x_a1, x_b0, x_b1, 1)
            __syncthreads();
            if (r0 + DW_CHUNK >= re) break;
            if (r0 + 3 * DW_CHUNK < re) DW_LOAD(r0 + 3 * DW_CHUNK, x_a0, x_a1, x_b0, x_b1)
            __builtin_amdgcn_sched_barrier(0);
            DW_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
            if (r0 + 2 * DW_CHUNK < re) DW_STAGE(r0 + 2 * DW_CHUNK, y_a0, y_a1, y_b0, y_b1, 0)
            __syncthreads();
        }
    }
    float* g = io.grad + (long)slab * L.slab_stride + net.w_off[j];
    const int k = k0 + wk * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + acc_row(r, half);
        if (n < hid && k < hid) g[(long)n * hid + k] = acc[r];
    }
}

// ---------------------------------------------------------------------------
// The same gradients for nets wider than 128 (the 256-wide heads), LDS-free: a workgroup still owns one 64 x 64 tile of
// one layer's dW_j for one row slab, but its four waves split the slab's ROWS instead of the tile, each accumulating the
// whole tile (4 x 4 accumulators of v_mfma_f32_16x16x4_f32) over its own k-steps of 4 rows, and a lane's two dwordx4
// loads (dz columns 4i..4i+3 and activation columns 4i..4i+3 of its row) ARE its operands for the k-step's 16 MFMAs —
// the column order of mlp_dw16_kernels.hip (tile t of a strip = the columns {4 i + t}: the gradient does not care which
// 16 columns are called a tile, the store at the end undoes it).  No staging, no barrier in the loop, DWT_PF k-steps in
// flight; the waves' partial tiles meet in LDS at the end in a fixed order.  (mlp_bwd_wide_kernel: 16-row chunks through
// LDS with two barriers each, 23 us per launch at 3 x 4096 rows for 11 us of matrix-pipe time.)
//
// blockIdx -> tile is XCD-aware: the tiles of one (slab, net) group read the same 2 * rows_per_slab * hid floats — four
// times each over the group — and consecutive block ids are dealt round-robin to the 8 XCDs, so a group's tiles take ids
// that are congruent mod 8: they meet in ONE XCD's L2 and the slab leaves HBM / the Infinity Cache once (the old mapping:
// 72.6 MB fetched per launch for 25 MB of operands).  Placement is a speed matter only.
// The first red_blocks blocks are the skinny-gradient reduction (mlp_bwd_skinny_reduce_kernel's work), as above.
// ---------------------------------------------------------------------------
#define DWT_PF 8
__global__ __launch_bounds__(256, 3) void mlp_bwd_wide64_kernel(const MlpLaunch L, const SkinnyLaunch S,
                                                                const float* __restrict__ ws, int red_blocks,
                                                                int red_per_net, int n_nets, int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) float dwt_smem[];      // 2 x [64 values][64 lanes]
    if ((int)blockIdx.x < red_blocks) {
        const int inet = blockIdx.x / red_per_net;
        if (inet < n_nets)
            skinny_reduce_block(L, S, ws, blockIdx.x - inet * red_per_net, inet, reinterpret_cast<float (*)[64]>(dwt_smem));
        return;
    }
    const int id = (int)blockIdx.x - red_blocks, xcd = id & 7, jj = id >> 3;
    const int t = jj % tiles_per_group, grp = 8 * (jj / tiles_per_group) + xcd;
    if (grp >= L.n_slabs * n_nets) return;
    const int znet = grp / L.n_slabs, slab = grp - znet * L.n_slabs;
    const nlbac_mlp& net = L.net[znet];
    const nlbac_mlp_io& io = L.io[znet];
    const int B = L.B, hid = net.hid, nwide = net.n_layers - 1;
    const int T = (hid + 63) >> 6, per_layer = T * T;
    if (t >= (nwide - 1) * per_layer) return;
    const int j = 1 + t / per_layer, tt = t % per_layer;
    const int n0 = (tt / T) * 64, k0 = (tt % T) * 64;
    const int rb = slab * L.rows_per_slab, re = min(B, rb + L.rows_per_slab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    // rows past the slab's end and columns past the width read as zeros (raw buffer loads out of range)
    const __amdgpu_buffer_rsrc_t rsA = rr_rsrc(io.dz + (long)j * ls, max(re, 0) * hid);
    const __amdgpu_buffer_rsrc_t rsB = rr_rsrc(io.acts + (long)(j - 1) * ls, max(re, 0) * hid);
    const int oob = (int)0x80000000;
    const int voA = (n0 + 4 * i + 3 < hid) ? (q * hid + n0 + 4 * i) * 4 : oob;
    const int voB = (k0 + 4 * i + 3 < hid) ? (q * hid + k0 + 4 * i) * 4 : oob;
    const int kstep_bytes = 16 * hid;                                   // 4 rows
    const int g_first = (rb >> 2) + wave, n_mine = (rb < re) ? (((re - rb + 3) >> 2) - wave + 3) / 4 : 0;     // k-steps g_first + 4 m
    f32x4 acc[4][4];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 va[DWT_PF], vb[DWT_PF];
    const int rounds = (n_mine + DWT_PF - 1) / DWT_PF;
    if (rounds > 0) {
#pragma unroll
        for (int u = 0; u < DWT_PF; ++u) {
            const int so = (g_first + 4 * u) * kstep_bytes;
            va[u] = rr_ldw(rsA, voA, so);
            vb[u] = rr_ldw(rsB, voB, so);
        }
        int g = g_first + 4 * DWT_PF;
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int u = 0; u < DWT_PF; ++u) {
                const f32x4 a = va[u], b = vb[u];
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
                const int so = g * kstep_bytes;           // (the set just used is refilled DWT_PF k-steps ahead)
                va[u] = rr_ldw(rsA, voA, so);
                vb[u] = rr_ldw(rsB, voB, so);
                __builtin_amdgcn_sched_barrier(0);
                g += 4;
            }
        }
    }
    // the four waves' partial tiles: (w0 + w1) + (w2 + w3), through LDS
    float* const red0 = dwt_smem, * const red1 = dwt_smem + 64 * 64;
    auto put = [&](float* red) __attribute__((always_inline)) {
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
                *reinterpret_cast<f32x4*>(red + ((ta * 4 + tb) * 64 + lane) * 4) = acc[ta][tb];
    };
    auto add = [&](const float* red) __attribute__((always_inline)) {
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[ta][tb] += *reinterpret_cast<const f32x4*>(red + ((ta * 4 + tb) * 64 + lane) * 4);
    };
    if (wave == 1) put(red0);
    if (wave == 3) put(red1);
    lds_barrier();
    if (wave == 0) add(red0);
    if (wave == 2) add(red1);
    lds_barrier();
    if (wave == 2) put(red0);
    lds_barrier();
    if (wave != 0) return;
    add(red0);
    // D[m][n] of a 16 x 16 x 4 MFMA: lane (q, i) register r holds m = 4 q + r (a dz lane column), n = i (an activation lane
    // column): dW[n0 + 4 m + ta][k0 + 4 i + tb] — the four tb of a lane are four consecutive floats of a row
    float* gW = io.grad + (long)slab * L.slab_stride + net.w_off[j];
    if (k0 + 4 * i + 3 < hid) {
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * (4 * q + r) + ta;
                if (n < hid)
                    *reinterpret_cast<f32x4*>(gW + (long)n * hid + k0 + 4 * i) = f32x4{acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]};
            }
    }
}

// Narrow nets (hid <= 128, the NODE fit): one workgroup owns the whole (padded 128x128) dW_j of a layer for its row
// slab, so dz and acts are read from HBM once per layer instead of once per 64-column tile pair (the fit streams
// ~1.3 GB of them per RK step - the one HBM-bound place of the update).  Each wave accumulates a 64x64 quadrant
// (2x2 MFMA tiles): four MFMAs per pair of LDS operand reads.  grid.x = layer, grid.y = row slab, grid.z = net.
#define DW128_LD 132
__global__ __launch_bounds__(256) void mlp_bwd_wide128_kernel(const MlpLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float dw_smem[];
    float (*sA)[DW_CHUNK][DW128_LD] = reinterpret_cast<float (*)[DW_CHUNK][DW128_LD]>(dw_smem);
    float (*sB)[DW_CHUNK][DW128_LD] = reinterpret_cast<float (*)[DW_CHUNK][DW128_LD]>(dw_smem + 2 * DW_CHUNK * DW128_LD);
    const nlbac_mlp& net = L.net[blockIdx.z];
    const nlbac_mlp_io& io = L.io[blockIdx.z];
    const int B = L.B, hid = net.hid, nwide = net.n_layers - 1;
    if ((int)blockIdx.x >= nwide - 1) return;
    const int j = 1 + blockIdx.x;
    const int slab = blockIdx.y;
    const int rb = slab * L.rows_per_slab, re = min(B, rb + L.rows_per_slab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    const float* dz = io.dz + (long)j * ls;
    const float* at = io.acts + (long)(j - 1) * ls;
    const int lr = tid >> 5, lc = (tid & 31) * 4;        // this thread stages rows lr, lr+8, lr+16, lr+24 of a chunk
    const bool c_ok = lc < hid;
    const float* pa = dz + (c_ok ? lc : 0);
    const float* pb = at + (c_ok ? lc : 0);

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

#define DW128_LOAD(r0_, A, Bv)                                                              \
    {                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                     \
            const int rr_ = min((r0_) + lr + 8 * q, re - 1);                                \
            A[q] = *reinterpret_cast<const float4*>(pa + (long)rr_ * hid);                  \
            Bv[q] = *reinterpret_cast<const float4*>(pb + (long)rr_ * hid);                 \
        }                                                                                   \
    }
#define DW128_STAGE(r0_, A, Bv, buf_)                                                       \
    {                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                     \
            const bool ok_ = c_ok && ((r0_) + lr + 8 * q < re);                             \
            *reinterpret_cast<float4*>(&sA[buf_][lr + 8 * q][lc]) = sel4(ok_, A[q]);        \
            *reinterpret_cast<float4*>(&sB[buf_][lr + 8 * q][lc]) = sel4(ok_, Bv[q]);       \
        }                                                                                   \
    }
#define DW128_MFMA(buf_)                                                                    \
    {                                                                                       \
        float a0[DW_CHUNK / 2], a1[DW_CHUNK / 2], b0[DW_CHUNK / 2], b1[DW_CHUNK / 2];       \
        _Pragma("unroll") for (int bb = 0; bb < DW_CHUNK / 2; ++bb) {                       \
            const float* ra = &sA[buf_][2 * bb + half][wn * 64 + (lane & 31)];              \
            const float* rbp = &sB[buf_][2 * bb + half][wk * 64 + (lane & 31)];             \
            a0[bb] = ra[0]; a1[bb] = ra[32]; b0[bb] = rbp[0]; b1[bb] = rbp[32];             \
        }                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        _Pragma("unroll") for (int bb = 0; bb < DW_CHUNK / 2; ++bb) {                       \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[bb], b0[bb], acc[0], 0, 0, 0); \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[bb], b1[bb], acc[1], 0, 0, 0); \
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[bb], b0[bb], acc[2], 0, 0, 0); \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[bb], b1[bb], acc[3], 0, 0, 0); \
        }                                                                                   \
    }
    if (rb < re) {      // (two register sets swapping roles, loop unrolled by two: see mlp_bwd_wide_kernel)
        float4 xa[4], xb[4], ya[4], yb[4];
        DW128_LOAD(rb, xa, xb)
        DW128_STAGE(rb, xa, xb, 0)
        DW128_LOAD(rb + DW_CHUNK, xa, xb)
        __syncthreads();
        for (int r0 = rb; r0 < re; r0 += 2 * DW_CHUNK) {
            if (r0 + 2 * DW_CHUNK < re) DW128_LOAD(r0 + 2 * DW_CHUNK, ya, yb)
            __builtin_amdgcn_sched_barrier(0);     // the loads are issued HERE (the scheduler would sink them to their use)
            DW128_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            if (r0 + DW_CHUNK < re) DW128_STAGE(r0 + DW_CHUNK, xa, xb, 1)
            __syncthreads();
            if (r0 + DW_CHUNK >= re) break;
            if (r0 + 3 * DW_CHUNK < re) DW128_LOAD(r0 + 3 * DW_CHUNK, xa, xb)
            __builtin_amdgcn_sched_barrier(0);
            DW128_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
            if (r0 + 2 * DW_CHUNK < re) DW128_STAGE(r0 + 2 * DW_CHUNK, ya, yb, 0)
            __syncthreads();
        }
    }
    float* g = io.grad + (long)slab * L.slab_stride + net.w_off[j];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = wk * 64 + (q & 1) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wn * 64 + (q >> 1) * 32 + acc_row(r, half);
            if (n < hid && k < hid) g[n * hid + k] = acc[q][r];
        }
    }
}

// ---------------------------------------------------------------------------
// backward, skinny weights and all biases (column reductions over the rows):
//   db_j[n]      = sum_b dz[j][b][n]
//   dW_0[n][i]   = sum_b dz[0][b][n] x[b][i]
//   dW_L[o][k]   = sum_b dy[b][o] acts[nwide-1][b][k],  db_L[o] = sum_b dy[b][o]
// Stage 1: grid = (row chunk, net), thread = hidden column; a barrier-free
// streaming loop over the chunk's rows (coalesced 1 KiB row reads of dz/acts,
// wave-uniform x/dy operands) writes one partial row-set per chunk:
//   ws[net][chunk][q][256],  q = [db_0..db_{nwide-1} | dW_0[:,i] | dW_L[o,:] | db_L]
// Stage 2: one thread per output element sums the chunks in order
// (deterministic) and scatters into grad slab 0.
// ---------------------------------------------------------------------------
#define SK_MAX_IN 16
#define SK_MAX_OUT 16
#define SK_MAX_Q (NLBAC_MAX_LAYERS + SK_MAX_IN + SK_MAX_OUT + 1)

#define SK_ROWS_LDS 128
// The row loop of one chunk for a net with NW wide layers: thread = hidden column.  Eight rows are in flight per
// thread (8 x (NW + 1) independent loads: the loop is latency bound), none of them redundant.
template <int NW>
__device__ __forceinline__ void skinny_rows(const float* __restrict__ aL, const float* __restrict__ dz0, long ls, int hid,
                                            int r0, int nr, const float (*sx)[SK_MAX_IN], const float (*sdy)[SK_MAX_OUT],
                                            int col, float (&dW0)[SK_MAX_IN], float (&dWL)[SK_MAX_OUT],
                                            float (&db)[NLBAC_MAX_LAYERS - 1], float& dbL) {
    constexpr int U = 8;
    for (int rr = 0; rr < nr; rr += U) {
        float a[U], z[U][NW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long ro = (long)(r0 + min(rr + u, nr - 1)) * hid;       // clamped: every load unconditional
            a[u] = aL[ro];
#pragma unroll
            for (int j = 0; j < NW; ++j) z[u][j] = dz0[(long)j * ls + ro];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr + u < nr) {                                            // (uniform)
                const int r = rr + u;
#pragma unroll
                for (int j = 0; j < NW; ++j) db[j] += z[u][j];
#pragma unroll
                for (int i = 0; i < SK_MAX_IN; i += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(&sx[r][i]);
                    dW0[i] = __builtin_fmaf(z[u][0], xv.x, dW0[i]);         dW0[i + 1] = __builtin_fmaf(z[u][0], xv.y, dW0[i + 1]);
                    dW0[i + 2] = __builtin_fmaf(z[u][0], xv.z, dW0[i + 2]); dW0[i + 3] = __builtin_fmaf(z[u][0], xv.w, dW0[i + 3]);
                }
#pragma unroll
                for (int o = 0; o < SK_MAX_OUT; o += 4) {
                    const float4 dv = *reinterpret_cast<const float4*>(&sdy[r][o]);
                    dWL[o] = __builtin_fmaf(dv.x, a[u], dWL[o]);         dWL[o + 1] = __builtin_fmaf(dv.y, a[u], dWL[o + 1]);
                    dWL[o + 2] = __builtin_fmaf(dv.z, a[u], dWL[o + 2]); dWL[o + 3] = __builtin_fmaf(dv.w, a[u], dWL[o + 3]);
                }
                if (col < SK_MAX_OUT) dbL += sdy[r][col];
            }
        }
    }
}

// NTHR: threads per block = columns covered; nets of <= 128 hidden units run 128-thread blocks (with 256, half the
// waves of every block would hold no column and still take the wave slots the kernel's register budget allows)
template <int NTHR>
__global__ __launch_bounds__(NTHR) void mlp_bwd_skinny_partial_kernel(const MlpLaunch L, const SkinnyLaunch S,
                                                                      float* __restrict__ ws) {
    // x / dy rows of the current 128-row block, zero-padded to fixed widths so the
    // inner loop has no data-dependent branches (every load is unconditional and
    // can be issued ahead; a guarded load would serialise on s_waitcnt per element)
    __shared__ __attribute__((aligned(16))) float sx[SK_ROWS_LDS][SK_MAX_IN];
    __shared__ __attribute__((aligned(16))) float sdy[SK_ROWS_LDS][SK_MAX_OUT];
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B, hid = net.hid, nwide = net.n_layers - 1;
    const int idim = net.in_dim, odim = net.out_dim;
    const int chunk = blockIdx.x;
    const int rb = chunk * S.rows_per_chunk, re = min(B, rb + S.rows_per_chunk);
    const int col = threadIdx.x;
    const bool live = col < hid;
    const int c = live ? col : 0;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    float dW0[SK_MAX_IN], dWL[SK_MAX_OUT], db[NLBAC_MAX_LAYERS - 1];
#pragma unroll
    for (int i = 0; i < SK_MAX_IN; ++i) dW0[i] = 0.f;
#pragma unroll
    for (int o = 0; o < SK_MAX_OUT; ++o) dWL[o] = 0.f;
#pragma unroll
    for (int j = 0; j < NLBAC_MAX_LAYERS - 1; ++j) db[j] = 0.f;
    float dbL = 0.f;
    const float* aL = io.acts + (long)(nwide - 1) * ls + c;
    const float* dz0 = io.dz + c;
    const int x0d = io.x0_dim;

    for (int r0 = rb; r0 < re; r0 += SK_ROWS_LDS) {
        const int nr = min(SK_ROWS_LDS, re - r0);
        __syncthreads();
        for (int idx = col; idx < nr * SK_MAX_IN; idx += NTHR) {
            const int r = idx / SK_MAX_IN, i = idx - r * SK_MAX_IN;
            float v = 0.f;
            if (i < idim)
                v = (i < x0d) ? io.x0[(long)(r0 + r) * io.x0_ld + i] : io.x1[(long)(r0 + r) * io.x1_ld + (i - x0d)];
            sx[r][i] = v;
        }
        for (int idx = col; idx < nr * SK_MAX_OUT; idx += NTHR) {
            const int r = idx / SK_MAX_OUT, o = idx - r * SK_MAX_OUT;
            sdy[r][o] = (o < odim) ? io.dy[(long)(r0 + r) * io.dy_ld + o] : 0.f;
        }
        __syncthreads();
        switch (nwide) {       // (uniform per block)
            case 1: skinny_rows<1>(aL, dz0, ls, hid, r0, nr, sx, sdy, col, dW0, dWL, db, dbL); break;
            case 2: skinny_rows<2>(aL, dz0, ls, hid, r0, nr, sx, sdy, col, dW0, dWL, db, dbL); break;
            case 3: skinny_rows<3>(aL, dz0, ls, hid, r0, nr, sx, sdy, col, dW0, dWL, db, dbL); break;
            case 4: skinny_rows<4>(aL, dz0, ls, hid, r0, nr, sx, sdy, col, dW0, dWL, db, dbL); break;
            default: skinny_rows<5>(aL, dz0, ls, hid, r0, nr, sx, sdy, col, dW0, dWL, db, dbL);
        }
    }

    float* w = ws + (long)blockIdx.y * S.net_stride + (long)chunk * skinny_nq(net) * 256 + col;
    int q = 0;
#pragma unroll
    for (int j = 0; j < NLBAC_MAX_LAYERS - 1; ++j)
        if (j < nwide) { w[(long)q * 256] = live ? db[j] : 0.f; ++q; }
#pragma unroll
    for (int i = 0; i < SK_MAX_IN; ++i)
        if (i < idim) { w[(long)q * 256] = live ? dW0[i] : 0.f; ++q; }
#pragma unroll
    for (int o = 0; o < SK_MAX_OUT; ++o)
        if (o < odim) { w[(long)q * 256] = live ? dWL[o] : 0.f; ++q; }
    w[(long)q * 256] = dbL;
}

// block `blk` = (quantity q, 64-column quarter) of net `inet`; thread = (chunk group cg, column): the four chunk groups
// of a block sum interleaved quarters of the chunk list, combined through LDS (`part`, 4 x 64 floats) in a fixed order
__device__ __forceinline__ void skinny_reduce_block(const MlpLaunch& L, const SkinnyLaunch& S, const float* __restrict__ ws,
                                                    int blk, int inet, float (*part)[64]) {
    const nlbac_mlp& net = L.net[inet];
    const nlbac_mlp_io& io = L.io[inet];
    const int hid = net.hid, nwide = net.n_layers - 1, idim = net.in_dim, odim = net.out_dim;
    const int nq = skinny_nq(net);
    const int q = blk >> 2, col = (blk & 3) * 64 + (threadIdx.x & 63), cg = threadIdx.x >> 6;
    if (q >= nq) return;
    const float* w = ws + (long)inet * S.net_stride + (long)q * 256 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int ch = cg;
    for (; ch + 12 < S.n_chunks; ch += 16) {       // four fixed interleaved chains per chunk group
        s0 += w[(long)(ch + 0) * nq * 256];
        s1 += w[(long)(ch + 4) * nq * 256];
        s2 += w[(long)(ch + 8) * nq * 256];
        s3 += w[(long)(ch + 12) * nq * 256];
    }
    for (; ch < S.n_chunks; ch += 4) s0 += w[(long)ch * nq * 256];
    part[cg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (cg != 0) return;
    const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    // slab 0 takes the sum; the entry's copies in the other slabs are ZEROED, so that whatever wrote them before — the
    // one-launch kernel of the narrow nets leaves a partial of every gradient in every slab — the slab sum is this call's
    long e = -1;
    if (q < nwide) { if (col < hid) e = net.b_off[q] + col; }
    else if (q < nwide + idim) { if (col < hid) e = net.w_off[0] + (long)col * idim + (q - nwide); }
    else if (q < nwide + idim + odim) { if (col < hid) e = net.w_off[nwide] + (long)(q - nwide - idim) * hid + col; }
    else if (col < odim) e = net.b_off[nwide] + col;
    if (e < 0) return;
    float* g = io.grad + e;
    g[0] = v;
    for (int sl = 1; sl < L.n_slabs; ++sl) g[(long)sl * L.slab_stride] = 0.f;
}

__global__ __launch_bounds__(256) void mlp_bwd_skinny_reduce_kernel(const MlpLaunch L, const SkinnyLaunch S,
                                                                    const float* __restrict__ ws) {
    __shared__ float part[4][64];
    skinny_reduce_block(L, S, ws, blockIdx.x, blockIdx.y, part);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int check_net(const nlbac_mlp& n, const char* who) {
    NLBAC_REQUIRE(n.n_layers >= 2 && n.n_layers <= NLBAC_MAX_LAYERS, "%s: n_layers %d out of [2,%d]", who, n.n_layers, NLBAC_MAX_LAYERS);
    NLBAC_REQUIRE(n.in_dim >= 1 && n.in_dim <= SK_MAX_IN, "%s: in_dim %d out of [1,%d]", who, n.in_dim, SK_MAX_IN);
    NLBAC_REQUIRE(n.out_dim >= 1 && n.out_dim <= SK_MAX_OUT, "%s: out_dim %d out of [1,%d]", who, n.out_dim, SK_MAX_OUT);
    NLBAC_REQUIRE(n.hid >= 4 && n.hid <= 256 && n.hid % 4 == 0, "%s: hid %d must be a multiple of 4 in [4,256]", who, n.hid);
    NLBAC_REQUIRE(n.params && n.packed, "%s: params/packed must be set", who);
    return 0;
}

// 1: all nets <= 4 column tiles, 2: all nets exactly 8, 0: anything else
static int tile_mode(const nlbac_mlp* nets, int n_nets) {
    bool all_le4 = true, all_8 = true;
    for (int i = 0; i < n_nets; ++i) {
        const int nt = (nets[i].hid + 31) >> 5;
        all_le4 = all_le4 && nt <= 4;
        all_8 = all_8 && nt == 8;
    }
    return all_le4 ? 1 : (all_8 ? 2 : 0);
}

// 256-wide nets: 8 waves x one column tile (default) or 4 waves x two tiles (NLBAC_MLP_WAVES8=0, kept for A/B runs)
static bool waves8() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_WAVES8"); return !(e && e[0] == '0'); }();
    return on;
}

// the LDS-free weight gradients of nets wider than 128 (mlp_bwd_wide64_kernel); NLBAC_MLP_DW64=0 keeps mlp_bwd_wide_kernel
static bool dw64_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_DW64"); return !(e && e[0] == '0'); }();
    return on;
}

// two-workgroups-per-CU forward variant: -1 = by grid size (default), 0 / 1 = forced (NLBAC_MLP_OCC, for A/B runs)
static int occ_mode() {
    static const int m = [] { const char* e = getenv("NLBAC_MLP_OCC"); return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1; }();
    return m;
}

static int fill_launch(MlpLaunch& L, const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B, const char* who) {
    NLBAC_REQUIRE(n_nets >= 1 && n_nets <= NLBAC_MAX_NETS, "%s: n_nets %d out of [1,%d]", who, n_nets, NLBAC_MAX_NETS);
    NLBAC_REQUIRE(B >= 1, "%s: B must be >= 1", who);
    memset(&L, 0, sizeof(L));
    int ld = 0;
    for (int i = 0; i < n_nets; ++i) {
        if (check_net(nets[i], who)) return -1;
        L.net[i] = nets[i];
        if (io) L.io[i] = io[i];
        int w = ((nets[i].hid + 31) & ~31);
        int ip = (nets[i].in_dim + 7) & ~7;
        if (ip > w) w = ip;
        if (w + 4 > ld) ld = w + 4;
    }
    L.B = B;
    L.ld = ld;
    return 0;
}

extern "C" int nlbac_mlp_pack_layout(nlbac_mlp* net) {
    NLBAC_REQUIRE(net, "nlbac_mlp_pack_layout: null net");
    const int nwide = net->n_layers - 1, hid = net->hid;
    long off = 0;
    for (int l = 0; l < NLBAC_MAX_LAYERS; ++l) { net->pf_off[l] = -1; net->pb_off[l] = -1; }
    // (nets whose every launch runs on the register-resident kernels get no 32x32x2 packs: pf_off / pb_off stay -1)
    const bool tile_packs = !nlbac_mlp_rr_serves_shape(net->n_layers, net->in_dim, hid, net->out_dim);
    for (int l = 0; l < nwide && tile_packs; ++l) {
        const int K = (l == 0) ? net->in_dim : hid;
        net->pf_off[l] = (int)off;
        off += (long)(((hid + 31) & ~31) >> 5) * (((K + 7) & ~7) >> 3) * 256;
        if (l >= 1) {
            net->pb_off[l] = (int)off;
            off += (long)(((K + 31) & ~31) >> 5) * (((hid + 7) & ~7) >> 3) * 256;
        }
    }
    net->rr_fwd_off = net->rr_bwd_off = -1;
    net->rr_kind = rr_kind_of(net->n_layers, hid);
    if (net->rr_kind == RR_KIND_CHAIN) {
        net->rr_fwd_off = (int)off;
        off += (long)(nwide - 1) * rr_layer_floats(hid);
        net->rr_bwd_off = (int)off;
        off += (long)(nwide - 1) * rr_layer_floats(hid);
    } else if (net->rr_kind == RR_KIND_PANEL) {
        net->rr_fwd_off = (int)off;
        off += (long)hid * hid;
        net->rr_bwd_off = (int)off;
        off += (long)hid * hid;
        off += 3L * rr_panel_l0_floats(hid);          // behind the panels: layer 0's A fragments (with its bias column), the
                                                      // last layer's and layer 0's transposed fragments (data backward)
    }
    net->packed_floats = (int)off;
    return (int)off;
}

extern "C" int nlbac_mlp_pack(const nlbac_mlp* nets, int n_nets, nlbac_stream_t s) {
    MlpLaunch L;
    if (fill_launch(L, nets, nullptr, n_nets, 1, "nlbac_mlp_pack")) return -1;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(64, n_nets), dim3(256), 0, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_mlp_pack");
    return 0;
}

static int mlp_fwd_launch(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B, const nlbac_gauss_head& G,
                          const char* who, nlbac_stream_t s) {
    MlpLaunch L;
    if (fill_launch(L, nets, io, n_nets, B, who)) return -1;
    for (int i = 0; i < n_nets; ++i) {
        NLBAC_REQUIRE(io[i].x0 && io[i].y, "%s: net %d needs x0 and y", who, i);
        NLBAC_REQUIRE(io[i].x0_dim == nets[i].in_dim || (io[i].x1 && io[i].x0_dim + io[i].x1_dim == nets[i].in_dim),
                      "%s: net %d input dims %d+%d != in_dim %d", who, i, io[i].x0_dim, io[i].x1_dim, nets[i].in_dim);
    }
    {   // nets with one hid x hid layer of 64 / 128 / 256 units run on the register-resident kernels (mlp_rr_kernels.hip)
        const int rr = nlbac_mlp_rr_fwd_launch(L, n_nets, G, who, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(!io[i].masks, "%s: net %d: ReLU mask words are written by the register-resident kernels only "
                      "(nlbac_mlp_masks_ok)", who, i);
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(nets[i].pf_off[0] >= 0, "%s: net %d has no 32x32x2 packs (its launches run on the register-resident kernels: "
                      "all nets of a launch must then have its width)", who, i);
    const size_t lds = (size_t)2 * NLBAC_MLP_TILE * L.ld * sizeof(float);
    const dim3 grid(nlbac_ceil_div(B, NLBAC_MLP_TILE), n_nets);
    switch (tile_mode(nets, n_nets)) {
        case 1: hipLaunchKernelGGL(mlp_fwd_kernel<1>, grid, dim3(256), lds, (hipStream_t)s, L, G); break;
        case 2:
            if (waves8() && lds <= 80 * 1024 &&
                (occ_mode() == 1 || (occ_mode() < 0 && (long)grid.x * grid.y > 256)))
                hipLaunchKernelGGL((mlp_fwd_kernel<3, 1>), grid, dim3(512), lds, (hipStream_t)s, L, G);
            else if (waves8()) hipLaunchKernelGGL(mlp_fwd_kernel<3>, grid, dim3(512), lds, (hipStream_t)s, L, G);
            else hipLaunchKernelGGL(mlp_fwd_kernel<2>, grid, dim3(256), lds, (hipStream_t)s, L, G);
            break;
        default: hipLaunchKernelGGL(mlp_fwd_kernel<0>, grid, dim3(256), lds, (hipStream_t)s, L, G);
    }
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int nlbac_mlp_fwd(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B, nlbac_stream_t s) {
    nlbac_gauss_head G;
    memset(&G, 0, sizeof(G));
    return mlp_fwd_launch(nets, io, n_nets, B, G, "nlbac_mlp_fwd", s);
}

extern "C" int nlbac_mlp_fwd_gauss(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B,
                                   const nlbac_gauss_head* head, nlbac_stream_t s) {
    const char* who = "nlbac_mlp_fwd_gauss";
    NLBAC_REQUIRE(head && head->eps && head->scale && head->bias && head->action && head->logp && head->n_u >= 1 &&
                      head->n_u <= MAX_NU, "%s: bad head", who);
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(nets[i].out_dim == 2 * head->n_u && io[i].y_ld >= 2 * head->n_u, "%s: net %d must have 2 n_u outputs", who, i);
    return mlp_fwd_launch(nets, io, n_nets, B, *head, who, s);
}

extern "C" int nlbac_mlp_fwd_head_ok(const nlbac_mlp* nets, int n_nets) {
    return (nets && n_nets >= 1 && nlbac_mlp_rrq_eligible(nets, n_nets)) ? 1 : 0;
}

extern "C" int nlbac_mlp_fwd_head(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B,
                                  const nlbac_gauss_head* head, nlbac_stream_t s) {
    const char* who = "nlbac_mlp_fwd_head";
    NLBAC_REQUIRE(head && head->cf_kind == 1 && !head->eps, "%s: a constraint head (cf_kind 1) without a Gaussian sample", who);
    const nlbac_gauss_head& G = *head;
    NLBAC_REQUIRE(nlbac_mlp_fwd_head_ok(nets, n_nets), "%s: these nets' forward does not evaluate constraint heads (nlbac_mlp_fwd_head_ok)", who);
    NLBAC_REQUIRE(G.cf_net >= 0 && G.cf_net < n_nets && nets[G.cf_net].out_dim == 1 && G.cf_nh == 7,
                  "%s: cf_net must be a scalar net of the launch, cf_nh == 7", who);
    NLBAC_REQUIRE(!G.cf_defer || G.cf_tiles, "%s: cf_defer needs cf_tiles", who);
    NLBAC_REQUIRE(G.cf_ps && G.cf_ps_next && G.cf_V && G.cf_hazards && G.cf_matr && G.cf_bmatr && G.cf_partials && (G.cf_tickets || G.cf_defer) &&
                      G.cf_sc && G.cf_dt > 0.f && G.cf_batch_size > 0.f, "%s: constraint head: null pointer / bad scalars", who);
    NLBAC_REQUIRE(G.cf_n_cbf == G.cf_nh && G.cf_n_clf == 1 && G.cf_backup_mode >= 1 && G.cf_backup_mode <= 2,
                  "%s: constraint head 1 publishes 2 n_hz + 1 columns (n_cbf == n_hz, one CLF term, a backup controller)", who);
    return mlp_fwd_launch(nets, io, n_nets, B, G, who, s);
}

static int mlp_bwd_data_launch(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B, const nlbac_dy_head& H,
                               const char* who, nlbac_stream_t s) {
    MlpLaunch L;
    if (fill_launch(L, nets, io, n_nets, B, who)) return -1;
    for (int i = 0; i < n_nets; ++i) {
        NLBAC_REQUIRE((io[i].dy || H.kind) && (io[i].acts || io[i].masks), "%s: net %d needs dy and acts (or masks)", who, i);
        NLBAC_REQUIRE(!io[i].skinny_ws || io[i].acts, "%s: net %d: skinny-gradient partials need acts", who, i);
    }
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(!io[i].dx || (io[i].dx_first >= 0 && io[i].dx_first < nets[i].in_dim),
                      "%s: net %d: dx_first %d out of [0, in_dim %d)", who, i, io[i].dx_first, nets[i].in_dim);
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(!io[i].skinny_ws || (B <= 32768 && io[i].x0 && io[i].dz && nets[i].hid <= 256),
                      "%s: net %d: skinny-gradient partials need x0, dz, hid <= 256 and B <= 32768", who, i);
    {   // nets with one hid x hid layer of 64 / 128 / 256 units run on the register-resident kernel (mlp_rr_kernels.hip)
        const int rr = nlbac_mlp_rr_bwd_launch(L, n_nets, H, who, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(io[i].acts, "%s: net %d: the LDS-tiled kernel gates with the saved activations (acts), not mask words", who, i);
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(nets[i].pf_off[0] >= 0, "%s: net %d has no 32x32x2 packs (its launches run on the register-resident kernels: "
                      "all nets of a launch must then have its width)", who, i);
    const size_t lds = ((size_t)2 * NLBAC_MLP_TILE * L.ld + 2 * NLBAC_MLP_TILE * 16) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(B, NLBAC_MLP_TILE), n_nets);
    switch (tile_mode(nets, n_nets)) {
        case 1: hipLaunchKernelGGL(mlp_bwd_data_kernel<1>, grid, dim3(256), lds, (hipStream_t)s, L, H); break;
        case 2: hipLaunchKernelGGL(mlp_bwd_data_kernel<2>, grid, dim3(256), lds, (hipStream_t)s, L, H); break;
        default: hipLaunchKernelGGL(mlp_bwd_data_kernel<0>, grid, dim3(256), lds, (hipStream_t)s, L, H);
    }
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int nlbac_mlp_bwd_data(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B, nlbac_stream_t s) {
    nlbac_dy_head H;
    memset(&H, 0, sizeof(H));
    return mlp_bwd_data_launch(nets, io, n_nets, B, H, "nlbac_mlp_bwd_data", s);
}

extern "C" int nlbac_mlp_bwd_data_head(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B,
                                       const nlbac_dy_head* head, nlbac_stream_t s) {
    const char* who = "nlbac_mlp_bwd_data_head";
    NLBAC_REQUIRE(head && head->kind >= 1 && head->kind <= 3 && head->B_norm >= 1, "%s: bad head", who);
    const nlbac_dy_head& H = *head;
    if (H.kind == 1) {
        NLBAC_REQUIRE(H.heads && H.eps && H.scale && H.alpha && H.n_u >= 1 && H.n_u <= MAX_NU, "%s: gauss head: null pointer / bad n_u", who);
        for (int i = 0; i < n_nets; ++i) NLBAC_REQUIRE(nets[i].out_dim == 2 * H.n_u, "%s: gauss head: net %d must have 2 n_u outputs", who, i);
    } else if (H.kind == 2) {
        NLBAC_REQUIRE((n_nets == 3 || n_nets == 4) && H.q1t && H.q2t && H.lt && H.nlogp && H.reward && H.constraint && H.mask &&
                          H.alpha && H.q[0] && H.q[1] && H.q[2] && H.dq[0] && H.dq[1] && H.dq[2] && H.partials && H.ticket && H.out,
                      "%s: td head: 3 nets (Q1, Q2, Lyapunov) or 4 (+ BarrierNet) and all pointers", who);
        NLBAC_REQUIRE(n_nets == 3 || (H.xt && H.xsig && H.xq && H.dxq && H.out_x), "%s: td head: the 4th net's pointers", who);
        for (int i = 0; i < n_nets; ++i) NLBAC_REQUIRE(nets[i].out_dim == 1, "%s: td head: scalar nets", who);
    } else {
        NLBAC_REQUIRE(H.n_prob >= 1 && H.n_prob <= 2 && n_nets >= 2 * H.n_prob && H.qa && H.qb && H.logp && H.alpha && H.dqa &&
                          H.dqb && H.partials && H.ticket && H.actor.sc,
                      "%s: actor head: 2 nets per controller (further nets take io.dy) and all pointers", who);
        for (int p = 0; p < H.n_prob; ++p)
            NLBAC_REQUIRE(H.actor.log_alpha[p] && H.actor.g_log_alpha[p], "%s: actor head: missing log_alpha pointers", who);
        for (int i = 0; i < 2 * H.n_prob; ++i) NLBAC_REQUIRE(nets[i].out_dim == 1, "%s: actor head: scalar nets", who);
        for (int i = 2 * H.n_prob + (H.cb_kind ? 1 : 0); i < n_nets; ++i) NLBAC_REQUIRE(io[i].dy, "%s: net %d needs dy", who, i);
        if (H.cb_kind) {
            NLBAC_REQUIRE(H.cb_kind == 1 && n_nets > 2 * H.n_prob && nets[2 * H.n_prob].out_dim == 1 && H.cb_nh >= 1 && H.cb_nh <= 16 &&
                              H.cb_ps_next && H.cb_matr && H.cb_bmatr && H.cb_hazards && H.cb_sc && H.cb_dps_next && H.cb_dV &&
                              H.cb_dt > 0.f && H.cb_batch > 0.f,
                          "%s: actor head: the constraint backward (cb_kind 1) needs a scalar net behind the Q pairs and all of its pointers", who);
        }
    }
    NLBAC_REQUIRE(!H.sums_defer || ((H.kind == 2 || H.kind == 3) && H.sums_tiles), "%s: sums_defer goes with kind 2 / 3 and sums_tiles", who);
    NLBAC_REQUIRE(!H.cb_defer || (H.cb_kind == 1 && H.cb_nh == 7 && H.cb_partials && H.cb_tiles && H.cb_stage && H.cb_auglag.n_cbf == 7 &&
                                  H.cb_auglag.n_clf == 1 && H.cb_auglag.backup_mode != 0 && H.cb_auglag.batch_size > 0.f),
                  "%s: cb_defer goes with cb_kind 1 (7 hazards, CLF term, backup controller) and needs cb_partials / cb_tiles / cb_auglag", who);
    for (int j = 0; j < 3; ++j) {
        const nlbac_head_sums& J = H.finish[j];
        if (!J.kind) continue;
        NLBAC_REQUIRE(J.kind >= 2 && J.kind <= 4 && J.partials && (J.n_tiles || J.kind == 4), "%s: finish[%d]: kind 2 / 3 / 4 with partials (and n_tiles)", who, j);
        NLBAC_REQUIRE(J.kind != 4 || J.sc, "%s: finish[%d]: the commit job needs sc", who, j);
        NLBAC_REQUIRE(J.kind != 2 || ((J.n_nets == 3 || J.n_nets == 4) && J.out && (J.n_nets == 3 || J.out_x)), "%s: finish[%d]: td sums need out (and out_x with 4 nets)", who, j);
        NLBAC_REQUIRE(J.kind != 3 || (J.n_nets >= 1 && J.n_nets <= 2 && J.B_norm >= 1 && J.actor.sc), "%s: finish[%d]: actor sums need n_prob, B_norm and the scalars block", who, j);
        if (J.kind == 3)
            for (int p = 0; p < J.n_nets; ++p)
                NLBAC_REQUIRE(J.actor.log_alpha[p] && J.actor.g_log_alpha[p], "%s: finish[%d]: missing log_alpha pointers", who, j);
    }
    return mlp_bwd_data_launch(nets, io, n_nets, B, H, who, s);
}

// rows per partial-sum chunk of the skinny-gradient reduction: short chunks so that >= 1 workgroup per CU streams
// rows concurrently (the loop is latency bound), capped at 2048 chunks; up to B = 32768 the chunk is the data backward's
// finest tile (NLBAC_SK_CHUNK rows), whose partial sums nlbac_mlp_bwd_data can leave itself
static inline int skinny_rows_per_chunk(int B) { return (B + 2047) / 2048 > NLBAC_SK_CHUNK ? (B + 2047) / 2048 : NLBAC_SK_CHUNK; }

extern "C" long nlbac_mlp_bwd_weights_ws_floats(const nlbac_mlp* nets, int n_nets, int B) {
    long per_net = 0;
    const int rpc = skinny_rows_per_chunk(B);
    int n_chunks = (B + rpc - 1) / rpc;
    if (rpc == NLBAC_SK_CHUNK) n_chunks = (n_chunks + 1) & ~1;      // (the 32-row kernels write two chunks per tile, the last one
                                                                    //  of a ragged batch possibly past ceil(B / 16): zeros)
    for (int i = 0; i < n_nets; ++i) {
        const long nq = (nets[i].n_layers - 1) + nets[i].in_dim + nets[i].out_dim + 1;
        if (nq * 256 * n_chunks > per_net) per_net = nq * 256 * n_chunks;
    }
    return per_net * n_nets;
}

extern "C" int nlbac_mlp_bwd_weights(const nlbac_mlp* nets, const nlbac_mlp_io* io, int n_nets, int B,
                                     int n_slabs, long slab_stride, float* ws, long ws_floats, nlbac_stream_t s) {
    MlpLaunch L;
    if (fill_launch(L, nets, io, n_nets, B, "nlbac_mlp_bwd_weights")) return -1;
    NLBAC_REQUIRE(n_slabs >= 1, "nlbac_mlp_bwd_weights: n_slabs must be >= 1");
    int max_blocks = 0, max_q = 0;
    for (int i = 0; i < n_nets; ++i) {
        NLBAC_REQUIRE(io[i].dy && io[i].acts && io[i].dz && io[i].grad && io[i].x0,
                      "nlbac_mlp_bwd_weights: net %d needs x0, dy, acts, dz, grad", i);
        const int T = (nets[i].hid + 63) >> 6;
        const int nb = (nets[i].n_layers - 2) * T * T;
        if (nb > max_blocks) max_blocks = nb;
        const int nq = (nets[i].n_layers - 1) + nets[i].in_dim + nets[i].out_dim + 1;
        if (nq > max_q) max_q = nq;
    }
    const long need = nlbac_mlp_bwd_weights_ws_floats(nets, n_nets, B);
    NLBAC_REQUIRE(ws && ws_floats >= need, "nlbac_mlp_bwd_weights: workspace too small (%ld < %ld floats)", ws_floats, need);
    int rps = nlbac_ceil_div(B, n_slabs);
    rps = (rps + DW_CHUNK - 1) / DW_CHUNK * DW_CHUNK;
    L.n_slabs = n_slabs; L.rows_per_slab = rps; L.slab_stride = slab_stride;
    SkinnyLaunch S;
    S.rows_per_chunk = skinny_rows_per_chunk(B);
    S.n_chunks = (B + S.rows_per_chunk - 1) / S.rows_per_chunk;
    S.net_stride = need / n_nets;
    bool all_narrow = true, partials_ready = S.rows_per_chunk == NLBAC_SK_CHUNK, reduced = false;
    for (int i = 0; i < n_nets; ++i) {
        all_narrow = all_narrow && nets[i].hid <= 128;
        // nlbac_mlp_bwd_data has already left this net's partial sums in its block of ws (nlbac_mlp_io::skinny_ws)
        partials_ready = partials_ready && io[i].skinny_ws == ws + (long)i * S.net_stride;
    }
    for (int i = 0; i < n_nets; ++i)
        NLBAC_REQUIRE(io[i].dz_first == 0 || partials_ready,
                      "nlbac_mlp_bwd_weights: dz_first > 0 (layer 0's dz rows were not stored) needs the data backward's partial sums");
    if (!partials_ready && nlbac_mlp_dw16_eligible(nets, n_nets, B))       // every layer's dW and db in one launch
        return nlbac_mlp_dw16_launch(L, n_nets, (hipStream_t)s);
    if (max_blocks > 0) {
        bool narrow = true;
        int max_layers = 0;
        for (int i = 0; i < n_nets; ++i) {
            narrow = narrow && nets[i].hid <= 128;
            if (nets[i].n_layers - 2 > max_layers) max_layers = nets[i].n_layers - 2;
        }
        if (narrow) {
            static bool attr_set = false;
            const size_t lds = (size_t)4 * DW_CHUNK * DW128_LD * sizeof(float);
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)mlp_bwd_wide128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_set = true;
            }
            hipLaunchKernelGGL(mlp_bwd_wide128_kernel, dim3(max_layers, n_slabs, n_nets), dim3(256), lds, (hipStream_t)s, L);
        } else if (dw64_enabled() && (long)B * 256 < (1L << 29)) {
            // (with the partial sums already there, the skinny reduction's blocks lead the GEMM tiles)
            const int red_per_net = max_q * 4, red_blocks = partials_ready ? red_per_net * n_nets : 0;
            const int groups8 = (n_slabs * n_nets + 7) / 8;
            hipLaunchKernelGGL(mlp_bwd_wide64_kernel, dim3(red_blocks + 8 * max_blocks * groups8), dim3(256),
                               (size_t)2 * 64 * 64 * sizeof(float), (hipStream_t)s, L, S, ws, red_blocks, red_per_net, n_nets,
                               max_blocks);
            reduced = partials_ready;
        } else {
            // (with the partial sums already there, the skinny reduction's blocks ride behind the GEMM tiles of slab 0)
            const int per_plane = max_blocks * n_slabs, red_per_net = max_q * 4;
            const int red_planes = partials_ready ? (red_per_net * n_nets + per_plane - 1) / per_plane : 0;
            hipLaunchKernelGGL(mlp_bwd_wide_kernel, dim3(max_blocks, n_slabs, n_nets + red_planes), dim3(256), 0,
                               (hipStream_t)s, L, S, ws, red_planes, red_per_net, n_nets);
            reduced = partials_ready;
        }
        NLBAC_CHECK_LAUNCH("nlbac_mlp_bwd_weights(wide)");
    }
    if (reduced) return 0;
    if (partials_ready) {
    } else if (all_narrow)
        hipLaunchKernelGGL(mlp_bwd_skinny_partial_kernel<128>, dim3(S.n_chunks, n_nets), dim3(128), 0, (hipStream_t)s, L, S, ws);
    else
        hipLaunchKernelGGL(mlp_bwd_skinny_partial_kernel<256>, dim3(S.n_chunks, n_nets), dim3(256), 0, (hipStream_t)s, L, S, ws);
    NLBAC_CHECK_LAUNCH("nlbac_mlp_bwd_weights(skinny partial)");
    hipLaunchKernelGGL(mlp_bwd_skinny_reduce_kernel, dim3(max_q * 4, n_nets), dim3(256), 0, (hipStream_t)s, L, S, ws);
    NLBAC_CHECK_LAUNCH("nlbac_mlp_bwd_weights(skinny reduce)");
    return 0;
}
