// Register-resident forward of the actor / critic nets (QNetwork, LyaNetwork, GaussianPolicy, BarrierNetwork:
// in -> hid -> hid -> out, U/sac_cbf_clf/model.py:37-114): the launches of nlbac_mlp_fwd / nlbac_mlp_fwd_gauss for nets
// whose one hid x hid layer has a panel pack (rr_device.h, RRPanel; hid = 64 / 128 / 256).
//
// A workgroup is still one 32-row tile of one net (grid.y = net), but its four waves no longer share a layer through LDS
// tiles and barriers: wave (rh, ch) owns rows 16 rh .. 16 rh + 15 and the output blocks of panel ch.  It computes layer 0
// for ALL units itself (K <= 16: a few dozen MFMAs, cheaper than exchanging it), keeps those activations in registers as
// the B operands of its panel's MFMA stream, and contributes its half of the skinny output layer; the two halves meet in
// LDS.  The LDS-tiled kernels run such a tile as a latency chain of ~16 us whatever the grid (a 3-net launch of 384
// workgroups: 33-38 us; 6 nets: 40 us); here a wave is ~620 MFMAs = 20k cycles of matrix-pipe time and a launch costs
// what its MFMAs cost.
#include "mlp_launch.h"
#include "rr_device.h"
#include <cstdlib>

#define MRR_MAX_IN 15       /* in_dim + the bias column <= 16: four k-steps of layer 0 */

#ifdef RR_TIMING      // ablation build: wave 0 of workgroup 0 stamps the shader clock behind the first net's outputs (as int64)
#define MSTAMP(k_) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long*>(io.y + (long)B * io.y_ld)[k_] = (long long)__builtin_readcyclecounter();
#else
#define MSTAMP(k_)
#endif

template <int NBH>
__global__ __launch_bounds__(256) void mlp_rr_fwd_kernel(const MlpLaunch L, const nlbac_gauss_head G) {
    constexpr int HID = 32 * NBH, NBA = 2 * NBH, KS = HID / 4, KSH = KS / 2;
    using P = RRPanel<NBH, KS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sO = smem;                               // [2][32][16] the panels' halves of the output layer
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), rh = wave >> 1, ch = wave & 1;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const int q = lane >> 4, r16 = lane & 15, m = 16 * rh + r16, grow = row0 + m;
    const bool row_ok = grow < B;
    const int idim = net.in_dim, odim = net.out_dim;
    const int KL0 = (idim + 4) >> 2;                      // k-steps of layer 0 over [x | 1] (1..4)
    const float* const params = net.params;

    MSTAMP(0)
    // ---- the panel's weight stream
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wcur = (net.rr_fwd_off + ch * (HID * HID / 2)) * 4;
    P panel;
    panel.prime(rs, voff, wcur);

    // ---- this lane's inputs: component 4 k0 + q of its row, a 1 in the bias column behind the last one
    float yv[4];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
        const int c = 4 * k0 + q;
        float v = (c == idim) ? 1.f : 0.f;
        if (row_ok && c < idim) v = (c < io.x0_dim) ? io.x0[(long)grow * io.x0_ld + c] : io.x1[(long)grow * io.x1_ld + (c - io.x0_dim)];
        yv[k0] = v;
    }
    // ---- this wave's half of the output layer's A fragments: lane (o, kq) supplies W_out[o][unit of k-step ks, quarter kq]
    float wo[KSH];
    {
        const float* wrow = params + net.w_off[2] + (long)min(r16, odim - 1) * HID + 16 * NBH * ch;
#pragma unroll
        for (int j = 0; j < NBH; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 16 * j + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[4 * j + r] = (r16 < odim) ? v[r] : 0.f;
        }
    }
    // the layer's biases enter as the C operands of this panel's blocks
    f32x4 cinit[NBH];
#pragma unroll
    for (int jo = 0; jo < NBH; ++jo) cinit[jo] = *reinterpret_cast<const f32x4*>(params + net.b_off[1] + 16 * (NBH * ch + jo) + 4 * q);
    MSTAMP(1)

    const long ls = io.acts_ls ? io.acts_ls : (long)B * HID;
    float* const a0row = (io.acts && ch == 0 && row_ok) ? io.acts + (long)grow * HID : nullptr;      // layer 0: one wave saves it
    float* const a1row = (io.acts && row_ok) ? io.acts + ls + (long)grow * HID + 16 * NBH * ch : nullptr;

    // ---- layer 0 for all units (bias folded into the product), ReLU in place
    float H0[KS];
    {
        // its A fragments come from the pack (four output blocks per load), layer 0's bias folded into the product
        const int l0 = (net.rr_bwd_off + HID * HID) * 4;
#pragma unroll
        for (int j4 = 0; j4 < NBA / 4; ++j4) {
            f32x4 w[4];
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) w[k0] = (k0 < KL0) ? rr_ldw(rs, voff, l0 + (k0 * (NBA / 4) + j4) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int jo = 4 * j4 + c;
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][c], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (KL0 > 1) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][c], yv[1], a, 0, 0, 0);
                if (KL0 > 2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][c], yv[2], a, 0, 0, 0);
                if (KL0 > 3) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3][c], yv[3], a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) H0[4 * jo + r] = rr_relu(a[r]);
            }
        }
    }
    MSTAMP(2)
    // ---- the hid x hid layer: this wave's panel; the last pair of blocks is finished inside the output product
    float H1[KSH];
    f32x4 acc[NBH];
    auto finish = [&](int jo, int r) __attribute__((always_inline)) {
        H1[4 * jo + r] = rr_relu(acc[jo][r]);
    };
    panel.run(acc, cinit, H0, rs, voff, wcur, [&](int) __attribute__((always_inline)) {},
              [&](int jo, int r) __attribute__((always_inline)) { finish(jo, r); });
    MSTAMP(3)
    // ---- this half of the output layer (one block, K = the panel's units); H1's last eight values just in time
    {
        const f32x4 o = P::template block<KSH>(wo, H1, [&](int ks) __attribute__((always_inline)) {
            // (NBH == 2: the pending pair IS the panel — value ks is finished right before k-step ks reads it)
            if (ks < 8) finish(NBH - 2 + (ks >> 2), ks & 3);
        });
        *reinterpret_cast<f32x4*>(sO + (ch * NLBAC_MLP_TILE + m) * 16 + 4 * q) = o;
    }
    MSTAMP(4)
    // ---- the saved activations leave in one burst BEHIND the weight stream: stores share the loads' in-order vmcnt
    //      queue, and one issued between two fragment loads makes the MFMAs that wait for the second load wait for the
    //      store's trip to HBM as well
    if (a0row) {
#pragma unroll
        for (int jo = 0; jo < NBA; ++jo)
            *reinterpret_cast<f32x4*>(a0row + 16 * jo + 4 * q) = f32x4{H0[4 * jo], H0[4 * jo + 1], H0[4 * jo + 2], H0[4 * jo + 3]};
    }
    if (a1row) {
#pragma unroll
        for (int jo = 0; jo < NBH; ++jo)
            *reinterpret_cast<f32x4*>(a1row + 16 * jo + 4 * q) = f32x4{H1[4 * jo], H1[4 * jo + 1], H1[4 * jo + 2], H1[4 * jo + 3]};
    }
    __syncthreads();
    MSTAMP(5)
    // ---- the two halves + bias -> y; the thread that writes a row's (mean | log_std) also draws the row's action and
    //      log-probability from it (nlbac_gauss_head: gauss_fwd_kernel's arithmetic, no launch of its own)
    if (tid < NLBAC_MLP_TILE && row0 + tid < B) {
        const int row = row0 + tid;
        const float* bo = params + net.b_off[2];
        float* y = io.y + (long)row * io.y_ld;
        for (int o = 0; o < odim; ++o) y[o] = (sO[tid * 16 + o] + sO[(NLBAC_MLP_TILE + tid) * 16 + o]) + bo[o];
        if (G.eps)
            gauss_fwd_row(y, G.eps, G.scale, G.bias, G.n_u, (long)blockIdx.y * B + row, G.action, G.action_ld, G.logp);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
static bool mrr_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_RR"); return !(e && e[0] == '0'); }();
    return on;
}

bool nlbac_mlp_rr_eligible(const nlbac_mlp* nets, int n_nets) {
    if (!mrr_enabled()) return false;
    for (int i = 0; i < n_nets; ++i) {
        const nlbac_mlp& n = nets[i];
        if (n.rr_kind != RR_KIND_PANEL || n.rr_fwd_off < 0 || n.hid != nets[0].hid) return false;
        if (n.hid != 64 && n.hid != 128 && n.hid != 256) return false;
        if (n.in_dim > MRR_MAX_IN || n.out_dim > 16) return false;
    }
    return true;
}

int nlbac_mlp_rr_fwd_launch(const MlpLaunch& L, int n_nets, const nlbac_gauss_head& G, const char* who, hipStream_t s) {
    if (!nlbac_mlp_rr_eligible(L.net, n_nets)) return 1;
    const int hid = L.net[0].hid;
    const size_t lds = (size_t)(2 * NLBAC_MLP_TILE * 16) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.B, NLBAC_MLP_TILE), n_nets);
    switch (hid) {
        case 64: hipLaunchKernelGGL(mlp_rr_fwd_kernel<2>, grid, dim3(256), lds, s, L, G); break;
        case 128: hipLaunchKernelGGL(mlp_rr_fwd_kernel<4>, grid, dim3(256), lds, s, L, G); break;
        default: hipLaunchKernelGGL(mlp_rr_fwd_kernel<8>, grid, dim3(256), lds, s, L, G);
    }
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}
