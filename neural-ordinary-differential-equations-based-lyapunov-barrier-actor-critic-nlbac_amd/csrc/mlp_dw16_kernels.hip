// Weight and bias gradients of narrow nets (hid <= 112: the NODE's f_net / g_net and the single-net NODEs) — every
// layer's dW_j and db_j in ONE launch that reads each dz / activation array from HBM exactly once:
//
//   dW_j[n][k] = sum_b dz[j][b][n] * acts[j-1][b][k]      1 <= j <= nwide-1   (hid x hid)
//   dW_0[n][i] = sum_b dz[0][b][n] * x[b][i]                                  (hid x in_dim)
//   dW_L[o][k] = sum_b dy[b][o]    * acts[nwide-1][b][k]                      (out_dim x hid)
//   db_j[n]    = sum_b dz[j][b][n],   db_L[o] = sum_b dy[b][o]
//
// (model.py:221-260's loss.backward() for the NODE parameters; the NODE fit streams ~1.1 GB of dz + activations per RK
// step through this — the one HBM-heavy place of the update.)
//
// grid = (row slab, net, layer slot), 256 threads.  The sum over rows is split over WAVES, not the output over tiles:
// every wave accumulates the whole gradient of its layer over its own rows on v_mfma_f32_16x16x4_f32 (one k-step = 4
// rows; up to 7 x 7 independent 16 x 16 accumulators, 196 AGPRs).  No operand is shared between waves, so nothing goes
// through LDS and no barrier sits in the loop: a lane loads its MFMA operands straight from global memory, PF k-steps
// ahead, into registers that ARE the operands.  The trick is the column order: within a k-step, lane (q, i) owns row
// 4s + q, and one dwordx4 load of columns 4i..4i+3 gives it its value for FOUR tiles — tile t of the first group is the
// column set {4i + t}, not {16t + i}; the gradient does not care which 16 columns are called a tile, only the store at
// the end does.  A 100-wide row is a dwordx4 (columns 0..63: tiles 0-3), a dwordx2 (64..95: tiles 4, 5) and a dword
// (96..111: tile 6): 7 tiles = 112 columns (the 32 x 32-tile kernel this replaces padded to 128 x 128, went through LDS
// and two barriers per 32 rows, and left the biases and the two skinny layers to a second pass over every dz).
// Raw-buffer loads return 0 out of range: rows past the end and lanes past the width need no select.
// The wave's k-steps are interleaved over the whole launch (k-step g of wave w of slab s: g = 4s + w + m * 4 n_slabs),
// so at any moment the chip reads one moving window of the arrays.  At the end the four partial gradients of a
// workgroup are summed through LDS in wave order and written to the slab's gradient block.
#include "common.h"
#include "mlp_launch.h"
#include "rr_device.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DW16_PF 8             // k-steps of operands in flight per wave (8 x 4 rows x 2 x 400 B = 25 KB per wave)
#define DW16_OOB ((int)0x80000000)

// tile t, lane column i  ->  the column it stands for (see above)
__device__ __forceinline__ int dw16_col(int t, int i) { return t < 4 ? 4 * i + t : (t < 6 ? 64 + 2 * i + (t - 4) : 96 + i); }

// One side of a product.  VEC: a dense (rows x hid) array read through a buffer resource (NT tiles: 4 / 6 / 7 for
// hid <= 64 / 96 / 112).  Otherwise (the first layer's inputs, the last layer's dy): one tile, up to 16 columns
// gathered from one or two strided sources.
template <int NT, bool VEC>
struct Dw16Side {
    static constexpr int T = VEC ? NT : 1;
    __amdgpu_buffer_rsrc_t rs;
    int vo4, vo2, vo1;                         // this lane's byte offsets within a row (DW16_OOB: past the width)
    const float* p0; const float* p1;          // (gathered)
    int ld0, ld1, split, width, nrows, q, i;
    float v[DW16_PF][T];

    __device__ __forceinline__ void init_vec(const float* base, int hid, int B, int q_, int i_) {
        rs = rr_rsrc(base, B * hid);
        q = q_; i = i_; width = hid; nrows = B;
        const int row = q * hid * 4;
        vo4 = (4 * i + 3 < hid) ? row + 16 * i : DW16_OOB;
        vo2 = (64 + 2 * i + 1 < hid) ? row + 256 + 8 * i : DW16_OOB;
        vo1 = (96 + i < hid) ? row + 384 + 4 * i : DW16_OOB;
    }
    __device__ __forceinline__ void init_gather(const float* a, int lda, int split_, const float* b, int ldb, int width_,
                                                int B, int q_, int i_) {
        p0 = a; p1 = b; ld0 = lda; ld1 = ldb; split = split_; width = width_; nrows = B; q = q_; i = i_;
    }
    // operands of k-step g (rows 4g .. 4g+3) into set u
    __device__ __forceinline__ void load(int u, int g) {
        if constexpr (VEC) {
            const int rb = g * 16 * width;                                 // byte offset of row 4g
            const f32x4 a = rr_ldw(rs, vo4 + rb, 0);
            v[u][0] = a[0]; v[u][1] = a[1]; v[u][2] = a[2]; v[u][3] = a[3];
            if constexpr (NT > 4) {
                // (the pair is cast as a whole: clang's __builtin_bit_cast of a vector ELEMENT reads element 0)
                const f32x2 b = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo2 + rb, 0, 0));
                v[u][4] = b[0];
                v[u][5] = b[1];
            }
            if constexpr (NT > 6) v[u][6] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo1 + rb, 0, 0));
        } else {
            const long row = 4L * g + q;
            float x = 0.f;
            if (i < width && row < nrows) x = (i < split) ? p0[row * ld0 + i] : p1[row * ld1 + (i - split)];
            v[u][0] = x;
        }
    }
};

template <int NT, bool AVEC, bool BVEC>
__device__ __forceinline__ void dw16_body(Dw16Side<NT, AVEC>& a, Dw16Side<NT, BVEC>& b, int g0, int gstep, int n_ksteps,
                                          float* __restrict__ gW, int ldw, float* __restrict__ gb, float* smem) {
    constexpr int NTA = Dw16Side<NT, AVEC>::T, NTB = Dw16Side<NT, BVEC>::T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
    f32x4 acc[NTA][NTB];
    float bs[NTA];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
        bs[ta] = 0.f;
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // this wave's k-steps: g0, g0 + gstep, ...; the count is rounded up to whole rings (k-steps past the end load zeros)
    const int mine = (g0 < n_ksteps) ? (n_ksteps - g0 + gstep - 1) / gstep : 0;
    const int rounds = (mine + DW16_PF - 1) / DW16_PF;
    if (rounds > 0) {
#pragma unroll
        for (int u = 0; u < DW16_PF; ++u) { a.load(u, g0 + u * gstep); b.load(u, g0 + u * gstep); }
        int g = g0 + DW16_PF * gstep;
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int u = 0; u < DW16_PF; ++u) {
                float av[NTA], bv[NTB];
#pragma unroll
                for (int t = 0; t < NTA; ++t) { av[t] = a.v[u][t]; bs[t] += av[t]; }
#pragma unroll
                for (int t = 0; t < NTB; ++t) bv[t] = b.v[u][t];
#ifndef DW16_NO_MFMA
#pragma unroll
                for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ta], bv[tb], acc[ta][tb], 0, 0, 0);
#endif
                // every accumulator is named as an AGPR tuple here: without the anchor the allocator keeps a row of them
                // in VGPRs and rotates the rest through v_accvgpr_mov each k-step (VALU slots the MFMAs wait behind)
#pragma unroll
                for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb) asm volatile("" : "+a"(acc[ta][tb]));
#ifndef DW16_NO_LOAD
                a.load(u, g);                    // (the set just used is refilled PF k-steps ahead)
                b.load(u, g);
#endif
                __builtin_amdgcn_sched_barrier(0);
                g += gstep;
            }
        }
    }

    // the four waves' partial gradients, summed in wave order through LDS
    float* red = smem;
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((ta * NTB + tb) * 4 + r) * 64 + lane] = acc[ta][tb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ta][tb][r] += red[((ta * NTB + tb) * 4 + r) * 64 + lane];
        }
    }
    __syncthreads();
    // column sums of the A side: lane (q, i) holds the sum over ITS rows of tile t's column — 16 partials per column
    float (*sBias)[NTA * 16] = reinterpret_cast<float (*)[NTA * 16]>(smem);       // [wave * 4 + q][t * 16 + i]
#pragma unroll
    for (int t = 0; t < NTA; ++t) sBias[wave * 4 + q][t * 16 + i] = bs[t];
    __syncthreads();
    if (tid < NTA * 16) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += sBias[k][tid];
        const int col = AVEC ? dw16_col(tid >> 4, tid & 15) : (tid & 15);
        if (col < a.width) gb[col] = v;
    }
    if (wave == 0) {
        // D[m][n] of a 16 x 16 x 4 MFMA: lane (q, i) register r holds row m = 4q + r (an A-side lane column), column
        // n = i (a B-side lane column)
#pragma unroll
        for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = AVEC ? dw16_col(ta, 4 * q + r) : 4 * q + r;
                    const int k = BVEC ? dw16_col(tb, i) : i;
                    if (n < a.width && k < b.width) gW[(long)n * ldw + k] = acc[ta][tb][r];
                }
    }
}

// layer slot -> layer: the hid x hid layers first (slots 0 .. nwide-2 -> layers 1 .. nwide-1), then the first and the last
// layer: the long workgroups of a launch are dispatched before the short ones
__device__ __forceinline__ int dw16_layer_of_slot(int slot, int nwide) {
    return slot < nwide - 1 ? slot + 1 : (slot == nwide - 1 ? 0 : nwide);
}

template <int NT>
__global__ __launch_bounds__(256) void mlp_dw16_kernel(const MlpLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float dw16_smem[];
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B, hid = net.hid, nwide = net.n_layers - 1;
    if ((int)blockIdx.z > nwide) return;                    // (uniform: nets of one launch may differ in depth)
    const int j = dw16_layer_of_slot(blockIdx.z, nwide);
#ifdef DW16_NO_EDGE
    if (j == 0 || j == nwide) return;
#endif
#ifdef DW16_NO_WIDE
    if (j != 0 && j != nwide) return;
#endif
    const int slab = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    const int g0 = slab * 4 + wave, gstep = L.n_slabs * 4, n_ksteps = (B + 3) >> 2;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * hid;
    float* g = io.grad + (long)slab * L.slab_stride;
    if (j == 0) {
        Dw16Side<NT, true> a; Dw16Side<NT, false> b;
        a.init_vec(io.dz, hid, B, q, i);
        b.init_gather(io.x0, io.x0_ld, io.x0_dim, io.x1, io.x1_ld, net.in_dim, B, q, i);
        dw16_body<NT, true, false>(a, b, g0, gstep, n_ksteps, g + net.w_off[0], net.in_dim, g + net.b_off[0], dw16_smem);
    } else if (j == nwide) {
        Dw16Side<NT, false> a; Dw16Side<NT, true> b;
        a.init_gather(io.dy, io.dy_ld, net.out_dim, nullptr, 0, net.out_dim, B, q, i);
        b.init_vec(io.acts + (long)(nwide - 1) * ls, hid, B, q, i);
        dw16_body<NT, false, true>(a, b, g0, gstep, n_ksteps, g + net.w_off[nwide], hid, g + net.b_off[nwide], dw16_smem);
    } else {
        Dw16Side<NT, true> a; Dw16Side<NT, true> b;
        a.init_vec(io.dz + (long)j * ls, hid, B, q, i);
        b.init_vec(io.acts + (long)(j - 1) * ls, hid, B, q, i);
        dw16_body<NT, true, true>(a, b, g0, gstep, n_ksteps, g + net.w_off[j], hid, g + net.b_off[j], dw16_smem);
    }
}

bool nlbac_mlp_dw16_eligible(const nlbac_mlp* nets, int n_nets, int B) {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_DW16"); return !(e && e[0] == '0'); }();
    if (!on) return false;
    for (int i = 0; i < n_nets; ++i)
        if (nets[i].hid > 112) return false;
    return true;
}

int nlbac_mlp_dw16_launch(const MlpLaunch& L, int n_nets, hipStream_t s) {
    int max_hid = 0, max_layers = 0;
    for (int i = 0; i < n_nets; ++i) {
        // (byte offsets into a layer's rows are 32-bit.  An error rather than the older kernels: those leave the skinny
        // gradients in slab 0 only, and a caller that alternates between the two would sum stale partials)
        // (... with headroom for the k-steps the ring requests past the end — DW16_PF * 4 * n_slabs k-steps of 4 rows — so
        //  that no byte offset passes 2^31: those loads must land out of range, not wrap)
        NLBAC_REQUIRE(((long)L.B + 16L * DW16_PF * L.n_slabs + 4) * L.net[i].hid < (1L << 29),
                      "nlbac_mlp_bwd_weights: %d rows x %d units (+ the ring's run-out) exceed 2^29 elements per layer", L.B, L.net[i].hid);
        if (L.net[i].hid > max_hid) max_hid = L.net[i].hid;
        if (L.net[i].n_layers > max_layers) max_layers = L.net[i].n_layers;
    }
    const dim3 grid(L.n_slabs, n_nets, max_layers);         // (slots 0 .. nwide = n_layers - 1; x fastest: a slot's slabs together)
    const size_t lds = (size_t)49 * 4 * 64 * sizeof(float);
    if (max_hid <= 64) hipLaunchKernelGGL(mlp_dw16_kernel<4>, grid, dim3(256), lds, s, L);
    else if (max_hid <= 96) hipLaunchKernelGGL(mlp_dw16_kernel<6>, grid, dim3(256), lds, s, L);
    else hipLaunchKernelGGL(mlp_dw16_kernel<7>, grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_mlp_bwd_weights(dw16)");
    return 0;
}
