// K1b: fused edge scorer on the f32 matrix cores (gfx950, v_mfma_f32_32x32x2_f32 = exact fp32).
//
// Reference: the `_edge_score` closures, model.py:29-34 / 115-122:
//     p_e = sigmoid( fc2( dropout( relu( fc1( [x_s * x_d | x_s - x_d] ) ) ) ) )
// which materialise [E,2H] features and [E,H] hidden activations (>= 10 KB per edge).  Here a
// workgroup owns a tile of 128 edges and all H hidden units and nothing per-edge but p_e (4 B)
// is written.  Algebraic split:
//     W1 [x*y | x-y] = W1a (x*y) + U[s] - U[d],   U = codes W1b^T  (node-level GEMM, host side)
// so the per-edge contraction is H x H instead of H x 2H.
//
// MFMA mapping: D[h][e] = sum_k W1a[h][k] * (x_s[k] x_d[k]):  A = W1a (rows h -> accumulator
// registers), B = the gathered Hadamard features (cols e -> lanes).  With the hidden units in
// registers the fc2 reduction over h is in-lane plus one cross-half shuffle.  Per 32-deep
// k-step a workgroup stages W1a^T[k0:k0+32][0:H] (coalesced, k-major) and the 128x32 feature
// tile ([k][e] image, conflict-free ds_read_b32 for the B operand) in LDS.
// Roofline: MFMA-bound (4 H^2 / 2 flops per edge after the split vs ~4 KB of L2 traffic).
#include <type_traits>

#include "sgs_common.h"

namespace sgs {
namespace {

constexpr int kT = 256;      // 4 waves = 2 edge groups x 2 hidden halves
constexpr int kBM = 64;      // edges per workgroup (32 per edge group)
constexpr int kBK = 16;      // k-step
using f32x16 = __attribute__((ext_vector_type(16))) float;

// W1 [H][2H] (fc1.weight) -> WaT [k][h] = W1[h][k], k < H
__global__ void __launch_bounds__(kT) transpose_w1a(const float* __restrict__ W1, int H, float* __restrict__ WaT) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: k block, by: h block
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int h = by + r, k = bx + tx;
        t[r][tx] = (h < H && k < H) ? W1[static_cast<int64_t>(h) * 2 * H + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = bx + r, h = by + tx;
        if (k < H && h < H) WaT[static_cast<int64_t>(k) * H + h] = t[tx][r];
    }
}

// W1 [H][2H] -> WT [k][h] = W1[h][k], k < 2H (endpoint-dropout variant: no node-level split)
__global__ void __launch_bounds__(kT) transpose_w1_full(const float* __restrict__ W1, int H, float* __restrict__ WT) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: k block (0 .. 2H), by: h block
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int h = by + r, k = bx + tx;
        t[r][tx] = (h < H && k < 2 * H) ? W1[static_cast<int64_t>(h) * 2 * H + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = bx + r, h = by + tx;
        if (k < 2 * H && h < H) WT[static_cast<int64_t>(k) * H + h] = t[tx][r];
    }
}

struct ScoreArgs {
    const float* codes;      // [N,H]
    const float* U;          // [N,H] = codes W1b^T
    const int64_t* src;      // [E]
    const int64_t* dst;      // [E]
    const int64_t* active;   // [n] edge ids (backward) or nullptr = identity
    int64_t n;               // rows processed (E forward, n_active backward)
    const int32_t* canon;    // paired forward (MODE 3): [n] ids of the edges processed here; each also produces the score of its mate
    const int32_t* mate;     // paired forward: [E] id of the reverse edge (d -> s) of edge (s -> d), or -1
    const int64_t* dyn_n;    // forward under sgs_dyn_edges_set: the live row count is read from this device word; `n` is then only the
                             // capacity the grid was sized for (HIP-graph replay of one captured step over partitions of any size)
    int64_t row_offset;      // global id of local edge 0 (edge-sharded graphs): dropout rows are global edge ids
    int H;
    const float* WaT;        // [H][H] k-major
    const float* b1;
    const float* w2;
    const float* b2;
    float drop_scale;
    uint32_t drop_thresh;
    uint64_t seed;
    const uint64_t* epoch;   // RNG epoch word (HIP-graph replay) or nullptr
    uint32_t site;
    int use_drop;
    float* p_out;            // forward: [n]
    const float* gp;         // backward: [n]
    float* dv;               // backward: [n,H]  dL/d(pre-activation)
    float* hdz;              // backward: [cdiv(n,64), H] per-tile column sums of dz * dropped hidden (rows sum to dw2)
    float* dz;               // backward: [n]
    float* feat;             // backward: [n,H]  x_s * x_d
    uint32_t* dvbits;        // backward (MODE 1), optional: [n, H/32] bit h of row r = [dropped hidden h of row r > 0] -- with dz and w2 that IS dv
    const uint32_t* inbits;  // MODE 4 / 5 (dfeat from the mask): [n, H/32]
    const float* indz;       // MODE 4 / 5: [n]
    int epd;                 // endpoint dropout (EdgeProbMLP, model.py:21-25): x = drop(A[src]), y = drop(A[dst]) per (edge, endpoint); K = 2H, no U term
    uint64_t seed_x, seed_y; // ... their mask streams (sites site_x / site_y, row = edge id)
    uint32_t site_x, site_y;
    float ep_scale;          // 1 / (1 - p_endpoint)
    uint32_t ep_thresh;
    const int32_t* sd;       // MODE 5: [n, 2] (src, dst) of every active row
    float* opart;            // MODE 5: [cdiv(n, 32) + N, H] run-end partial sums of dfeat * codes[dst] (see the kernel)
    // bf16x6 kernels: start-up stagger of every CU's second resident workgroup, in 64-cycle sleeps (0 = none; see the kernel), and the probe
    // knobs of tools/stagger_probe.py / stagger_trace.py (0 / NULL in the product path)
    int stagger, prio;
    unsigned long long* trace;
};

// Workgroup = 64 edges x all HP = 32*NT hidden units; wave (eg, hh) owns edges 32eg..32eg+31 and
// hidden units hh*HP/2 .. (hh+1)*HP/2 - 1, i.e. NT/2 accumulator tiles of 32x32 (<= 64 registers),
// small enough that three independent workgroups share a CU and hide each other's tile staging,
// prologue and epilogue behind MFMAs.  EXACT: H == 32*NT (the production H = 256): every bounds
// check on k / h folds away.
// EPD (endpoint dropout, EdgeProbMLP with dropout > 0): the endpoint codes are masked per (edge, endpoint) before the features are
// formed, so the node-level split U[s] - U[d] does not exist: the contraction runs over all 2H features -- k < H: x_m * y_m against
// W1[:, k], k >= H: x_m - y_m against W1[:, k] (WaT is then the full [2H][H] transpose) -- and BWD writes feat [n, 2H].
template <int NT, bool BWD, bool EXACT, bool EPD = false>
__global__ void __launch_bounds__(kT, 3) edge_score_kernel(ScoreArgs a) {
    constexpr int HP = 32 * NT;
    constexpr int NTW = NT / 2;                                   // tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two LDS stages of {W1a^T tile [kBK][HP], feature tile [kBK][64]}, endpoint ids, fc2 partials
    float* Wt_s0 = reinterpret_cast<float*>(smem);
    float* Ft_s0 = Wt_s0 + 2 * kBK * HP;
    int* s_idx = reinterpret_cast<int*>(Ft_s0 + 2 * kBK * kBM);   // [64]
    int* d_idx = s_idx + kBM;                                     // [64]
    float* zpart = reinterpret_cast<float*>(d_idx + kBM);         // [2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int eg = wave & 1, hh = wave >> 1;
    const int H = EXACT ? HP : a.H;
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kBM;
    if (a.dyn_n) {
        const int64_t live_n = *a.dyn_n;                          // never more than the capacity the launch was sized for
        if (live_n < a.n) a.n = live_n;
        if (row0 >= a.n) return;                                  // (uniform per workgroup)
    }

    if (tid < kBM) {
        const int64_t r = row0 + tid;
        int s = 0, d = 0;
        if (r < a.n) {
            const int64_t e = a.active ? a.active[r] : r;
            s = static_cast<int>(a.src[e]);
            d = static_cast<int>(a.dst[e]);
        }
        s_idx[tid] = s;
        d_idx[tid] = d;
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int kh = lane >> 5, l31 = lane & 31;
    __syncthreads();   // s_idx / d_idx visible

    // Software pipeline: the NEXT k-step's W rows and endpoint rows are fetched into registers while
    // the matrix cores work on the current LDS stage; they are multiplied / transposed into the
    // other stage mid-phase.  One barrier per k-step.
    constexpr int kWV = (kBK * (HP / 4) + kT - 1) / kT;      // float4 W loads per thread per k-step (<= 4)
    float4 wreg[kWV], xreg, yreg;
    const int fe = tid & (kBM - 1), fc = tid >> 6;           // this thread's (edge, float4 chunk) of the feature tile
    const int my_s = s_idx[fe], my_d = d_idx[fe];
    const int KT = EPD ? 2 * H : H;                          // contraction length
    uint32_t rkx = 0u, rky = 0u;                             // EPD: this thread's edge's mask rows for the two endpoints
    if (EPD) {
        const int64_t r_ = row0 + fe;
        const int64_t e_ = r_ < a.n ? (a.active ? a.active[r_] : r_) : 0;
        rkx = dropout_row_key(fold_epoch(a.seed_x, a.epoch), a.site_x, static_cast<uint64_t>(a.row_offset + e_));
        rky = dropout_row_key(fold_epoch(a.seed_y, a.epoch), a.site_y, static_cast<uint64_t>(a.row_offset + e_));
    }
    auto ep_mask = [&](float4& v, uint32_t rk, int kk) {     // v <- v * keep / (1 - p) on columns kk .. kk + 3 (kk % 4 == 0)
        const uint32_t b0 = dropout_pair_bits(rk, static_cast<uint32_t>(kk >> 1)), b1 = dropout_pair_bits(rk, static_cast<uint32_t>((kk >> 1) + 1));
        v.x = (b0 & 0xFFFFu) >= a.ep_thresh ? v.x * a.ep_scale : 0.f;
        v.y = (b0 >> 16) >= a.ep_thresh ? v.y * a.ep_scale : 0.f;
        v.z = (b1 & 0xFFFFu) >= a.ep_thresh ? v.z * a.ep_scale : 0.f;
        v.w = (b1 >> 16) >= a.ep_thresh ? v.w * a.ep_scale : 0.f;
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < kWV; ++j) {
            const int i = j * kT + tid;
            const int k = i / (HP / 4), h4 = (i % (HP / 4)) * 4;
            wreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EXACT || (i < kBK * (HP / 4) && k0 + k < KT && h4 < H))
                wreg[j] = *reinterpret_cast<const float4*>(a.WaT + static_cast<int64_t>(k0 + k) * H + h4);
        }
        const int kf = k0 + 4 * fc;                              // feature index (0 .. KT - 1)
        const int kk = (EPD && kf >= H) ? kf - H : kf;           // column of the codes it is formed from
        xreg = make_float4(0.f, 0.f, 0.f, 0.f);
        yreg = xreg;
        if (EXACT || kf < KT) {
            xreg = *reinterpret_cast<const float4*>(a.codes + static_cast<int64_t>(my_s) * H + kk);
            yreg = *reinterpret_cast<const float4*>(a.codes + static_cast<int64_t>(my_d) * H + kk);
            if (EPD && a.epd) { ep_mask(xreg, rkx, kk); ep_mask(yreg, rky, kk); }
        }
    };
    auto commit = [&](int k0, int buf) {
        float* Wt_s = Wt_s0 + buf * kBK * HP;
        float* Ft_s = Ft_s0 + buf * kBK * kBM;
#pragma unroll
        for (int j = 0; j < kWV; ++j) {
            const int i = j * kT + tid;
            if (EXACT || i < kBK * (HP / 4)) {
                const int k = i / (HP / 4), h4 = (i % (HP / 4)) * 4;
                *reinterpret_cast<float4*>(Wt_s + k * HP + h4) = wreg[j];
            }
        }
        const bool diff = EPD && (k0 + 4 * fc) >= H;              // second half of the features: x_m - y_m
        const float4 f = diff ? make_float4(xreg.x - yreg.x, xreg.y - yreg.y, xreg.z - yreg.z, xreg.w - yreg.w)
                              : make_float4(xreg.x * yreg.x, xreg.y * yreg.y, xreg.z * yreg.z, xreg.w * yreg.w);
        if (BWD) {
            const int64_t r = row0 + fe;
            const int kk = k0 + 4 * fc;
            if (r < a.n && (EXACT || kk < KT)) *reinterpret_cast<float4*>(a.feat + r * KT + kk) = f;
        }
        Ft_s[(4 * fc + 0) * kBM + fe] = f.x;
        Ft_s[(4 * fc + 1) * kBM + fe] = f.y;
        Ft_s[(4 * fc + 2) * kBM + fe] = f.z;
        Ft_s[(4 * fc + 3) * kBM + fe] = f.w;
    };
    // MFMA operands are register double-buffered one k2-step ahead of their use.
    float av[2][NTW], bv[2];
    auto lds_operands = [&](int buf, int kk, int slot) {
        const float* Wt_s = Wt_s0 + buf * kBK * HP + hh * (HP / 2);
        const float* Ft_s = Ft_s0 + buf * kBK * kBM;
        bv[slot] = Ft_s[(kk + kh) * kBM + 32 * eg + l31];
#pragma unroll
        for (int t = 0; t < NTW; ++t) av[slot][t] = Wt_s[(kk + kh) * HP + 32 * t + l31];
    };

    fetch(0);
    commit(0, 0);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int k0 = 0; k0 < KT; k0 += kBK) {
        const bool more = k0 + kBK < KT;
        if (more) fetch(k0 + kBK);                 // global loads in flight during the MFMAs
        lds_operands(cur, 0, 0);
#pragma unroll
        for (int st = 0; st < kBK / 2; ++st) {
            if (st + 1 < kBK / 2) lds_operands(cur, 2 * (st + 1), (st + 1) & 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1][t], bv[st & 1], acc[t], 0, 0, 0);
            if (st == kBK / 4 - 1 && more) commit(k0 + kBK, cur ^ 1);
        }
        __syncthreads();                           // next stage complete; everyone done with this one
        cur ^= 1;
    }

    // ---- epilogue: lane = edge (l31) x half (kh); acc[t][r] is hidden unit
    //      hh*HP/2 + 32t + (r&3) + 8(r>>2) + 4kh of edge row0 + 32eg + l31
    const int el = 32 * eg + l31;
    const int64_t r = row0 + el;
    const bool live = r < a.n;
    const int64_t eg_id = live ? (a.active ? a.active[r] : r) : 0;   // global edge id: dropout row
    const uint32_t rkey = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + eg_id));
    const float* Us = a.U + static_cast<int64_t>(s_idx[el]) * H;
    const float* Ud = a.U + static_cast<int64_t>(d_idx[el]) * H;
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hb = hh * (HP / 2) + 32 * t + 8 * g + 4 * kh;
            if (EXACT || hb < H) {
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 us = EPD ? z4 : *reinterpret_cast<const float4*>(Us + hb);          // EPD: no node-level term
                const float4 ud = EPD ? z4 : *reinterpret_cast<const float4*>(Ud + hb);
                const float4 bb = *reinterpret_cast<const float4*>(a.b1 + hb);
                const float4 ww = *reinterpret_cast<const float4*>(a.w2 + hb);
                const float u4[4] = {us.x - ud.x, us.y - ud.y, us.z - ud.z, us.w - ud.w};
                const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
                const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
                uint32_t bits[2] = {0u, 0u};
                if (a.use_drop) {
                    bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
                    bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (acc[t][4 * g + j] + u4[j]) + b4[j];
                    float m = v > 0.f ? 1.f : 0.f;                 // relu'
                    if (a.use_drop) {
                        const uint32_t draw = (j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu);
                        m = draw >= a.drop_thresh ? m * a.drop_scale : 0.f;
                    }
                    const float hd = v * m;                          // dropout(relu(v))
                    z = fmaf(w4[j], hd, z);
                    if (BWD) acc[t][4 * g + j] = hd;                 // kept for the second epilogue pass
                }
            }
        }
    }
    z += __shfl_xor(z, 32, 64);
    if (kh == 0) zpart[hh * kBM + el] = z;           // fc2 partial over this wave's hidden half
    __syncthreads();
    z = (zpart[el] + zpart[kBM + el]) + a.b2[0];
    const float p = 1.0f / (1.0f + expf(-z));
    if (!BWD) {
        if (live && hh == 0 && kh == 0) a.p_out[r] = p;
        return;
    }
    // ---- backward epilogue: dz = gp p (1-p);  dv = dz w2 relu' keep scale;  hdz = dz hd
    // hdz is only ever column-summed (d w2 = sum_e dz_e hd_e): instead of writing [n, H] and reading it back, every
    // workgroup reduces its 64 edges in registers / LDS and writes ONE row of partial sums, hdz_part[blockIdx.x][:].
    const float dzv = live ? a.gp[r] * p * (1.0f - p) : 0.f;                     // 0 on the padding rows of the last tile
    if (live && hh == 0 && kh == 0) a.dz[r] = dzv;
    float* colpart = Wt_s0;                                                      // [2 edge groups][HP]: the operand stages are dead
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int hb = hh * (HP / 2) + 32 * t + 8 * g + 4 * kh;
            if (EXACT || hb < H) {
                const float4 ww = *reinterpret_cast<const float4*>(a.w2 + hb);
                const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
                float dv4[4], hz4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float hd = acc[t][4 * g + j];
                    // hd > 0  <=>  v > 0 and kept; then d hd / d v = scale (or 1 without dropout)
                    const float m = hd > 0.f ? (a.use_drop ? a.drop_scale : 1.f) : 0.f;
                    dv4[j] = dzv * w4[j] * m;
                    hz4[j] = dzv * hd;
                }
                if (live) *reinterpret_cast<float4*>(a.dv + r * H + hb) = make_float4(dv4[0], dv4[1], dv4[2], dv4[3]);
                // sum over the 32 edges of this half-wave (lanes with equal kh): fixed xor tree
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sum = hz4[j];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (l31 == 0) colpart[eg * HP + hb + j] = sum;
                }
            }
        }
    }
    __syncthreads();
    for (int h = tid; h < H; h += kT) a.hdz[static_cast<int64_t>(blockIdx.x) * H + h] = colpart[h] + colpart[HP + h];
}

// ---------------------------------------------------------------------------------------------
// Forward variant B ("stream"): no LDS tiles, no per-k-step barriers.  Both MFMA operands are streamed
// from L2 straight into registers:
//   * W1a is pre-packed as Wp[tile t][j4][lane][4]: the A operands of lane (l31, kh) for the four
//     k2-steps j = 4 j4 .. 4 j4 + 3 of tile t (k = 8 j4 + 2 jj + kh, h = 32 t + l31) are ONE 16-byte load;
//   * the node codes are re-laid as Ceo[n][kh][H/2] (even / odd k split) so that the B operands of lane
//     (edge l31, kh) for the same four k2-steps are one 16-byte load of x and one of y.
// Waves never wait for each other until the final fc2 combine, so three waves per SIMD keep the matrix
// pipe fed while others gather; operand registers are double-buffered one j4-iteration (4 NTW MFMAs) ahead.
// Requires H % 64 == 0 (H = 64, 128, 256); other sizes use the LDS-tiled kernel above.
__global__ void __launch_bounds__(kT) pack_w1a_stream(const float* __restrict__ W1, int H, float* __restrict__ Wp) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;     // one output float
    if (i >= static_cast<int64_t>(H) * H) return;
    const int jj = i & 3, lane = (i >> 2) & 63;
    const int64_t rest = i >> 8;                  // t * (H/8) + j4
    const int j4 = static_cast<int>(rest % (H / 8)), t = static_cast<int>(rest / (H / 8));
    const int kh = lane >> 5, l31 = lane & 31;
    const int k = 8 * j4 + 2 * jj + kh, h = 32 * t + l31;
    Wp[i] = W1[static_cast<int64_t>(h) * 2 * H + k];
}
__global__ void __launch_bounds__(kT) pack_codes_eo(const float* __restrict__ codes, int64_t N, int H, float* __restrict__ Ceo) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (i >= N * H) return;
    const int64_t n = i / H;
    const int k = static_cast<int>(i - n * H);
    Ceo[n * H + (k & 1) * (H / 2) + (k >> 1)] = codes[i];
}

// both re-layouts in one launch (the forward is launch-latency sensitive at partition scale): the first n_w workgroups
// pack W1a, the rest re-lay the codes
__global__ void __launch_bounds__(kT) pack_stream_operands(const float* __restrict__ W1, int H, float* __restrict__ Wp, int n_w,
                                                          const float* __restrict__ codes, int64_t N, float* __restrict__ Ceo) {
    if (static_cast<int>(blockIdx.x) < n_w) {
        const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;     // one output float
        if (i >= static_cast<int64_t>(H) * H) return;
        const int jj = i & 3, lane = (i >> 2) & 63;
        const int64_t rest = i >> 8;                  // t * (H/8) + j4
        const int j4 = static_cast<int>(rest % (H / 8)), t = static_cast<int>(rest / (H / 8));
        const int kh = lane >> 5, l31 = lane & 31;
        const int k = 8 * j4 + 2 * jj + kh, h = 32 * t + l31;
        Wp[i] = W1[static_cast<int64_t>(h) * 2 * H + k];
    } else {
        const int64_t i = (static_cast<int64_t>(blockIdx.x) - n_w) * kT + threadIdx.x;
        if (i >= N * H) return;
        const int64_t n = i / H;
        const int k = static_cast<int>(i - n * H);
        Ceo[n * H + (k & 1) * (H / 2) + (k >> 1)] = codes[i];
    }
}

template <int NT>
__global__ void __launch_bounds__(kT, 3) edge_score_stream_kernel(ScoreArgs a, const float* __restrict__ Wp, const float* __restrict__ Ceo) {
    constexpr int H = 32 * NT;
    constexpr int NTW = NT / 2;
    constexpr int NJ4 = H / 8;
    __shared__ float zpart[2][kBM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int eg = wave & 1, hh = wave >> 1;
    const int kh = lane >> 5, l31 = lane & 31;
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kBM;
    if (a.dyn_n) {
        const int64_t live_n = *a.dyn_n;
        if (live_n < a.n) a.n = live_n;
        if (row0 >= a.n) return;
    }
    const int el = 32 * eg + l31;
    const int64_t r = row0 + el;
    const bool live = r < a.n;
    int s = 0, d = 0;
    int64_t eg_id = 0;
    if (live) {
        eg_id = a.active ? a.active[r] : r;
        s = static_cast<int>(a.src[eg_id]);
        d = static_cast<int>(a.dst[eg_id]);
    }
    const float4* xp = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(s) * H + kh * (H / 2));
    const float4* yp = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(d) * H + kh * (H / 2));
    const float4* wp = reinterpret_cast<const float4*>(Wp) + (static_cast<int64_t>(hh) * NTW * NJ4) * 64 + lane;

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    float4 A0[NTW], A1[NTW], x0, y0, x1, y1;
    auto load = [&](int j4, float4 (&A)[NTW], float4& x, float4& y) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) A[t] = wp[(static_cast<int64_t>(t) * NJ4 + j4) * 64];
        x = xp[j4];
        y = yp[j4];
    };
    auto mma = [&](const float4 (&A)[NTW], const float4& x, const float4& y) {
        const float b[4] = {x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float av = jj == 0 ? A[t].x : jj == 1 ? A[t].y : jj == 2 ? A[t].z : A[t].w;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[jj], acc[t], 0, 0, 0);
            }
        }
    };
    // (Measured alternative, rejected: unconditional refills of both stages pinned with sched_barrier give exact
    //  vmcnt(8)/vmcnt(5) waits instead of the vmcnt(0) the conditional refill below forces at its join, but every stage
    //  then gets only one 16-MFMA group of cover and the kernel drops from 102 to 78 TFLOP/s: with ~13 TB/s of L1->L2
    //  requests in flight the load latency is well above one group, and the schedule below issues stage 1 in the MIDDLE of
    //  stage 0's MFMA group, which is what matters.)
    load(0, A0, x0, y0);
#pragma unroll 1
    for (int j4 = 0; j4 < NJ4; j4 += 2) {
        load(j4 + 1, A1, x1, y1);                    // NJ4 is even (H % 16 == 0)
        mma(A0, x0, y0);
        if (j4 + 2 < NJ4) load(j4 + 2, A0, x0, y0);
        mma(A1, x1, y1);
    }

    // ---- epilogue: hidden unit hh*H/2 + 32t + (q&3) + 8(q>>2) + 4kh.  Sixteen (tile, group) steps per wave, each
    // needing the endpoint rows of U (two gathers) + b1 + w2.  The loads of step i+1 are issued before step i is
    // processed (two register stages; the A/B operand stages are dead by now), so a wave
    // exposes ONE gather latency here instead of sixteen; letting the compiler hoist freely blows the register budget.
    const uint32_t rkey = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + eg_id));
    const float* Us = a.U + static_cast<int64_t>(s) * H + hh * (H / 2) + 4 * kh;
    const float* Ud = a.U + static_cast<int64_t>(d) * H + hh * (H / 2) + 4 * kh;
    const float* b1p = a.b1 + hh * (H / 2) + 4 * kh;
    const float* w2p = a.w2 + hh * (H / 2) + 4 * kh;
    struct Epi { float4 us, ud, bb, ww; };
    auto eload = [&](int i, Epi& L) {                // i = 4 t + g -> offset 32 t + 8 g = 8 i
        L.us = *reinterpret_cast<const float4*>(Us + 8 * i);
        L.ud = *reinterpret_cast<const float4*>(Ud + 8 * i);
        L.bb = *reinterpret_cast<const float4*>(b1p + 8 * i);
        L.ww = *reinterpret_cast<const float4*>(w2p + 8 * i);
    };
    float z = 0.f;
    auto estep = [&](int i, const Epi& L) {
        const int t = i >> 2, g = i & 3;
        const int hb = hh * (H / 2) + 8 * i + 4 * kh;
        const float u4[4] = {L.us.x - L.ud.x, L.us.y - L.ud.y, L.us.z - L.ud.z, L.us.w - L.ud.w};
        const float b4[4] = {L.bb.x, L.bb.y, L.bb.z, L.bb.w};
        const float w4[4] = {L.ww.x, L.ww.y, L.ww.z, L.ww.w};
        uint32_t bits[2] = {0u, 0u};
        if (a.use_drop) {
            bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
            bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = (acc[t][4 * g + j] + u4[j]) + b4[j];
            float m = v > 0.f ? 1.f : 0.f;
            if (a.use_drop) {
                const uint32_t draw = (j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu);
                m = draw >= a.drop_thresh ? m * a.drop_scale : 0.f;
            }
            z = fmaf(w4[j], v * m, z);
        }
    };
    const int Hrt = a.H;
    Epi L0, L1;
    eload(0, L0);
#pragma unroll
    for (int i = 0; i < 4 * NTW; i += 2) {
        // the (always true) run-time bounds keep every step in its own basic block: the next step's loads are issued
        // at its head and cannot be hoisted further, which bounds the live registers to two stages
        if (8 * i < Hrt) {
            eload(i + 1, L1);
            estep(i, L0);
        }
        if (8 * i + 8 < Hrt) {
            if (i + 2 < 4 * NTW) eload(i + 2, L0);
            estep(i + 1, L1);
        }
    }
    z += __shfl_xor(z, 32, 64);
    if (kh == 0) zpart[hh][el] = z;
    __syncthreads();
    if (live && hh == 0 && kh == 0) {
        const float zz = (zpart[0][el] + zpart[1][el]) + a.b2[0];
        a.p_out[r] = 1.0f / (1.0f + expf(-zz));
    }
}

// ---------------------------------------------------------------------------------------------
// Forward variant D ("stream, 64-edge wave tile").  tools/ceilings/ceilings.hip isolates the main loop of variant B and
// shows that it -- not the gathers, the dropout hash, the epilogue or the tail -- caps the kernel: with a 32-edge x
// 128-hidden wave tile every 16 MFMAs need 6 operand loads, and the L1/L2 request stream (~13 TB/s) inflates the load
// latency beyond what three waves per SIMD can cover (109 TFLOP/s for the bare loop).  Doubling the edges per wave
// (64 x 128: 8 accumulators, 2 waves per SIMD) reuses every A value twice: 8 loads per 32 MFMAs, 130 TFLOP/s for the
// bare loop.  Workgroup = 128 edges x all H hidden units: wave = (edge pair-group, hidden half).
constexpr int kBM2 = 128;
// BWD: the backward core on the same loop (recompute, then dv / feat / dz / per-64-edge-tile sums of dz * hidden).
template <int NT, bool BWD>
__global__ void __launch_bounds__(kT, 2) edge_score_stream64_kernel(ScoreArgs a, const float* __restrict__ Wp, const float* __restrict__ Ceo) {
    constexpr int H = 32 * NT;
    constexpr int NTW = NT / 2;
    constexpr int NJ4 = H / 8;
    __shared__ float zpart[2][kBM2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ep = wave & 1, hh = wave >> 1;
    const int kh = lane >> 5, l31 = lane & 31;
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kBM2;
    if (a.dyn_n) {
        const int64_t live_n = *a.dyn_n;
        if (live_n < a.n) a.n = live_n;
        if (row0 >= a.n) return;
    }
    int el[2], s[2], d[2];
    int64_t eg_id[2];
    bool live[2];
    const float4 *xp[2], *yp[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        el[g] = 64 * ep + 32 * g + l31;
        const int64_t r = row0 + el[g];
        live[g] = r < a.n;
        s[g] = 0; d[g] = 0; eg_id[g] = 0;
        if (live[g]) {
            eg_id[g] = a.active ? a.active[r] : r;
            s[g] = static_cast<int>(a.src[eg_id[g]]);
            d[g] = static_cast<int>(a.dst[eg_id[g]]);
        }
        xp[g] = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(s[g]) * H + kh * (H / 2));
        yp[g] = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(d[g]) * H + kh * (H / 2));
    }
    const float4* wp = reinterpret_cast<const float4*>(Wp) + (static_cast<int64_t>(hh) * NTW * NJ4) * 64 + lane;

    f32x16 acc[2][NTW];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[g][t][q] = 0.f;

    float4 A0[NTW], A1[NTW], x0[2], y0[2], x1[2], y1[2];
    auto load = [&](int j4, float4 (&A)[NTW], float4 (&x)[2], float4 (&y)[2]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) A[t] = wp[(static_cast<int64_t>(t) * NJ4 + j4) * 64];
#pragma unroll
        for (int g = 0; g < 2; ++g) { x[g] = xp[g][j4]; y[g] = yp[g][j4]; }
    };
    auto mma = [&](int j4, const float4 (&A)[NTW], const float4 (&x)[2], const float4 (&y)[2]) {
        if (BWD && hh == 0) {
            // feat[e, k] = x_s[k] x_d[k] for the weight gradient: this lane holds k = 8 j4 + 2 jj + kh (jj = 0..3); the two
            // half-waves swap two values each so that every lane stores four CONSECUTIVE k of its edge as one 16-byte row piece
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float o0 = x[g].x * y[g].x, o1 = x[g].y * y[g].y, o2 = x[g].z * y[g].z, o3 = x[g].w * y[g].w;
                const float v0 = __shfl_xor(kh ? o0 : o2, 32, 64), v1 = __shfl_xor(kh ? o1 : o3, 32, 64);
                const float4 row = kh ? make_float4(v0, o2, v1, o3) : make_float4(o0, v0, o1, v1);
                if (live[g]) *reinterpret_cast<float4*>(a.feat + (row0 + el[g]) * H + 8 * j4 + 4 * kh) = row;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float bx = jj == 0 ? x[g].x : jj == 1 ? x[g].y : jj == 2 ? x[g].z : x[g].w;
                const float by = jj == 0 ? y[g].x : jj == 1 ? y[g].y : jj == 2 ? y[g].z : y[g].w;
                const float b = bx * by;
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const float av = jj == 0 ? A[t].x : jj == 1 ? A[t].y : jj == 2 ? A[t].z : A[t].w;
                    acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[g][t], 0, 0, 0);
                }
            }
        }
    };
    load(0, A0, x0, y0);
#pragma unroll 1
    for (int j4 = 0; j4 < NJ4; j4 += 2) {
        load(j4 + 1, A1, x1, y1);                    // NJ4 is even (H % 16 == 0)
        mma(j4, A0, x0, y0);
        if (j4 + 2 < NJ4) load(j4 + 2, A0, x0, y0);
        mma(j4 + 1, A1, x1, y1);
    }

    // ---- epilogue (as variant B, once per edge group): hidden unit hh*H/2 + 8 i + (j) + 4 kh for step i = 4 t + g4
    const int Hrt = a.H;
    const float* b1p = a.b1 + hh * (H / 2) + 4 * kh;
    const float* w2p = a.w2 + hh * (H / 2) + 4 * kh;
    struct Epi { float4 us, ud, bb, ww; };
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t rkey = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + eg_id[g]));
        const float* Us = a.U + static_cast<int64_t>(s[g]) * H + hh * (H / 2) + 4 * kh;
        const float* Ud = a.U + static_cast<int64_t>(d[g]) * H + hh * (H / 2) + 4 * kh;
        auto eload = [&](int i, Epi& L) {
            L.us = *reinterpret_cast<const float4*>(Us + 8 * i);
            L.ud = *reinterpret_cast<const float4*>(Ud + 8 * i);
            L.bb = *reinterpret_cast<const float4*>(b1p + 8 * i);
            L.ww = *reinterpret_cast<const float4*>(w2p + 8 * i);
        };
        float z = 0.f;
        auto estep = [&](int i, const Epi& L) {
            const int t = i >> 2, g4 = i & 3;
            const int hb = hh * (H / 2) + 8 * i + 4 * kh;
            const float u4[4] = {L.us.x - L.ud.x, L.us.y - L.ud.y, L.us.z - L.ud.z, L.us.w - L.ud.w};
            const float b4[4] = {L.bb.x, L.bb.y, L.bb.z, L.bb.w};
            const float w4[4] = {L.ww.x, L.ww.y, L.ww.z, L.ww.w};
            uint32_t bits[2] = {0u, 0u};
            if (a.use_drop) {
                bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
                bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = (acc[g][t][4 * g4 + j] + u4[j]) + b4[j];
                float m = v > 0.f ? 1.f : 0.f;
                if (a.use_drop) {
                    const uint32_t draw = (j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu);
                    m = draw >= a.drop_thresh ? m * a.drop_scale : 0.f;
                }
                const float hd = v * m;                              // dropout(relu(v))
                z = fmaf(w4[j], hd, z);
                if (BWD) acc[g][t][4 * g4 + j] = hd;                 // kept for the second epilogue pass
            }
        };
        Epi L0, L1;
        eload(0, L0);
#pragma unroll
        for (int i = 0; i < 4 * NTW; i += 2) {
            if (8 * i < Hrt) {                       // always true: one basic block per step bounds the live registers
                eload(i + 1, L1);
                estep(i, L0);
            }
            if (8 * i + 8 < Hrt) {
                if (i + 2 < 4 * NTW) eload(i + 2, L0);
                estep(i + 1, L1);
            }
        }
        z += __shfl_xor(z, 32, 64);
        if (kh == 0) zpart[hh][el[g]] = z;
    }
    __syncthreads();
    if (!BWD) {
        if (hh == 0 && kh == 0) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (live[g]) {
                    const float zz = (zpart[0][el[g]] + zpart[1][el[g]]) + a.b2[0];
                    a.p_out[row0 + el[g]] = 1.0f / (1.0f + expf(-zz));
                }
            }
        }
        return;
    }
    // ---- backward epilogue: dz = gp p (1-p);  dv = dz w2 relu' keep scale;  per-64-edge-tile sums of dz * hidden
    float dzv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float zz = (zpart[0][el[g]] + zpart[1][el[g]]) + a.b2[0];
        const float p = 1.0f / (1.0f + expf(-zz));
        dzv[g] = live[g] ? a.gp[row0 + el[g]] * p * (1.0f - p) : 0.f;             // 0 on the padding rows of the last tile
        if (live[g] && hh == 0 && kh == 0) a.dz[row0 + el[g]] = dzv[g];
    }
    // this wave's 64 edges are exactly one tile of sgs_edge_score_bwd_tile() rows: row 2 blockIdx + ep of hdz_part, its hidden half
    const int64_t prow = 2 * static_cast<int64_t>(blockIdx.x) + ep;
    const bool prow_ok = prow * 64 < a.n;
    const float dscale = a.use_drop ? a.drop_scale : 1.f;
#pragma unroll
    for (int i = 0; i < 4 * NTW; ++i) {
        if (8 * i < Hrt) {                           // always true: one basic block per step (register budget, as above)
            const int t = i >> 2, g4 = i & 3;
            const int hb = hh * (H / 2) + 8 * i + 4 * kh;
            const float4 ww = *reinterpret_cast<const float4*>(w2p + 8 * i);
            const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
            float hs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float dv4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float hd = acc[g][t][4 * g4 + j];
                    dv4[j] = dzv[g] * w4[j] * (hd > 0.f ? dscale : 0.f);     // hd > 0 <=> v > 0 and kept
                    hs[j] = fmaf(dzv[g], hd, hs[j]);
                }
                if (live[g]) *reinterpret_cast<float4*>(a.dv + (row0 + el[g]) * H + hb) = make_float4(dv4[0], dv4[1], dv4[2], dv4[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = hs[j];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);   // the 32 lanes of this half-wave (equal kh)
                if (l31 == 0 && prow_ok) a.hdz[prow * H + hb + j] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward variant E ("bf16x6"): the same fp32 contraction on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into
// three bf16 pieces by round-to-nearest residuals, v = v1 + v2 + v3 with |v2| <= 2^-8 |v|, |v3| <= 2^-16 |v| (the last
// residual has at most 8 significant bits left, so it converts exactly); the product w f is then the sum of nine bf16 x bf16
// products, each exact in the fp32 accumulator, and the six of order <= 2 (w1 f1, w1 f2, w2 f1, w1 f3, w2 f2, w3 f1) are
// kept: the three dropped ones are <= 2^-23 |w f| together, i.e. below the rounding error of ONE fp32 multiply, so the
// result is fp32-faithful (tests/test_gpu_edge_score.py measures it against an fp64 evaluation next to the fp32 MFMA kernels).
// v_mfma_f32_32x32x16_bf16 retires 16x the flops per cycle of v_mfma_f32_32x32x2_f32, so six of them per fp32 product leave
// 16/6 = 2.7x the fp32 matrix rate.
//   * W1a is split once per launch into Wp16[kc][tile t][piece][lane][8 bf16]: the A operand (hidden rows) of lane
//     (l31, kh) for k = 16 kc + 8 kh .. + 7 is one 16-byte word per piece, and a k-chunk of all H hidden units is one
//     contiguous block (24 KiB at H = 256) that the workgroup copies into LDS linearly (double-buffered, one barrier per
//     chunk) and every wave reads back lane-linearly (ds_read_b128, conflict-free);
//   * a wave owns 32 edges and ALL H hidden units (8 accumulator tiles = 128 registers): its B operand is eight consecutive
//     fp32 of each endpoint's row of the plain node codes (two 16-byte loads per endpoint per chunk, no re-layout), multiplied
//     and split in registers (52 vector instructions per 48 MFMAs), and the fc2 reduction never leaves the wave.
// Requires H % 128 == 0.
// rowscale (TRANSPOSED only): the operand is diag(rowscale * scale) W1a, i.e. entry (k, h) is scaled by rowscale[k] * scale before it is
// split -- the mask form of dfeat folds fc2's weights and the dropout scale into the matrix (MODE 4).
template <bool TRANSPOSED = false>      // TRANSPOSED: the pieces of W1a^T (row h of the operand = column h of W1a): the row-GEMM mode's operand
__global__ void __launch_bounds__(kT) pack_w1a_bf16x3(const float* __restrict__ W1, int H, uint4* __restrict__ Wp16,
                                                     const float* __restrict__ rowscale = nullptr, float scale = 1.f) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;       // one (kc, t, lane)
    const int NTl = H / 32;
    if (i >= static_cast<int64_t>(H / 16) * NTl * 64) return;
    const int lane = i & 63;
    const int64_t rest = i >> 6;
    const int t = static_cast<int>(rest % NTl), kc = static_cast<int>(rest / NTl);
    const int h = 32 * t + (lane & 31), k0 = 16 * kc + 8 * (lane >> 5);
    uint32_t p[3][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float w0 = TRANSPOSED ? W1[static_cast<int64_t>(k0 + 2 * m) * 2 * H + h] : W1[static_cast<int64_t>(h) * 2 * H + k0 + 2 * m];
        float w1 = TRANSPOSED ? W1[static_cast<int64_t>(k0 + 2 * m + 1) * 2 * H + h] : W1[static_cast<int64_t>(h) * 2 * H + k0 + 2 * m + 1];
        if (TRANSPOSED && rowscale) { w0 *= rowscale[k0 + 2 * m] * scale; w1 *= rowscale[k0 + 2 * m + 1] * scale; }
        split3(w0, w1, p[0][m], p[1][m], p[2][m]);
    }
    uint4* o = Wp16 + (static_cast<int64_t>(kc) * NTl + t) * 3 * 64 + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * 64] = make_uint4(p[c][0], p[c][1], p[c][2], p[c][3]);
}

// Measured (MI355X, E = 351 194, H = 256; launch = pack + kernel): 0.25-0.29 ms by device against 0.42-0.47 ms for variant D
// in the same process, i.e. 1.1 PFLOP/s of bf16 MFMA work executed.  Timing probes with one ingredient removed each (WRONG
// results, timing only; they are not kept in the source) moved the launch by: same-row U gathers -4 %, no feature loads
// -4 %, no W staging -7 %, no barriers +-0, no operand-split arithmetic -1..-4 %, consecutive MFMAs on different accumulators
// +-0, every tile reading ONE tile's fragments +10 % (slower), second-slot workgroups started half a lifetime late +-0,
// 8-wave workgroups (half the W traffic) +2 %, two chunks per phase (twice the prefetch distance, 96 KiB of LDS) +5 %,
// and no epilogue at all -19 %.  No stall source explains the distance to the MFMA bound (0.11 ms at 2.4 GHz): the kernel
// runs at an effective 2.05 GHz under the counters and the guide's own bf16 loops on random data hold 1.5-1.7 GHz, so it
// is priced against the clock the chip gives a dense bf16 MFMA stream rather than against stalls.  Counters:
// SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x the 8.43 M MFMAs issued, 42 % of the SIMD cycles; no LDS bank conflicts.
// MODE 1: the backward core on the same loop (recompute over the active rows, then dv / feat / dz / per-64-edge-tile sums of
// dz * hidden, exactly what edge_score_kernel<NT, true> produces).
// MODE 2: the loop as a row GEMM, out[r, :] = in[r, :] . M for a dense [n, H] input (a.codes -> a.feat): the backward's
// dfeat = dv W1a, with `Wp16` packed from the TRANSPOSED matrix (pack_w1a_bf16x3<true>); no gather, no product, the
// accumulators are the result.
// MODE 3: the PAIRED forward.  fc1's heavy half W1a (x_s * x_d) is symmetric in the endpoints: on an undirected graph stored in
// both directions (every dataset of the reference: datasets.py:189-190 to_undirected) the edge (s -> d) and its mate (d -> s)
// share that contraction bit for bit (a * b == b * a in fp32, same operands, same order of accumulation) and differ only in the
// sign of the node-level term U[s] - U[d] and in their dropout rows.  So only the canonical edge of every mated pair (and every
// unmated edge) runs the main loop -- about half of the candidate edges -- and the epilogue finishes BOTH scores from the one
// set of accumulators: the same p as MODE 0, bit for bit, at ~0.6 x the time.
// acc <- (acc << 1) | [this lane's bit of `mask`] as ONE vector instruction: v_addc_co_u32 acc, acc + acc + carry-in, the carry-in taken
// per lane from a 64-bit lane mask (the result of a vector compare).  The scorer's forward epilogue builds its ReLU x dropout mask words
// with it: one instruction per hidden unit instead of min / shift / or.
// SALU_MASK: `mask` was produced by a SCALAR instruction (the s_and_b64 of two compare results, in the dropout variants): the two-wait-state
// rule is about a VECTOR instruction's SGPR result read by the next vector instruction, so no pad is needed -- eight s_nop per epilogue step.
template <bool SALU_MASK = false>
__device__ __forceinline__ uint32_t shift_in_bit(uint32_t acc, uint64_t mask) {
    uint32_t out;
    uint64_t carry_out;
    if constexpr (SALU_MASK) {
        asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(carry_out) : "v"(acc), "s"(mask));
        return out;
    }
    // (s_nop 1: gfx950 wants two wait states between a vector compare's SGPR result and a vector instruction that reads it as a mask; the
    //  compiler pads its own instructions but does not look inside inline assembly -- without the pad the mask words of a no-dropout
    //  forward came out wrong in a few lanes)
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(carry_out) : "v"(acc), "s"(mask));
    return out;
}

// Stamps INSIDE a phase of the main loop, and the no-barrier timing probe, exist only in a -DSGS_PHASE_PROBE=1 build (SGS_PHASE_PROBE=1
// python sgs-gnn_amd/build.py): their branches cut the phase into basic blocks, across which the compiler sinks the operand split out of the
// MFMA gaps it was placed in.
#ifndef SGS_PHASE_PROBE
#define SGS_PHASE_PROBE 0
#endif
#if SGS_PHASE_PROBE
#define SGS_PHASE_STAMP(on, k, pre) do { if (on) asm volatile(pre "s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst[k]) :: "memory"); } while (0)
#define SGS_PHASE_BARRIER() do { if (!(a.prio & 4)) __syncthreads(); } while (0)
#else
#define SGS_PHASE_STAMP(on, k, pre) do { (void)(on); } while (0)
#define SGS_PHASE_BARRIER() __syncthreads()
#endif
#define SGS_STAMP(k) do { if (a.trace && tid == 0) a.trace[8 * blockIdx.x + (k)] = __builtin_readcyclecounter(); } while (0)

template <int NT, int NW, int MODE = 0>
__global__ void __launch_bounds__(64 * NW, 8 / NW) edge_score_bf16x6_kernel(ScoreArgs a, const uint4* __restrict__ Wp16) {
    constexpr bool BWD = MODE == 1, FUSED = MODE == 5, GEMMB = MODE == 4 || MODE == 5, GEMM = MODE == 2 || GEMMB, PAIR = MODE == 3;
    constexpr int H = 32 * NT;
    constexpr int NPH = H / 16;              // phases: one 16-deep k-chunk each, one barrier per phase
    constexpr int CH = NT * 3 * 64;          // 16-byte words per k-chunk of W1a (all hidden units, three pieces)
    constexpr int TH = 64 * NW;              // NW waves x 32 edges per workgroup, all reading the same staged chunks of W1a
    constexpr int SPT = CH / TH;             // 16-byte words copied per thread per phase
    static_assert(CH % TH == 0 && (SPT == 3 || SPT == 6), "unsupported shape");
    __shared__ uint4 wl[2][CH];
    __shared__ __attribute__((aligned(16))) float bw[2][H];          // b1, w2 for the epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, l31 = lane & 31;
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * (32 * NW);
    if (a.dyn_n) {
        const int64_t live_n = *a.dyn_n;
        if (live_n < a.n) a.n = live_n;
        if (row0 >= a.n) return;                                  // (uniform per workgroup, before any barrier)
    }
    // Two workgroups share a CU, one wave of each per SIMD.  Started together they stay in step: both in the MFMA loop (sharing the matrix
    // pipe), then both in the epilogue (the pipe idle; counters, round 3: busy 42 % of the cycles).  The workgroups that take the SECOND
    // slot of every CU at kernel start (256 .. 511 in dispatch order) therefore sleep for about half a main loop first; every later
    // workgroup starts when a slot frees and inherits the offset.  Shader-clock stamps (tools/stagger_trace.py, E = 351 194): main loop
    // 60.6k -> 46.2k cycles per workgroup, kernel 234 -> 199 us.  Only for launches of >= 1 024 live workgroups (two per slot): the sleep
    // idles half the chip for ~20 us once.
    if (a.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512 && a.n >= static_cast<int64_t>(1024) * 32 * NW)
        for (int w = 0; w < a.stagger; ++w) __builtin_amdgcn_s_sleep(1);
    if ((a.prio & 3) == 1) __builtin_amdgcn_s_setprio(1);
    SGS_STAMP(0);
    if (a.trace && tid == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[8 * blockIdx.x + 4] = (static_cast<unsigned long long>(xcc) << 32) | hw;
    }
    if (!GEMM)
        for (int i = tid; i < H; i += TH) {                                               // (visible after the first barrier below)
            bw[0][i] = a.b1[i];
            // forward modes: fc2's weight with the dropout scale folded in (z += (w2 / (1 - p)) * relu(v) on the kept units)
            bw[1][i] = (!BWD && a.use_drop) ? a.w2[i] * a.drop_scale : a.w2[i];
        }
    const int64_t r = row0 + 32 * wave + l31;
    const bool live = r < a.n;
    int s = 0, d = 0;
    int64_t eg_id = 0;
    if (live && !GEMM) {
        eg_id = PAIR ? static_cast<int64_t>(a.canon[r]) : (a.active ? a.active[r] : r);
        s = static_cast<int>(a.src[eg_id]);
        d = static_cast<int>(a.dst[eg_id]);
    }
    const int64_t xrow = GEMM ? (live ? r : 0) : static_cast<int64_t>(s);          // MODE 2 reads its own row (row 0 on the padding rows)
    const float4* xp = reinterpret_cast<const float4*>(a.codes + xrow * H + 8 * kh);
    const float4* yp = reinterpret_cast<const float4*>(a.codes + static_cast<int64_t>(d) * H + 8 * kh);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    // MODE 4: the input row is a MASK (bit h of row r); lane (l31, kh) needs bits 16 kc + 8 kh .. + 7 of its row per chunk = byte 2 kc + kh
    const uint8_t* bp = GEMMB ? reinterpret_cast<const uint8_t*>(a.inbits + (live ? r : 0) * (H / 32)) + kh : nullptr;
    struct Feat { float4 xa, xb, ya, yb; uint32_t mb; };
    // W staging by LDS-DMA (global_load_lds_dwordx4: no register round trip, no ds_write pass), lane-linear: lane L of wave w moves 16-byte
    // word k TH + 64 w + L of the chunk to the same word of the stage.  Issued from inline assembly ON PURPOSE: next to a
    // __builtin_amdgcn_global_load_lds in flight hipcc 7.2 waits vmcnt(0) before every ds_read of the OTHER stage (it cannot tell the
    // stages apart), which drains the prefetch as soon as it is issued.  The assembly's loads are invisible to the compiler's counters;
    // every phase retires them itself (s_waitcnt vmcnt(0) ahead of the barrier that publishes the stage).  M0 = the LDS byte address.
    const uint32_t wl_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)(&wl[0][0])));
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t voff = static_cast<uint32_t>(tid) * 16u;
    // (M0 is written in the statement that uses it and never restored: nothing else in these kernels reads it -- checked in the ISA.)
    auto dma_one = [&](int ph, int stage, int k) {
        const uint4* gb = Wp16 + static_cast<int64_t>(ph) * CH + k * TH;                           // (wave-uniform)
        const uint32_t ld = wl_lds + static_cast<uint32_t>((stage * CH + k * TH) * 16) + wave_u * 1024u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(ld), "s"(gb) : "memory");
    };
    auto dma = [&](int ph, int stage) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) dma_one(ph, stage, k);
    };
    auto fload_x = [&](int kc, Feat& f) {
        if constexpr (GEMMB) { f.mb = bp[2 * kc]; return; }
        f.xa = xp[4 * kc]; f.xb = xp[4 * kc + 1];
    };
    auto fload_y = [&](int kc, Feat& f) {
        if constexpr (!GEMM) { f.ya = yp[4 * kc]; f.yb = yp[4 * kc + 1]; }
    };
    auto fload = [&](int kc, Feat& f) { fload_x(kc, f); fload_y(kc, f); };
    struct WF { uint4 q1, q2, q3; };
    Feat fa;
    dma(0, 0);
    fload(0, fa);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SGS_STAMP(1);
    // Software pipeline over the k-chunks.  Phase ph runs the 6 NT MFMAs of chunk ph on operand pieces that the PREVIOUS phase split, and in
    // the gaps behind its first MFMAs (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles: six or seven vector instructions fit)
    // splits the fp32 features of chunk ph + 1, then refills their registers with chunk ph + 2.  (Round 3, shader-clock stamps: with the
    // whole split ahead of the MFMAs and the first W fragments read just before them, a phase took ~2 800 cycles for 1 536 cycles of MFMAs.)
    struct Pieces { u32x4 F1, F2, F3; };
    struct SplitState { float a, b; uint32_t p1; float4 pa, pb; };
    // slot 2 j: products of pair j, its first piece and the remainders; slot 2 j + 1: second and third piece.  (split3(), in two halves.)
    auto split_slot = [&](int slot, Feat& f, int kc_, bool wr, Pieces& P, SplitState& st) {
        const int j = slot >> 1;
        if constexpr (GEMMB) {
            if (slot != 0) return;
            const uint32_t b = f.mb;                 // eight mask bits -> eight bf16 ones / zeros: ONE exact piece (3 MFMAs per tile)
#pragma unroll
            for (int m = 0; m < 4; ++m) P.F1[m] = ((b >> (2 * m)) & 1u ? 0x3F80u : 0u) | ((b >> (2 * m + 1)) & 1u ? 0x3F800000u : 0u);
            P.F2 = P.F1; P.F3 = P.F1;
        } else {
            if ((slot & 1) == 0) {
                const float4 x = j < 2 ? f.xa : f.xb, y = j < 2 ? f.ya : f.yb;
                const float x0 = (j & 1) ? x.z : x.x, x1 = (j & 1) ? x.w : x.y, y0 = (j & 1) ? y.z : y.x, y1 = (j & 1) ? y.w : y.y;
                st.a = GEMM ? x0 : x0 * y0;
                st.b = GEMM ? x1 : x1 * y1;
                if constexpr (BWD) {                 // feat[e, k] = x_s[k] x_d[k], k = 16 kc + 8 kh .. + 7, for the weight gradient
                    float4& pp = j < 2 ? st.pa : st.pb;
                    if (j & 1) { pp.z = st.a; pp.w = st.b; } else { pp.x = st.a; pp.y = st.b; }
                    if ((j & 1) && live && wr) reinterpret_cast<float4*>(a.feat + r * H + 16 * kc_ + 8 * kh)[j >> 1] = pp;
                }
                const uint32_t p1 = pk_bf16(st.a, st.b);
                st.a -= __uint_as_float(p1 << 16);
                st.b -= __uint_as_float(p1 & 0xFFFF0000u);
                P.F1[j] = p1;
            } else {
                const uint32_t p2 = pk_bf16(st.a, st.b);
                st.a -= __uint_as_float(p2 << 16);
                st.b -= __uint_as_float(p2 & 0xFFFF0000u);
                P.F2[j] = p2;
                P.F3[j] = pk_bf16(st.a, st.b);
            }
        }
    };
    constexpr int kSlots = GEMMB ? 1 : 8;
    auto split_feat = [&](Feat& f, int kc_, bool wr, Pieces& P) {
        SplitState st;
#pragma unroll
        for (int q = 0; q < kSlots; ++q) split_slot(q, f, kc_, wr, P, st);
    };
    unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // probe: stamps inside one phase (a.trace only)
    // One phase.  On entry `A` / `B` hold the W fragments of tiles 0 / 1 of the current stage (read at the END of the previous phase, behind
    // its last tile's MFMAs); fragments of tile t + 2 are read behind the MFMAs of tile t.  Before the LAST tile: the DMA of the next stage
    // is retired, one barrier publishes it (and says every wave is through reading the stage after next's victim), and the next phase's
    // first fragments are requested -- A is free (its last user, tile NT - 2, has issued), the odd set alternates between B and `Cn`.
    auto chunk = [&](const uint4* wcur, const uint4* wnx, int stage_next, WF& A, WF& B, WF& Cn, const Pieces& C, Pieces& Nx, Feat& f, int kc1,
                     int kn_, bool wr, bool tr) {
        const bf16x8 f1 = __builtin_bit_cast(bf16x8, C.F1), f2 = __builtin_bit_cast(bf16x8, C.F2), f3 = __builtin_bit_cast(bf16x8, C.F3);
        auto wload = [&](const uint4* wsrc, int t, WF& w) {
            w.q1 = wsrc[(t * 3 + 0) * 64 + lane];
            w.q2 = wsrc[(t * 3 + 1) * 64 + lane];
            w.q3 = wsrc[(t * 3 + 2) * 64 + lane];
        };
        constexpr int kPerTile = GEMMB ? 3 : 6;
        auto one = [&](int t, const WF& w, int m) {          // smallest terms first
            const bf16x8 w1 = __builtin_bit_cast(bf16x8, w.q1), w2 = __builtin_bit_cast(bf16x8, w.q2), w3 = __builtin_bit_cast(bf16x8, w.q3);
            if constexpr (GEMMB) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m == 0 ? w3 : m == 1 ? w2 : w1, f1, acc[t], 0, 0, 0);
            } else {
                const bf16x8 wm = m == 0 ? w3 : (m == 1 || m == 3) ? w2 : w1;
                const bf16x8 fm = (m == 0 || m == 3 || m == 5) ? f1 : (m == 1 || m == 4) ? f2 : f3;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, fm, acc[t], 0, 0, 0);
            }
        };
        SplitState st;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == NT - 1) {
                SGS_PHASE_STAMP(tr, 1, "s_waitcnt lgkmcnt(0)\n\t");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // the next stage has landed (this wave's part)
                SGS_PHASE_STAMP(tr, 2, "");
                SGS_PHASE_BARRIER();
                SGS_PHASE_STAMP(tr, 3, "");
                wload(wnx, 0, A);
                wload(wnx, 1, Cn);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < kPerTile; ++m) {
                const int slot = t * kPerTile + m;
                if (t & 1) one(t, B, m); else one(t, A, m);
                // one piece of work per MFMA gap: the split, then the refill of the (now dead) raw features two chunks ahead, then the DMA of
                // the next phase's W, one 4 KiB instruction per gap (behind the feature loads: vmcnt retires in order)
                if (slot < kSlots + 2 + SPT) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot < kSlots) split_slot(slot, f, kc1, wr, Nx, st);
                    else if (slot == kSlots) fload_x(kn_, f);
                    else if (slot == kSlots + 1) fload_y(kn_, f);
                    else dma_one(kc1, stage_next, slot - kSlots - 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (t + 2 < NT) { if (t & 1) wload(wcur, t + 2, B); else wload(wcur, t + 2, A); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    Pieces pc, pn_;
    WF wa, wb, wc;
    {
        const uint4* w0 = wl[0];
        wa.q1 = w0[lane]; wa.q2 = w0[64 + lane]; wa.q3 = w0[128 + lane];
        wb.q1 = w0[192 + lane]; wb.q2 = w0[256 + lane]; wb.q3 = w0[320 + lane];
    }
    split_feat(fa, 0, true, pc);
    fload(NPH > 1 ? 1 : 0, fa);
    static_assert(NPH % 2 == 0, "the phase loop is unrolled by two (the odd fragment set and the operand pieces alternate)");
#pragma unroll 1
    for (int ph = 0; ph < NPH; ph += 2) {
        // (the last phases reload / re-split their own chunks: no branches around the loads)
        const bool tr = a.trace != nullptr && ph == NPH / 2;
        SGS_PHASE_STAMP(tr, 0, "");
        chunk(wl[0], wl[1], 1, wa, wb, wc, pc, pn_, fa, ph + 1, ph + 2 < NPH ? ph + 2 : NPH - 1, true, tr);
        SGS_PHASE_STAMP(tr, 4, "");
        chunk(wl[1], wl[0], 0, wa, wc, wb, pn_, pc, fa, ph + 2 < NPH ? ph + 2 : NPH - 1, ph + 3 < NPH ? ph + 3 : NPH - 1, ph + 2 < NPH, false);
    }
#if SGS_PHASE_PROBE
    if (a.trace && lane == 0)
        for (int k = 0; k < 5; ++k) a.trace[8 * (static_cast<int64_t>(gridDim.x) + 4 * blockIdx.x + wave) + k] = pst[k];
#endif
    __syncthreads();          // (MODE 5 reuses the stages as wave-private tiles; the epilogues' b1 / w2 reads need no more than the first barrier)
    if constexpr (FUSED) {
        // MODE 5: dfeat[r, :] = dz[r] * acc never reaches memory as such.  The two endpoint reductions of the scorer backward need
        //   d codes[src r, :] += dfeat[r, :] * codes[dst r, :]      and      d codes[dst r, :] += dfeat[r, :] * codes[src r, :].
        // The active rows are sorted by source (a drawn subset of a row-sorted edge list, in edge order), so the first sum runs over
        // CONSECUTIVE rows: it is reduced here, per wave, as a segmented sum down the 32 rows of the tile (through LDS, one column per
        // lane, fixed order) and only the RUN-END rows -- the last row of a source inside a wave -- write their partial sum, into slot
        // (r >> 5) + src of `opart` (strictly increasing along the run ends, so no two collide; a node with b - a out-rows has
        // ((b - 1) >> 5) - (a >> 5) + 1 of them, which the follow-up reduction adds in order).  The second sum gathers by destination:
        // its rows G[r, :] = dfeat[r, :] * codes[src r, :] are written out (a.feat) and reduced by scorer_bwd_reduce.
        // Against writing dfeat and reading it back twice: one [n, H] read and both codes gathers of the reduction are gone.
        const float rs = live ? a.indz[r] : 0.f;
        int sr = 0, dr = 0;
        if (live) { sr = a.sd[2 * r]; dr = a.sd[2 * r + 1]; }
        const int nxt = __shfl(sr, (lane & 32) | ((l31 + 1) & 31), 64);
        const bool endrow = live && (l31 == 31 || r + 1 >= a.n || nxt != sr);
        const uint32_t endmask = static_cast<uint32_t>(__ballot(endrow && kh == 0));      // bit = row of the tile (lanes 0..31)
        constexpr int LS = 68;                                                           // padded row: 16-byte stores of 8 rows hit 32 distinct banks
        // the W stages are dead (the loop ended on a barrier): at H = 256 they hold the four waves' tiles; at H = 128 they are too small
        constexpr bool kInWl = sizeof(wl) >= static_cast<size_t>(NW) * 32 * LS * sizeof(float);
        __shared__ float ot_own[kInWl ? 1 : NW * 32 * LS];
        float* ot = (kInWl ? reinterpret_cast<float*>(&wl[0][0]) : ot_own) + wave * (32 * LS);
        const float* cs = a.codes + static_cast<int64_t>(sr) * H + 4 * kh;
        const float* cd = a.codes + static_cast<int64_t>(dr) * H + 4 * kh;
        const int64_t wrow = (row0 + 32 * wave) >> 5;                                    // this wave's index among all 32-row tiles
#pragma unroll
        for (int Q = 0; Q < NT / 2; ++Q) {                                               // 64 columns at a time
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const int i = 8 * Q + ii, t = i >> 2, g4 = i & 3;
                const float4 s4 = *reinterpret_cast<const float4*>(cs + 8 * i);
                const float4 d4 = *reinterpret_cast<const float4*>(cd + 8 * i);
                const float f0 = rs * acc[t][4 * g4], f1 = rs * acc[t][4 * g4 + 1], f2 = rs * acc[t][4 * g4 + 2], f3 = rs * acc[t][4 * g4 + 3];
                if (live) *reinterpret_cast<float4*>(a.feat + r * H + 8 * i + 4 * kh) = make_float4(f0 * s4.x, f1 * s4.y, f2 * s4.z, f3 * s4.w);
                *reinterpret_cast<float4*>(ot + l31 * LS + 8 * ii + 4 * kh) = make_float4(f0 * d4.x, f1 * d4.y, f2 * d4.z, f3 * d4.w);
            }
            // the tile is wave-private and a wave's LDS operations execute in issue order: no workgroup barrier, only the compiler is
            // told not to move the reads above the writes (or the next pass's writes above these reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float col[32];
#pragma unroll
            for (int row = 0; row < 32; ++row) col[row] = ot[row * LS + lane];            // all 32 reads in flight before the (branchy) walk below
            float run = 0.f;
#pragma unroll
            for (int row = 0; row < 32; ++row) {
                run += col[row];
                if ((endmask >> row) & 1u) {                                             // (wave-uniform)
                    const int64_t slot = wrow + __builtin_amdgcn_readlane(sr, row);
                    a.opart[slot * H + 64 * Q + lane] = run;
                    run = 0.f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
    if constexpr (GEMM) {                            // out[r, 8 i + 4 kh + j] = accumulator register 4 g4 + j of tile t (i = 4 t + g4)
        if (live) {
            const float rs = GEMMB ? a.indz[r] : 1.f;        // MODE 4: the row's dz was left out of the mask operand
#pragma unroll
            for (int i = 0; i < 4 * NT; ++i)
                *reinterpret_cast<float4*>(a.feat + r * H + 8 * i + 4 * kh) =
                    GEMMB ? make_float4(rs * acc[i >> 2][4 * (i & 3)], rs * acc[i >> 2][4 * (i & 3) + 1], rs * acc[i >> 2][4 * (i & 3) + 2],
                                        rs * acc[i >> 2][4 * (i & 3) + 3])
                          : make_float4(acc[i >> 2][4 * (i & 3)], acc[i >> 2][4 * (i & 3) + 1], acc[i >> 2][4 * (i & 3) + 2], acc[i >> 2][4 * (i & 3) + 3]);
        }
        return;
    }

    // ---- epilogue (as variants B / D): hidden unit 8 i + j + 4 kh for step i = 4 t + g4; the wave holds every hidden unit.
    // Counters showed the waves of this kernel waiting on memory for 60 % of their life, and most of that here: with one step
    // of look-ahead each of the 32 steps exposed a gather latency.  The main loop's operand registers are dead now, so the
    // endpoint rows of U are gathered kPF steps ahead, and b1 / w2 (the same for every edge) come from LDS.
    constexpr int kPF = BWD ? 8 : (PAIR ? 5 : 6);          // (forward: four straight-line copies of the step loop share the register file)
    // Every memory operation of the main loop is retired before the epilogue begins.  The last phase issues loads whose results nothing
    // reads (it "reloads its own chunks" to stay branch-free); their destination registers are dead after the loop and get new tenants
    // here, and a build of round 3 produced scores that differed from run to run in a few lanes of a wave by one hidden unit's bias term
    // (N = 33 869: the codes table is not L2-resident, so those loads return late).  Draining costs a fraction of a microsecond per wave.
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_sched_barrier(0);
    const int Hrt = a.H;
    const uint32_t rkey = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + eg_id));
    int64_t mate_id = -1;                                         // MODE 3: the reverse edge, finished from the same accumulators
    uint32_t rkey2 = 0u;
    float z2 = 0.f;
    if constexpr (PAIR) {
        if (live) mate_id = a.mate[eg_id];
        rkey2 = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + (mate_id >= 0 ? mate_id : 0)));
    }
    const float* Us = a.U + static_cast<int64_t>(s) * H + 4 * kh;
    const float* Ud = a.U + static_cast<int64_t>(d) * H + 4 * kh;
    float4 us[kPF], ud[kPF];
    if constexpr (BWD) {
#pragma unroll
        for (int i = 0; i < kPF; ++i) {
            us[i] = *reinterpret_cast<const float4*>(Us + 8 * i);
            ud[i] = *reinterpret_cast<const float4*>(Ud + 8 * i);
        }
    }
    if constexpr (!BWD) {
        if ((a.prio & 3) == 1) __builtin_amdgcn_s_setprio(0);
        if ((a.prio & 3) == 2) __builtin_amdgcn_s_setprio(2);
        SGS_STAMP(2);
        // ---- forward epilogue (MODE 0 and the paired MODE 3), written for the vector-instruction count: the two waves of a SIMD run it at
        // the same time, so its instructions are not hidden behind anyone's MFMAs.  Per hidden unit and edge: v = (acc + b1) +/- (U[s] - U[d])
        // (acc + b1 shared by the pair), two compares (v > 0; draw >= threshold, the 16-bit draw selected by the compare itself), their AND on
        // the scalar unit, one select, one FMA into one of four partial sums (w2 / (1 - p) comes pre-scaled from LDS), one carry-in add for the
        // mask bit.  Steps run from the last hidden unit down: the mask words fill from the top, so bit 8 g4 + j lands in place.
        const bool want_bits = a.dvbits != nullptr;
        const bool use_drop = a.use_drop != 0;
        const uint32_t thr = a.drop_thresh;
        float zA[2] = {0.f, 0.f}, zB[2] = {0.f, 0.f};      // two partial sums per edge (even / odd hidden units)
        uint32_t fb[NT], fb2[PAIR ? NT : 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = 0u;
#pragma unroll
        for (int t = 0; t < (PAIR ? NT : 1); ++t) fb2[t] = 0u;
#pragma unroll
        for (int k = 0; k < kPF; ++k) {                 // the ring starts with the LAST kPF steps
            us[k] = *reinterpret_cast<const float4*>(Us + 8 * (4 * NT - 1 - k));
            ud[k] = *reinterpret_cast<const float4*>(Ud + 8 * (4 * NT - 1 - k));
        }
        // four straight-line copies of the loop (dropout on / off x mask kept / not), chosen once per wave: the two questions cost a
        // compare and a branch per hidden unit when asked inside
        auto steps = [&](auto DROP_, auto BITS_) {
            constexpr bool DROP = decltype(DROP_)::value, BITS = decltype(BITS_)::value;
#pragma unroll
            for (int ii = 0; ii < 4 * NT; ++ii) {
                if (8 * ii < Hrt) {                      // always true: one basic block per step keeps the look-ahead at kPF steps
                    const int i = 4 * NT - 1 - ii, t = i >> 2, g4 = i & 3;
                    const int hb = 8 * i + 4 * kh;
                    const float4 bb = *reinterpret_cast<const float4*>(&bw[0][hb]);
                    const float4 ww = *reinterpret_cast<const float4*>(&bw[1][hb]);
                    const float4 su = us[ii % kPF], du = ud[ii % kPF];
                    const float d4[4] = {su.x - du.x, su.y - du.y, su.z - du.z, su.w - du.w};
                    const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
                    const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
                    if (ii + kPF < 4 * NT) {
                        us[ii % kPF] = *reinterpret_cast<const float4*>(Us + 8 * (i - kPF));
                        ud[ii % kPF] = *reinterpret_cast<const float4*>(Ud + 8 * (i - kPF));
                    }
                    uint32_t bits[2] = {0u, 0u}, bits2[2] = {0u, 0u};
                    if constexpr (DROP) {
                        bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
                        bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
                        if constexpr (PAIR) {
                            bits2[0] = dropout_pair_bits(rkey2, static_cast<uint32_t>(hb >> 1));
                            bits2[1] = dropout_pair_bits(rkey2, static_cast<uint32_t>((hb >> 1) + 1));
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 3 - jj;
                        const float tb = acc[t][4 * g4 + j] + b4[j];
                        const float v = tb + d4[j];
                        // the mask word of the wave is the AND of the two compares' own lane masks (a ballot of a COMPARE is the compare's
                        // SGPR result; a ballot of their conjunction was rebuilt from a 0 / 1 select and one more compare: two more vector
                        // instructions per unit and direction)
                        const bool pos = v > 0.f;
                        bool on = pos;
                        uint64_t onm = BITS ? __builtin_amdgcn_ballot_w64(pos) : 0ull;
                        if constexpr (DROP) {
                            const bool kept = ((j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu)) >= thr;
                            on = pos & kept;
                            if constexpr (BITS) onm &= __builtin_amdgcn_ballot_w64(kept);
                        }
                        zA[j & 1] = fmaf(w4[j], on ? v : 0.f, zA[j & 1]);
                        if constexpr (BITS) fb[t] = shift_in_bit<DROP>(fb[t], onm);
                        if constexpr (PAIR) {
                            const float v2 = tb - d4[j];                     // the mate: the node-level term with the opposite sign
                            const bool pos2 = v2 > 0.f;
                            bool on2 = pos2;
                            uint64_t onm2 = BITS ? __builtin_amdgcn_ballot_w64(pos2) : 0ull;
                            if constexpr (DROP) {
                                const bool kept2 = ((j & 1) ? (bits2[j >> 1] >> 16) : (bits2[j >> 1] & 0xFFFFu)) >= thr;
                                on2 = pos2 & kept2;
                                if constexpr (BITS) onm2 &= __builtin_amdgcn_ballot_w64(kept2);
                            }
                            zB[j & 1] = fmaf(w4[j], on2 ? v2 : 0.f, zB[j & 1]);
                            if constexpr (BITS) fb2[PAIR ? t : 0] = shift_in_bit<DROP>(fb2[PAIR ? t : 0], onm2);
                        }
                    }
                    if constexpr (BITS) {
                        if (g4 != 0) {                                       // the other half-wave's nibble goes between two groups
                            fb[t] <<= 4;
                            if constexpr (PAIR) fb2[PAIR ? t : 0] <<= 4;
                        }
                    }
                }
            }
        };
        if (use_drop) { if (want_bits) steps(std::true_type{}, std::true_type{}); else steps(std::true_type{}, std::false_type{}); }
        else          { if (want_bits) steps(std::false_type{}, std::true_type{}); else steps(std::false_type{}, std::false_type{}); }
        float z = zA[0] + zA[1];
        float zm = zB[0] + zB[1];
        if (want_bits) {                                                 // join the two kh halves of every word, one lane stores the row
#pragma unroll
            for (int t = 0; t < NT; ++t) { fb[t] <<= 4 * kh; fb[t] |= __shfl_xor(fb[t], 32, 64); }
            if (live && kh == 0) {
                uint4* bo = reinterpret_cast<uint4*>(a.dvbits + eg_id * NT);
#pragma unroll
                for (int t = 0; t < NT; t += 4) bo[t >> 2] = make_uint4(fb[t], fb[t + 1], fb[t + 2], fb[t + 3]);
            }
            if constexpr (PAIR) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { fb2[t] <<= 4 * kh; fb2[t] |= __shfl_xor(fb2[t], 32, 64); }
                if (live && kh == 0 && mate_id >= 0) {
                    uint4* bo = reinterpret_cast<uint4*>(a.dvbits + mate_id * NT);
#pragma unroll
                    for (int t = 0; t < NT; t += 4) bo[t >> 2] = make_uint4(fb2[t], fb2[t + 1], fb2[t + 2], fb2[t + 3]);
                }
            }
        }
        z += __shfl_xor(z, 32, 64);
        if constexpr (PAIR) {
            zm += __shfl_xor(zm, 32, 64);
            if (live && kh == 0) {
                a.p_out[eg_id] = 1.0f / (1.0f + expf(-(z + a.b2[0])));
                if (mate_id >= 0) a.p_out[mate_id] = 1.0f / (1.0f + expf(-(zm + a.b2[0])));
            }
        } else {
            if (live && kh == 0) a.p_out[r] = 1.0f / (1.0f + expf(-(z + a.b2[0])));
        }
        SGS_STAMP(3);
        return;
    }
    float z = 0.f;
    // forward with `dvbits`: the ReLU x dropout mask of every scored edge, one bit per hidden unit (bit h of edge e in word h / 32 of row e)
    // -- with it the backward needs no recompute for dz and dv (sgs_edge_score_fwd_mask)
    const bool want_bits = !BWD && a.dvbits != nullptr;
    uint32_t fb[NT], fb2[PAIR ? NT : 1];             // whole rows in registers, two 16-byte stores per row at the end
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[t] = 0u;
#pragma unroll
    for (int t = 0; t < (PAIR ? NT : 1); ++t) fb2[t] = 0u;
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {
        if (8 * i < Hrt) {                           // always true: one basic block per step keeps the look-ahead at kPF steps
            const int t = i >> 2, g4 = i & 3;
            const int hb = 8 * i + 4 * kh;
            const float4 bb = *reinterpret_cast<const float4*>(&bw[0][hb]);
            const float4 ww = *reinterpret_cast<const float4*>(&bw[1][hb]);
            const float4 su = us[i % kPF], du = ud[i % kPF];
            const float u4[4] = {su.x - du.x, su.y - du.y, su.z - du.z, su.w - du.w};
            const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
            const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
            if (i + kPF < 4 * NT) {
                us[i % kPF] = *reinterpret_cast<const float4*>(Us + 8 * (i + kPF));
                ud[i % kPF] = *reinterpret_cast<const float4*>(Ud + 8 * (i + kPF));
            }
            uint32_t bits[2] = {0u, 0u};
            if (a.use_drop) {
                bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
                bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
            }
            if constexpr (PAIR) {
                // the mate (d -> s): same accumulators, node-level term U[d] - U[s] (the exact negation), its own dropout row
                const float n4[4] = {du.x - su.x, du.y - su.y, du.z - su.z, du.w - su.w};
                uint32_t bits2[2] = {0u, 0u};
                if (a.use_drop) {
                    bits2[0] = dropout_pair_bits(rkey2, static_cast<uint32_t>(hb >> 1));
                    bits2[1] = dropout_pair_bits(rkey2, static_cast<uint32_t>((hb >> 1) + 1));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v2 = (acc[t][4 * g4 + j] + n4[j]) + b4[j];
                    float m2 = v2 > 0.f ? 1.f : 0.f;
                    if (a.use_drop) {
                        const uint32_t draw2 = (j & 1) ? (bits2[j >> 1] >> 16) : (bits2[j >> 1] & 0xFFFFu);
                        m2 = draw2 >= a.drop_thresh ? m2 * a.drop_scale : 0.f;
                    }
                    z2 = fmaf(w4[j], v2 * m2, z2);
                    if (want_bits) fb2[PAIR ? t : 0] |= min(__float_as_uint(m2), 1u) << (8 * g4 + j);      // m2 is +0.0 or a positive factor
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (same order of additions as the forward epilogue above: the recomputed ReLU mask must equal the kept one bit for bit,
                //  and a unit with |v| ~ 1e-8 flips with the order)
                const float v = (acc[t][4 * g4 + j] + b4[j]) + u4[j];
                float m = v > 0.f ? 1.f : 0.f;
                if (a.use_drop) {
                    const uint32_t draw = (j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu);
                    m = draw >= a.drop_thresh ? m * a.drop_scale : 0.f;
                }
                const float hd = v * m;                              // dropout(relu(v))
                z = fmaf(w4[j], hd, z);
                if (BWD) acc[t][4 * g4 + j] = hd;                    // kept for the second pass
                // the factor m is +0.0 (v <= 0 or dropped) or positive: its bit pattern is nonzero exactly where hd > 0 (hd itself may be -0.0).
                // Compile-time positions; the kh nibble shift is applied once per word below
                if (want_bits) fb[t] |= min(__float_as_uint(m), 1u) << (8 * g4 + j);
            }
        }
    }
    if (want_bits) {                                                 // join the two kh halves of every word, one lane stores the row
#pragma unroll
        for (int t = 0; t < NT; ++t) { fb[t] <<= 4 * kh; fb[t] |= __shfl_xor(fb[t], 32, 64); }
        if (live && kh == 0) {
            uint4* bo = reinterpret_cast<uint4*>(a.dvbits + eg_id * NT);
#pragma unroll
            for (int t = 0; t < NT; t += 4) bo[t >> 2] = make_uint4(fb[t], fb[t + 1], fb[t + 2], fb[t + 3]);
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int t = 0; t < NT; ++t) { fb2[t] <<= 4 * kh; fb2[t] |= __shfl_xor(fb2[t], 32, 64); }
            if (live && kh == 0 && mate_id >= 0) {
                uint4* bo = reinterpret_cast<uint4*>(a.dvbits + mate_id * NT);
#pragma unroll
                for (int t = 0; t < NT; t += 4) bo[t >> 2] = make_uint4(fb2[t], fb2[t + 1], fb2[t + 2], fb2[t + 3]);
            }
        }
    }
    z += __shfl_xor(z, 32, 64);
    if constexpr (PAIR) {
        z2 += __shfl_xor(z2, 32, 64);
        if (live && kh == 0) {
            a.p_out[eg_id] = 1.0f / (1.0f + expf(-(z + a.b2[0])));
            if (mate_id >= 0) a.p_out[mate_id] = 1.0f / (1.0f + expf(-(z2 + a.b2[0])));
        }
        return;
    }
    if (!BWD) {
        if (live && kh == 0) {
            const float zz = z + a.b2[0];
            a.p_out[r] = 1.0f / (1.0f + expf(-zz));
        }
        return;
    }
    // ---- backward epilogue: dz = gp p (1-p);  dv = dz w2 relu' keep scale;  per-64-edge-tile sums of dz * hidden
    const float zz = z + a.b2[0];
    const float pr = 1.0f / (1.0f + expf(-zz));
    const float dzv = live ? a.gp[r] * pr * (1.0f - pr) : 0.f;                    // 0 on the padding rows of the last tile
    if (live && kh == 0) a.dz[r] = dzv;
    const float dscale = a.use_drop ? a.drop_scale : 1.f;
    // a wave holds 32 edges; a row of hdz_part is 64 (sgs_edge_score_bwd_tile()): waves 2m and 2m+1 meet in LDS (the W
    // buffers are free: the main loop ended on a barrier)
    float* hsum = reinterpret_cast<float*>(&wl[0][0]);                            // [NW][H]
    // dv[r, h] = dz[r] * w2[h] * [hd > 0] * scale: one bit per entry carries it (with dz and w2) -- `dvbits` gets bit h of row r in
    // word h / 32, and the consumers (MODE 4, gemm_tn's mask operand, the endpoint reduction) never read a [n, H] fp32 dv
    uint32_t mbits[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) mbits[t] = 0u;
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {
        if (8 * i < Hrt) {                           // always true: one basic block per step (register budget, as above)
            const int t = i >> 2, g4 = i & 3;
            const int hb = 8 * i + 4 * kh;
            const float4 ww = *reinterpret_cast<const float4*>(&bw[1][hb]);
            const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
            float dv4[4], hs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hd = acc[t][4 * g4 + j];
                dv4[j] = dzv * w4[j] * (hd > 0.f ? dscale : 0.f);                 // hd > 0 <=> v > 0 and kept
                hs[j] = dzv * hd;
                mbits[t] |= (hd > 0.f ? 1u : 0u) << (8 * g4 + 4 * kh + j);       // h & 31 = 8 (i & 3) + 4 kh + j, h >> 5 = i >> 2 = t
            }
            if (live && a.dv) *reinterpret_cast<float4*>(a.dv + r * H + hb) = make_float4(dv4[0], dv4[1], dv4[2], dv4[3]);
#pragma unroll
            // the 32 lanes of this half-wave (equal kh) on the DPP path, result in lanes 16..31 (as __shfl_xor steps this loop was 640
            // ds_bpermute_b32 with a wait each: a third of a tile's time)
            for (int j = 0; j < 4; ++j) hs[j] = half_wave_sum_hi(hs[j]);
            if (l31 == 31) *reinterpret_cast<float4*>(hsum + wave * H + hb) = make_float4(hs[0], hs[1], hs[2], hs[3]);
        }
    }
    if (a.dvbits) {
#pragma unroll
        for (int t = 0; t < NT; ++t) mbits[t] |= __shfl_xor(mbits[t], 32, 64);    // the two halves (kh) of every byte
        if (live && kh == 0) {
            uint4* bo = reinterpret_cast<uint4*>(a.dvbits + r * NT);
#pragma unroll
            for (int t = 0; t < NT; t += 4) bo[t >> 2] = make_uint4(mbits[t], mbits[t + 1], mbits[t + 2], mbits[t + 3]);
        }
    }
    __syncthreads();
    // tile (2 blockIdx NW/2 ... ): waves (2m, 2m+1) -> hdz_part row blockIdx * NW/2 + m
    for (int idx = tid; idx < (NW / 2) * H; idx += TH) {
        const int m = idx / H, h = idx - m * H;
        const int64_t prow = static_cast<int64_t>(blockIdx.x) * (NW / 2) + m;
        if (prow * 64 < a.n) a.hdz[prow * H + h] = hsum[(2 * m) * H + h] + hsum[(2 * m + 1) * H + h];
    }
}

// ---------------------------------------------------------------------------------------------
// Forward variant C ("weight-stationary"): PMC on variant B showed 118 M L2 requests per launch
// (~13 TB/s), two thirds of them re-fetching W1a.  Here a PERSISTENT workgroup keeps one hidden-half of
// the packed W1a (H/2 x H fp32 = 128 KiB at H = 256; the CU has 160 KiB of LDS) resident for its whole
// life and its waves pull 32-edge tiles from an atomic counter; only the endpoint codes stream from L2.
// A operands come from LDS with one ds_read_b128 per four k2-steps (conflict-free, lane-linear), B
// operands as in variant B.  No barrier after the initial fill.  Each workgroup produces the fc2 partial
// sum of ITS hidden half, zpart[hh][e]; edge_score_finish adds the halves and applies the sigmoid.
template <int NT, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) edge_score_wres_kernel(ScoreArgs a, const float* __restrict__ Wp,
                                                                                  const float* __restrict__ Ceo,
                                                                                  unsigned int* __restrict__ tile_ctr,
                                                                                  float* __restrict__ zpart) {
    constexpr int H = 32 * NT;
    constexpr int NTW = NT / 2;
    constexpr int NJ4 = H / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* Ws = reinterpret_cast<float4*>(smem);                  // [NTW][NJ4][64] float4 = this half of Wp
    const int tid = threadIdx.x, lane = tid & 63;
    const int hh = blockIdx.x & 1;
    const int kh = lane >> 5, l31 = lane & 31;
    {   // one-time fill: a straight copy of the half (contiguous in Wp)
        const float4* src = reinterpret_cast<const float4*>(Wp) + static_cast<int64_t>(hh) * NTW * NJ4 * 64;
        for (int i = tid; i < NTW * NJ4 * 64; i += THREADS) Ws[i] = src[i];
    }
    __syncthreads();
    const int64_t n_tiles = (a.n + 31) >> 5;
    const int Hrt = a.H;
    for (;;) {
        unsigned int t32 = 0;
        if (lane == 0) t32 = atomicAdd(&tile_ctr[hh], 1u);
        t32 = __builtin_amdgcn_readfirstlane(t32);
        if (static_cast<int64_t>(t32) >= n_tiles) break;           // every wave reaches this exit
        const int64_t r = static_cast<int64_t>(t32) * 32 + l31;
        const bool live = r < a.n;
        int s = 0, d = 0;
        int64_t eg_id = 0;
        if (live) {
            eg_id = a.active ? a.active[r] : r;
            s = static_cast<int>(a.src[eg_id]);
            d = static_cast<int>(a.dst[eg_id]);
        }
        const float4* xp = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(s) * H + kh * (H / 2));
        const float4* yp = reinterpret_cast<const float4*>(Ceo + static_cast<int64_t>(d) * H + kh * (H / 2));
        f32x16 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
        float4 A0[NTW], A1[NTW], x0, y0, x1, y1;
        auto load = [&](int j4, float4 (&A)[NTW], float4& x, float4& y) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) A[t] = Ws[(t * NJ4 + j4) * 64 + lane];
            x = xp[j4];
            y = yp[j4];
        };
        auto mma = [&](const float4 (&A)[NTW], const float4& x, const float4& y) {
            const float b[4] = {x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const float av = jj == 0 ? A[t].x : jj == 1 ? A[t].y : jj == 2 ? A[t].z : A[t].w;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[jj], acc[t], 0, 0, 0);
                }
            }
        };
        load(0, A0, x0, y0);
#pragma unroll 1
        for (int j4 = 0; j4 < NJ4; j4 += 2) {
            load(j4 + 1, A1, x1, y1);
            mma(A0, x0, y0);
            if (j4 + 2 < NJ4) load(j4 + 2, A0, x0, y0);
            mma(A1, x1, y1);
        }
        // epilogue for this wave's hidden half: hidden unit hh*H/2 + 32t + (q&3) + 8(q>>2) + 4kh
        const uint32_t rkey = dropout_row_key(fold_epoch(a.seed, a.epoch), a.site, static_cast<uint64_t>(a.row_offset + eg_id));
        const float* Us = a.U + static_cast<int64_t>(s) * H;
        const float* Ud = a.U + static_cast<int64_t>(d) * H;
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hb = hh * (H / 2) + 32 * t + 8 * g + 4 * kh;
                if (hb < Hrt) {      // always true; keeps hipcc from hoisting all the float4 loads at once
                    const float4 us = *reinterpret_cast<const float4*>(Us + hb);
                    const float4 ud = *reinterpret_cast<const float4*>(Ud + hb);
                    const float4 bb = *reinterpret_cast<const float4*>(a.b1 + hb);
                    const float4 ww = *reinterpret_cast<const float4*>(a.w2 + hb);
                    const float u4[4] = {us.x - ud.x, us.y - ud.y, us.z - ud.z, us.w - ud.w};
                    const float b4[4] = {bb.x, bb.y, bb.z, bb.w};
                    const float w4[4] = {ww.x, ww.y, ww.z, ww.w};
                    uint32_t bits[2] = {0u, 0u};
                    if (a.use_drop) {
                        bits[0] = dropout_pair_bits(rkey, static_cast<uint32_t>(hb >> 1));
                        bits[1] = dropout_pair_bits(rkey, static_cast<uint32_t>((hb >> 1) + 1));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = (acc[t][4 * g + j] + u4[j]) + b4[j];
                        float m = v > 0.f ? 1.f : 0.f;
                        if (a.use_drop) {
                            const uint32_t draw = (j & 1) ? (bits[j >> 1] >> 16) : (bits[j >> 1] & 0xFFFFu);
                            m = draw >= a.drop_thresh ? m * a.drop_scale : 0.f;
                        }
                        z = fmaf(w4[j], v * m, z);
                    }
                }
            }
        }
        z += __shfl_xor(z, 32, 64);
        if (live && kh == 0) zpart[static_cast<int64_t>(hh) * a.n + r] = z;
    }
}

__global__ void __launch_bounds__(kT) edge_score_finish(const float* __restrict__ zpart, int64_t n, const float* __restrict__ b2,
                                                       float* __restrict__ p_out) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
    if (r >= n) return;
    const float z = (zpart[r] + zpart[n + r]) + b2[0];
    p_out[r] = 1.0f / (1.0f + expf(-z));
}

// out[v,:] = sum_{k in out-row v} sgn_out * Mo[out_eid[k],:] (* T[out_dst[k],:])
//          + sum_{k in in-row v}  sgn_in  * Mi[in_eid[k],:]  (* T[in_src[k],:])
// Scatter of per-edge gradient rows to both endpoints as a deterministic gather over the two
// CSR orientations of the active edge list (no float atomics).
template <int VEC, bool HAS_T>
__global__ void __launch_bounds__(kT) endpoint_reduce(const float* __restrict__ Mo, const float* __restrict__ Mi,
                                                     const float* __restrict__ T, int64_t N, int64_t H,
                                                     const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                     const int* __restrict__ in_eid, const int* __restrict__ out_ptr,
                                                     const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                     float sgn_out, float sgn_in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6;   // one wave per node
    if (v >= N) return;
    for (int64_t c0 = static_cast<int64_t>(lane) * VEC; c0 < H; c0 += 64 * VEC) {
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int dir = 0; dir < 2; ++dir) {
            const int* ptr = dir == 0 ? out_ptr : in_ptr;
            const int* col = dir == 0 ? out_dst : in_src;
            const int* eid = dir == 0 ? out_eid : in_eid;
            const float sg = dir == 0 ? sgn_out : sgn_in;
            const float* M = dir == 0 ? Mo : Mi;
            const int b = ptr[v], e = ptr[v + 1];
            for (int k = b; k < e; ++k) {
                float m[VEC], t[VEC];
                if (VEC == 4) {
                    *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(M + static_cast<int64_t>(eid[k]) * H + c0);
                    if (HAS_T) *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(T + static_cast<int64_t>(col[k]) * H + c0);
                } else {
                    m[0] = M[static_cast<int64_t>(eid[k]) * H + c0];
                    if (HAS_T) t[0] = T[static_cast<int64_t>(col[k]) * H + c0];
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = fmaf(sg * m[j], HAS_T ? t[j] : 1.0f, acc[j]);
            }
        }
        if (VEC == 4) *reinterpret_cast<float4*>(out + v * H + c0) = *reinterpret_cast<float4*>(acc);
        else out[v * H + c0] = acc[0];
    }
}

// Small-N variant: a 4-wave workgroup per node; the node's out-row then in-row entries form one list
// that the waves stride with four independent row gathers in flight each; partial sums meet in LDS
// in a fixed order (deterministic).
// NW waves per node: 4 for moderate rows, 16 when rows are long -- a power-law partition has hub nodes with thousands of
// incident sampled edges, and with 4 waves those few workgroups set the kernel's duration (92 us -> see profiles).
template <int VEC, bool HAS_T, int NW>
__global__ void __launch_bounds__(64 * NW) endpoint_reduce_rowblock(const float* __restrict__ Mo, const float* __restrict__ Mi,
                                                              const float* __restrict__ T, int64_t N, int64_t H,
                                                              const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                              const int* __restrict__ in_eid, const int* __restrict__ out_ptr,
                                                              const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                              float sgn_out, float sgn_in, float* __restrict__ out) {
    using V = typename std::conditional<VEC == 4, float4, float>::type;
    __shared__ float part[NW][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t v = blockIdx.x;
    const int ob = out_ptr[v], no = out_ptr[v + 1] - ob;
    const int ib = in_ptr[v], ni = in_ptr[v + 1] - ib;
    const int total = no + ni;
    for (int64_t cbase = 0; cbase < H; cbase += 64 * VEC) {
        const int64_t c0 = cbase + static_cast<int64_t>(lane) * VEC;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        if (c0 < H) {
            for (int k0 = wave; k0 < total; k0 += 4 * NW) {
                // indices first, then every row gather, all unconditional (entries past the row: clamped to its last entry, sign 0) --
                // under `if (k < total)` each entry's index wait (vmcnt(0)) also waited for the previous entry's rows
                float m[4][VEC], t[4][VEC], sg[4];
                int er[4], cr[4];
                const float* Mp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + NW * u;
                    const int kc = k < total ? k : total - 1;
                    const bool isout = kc < no;
                    const int idx = isout ? ob + kc : ib + (kc - no);
                    er[u] = (isout ? out_eid : in_eid)[idx];
                    if (HAS_T) cr[u] = (isout ? out_dst : in_src)[idx];
                    Mp[u] = isout ? Mo : Mi;
                    sg[u] = k < total ? (isout ? sgn_out : sgn_in) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    *reinterpret_cast<V*>(m[u]) = *reinterpret_cast<const V*>(Mp[u] + static_cast<int64_t>(er[u]) * H + c0);
                    if (HAS_T) *reinterpret_cast<V*>(t[u]) = *reinterpret_cast<const V*>(T + static_cast<int64_t>(cr[u]) * H + c0);
                    else {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) t[u][j] = 1.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(sg[u] * m[u][j], t[u][j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) part[wave][lane * VEC + j] = acc[j];
        __syncthreads();
        const int tt = threadIdx.x;
        if (tt < 64 * VEC && cbase + tt < H) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < NW; g += 4) sum += (part[g][tt] + part[g + 1][tt]) + (part[g + 2][tt] + part[g + 3][tt]);
            out[v * H + cbase + tt] = sum;
        }
        __syncthreads();
    }
}

// The backward WITHOUT a recompute (the forward kept the mask: sgs_edge_score_fwd_mask).  Per active row r (edge e = active[r]):
//   dz[r] = gp[r] p[e] (1 - p[e]),   bits[r, :] = maskbits[e, :],   feat[r, :] = codes[src e, :] * codes[dst e, :]
// -- everything the three consumers of the mask form need, in one HBM-bound pass (the feat rows are the bytes: 4 H per row).
__global__ void __launch_bounds__(kT) scorer_bwd_prep(const float* __restrict__ codes, int64_t H, const int64_t* __restrict__ src,
                                                     const int64_t* __restrict__ dst, const int64_t* __restrict__ active, int64_t n,
                                                     const float* __restrict__ gp, const float* __restrict__ p,
                                                     const uint32_t* __restrict__ maskbits, float* __restrict__ dz,
                                                     uint32_t* __restrict__ bits, float* __restrict__ feat, int32_t* __restrict__ sd) {
    constexpr int R = 4;                                                  // rows per wave: their index chains and gathers in flight together
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> 6) * R;
    if (r0 >= n) return;
    const int wpr = static_cast<int>(H >> 5);
    int64_t e[R], s[R], d[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int64_t r = r0 + u < n ? r0 + u : n - 1;                    // (rows past the end: the last row again, not stored)
        e[u] = active ? active[r] : r;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) { s[u] = src[e[u]]; d[u] = dst[e[u]]; }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (r0 + u >= n) break;
        if (lane < wpr) bits[(r0 + u) * wpr + lane] = maskbits[e[u] * wpr + lane];
        if (lane == 63) { const float pe = p[e[u]]; dz[r0 + u] = gp[r0 + u] * pe * (1.0f - pe); }
        if (sd && lane == 62) { sd[2 * (r0 + u)] = static_cast<int32_t>(s[u]); sd[2 * (r0 + u) + 1] = static_cast<int32_t>(d[u]); }
    }
    if (!feat) return;                                                    // the fused backward gathers the rows where it needs them
    for (int64_t c0 = static_cast<int64_t>(lane) * 4; c0 < H; c0 += 256) {
        float4 a[R], b[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            a[u] = *reinterpret_cast<const float4*>(codes + s[u] * H + c0);
            b[u] = *reinterpret_cast<const float4*>(codes + d[u] * H + c0);
        }
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (r0 + u < n)
                *reinterpret_cast<float4*>(feat + (r0 + u) * H + c0) = make_float4(a[u].x * b[u].x, a[u].y * b[u].y, a[u].z * b[u].z, a[u].w * b[u].w);
    }
}

// d fc2.weight without the hidden activations: hd[e, h] = bit[e, h] * scale * (W1a feat_e + U[s] - U[d] + b1)[h], so
//   dw2[h] = sum_e dz_e hd[e, h] = scale * ( sum_k W1a[h, k] T[h, k]  +  sum_v U[v, h] R[v, h]  +  b1[h] c[h] )
// with T = mask^T diag(dz) feat (the weight-gradient GEMM before its row factor), R[v, :] = (sum_out - sum_in) dz_e bit[e, :] (the d U
// reduction before its column factor) and c = mask^T dz (the column sums before theirs).  One workgroup per hidden unit.
__global__ void __launch_bounds__(kT) scorer_dw2_from_parts(const float* __restrict__ W1, const float* __restrict__ Traw, const float* __restrict__ U,
                                                           const float* __restrict__ Rraw, const float* __restrict__ b1,
                                                           const float* __restrict__ craw, int64_t N, int H, float scale,
                                                           float* __restrict__ dw2) {
    __shared__ float red[kT / 64];
    const int h = blockIdx.x;
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < H; k += kT) a = fmaf(W1[static_cast<int64_t>(h) * 2 * H + k], Traw[static_cast<int64_t>(h) * H + k], a);
    for (int64_t v = threadIdx.x; v < N; v += kT) b = fmaf(U[v * H + h], Rraw[v * H + h], b);
    const float ra = block_sum(a, red);
    const float rb = block_sum(b, red);
    if (threadIdx.x == 0) dw2[h] = scale * ((ra + rb) + b1[h] * craw[h]);
}

// The scorer backward's two endpoint reductions in ONE pass over the incident-edge lists of a node v:
//   out_codes[v,:] = sum_k dfeat[e_k,:] * codes[other_k,:]          (both orientations, sign +)
//   out_U[v,:]     = sum_{k in out-row} dv[e_k,:] - sum_{k in in-row} dv[e_k,:]
// (U[s] enters the pre-activation with +, U[d] with -).  Same row walk, same fixed summation order as two calls of
// endpoint_reduce_rowblock; the edge ids / other-endpoint ids are read once and three row gathers are in flight per entry.
// BITS: the dv rows come as mask bits + dz (see sgs_edge_score_bwd_core_bits): out_U[v, c] = w2[c] * scale * (sum_out - sum_in) dz[e] bit[e, c]
constexpr int kEpU = 4;          // entries in flight per wave
template <int NW, bool BITS = false>
__global__ void __launch_bounds__(64 * NW) endpoint_reduce_pair_rowblock(const float* __restrict__ dfeat, const float* __restrict__ dv,
                                                                        const float* __restrict__ codes, int64_t N, int64_t H,
                                                                        const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                                        const int* __restrict__ in_eid, const int* __restrict__ out_ptr,
                                                                        const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                                        float* __restrict__ out_codes, float* __restrict__ out_U,
                                                                        const uint32_t* __restrict__ bits = nullptr,
                                                                        const float* __restrict__ dz = nullptr,
                                                                        const float* __restrict__ w2 = nullptr, float scale = 1.f,
                                                                        float* __restrict__ out_Uraw = nullptr) {
    __shared__ float part[2][NW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t v = blockIdx.x;
    const int ob = out_ptr[v], no = out_ptr[v + 1] - ob;
    const int ib = in_ptr[v], ni = in_ptr[v + 1] - ib;
    const int total = no + ni;
    for (int64_t cbase = 0; cbase < H; cbase += 256) {
        const int64_t c0 = cbase + static_cast<int64_t>(lane) * 4;
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < H) {
            for (int k0 = wave; k0 < total; k0 += kEpU * NW) {      // kEpU entries (3 row gathers each) in flight per wave
                // three phases, every load unconditional (entries past the row are clamped to its last entry and weighted 0): with the
                // loads under `if (k < total)` the compiler waits for entry u's indices with vmcnt(0), i.e. also for the row gathers of
                // entry u - 1, and the entries go through memory one after the other
                int er[kEpU], cr[kEpU];
                float sg[kEpU], ok[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const int k = k0 + NW * u;
                    const int kc = k < total ? k : total - 1;
                    const bool isout = kc < no;
                    const int idx = isout ? ob + kc : ib + (kc - no);
                    er[u] = (isout ? out_eid : in_eid)[idx];
                    cr[u] = (isout ? out_dst : in_src)[idx];
                    ok[u] = k < total ? 1.f : 0.f;
                    sg[u] = k < total ? (isout ? 1.f : -1.f) : 0.f;
                }
                float m1[kEpU][4], m2[kEpU][4], t[kEpU][4];
                uint32_t wd[kEpU];
                float dzr[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    *reinterpret_cast<float4*>(m1[u]) = *reinterpret_cast<const float4*>(dfeat + static_cast<int64_t>(er[u]) * H + c0);
                    *reinterpret_cast<float4*>(t[u]) = *reinterpret_cast<const float4*>(codes + static_cast<int64_t>(cr[u]) * H + c0);
                    if constexpr (BITS) {
                        wd[u] = bits[static_cast<int64_t>(er[u]) * (H >> 5) + (c0 >> 5)];
                        dzr[u] = dz[er[u]];
                    } else {
                        *reinterpret_cast<float4*>(m2[u]) = *reinterpret_cast<const float4*>(dv + static_cast<int64_t>(er[u]) * H + c0);
                    }
                }
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    if constexpr (BITS) {
                        const uint32_t w = wd[u] >> (c0 & 31);
#pragma unroll
                        for (int j = 0; j < 4; ++j) m2[u][j] = (w >> j) & 1u ? dzr[u] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a1[j] = fmaf(ok[u] * m1[u][j], t[u][j], a1[j]);
                        a2[j] = fmaf(sg[u], m2[u][j], a2[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[0][wave][lane * 4 + j] = a1[j]; part[1][wave][lane * 4 + j] = a2[j]; }
        __syncthreads();
        for (int tt = threadIdx.x; tt < 512; tt += 64 * NW) {
            const int which = tt >> 8, c = tt & 255;
            if (cbase + c < H) {
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < NW; g += 4) sum += (part[which][g][c] + part[which][g + 1][c]) + (part[which][g + 2][c] + part[which][g + 3][c]);
                if (BITS && which) {
                    if (out_Uraw) out_Uraw[v * H + cbase + c] = sum;      // before the column factor: a term of d fc2.weight (scorer_dw2_from_parts)
                    sum *= w2[cbase + c] * scale;
                }
                (which ? out_U : out_codes)[v * H + cbase + c] = sum;
            }
        }
        __syncthreads();
    }
}

// The endpoint reductions of the FUSED scorer backward (after MODE 5 of the bf16x6 loop), one workgroup per node v:
//   out_codes[v, :] = sum over v's run-end slots of opart  (the by-source half, already reduced per 32-row tile)
//                   + sum_{k in in-row v} G[in_eid[k], :]  (the by-destination half: G = dfeat * codes[src])
//   R[v, :]         = sum_{k in out-row v} dz[e] bit[e, :] - sum_{k in in-row v} dz[e] bit[e, :];   out_U = R * w2 * scale, out_Uraw = R
// Needs the active rows sorted by source: out-row v is then the rows out_ptr[v] .. out_ptr[v + 1] - 1 themselves.
template <int NW>
__global__ void __launch_bounds__(64 * NW) scorer_bwd_reduce(const float* __restrict__ G, const float* __restrict__ opart,
                                                            const uint32_t* __restrict__ bits, const float* __restrict__ dz,
                                                            const float* __restrict__ w2, float scale, int64_t N, int64_t H,
                                                            const int* __restrict__ in_ptr, const int* __restrict__ in_eid,
                                                            const int* __restrict__ out_ptr, float* __restrict__ out_codes,
                                                            float* __restrict__ out_U, float* __restrict__ out_Uraw) {
    __shared__ float part[2][NW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t v = blockIdx.x;
    const int ob = out_ptr[v], no = out_ptr[v + 1] - ob;
    const int ib = in_ptr[v], ni = in_ptr[v + 1] - ib;
    const int w0 = ob >> 5, np = no > 0 ? ((ob + no - 1) >> 5) - w0 + 1 : 0;          // run-end slots of this node: tiles w0 .. w0 + np - 1
    const int wpr = static_cast<int>(H >> 5);
    for (int64_t cbase = 0; cbase < H; cbase += 256) {
        const int64_t c0 = cbase + static_cast<int64_t>(lane) * 4;
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < H) {
            const int wsel = static_cast<int>(c0 >> 5), wsh = static_cast<int>(c0 & 31);
            // (1) in-row: G rows + mask bits (sign -).  kEpU entries in flight per wave, every load unconditional (entries past the row
            //     are clamped to its last entry and weighted 0)
            for (int k0 = wave; k0 < ni; k0 += kEpU * NW) {
                int er[kEpU];
                float ok[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const int k = k0 + NW * u;
                    er[u] = in_eid[ib + (k < ni ? k : ni - 1)];
                    ok[u] = k < ni ? 1.f : 0.f;
                }
                float4 m1[kEpU];
                uint32_t wd[kEpU];
                float dzr[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    m1[u] = *reinterpret_cast<const float4*>(G + static_cast<int64_t>(er[u]) * H + c0);
                    wd[u] = bits[static_cast<int64_t>(er[u]) * wpr + wsel];
                    dzr[u] = dz[er[u]];
                }
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const uint32_t w = wd[u] >> wsh;
                    const float dd = ok[u] * dzr[u];
                    a1[0] = fmaf(ok[u], m1[u].x, a1[0]); a1[1] = fmaf(ok[u], m1[u].y, a1[1]);
                    a1[2] = fmaf(ok[u], m1[u].z, a1[2]); a1[3] = fmaf(ok[u], m1[u].w, a1[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a2[j] -= (w >> j) & 1u ? dd : 0.f;
                }
            }
            // (2) out-row: mask bits only (sign +); the rows themselves are ob .. ob + no - 1
            for (int k0 = wave; k0 < no; k0 += kEpU * NW) {
                uint32_t wd[kEpU];
                float dzr[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const int k = k0 + NW * u;
                    const int64_t e = ob + (k < no ? k : no - 1);
                    wd[u] = bits[e * wpr + wsel];
                    dzr[u] = dz[e] * (k < no ? 1.f : 0.f);
                }
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const uint32_t w = wd[u] >> wsh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) a2[j] += (w >> j) & 1u ? dzr[u] : 0.f;
                }
            }
            // (3) the by-source half: this node's run-end partial sums, tile after tile
            for (int k0 = wave; k0 < np; k0 += kEpU * NW) {
                float4 m1[kEpU];
                float ok[kEpU];
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    const int k = k0 + NW * u;
                    const int64_t slot = static_cast<int64_t>(w0 + (k < np ? k : np - 1)) + v;
                    m1[u] = *reinterpret_cast<const float4*>(opart + slot * H + c0);
                    ok[u] = k < np ? 1.f : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kEpU; ++u) {
                    a1[0] = fmaf(ok[u], m1[u].x, a1[0]); a1[1] = fmaf(ok[u], m1[u].y, a1[1]);
                    a1[2] = fmaf(ok[u], m1[u].z, a1[2]); a1[3] = fmaf(ok[u], m1[u].w, a1[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[0][wave][lane * 4 + j] = a1[j]; part[1][wave][lane * 4 + j] = a2[j]; }
        __syncthreads();
        for (int tt = threadIdx.x; tt < 512; tt += 64 * NW) {
            const int which = tt >> 8, c = tt & 255;
            if (cbase + c < H) {
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < NW; g += 4) sum += (part[which][g][c] + part[which][g + 1][c]) + (part[which][g + 2][c] + part[which][g + 3][c]);
                if (which) {
                    if (out_Uraw) out_Uraw[v * H + cbase + c] = sum;
                    sum *= w2[cbase + c] * scale;
                }
                (which ? out_U : out_codes)[v * H + cbase + c] = sum;
            }
        }
        __syncthreads();
    }
}

// Endpoint reductions of the endpoint-dropout scorer (EdgeProbMLP with dropout): with x_m = A[s] * kx / (1 - p), y_m = A[d] * ky / (1 - p) and
// dfeat2 [n, 2H] = [d (x_m * y_m) | d (x_m - y_m)],
//   d A[v, :] = sum_{e in out-row v} (dfa[e] * y_m(e) + dfb[e]) * kx(e) / (1 - p)  +  sum_{e in in-row v} (dfa[e] * x_m(e) - dfb[e]) * ky(e) / (1 - p)
// (masks recomputed from the hash: row = ORIGINAL edge id, active[r] of row r).  One 4-wave workgroup per node, fixed summation order.
__global__ void __launch_bounds__(256) epd_endpoint_reduce(const float* __restrict__ dfeat2, const float* __restrict__ A, int64_t N, int64_t H,
                                                          const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                          const int* __restrict__ in_eid, const int* __restrict__ out_ptr,
                                                          const int* __restrict__ out_dst, const int* __restrict__ out_eid,
                                                          const int64_t* __restrict__ active, int64_t row_offset, uint64_t seed_x, uint32_t site_x,
                                                          uint64_t seed_y, uint32_t site_y, const uint64_t* __restrict__ epoch, uint32_t thresh,
                                                          float scale, int use_drop, float* __restrict__ dA) {
    constexpr int NW = 4;
    __shared__ float part[NW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t v = blockIdx.x;
    const int ob = out_ptr[v], no = out_ptr[v + 1] - ob;
    const int ib = in_ptr[v], ni = in_ptr[v + 1] - ib;
    const int total = no + ni;
    const uint64_t sx = fold_epoch(seed_x, epoch), sy = fold_epoch(seed_y, epoch);
    for (int64_t cbase = 0; cbase < H; cbase += 256) {
        const int64_t c0 = cbase + static_cast<int64_t>(lane) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < H) {
            for (int k = wave; k < total; k += NW) {
                const bool isout = k < no;
                const int idx = isout ? ob + k : ib + (k - no);
                const int r = (isout ? out_eid : in_eid)[idx];
                const int o = (isout ? out_dst : in_src)[idx];
                const int64_t e = row_offset + (active ? active[r] : static_cast<int64_t>(r));
                const float4 fa = *reinterpret_cast<const float4*>(dfeat2 + static_cast<int64_t>(r) * 2 * H + c0);
                const float4 fb = *reinterpret_cast<const float4*>(dfeat2 + static_cast<int64_t>(r) * 2 * H + H + c0);
                const float4 ao = *reinterpret_cast<const float4*>(A + static_cast<int64_t>(o) * H + c0);
                float ko[4] = {1.f, 1.f, 1.f, 1.f}, km[4] = {1.f, 1.f, 1.f, 1.f};    // masks (x scale) of the OTHER endpoint and of this one
                if (use_drop) {
                    const uint32_t rx = dropout_row_key(sx, site_x, static_cast<uint64_t>(e)), ry = dropout_row_key(sy, site_y, static_cast<uint64_t>(e));
                    const uint32_t rmine = isout ? rx : ry, rother = isout ? ry : rx;
                    const uint32_t m0 = dropout_pair_bits(rmine, static_cast<uint32_t>(c0 >> 1)), m1 = dropout_pair_bits(rmine, static_cast<uint32_t>((c0 >> 1) + 1));
                    const uint32_t o0 = dropout_pair_bits(rother, static_cast<uint32_t>(c0 >> 1)), o1 = dropout_pair_bits(rother, static_cast<uint32_t>((c0 >> 1) + 1));
                    km[0] = (m0 & 0xFFFFu) >= thresh ? scale : 0.f; km[1] = (m0 >> 16) >= thresh ? scale : 0.f;
                    km[2] = (m1 & 0xFFFFu) >= thresh ? scale : 0.f; km[3] = (m1 >> 16) >= thresh ? scale : 0.f;
                    ko[0] = (o0 & 0xFFFFu) >= thresh ? scale : 0.f; ko[1] = (o0 >> 16) >= thresh ? scale : 0.f;
                    ko[2] = (o1 & 0xFFFFu) >= thresh ? scale : 0.f; ko[3] = (o1 >> 16) >= thresh ? scale : 0.f;
                }
                const float fav[4] = {fa.x, fa.y, fa.z, fa.w}, fbv[4] = {fb.x, fb.y, fb.z, fb.w}, aov[4] = {ao.x, ao.y, ao.z, ao.w};
                const float sg = isout ? 1.f : -1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += (fav[j] * (aov[j] * ko[j]) + sg * fbv[j]) * km[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wave][lane * 4 + j] = acc[j];
        __syncthreads();
        const int tt = threadIdx.x;
        if (cbase + tt < H) dA[v * H + cbase + tt] = (part[0][tt] + part[1][tt]) + (part[2][tt] + part[3][tt]);
        __syncthreads();
    }
}

inline size_t score_smem_bytes(int NT) { return 2 * (static_cast<size_t>(kBK) * 32 * NT * 4 + kBK * kBM * 4) + 2 * kBM * 4 + 2 * kBM * 4; }

template <bool BWD, bool EPD = false>
int launch_score(const ScoreArgs& a, hipStream_t stream) {
    const int H = a.H;
    const int NT = H <= 64 ? 2 : H <= 128 ? 4 : 8;     // hidden units padded to 64 / 128 / 256
    const dim3 grid(static_cast<unsigned>(cdiv(a.n, kBM))), blk(kT);
    const size_t sm = score_smem_bytes(NT);
    // 2 stages x (16 KiB W + 4 KiB features) at H = 256: 41 KiB per workgroup, three per CU
    // (an EXACT specialisation that folds the bounds checks makes hipcc 7.2 spill ~200 VGPRs at this
    //  register budget; the generic variant allocates 151 VGPRs, no scratch, 3 waves/SIMD.)
    const bool exact = false;
#define SGS_SCORE_CASE(NT_, EX_)                                                                                   \
    do {                                                                                                            \
        static bool raised = false;                                                                                 \
        if (!raised) {                                                                                              \
            SGS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_score_kernel<NT_, BWD, EX_, EPD>),   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sm)));      \
            raised = true;                                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL((edge_score_kernel<NT_, BWD, EX_, EPD>), grid, blk, sm, stream, a);                      \
    } while (0)
    switch (NT) {
        case 2: if (exact) SGS_SCORE_CASE(2, true); else SGS_SCORE_CASE(2, false); break;
        case 4: if (exact) SGS_SCORE_CASE(4, true); else SGS_SCORE_CASE(4, false); break;
        default: if (exact) SGS_SCORE_CASE(8, true); else SGS_SCORE_CASE(8, false); break;
    }
#undef SGS_SCORE_CASE
    SGS_LAUNCH_OK();
    return SGS_OK;
}

inline int check_common(const char* who, int64_t N, int64_t H, int64_t E, float p_drop) {
    SGS_REQUIRE(N >= 0 && E >= 0 && N < (int64_t(1) << 31), SGS_EINVAL, "%s: bad sizes", who);
    SGS_REQUIRE(H >= 4 && H <= 256 && H % 4 == 0, SGS_EINVAL, "%s: hidden size H=%lld unsupported (need 4 <= H <= 256, H %% 4 == 0)",
                who, (long long)H);
    SGS_REQUIRE(p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "%s: bad dropout probability", who);
    return SGS_OK;
}

}  // namespace
}  // namespace sgs

using namespace sgs;

// start-up stagger of the bf16x6 kernels (see the kernel): 80 sleeps of 64 cycles per 32 hidden units ~ half a main loop
static int g_probe_stagger = -1;                         // >= 0: forced by sgs_edge_score_probe_set
static int g_probe_prio = 0;
static unsigned g_stagger_modes = (1u << 0) | (1u << 3); // kernel MODEs that use it
static unsigned long long* g_probe_trace = nullptr;
static void bf16x6_launch_knobs(ScoreArgs& a, int mode, int64_t H) {
    a.stagger = (g_stagger_modes >> mode) & 1u ? (g_probe_stagger >= 0 ? g_probe_stagger : static_cast<int>(80 * (H / 32))) : 0;
    a.prio = g_probe_prio;
    a.trace = g_probe_trace;
}

extern "C" {

size_t sgs_edge_score_workspace_bytes(int64_t N, int64_t H, int64_t E) {
    if (N < 0) N = 0;
    if (H < 0) H = 0;
    if (E < 0) E = 0;
    return carve_bytes(static_cast<size_t>(H) * H, 4) + carve_bytes(static_cast<size_t>(N) * H, 4) + carve_bytes(2 * static_cast<size_t>(E), 4) +
           carve_bytes(64, 4) + carve_bytes(static_cast<size_t>(H) * H, 8) + 256;
}

// 0 = LDS-tiled kernel, 1 = register-streaming kernel, 2 = weight-stationary persistent kernel (forward, H % 64 == 0).
// A/B switch for benchmarks.  Measured (MI355X, E = 351 194, H = 256, same process): 0 -> 95, 1 -> 100.7, 2 -> 99.5 TFLOP/s;
// whole-step throughput is equal within noise, so the fastest kernel is the default.
static int g_bwd_variant = -1;     // -1: automatic (4 at H % 128 == 0 and >= 65 536 active rows, else 0); 0: LDS-tiled core; 3: 64-edge streaming core (A/B: slower); 4: bf16x6 loop
static int g_score_variant = -1;   // -1: automatic (when the launch fills the chip with 128-edge workgroups: 4 if H % 128 == 0, else 3; otherwise 1)
void sgs_edge_score_set_variant(int v) { g_score_variant = v; }
void sgs_edge_score_set_bwd_variant(int v) { g_bwd_variant = v; }
int sgs_edge_score_get_variant(void) { return g_score_variant; }
int sgs_edge_score_get_bwd_variant(void) { return g_bwd_variant; }
int sgs_edge_score_bwd_tile(void) { return kBM; }

int sgs_edge_score_fwd(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                       int64_t edge_id_offset, const float* W1, const float* b1, const float* w2, const float* b2, float p_drop,
                       uint64_t seed, uint32_t site, float* p_out, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_common("sgs_edge_score_fwd", N, H, E, p_drop)) return rc;
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(codes && U && edge_index && W1 && b1 && w2 && b2 && p_out, SGS_EINVAL, "sgs_edge_score_fwd: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(N, H, E), SGS_EWORKSPACE, "sgs_edge_score_fwd: workspace too small");
    Carver cv(ws);
    float* WaT = cv.take<float>(static_cast<size_t>(H) * H);
    float* Ceo = cv.take<float>(static_cast<size_t>(N) * H);
    ScoreArgs a{};
    a.codes = codes; a.U = U; a.src = edge_index; a.dst = edge_index + E; a.active = nullptr; a.n = E; a.H = static_cast<int>(H);
    a.row_offset = edge_id_offset;
    a.WaT = WaT; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.drop_scale = 1.0f / (1.0f - p_drop); a.drop_thresh = dropout_thresh(p_drop); a.seed = seed; a.epoch = epoch_ptr(); a.site = site;
    a.use_drop = p_drop > 0.f; a.p_out = p_out;
    a.dyn_n = dyn_edges_ptr();
    int variant = g_score_variant;
    if (variant < 0) variant = (cdiv(E, kBM2) >= 512) ? (H % 128 == 0 ? 4 : 3) : 1;          // 512 = 2 resident workgroups x 256 CUs
    if (variant == 2 && a.dyn_n) variant = 3;      // the persistent kernel's tile queue is sized on the host
    if (variant == 2 && H % 64 == 0 && N > 0) {
        float* zpart = cv.take<float>(2 * static_cast<size_t>(E));
        unsigned int* ctr = cv.take<unsigned int>(64);
        if (int rc = zero_async(ctr, 256, stream)) return rc;
        {
            const int n_w = static_cast<int>(cdiv(H * H, kT));
            hipLaunchKernelGGL(pack_stream_operands, dim3(static_cast<unsigned>(n_w + cdiv(N * H, kT))), dim3(kT), 0, stream, W1,
                               static_cast<int>(H), WaT, n_w, codes, N, Ceo);
        }
        constexpr int TH = 768;                                             // 12 waves = 3 per SIMD (168-register budget; 16 waves spill)
        const size_t sm = static_cast<size_t>(H / 2) * H * 4;             // this half of W1a, packed
        const int64_t n_tiles = cdiv(E, 32);
        int64_t nwg = 256;                                                // persistent: one workgroup per CU (128 per hidden half)
        if (nwg > 2 * cdiv(n_tiles, TH / 64)) nwg = 2 * cdiv(n_tiles, TH / 64);
        if (nwg < 2) nwg = 2;
#define SGS_WRES_CASE(NT_)                                                                                              \
        do {                                                                                                                \
            static bool raised = false;                                                                                     \
            if (!raised) {                                                                                                  \
                SGS_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_score_wres_kernel<NT_, TH>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sm)));          \
                raised = true;                                                                                              \
            }                                                                                                               \
            hipLaunchKernelGGL((edge_score_wres_kernel<NT_, TH>), dim3(static_cast<unsigned>(nwg)), dim3(TH), sm, stream, a, WaT, Ceo, \
                               ctr, zpart);                                                                                 \
        } while (0)
        if (H == 256) SGS_WRES_CASE(8); else if (H == 128) SGS_WRES_CASE(4); else SGS_WRES_CASE(2);
#undef SGS_WRES_CASE
        hipLaunchKernelGGL(edge_score_finish, dim3(cdiv(E, kT)), dim3(kT), 0, stream, zpart, E, b2, p_out);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (variant == 4 && H % 128 == 0 && N > 0) {
        cv.take<float>(2 * static_cast<size_t>(E));
        cv.take<unsigned int>(64);
        uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
        hipLaunchKernelGGL(pack_w1a_bf16x3<false>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                           static_cast<int>(H), Wp16);
        const dim3 grid(static_cast<unsigned>(cdiv(E, 128))), blk(256);       // 4 waves x 32 edges; two workgroups per CU
        bf16x6_launch_knobs(a, 0, H);
        if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4>), grid, blk, 0, stream, a, Wp16);
        else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4>), grid, blk, 0, stream, a, Wp16);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (variant == 4) variant = 3;
    if (variant == 3 && H % 64 == 0 && N > 0) {
        {
            const int n_w = static_cast<int>(cdiv(H * H, kT));
            hipLaunchKernelGGL(pack_stream_operands, dim3(static_cast<unsigned>(n_w + cdiv(N * H, kT))), dim3(kT), 0, stream, W1,
                               static_cast<int>(H), WaT, n_w, codes, N, Ceo);
        }
        const dim3 grid(static_cast<unsigned>(cdiv(E, kBM2))), blk(kT);
        if (H == 256)      hipLaunchKernelGGL((edge_score_stream64_kernel<8, false>), grid, blk, 0, stream, a, WaT, Ceo);
        else if (H == 128) hipLaunchKernelGGL((edge_score_stream64_kernel<4, false>), grid, blk, 0, stream, a, WaT, Ceo);
        else               hipLaunchKernelGGL((edge_score_stream64_kernel<2, false>), grid, blk, 0, stream, a, WaT, Ceo);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    if (variant == 1 && H % 64 == 0 && N > 0) {
        {
            const int n_w = static_cast<int>(cdiv(H * H, kT));
            hipLaunchKernelGGL(pack_stream_operands, dim3(static_cast<unsigned>(n_w + cdiv(N * H, kT))), dim3(kT), 0, stream, W1,
                               static_cast<int>(H), WaT, n_w, codes, N, Ceo);
        }
        const dim3 grid(static_cast<unsigned>(cdiv(E, kBM))), blk(kT);
        if (H == 256)      hipLaunchKernelGGL((edge_score_stream_kernel<8>), grid, blk, 0, stream, a, WaT, Ceo);
        else if (H == 128) hipLaunchKernelGGL((edge_score_stream_kernel<4>), grid, blk, 0, stream, a, WaT, Ceo);
        else               hipLaunchKernelGGL((edge_score_stream_kernel<2>), grid, blk, 0, stream, a, WaT, Ceo);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    hipLaunchKernelGGL(transpose_w1a, dim3(cdiv(H, 32), cdiv(H, 32)), dim3(kT), 0, stream, W1, static_cast<int>(H), WaT);
    return launch_score<false>(a, stream);
}

/* Paired forward (sgs_hip.h): only the `M` canonical edges run the H x H contraction; each also finishes its mate's score. */
int sgs_edge_score_paired_supported(int64_t H) { return (H == 128 || H == 256) ? 1 : 0; }

static int fwd_bf16x6_impl(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                           int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                           const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, uint32_t* maskbits,
                           void* ws, size_t ws_bytes, sgs_stream_t stream_);

int sgs_edge_score_fwd_paired(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                              int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                              const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, void* ws,
                              size_t ws_bytes, sgs_stream_t stream_) {
    SGS_REQUIRE((canon && mate) || E == 0 || M == 0, SGS_EINVAL, "sgs_edge_score_fwd_paired: null pointer");
    return fwd_bf16x6_impl(codes, U, N, H, edge_index, E, edge_id_offset, canon, M, mate, W1, b1, w2, b2, p_drop, seed, site, p_out, nullptr, ws,
                           ws_bytes, stream_);
}

/* The bf16x6 forward (paired when canon / mate are given, every edge otherwise) that also writes the ReLU x dropout MASK of every scored
 * edge: maskbits [E, H/32], bit h of edge e in word h / 32 of row e.  With the mask and p the backward needs no recompute for dz and dv
 * (sgs_edge_score_bwd_prep); the scores are the plain forward's bit for bit. */
int sgs_edge_score_fwd_mask(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                            int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                            const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, uint32_t* maskbits,
                            void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    SGS_REQUIRE(maskbits || E == 0, SGS_EINVAL, "sgs_edge_score_fwd_mask: null pointer");
    SGS_REQUIRE((canon != nullptr) == (mate != nullptr), SGS_EINVAL, "sgs_edge_score_fwd_mask: canon and mate come together");
    return fwd_bf16x6_impl(codes, U, N, H, edge_index, E, edge_id_offset, canon, canon ? M : E, mate, W1, b1, w2, b2, p_drop, seed, site, p_out,
                           maskbits, ws, ws_bytes, stream_);
}

static int fwd_bf16x6_impl(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                           int64_t edge_id_offset, const int32_t* canon, int64_t M, const int32_t* mate, const float* W1, const float* b1,
                           const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site, float* p_out, uint32_t* maskbits,
                           void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_common("sgs_edge_score_fwd_paired", N, H, E, p_drop)) return rc;
    SGS_REQUIRE(sgs_edge_score_paired_supported(H), SGS_EINVAL, "sgs_edge_score_fwd_paired: H must be 128 or 256");
    SGS_REQUIRE(M >= 0 && M <= E, SGS_EINVAL, "sgs_edge_score_fwd_paired: bad canonical count");
    if (E == 0 || M == 0) return SGS_OK;
    SGS_REQUIRE(codes && U && edge_index && W1 && b1 && w2 && b2 && p_out && N > 0, SGS_EINVAL, "sgs_edge_score_fwd_paired: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(N, H, E), SGS_EWORKSPACE, "sgs_edge_score_fwd_paired: workspace too small");
    Carver cv(ws);
    cv.take<float>(static_cast<size_t>(H) * H);
    cv.take<float>(static_cast<size_t>(N) * H);
    cv.take<float>(2 * static_cast<size_t>(E));
    cv.take<unsigned int>(64);
    uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
    ScoreArgs a{};
    a.codes = codes; a.U = U; a.src = edge_index; a.dst = edge_index + E; a.active = nullptr; a.n = M; a.H = static_cast<int>(H);
    a.row_offset = edge_id_offset;
    a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.drop_scale = 1.0f / (1.0f - p_drop); a.drop_thresh = dropout_thresh(p_drop); a.seed = seed; a.epoch = epoch_ptr(); a.site = site;
    a.use_drop = p_drop > 0.f; a.p_out = p_out; a.dvbits = maskbits;
    a.canon = canon; a.mate = mate;
    hipLaunchKernelGGL(pack_w1a_bf16x3<false>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                       static_cast<int>(H), Wp16);
    const dim3 grid(static_cast<unsigned>(cdiv(M, 128))), blk(256);
    if (canon) {
        a.dyn_n = dyn_edges_ptr() ? dyn_edges_ptr() + 1 : nullptr;   // word 1 of the registered dims: the live number of canonical edges
        bf16x6_launch_knobs(a, 3, H);
        // (probe: prio bit 6 pads the launch with dynamic LDS so that ONE workgroup fits a CU)
        if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4, 3>), grid, blk, (a.prio & 64) ? 40960 : 0, stream, a, Wp16);
        else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4, 3>), grid, blk, 0, stream, a, Wp16);
    } else {                                                         // every edge runs the contraction (no mates: a directed edge list)
        a.dyn_n = dyn_edges_ptr();
        bf16x6_launch_knobs(a, 0, H);
        if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4>), grid, blk, 0, stream, a, Wp16);
        else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4>), grid, blk, 0, stream, a, Wp16);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* Backward core over the active rows: recomputes the hidden layer and writes
 * dv [n,H] = dL/d(fc1 pre-activation), hdz_part [cdiv(n, 64), H] = per-64-edge-tile column sums of dz * hidden,
 * dz [n], feat [n,H] = x_s*x_d. */
static int bwd_core_impl(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                         int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1,
                         const float* b1, const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site,
                         float* dv, uint32_t* dvbits, float* hdz_part, float* dz, float* feat, void* ws, size_t ws_bytes,
                         sgs_stream_t stream_);

int sgs_edge_score_bwd_core(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                            int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1,
                            const float* b1, const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site,
                            float* dv, float* hdz_part, float* dz, float* feat, void* ws, size_t ws_bytes,
                            sgs_stream_t stream_) {
    SGS_REQUIRE(dv || n_active == 0, SGS_EINVAL, "sgs_edge_score_bwd_core: null pointer");
    return bwd_core_impl(codes, U, N, H, edge_index, E, edge_id_offset, active_eid, n_active, grad_p, W1, b1, w2, b2, p_drop, seed, site, dv,
                         nullptr, hdz_part, dz, feat, ws, ws_bytes, stream_);
}

/* The mask form of the backward (H = 128 or 256): dv[r, h] = dz[r] * w2[h] * [hidden h of row r survived ReLU and dropout] / (1 - p), so the
 * core writes ONE BIT per entry (dvbits [n, H/32], bit h of row r in word h / 32) instead of the fp32 [n, H] matrix, and the three
 * consumers rebuild what they need from bits + dz + w2: sgs_edge_score_bwd_dfeat_bits, sgs_gemm_tn_mask, sgs_endpoint_reduce_pair_bits.
 * A 0 / 1 operand is exact in bf16, so those two contractions issue 3 bf16 MFMA products per fp32 product instead of 6. */
int sgs_edge_score_bwd_bits_supported(int64_t H) { return (H == 128 || H == 256) ? 1 : 0; }

int sgs_edge_score_bwd_core_bits(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                                 int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1,
                                 const float* b1, const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site,
                                 uint32_t* dvbits, float* hdz_part, float* dz, float* feat, void* ws, size_t ws_bytes,
                                 sgs_stream_t stream_) {
    SGS_REQUIRE(sgs_edge_score_bwd_bits_supported(H) && N > 0, SGS_EINVAL, "sgs_edge_score_bwd_core_bits: H=%lld unsupported (128 or 256)", (long long)H);
    SGS_REQUIRE(dvbits || n_active == 0, SGS_EINVAL, "sgs_edge_score_bwd_core_bits: null pointer");
    return bwd_core_impl(codes, U, N, H, edge_index, E, edge_id_offset, active_eid, n_active, grad_p, W1, b1, w2, b2, p_drop, seed, site, nullptr,
                         dvbits, hdz_part, dz, feat, ws, ws_bytes, stream_);
}

static int bwd_core_impl(const float* codes, const float* U, int64_t N, int64_t H, const int64_t* edge_index, int64_t E,
                         int64_t edge_id_offset, const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1,
                         const float* b1, const float* w2, const float* b2, float p_drop, uint64_t seed, uint32_t site,
                         float* dv, uint32_t* dvbits, float* hdz_part, float* dz, float* feat, void* ws, size_t ws_bytes,
                         sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_common("sgs_edge_score_bwd_core", N, H, E, p_drop)) return rc;
    SGS_REQUIRE(n_active >= 0 && (active_eid || n_active == E), SGS_EINVAL,
                "sgs_edge_score_bwd_core: n_active must equal E when active_eid is NULL");
    if (n_active == 0) return SGS_OK;
    SGS_REQUIRE(codes && U && edge_index && grad_p && W1 && b1 && w2 && b2 && (dv || dvbits) && hdz_part && dz && feat, SGS_EINVAL,
                "sgs_edge_score_bwd_core: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(N, H, 0), SGS_EWORKSPACE, "sgs_edge_score_bwd_core: workspace too small");
    Carver cv(ws);
    float* WaT = cv.take<float>(static_cast<size_t>(H) * H);
    float* Ceo = cv.take<float>(static_cast<size_t>(N) * H);
    // The 64-edge streaming loop is available for the backward core too (variant 3), but it is NOT the default: measured
    // (tools/bwd_probe.py) 323 / 656 / 1092 us against 227 / 558 / 997 us for the LDS-tiled core at 100 k / 262 k / 500 k active
    // rows.  The backward is bound by what it writes (dv and feat, 2 KB per row) and the tiled core produces the feature tile
    // in LDS anyway, while the streaming loop has to rebuild row pieces from registers with cross-lane swaps.
    // automatic: the bf16x6 loop once the launch fills the chip (measured, tools/bwd_probe.py: 160 vs 230 us at 100 k active rows,
    // 327 vs 574 at 262 k, 618 vs 1009 at 500 k), the LDS-tiled core below that
    const int bwd_variant = dvbits ? 4 : (g_bwd_variant >= 0 ? g_bwd_variant : ((H % 128 == 0 && cdiv(n_active, 128) >= 512) ? 4 : 0));
    if (bwd_variant == 4 && H % 128 == 0 && N > 0) {
        // the recompute on the bf16x6 loop (forward variant 4): same outputs
        cv.take<float>(0);
        cv.take<unsigned int>(64);
        uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
        hipLaunchKernelGGL(pack_w1a_bf16x3<false>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                           static_cast<int>(H), Wp16);
        ScoreArgs b{};
        b.codes = codes; b.U = U; b.src = edge_index; b.dst = edge_index + E; b.active = active_eid; b.n = n_active;
        b.row_offset = edge_id_offset;
        b.H = static_cast<int>(H); b.WaT = nullptr; b.b1 = b1; b.w2 = w2; b.b2 = b2;
        b.drop_scale = 1.0f / (1.0f - p_drop); b.drop_thresh = dropout_thresh(p_drop); b.seed = seed; b.epoch = epoch_ptr(); b.site = site;
        b.use_drop = p_drop > 0.f; b.gp = grad_p; b.dv = dv; b.dvbits = dvbits; b.hdz = hdz_part; b.dz = dz; b.feat = feat;
        const dim3 grid(static_cast<unsigned>(cdiv(n_active, 128))), blk(256);
        bf16x6_launch_knobs(b, 1, H);
        if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4, 1>), grid, blk, 0, stream, b, Wp16);
        else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4, 1>), grid, blk, 0, stream, b, Wp16);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    const bool stream64 = H % 64 == 0 && bwd_variant == 3;
    if (stream64) {
        const int n_w = static_cast<int>(cdiv(H * H, kT));
        hipLaunchKernelGGL(pack_stream_operands, dim3(static_cast<unsigned>(n_w + cdiv(N * H, kT))), dim3(kT), 0, stream, W1,
                           static_cast<int>(H), WaT, n_w, codes, N, Ceo);
    } else {
        hipLaunchKernelGGL(transpose_w1a, dim3(cdiv(H, 32), cdiv(H, 32)), dim3(kT), 0, stream, W1, static_cast<int>(H), WaT);
    }
    ScoreArgs a{};
    a.codes = codes; a.U = U; a.src = edge_index; a.dst = edge_index + E; a.active = active_eid; a.n = n_active;
    a.row_offset = edge_id_offset;
    a.H = static_cast<int>(H); a.WaT = WaT; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.drop_scale = 1.0f / (1.0f - p_drop); a.drop_thresh = dropout_thresh(p_drop); a.seed = seed; a.epoch = epoch_ptr(); a.site = site;
    a.use_drop = p_drop > 0.f; a.gp = grad_p; a.dv = dv; a.hdz = hdz_part; a.dz = dz; a.feat = feat;
    if (stream64) {
        const dim3 grid(static_cast<unsigned>(cdiv(n_active, kBM2))), blk(kT);
        if (H == 256)      hipLaunchKernelGGL((edge_score_stream64_kernel<8, true>), grid, blk, 0, stream, a, WaT, Ceo);
        else if (H == 128) hipLaunchKernelGGL((edge_score_stream64_kernel<4, true>), grid, blk, 0, stream, a, WaT, Ceo);
        else               hipLaunchKernelGGL((edge_score_stream64_kernel<2, true>), grid, blk, 0, stream, a, WaT, Ceo);
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    return launch_score<true>(a, stream);
}

/* dfeat [n,H] = dv [n,H] . W1[:, :H]  (the scorer backward's gradient wrt the Hadamard features).  The bf16x6 loop as a row GEMM at
 * H = 128 or 256 (sgs_edge_score_bwd_dfeat_supported; other sizes: the caller uses a library GEMM). */
int sgs_edge_score_bwd_dfeat_supported(int64_t H) { return (H == 128 || H == 256) ? 1 : 0; }

int sgs_edge_score_bwd_dfeat(const float* dv, int64_t n, int64_t H, const float* W1, float* dfeat, void* ws, size_t ws_bytes,
                             sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0 && H > 0, SGS_EINVAL, "sgs_edge_score_bwd_dfeat: bad sizes");
    SGS_REQUIRE(sgs_edge_score_bwd_dfeat_supported(H), SGS_EINVAL, "sgs_edge_score_bwd_dfeat: H=%lld unsupported (128 or 256)", (long long)H);
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(dv && W1 && dfeat, SGS_EINVAL, "sgs_edge_score_bwd_dfeat: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(0, H, 0), SGS_EWORKSPACE, "sgs_edge_score_bwd_dfeat: workspace too small");
    Carver cv(ws);
    cv.take<float>(static_cast<size_t>(H) * H);
    cv.take<float>(0);
    cv.take<float>(0);
    cv.take<unsigned int>(64);
    uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
    hipLaunchKernelGGL(pack_w1a_bf16x3<true>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                       static_cast<int>(H), Wp16);
    ScoreArgs a{};
    a.codes = dv; a.n = n; a.H = static_cast<int>(H); a.feat = dfeat;
    const dim3 grid(static_cast<unsigned>(cdiv(n, 128))), blk(256);
    bf16x6_launch_knobs(a, 2, H);
    if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4, 2>), grid, blk, 0, stream, a, Wp16);
    else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4, 2>), grid, blk, 0, stream, a, Wp16);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* dfeat [n,H] = dv . W1[:, :H] from the mask form of dv: out[r, :] = dz[r] * (bits[r, :] . diag(w2 / (1 - p)) W1a) -- the 0 / 1 operand
 * is one exact bf16 piece, the scaled matrix is split three ways: 3 MFMA products per fp32 product. */
int sgs_edge_score_bwd_dfeat_bits(const uint32_t* dvbits, const float* dz, int64_t n, int64_t H, const float* W1, const float* w2, float p_drop,
                                  float* dfeat, void* ws, size_t ws_bytes, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0 && H > 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_edge_score_bwd_dfeat_bits: bad arguments");
    SGS_REQUIRE(sgs_edge_score_bwd_bits_supported(H), SGS_EINVAL, "sgs_edge_score_bwd_dfeat_bits: H=%lld unsupported (128 or 256)", (long long)H);
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(dvbits && dz && W1 && w2 && dfeat, SGS_EINVAL, "sgs_edge_score_bwd_dfeat_bits: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(0, H, 0), SGS_EWORKSPACE, "sgs_edge_score_bwd_dfeat_bits: workspace too small");
    Carver cv(ws);
    cv.take<float>(static_cast<size_t>(H) * H);
    cv.take<float>(0);
    cv.take<float>(0);
    cv.take<unsigned int>(64);
    uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
    hipLaunchKernelGGL(pack_w1a_bf16x3<true>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                       static_cast<int>(H), Wp16, w2, 1.0f / (1.0f - p_drop));
    ScoreArgs a{};
    a.inbits = dvbits; a.indz = dz; a.n = n; a.H = static_cast<int>(H); a.feat = dfeat;
    const dim3 grid(static_cast<unsigned>(cdiv(n, 128))), blk(256);
    bf16x6_launch_knobs(a, 4, H);
    if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4, 4>), grid, blk, 0, stream, a, Wp16);
    else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4, 4>), grid, blk, 0, stream, a, Wp16);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_endpoint_reduce(const float* M_out, const float* M_in, const float* T, int64_t N, int64_t H, int64_t nnz,
                        const int32_t* in_ptr, const int32_t* in_src,
                        const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                        float sign_out, float sign_in, float* out, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && H >= 0, SGS_EINVAL, "sgs_endpoint_reduce: bad sizes");
    if (N == 0 || H == 0) return SGS_OK;
    SGS_REQUIRE(M_out && M_in && in_ptr && out_ptr && out, SGS_EINVAL, "sgs_endpoint_reduce: null pointer");
    const bool v4 = H % 4 == 0 && (reinterpret_cast<uintptr_t>(M_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(M_in) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (!T || (reinterpret_cast<uintptr_t>(T) & 15) == 0);
    const dim3 blk(kT);
    if (nnz >= 8 * N) {                       // long rows: a workgroup per node (partitions, and whole graphs of average degree >= 8)
        const dim3 grid(static_cast<unsigned>(N));
        const bool wide = nnz >= 64 * N;       // long rows (hubs with thousands of entries): 16 waves per node
#define SGS_EPR(VEC_, HT_)                                                                                                   \
        do {                                                                                                                    \
            if (wide) hipLaunchKernelGGL((endpoint_reduce_rowblock<VEC_, HT_, 16>), grid, dim3(1024), 0, stream, M_out, M_in, T, N, H, in_ptr, \
                                         in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);                  \
            else      hipLaunchKernelGGL((endpoint_reduce_rowblock<VEC_, HT_, 4>), grid, blk, 0, stream, M_out, M_in, T, N, H, in_ptr,        \
                                         in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);                  \
        } while (0)
        if (v4) { if (T) SGS_EPR(4, true); else SGS_EPR(4, false); }
        else    { if (T) SGS_EPR(1, true); else SGS_EPR(1, false); }
#undef SGS_EPR
        SGS_LAUNCH_OK();
        return SGS_OK;
    }
    const dim3 grid(static_cast<unsigned>(cdiv(N * 64, kT)));
    if (v4) {
        if (T) hipLaunchKernelGGL((endpoint_reduce<4, true>), grid, blk, 0, stream, M_out, M_in, T, N, H, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);
        else   hipLaunchKernelGGL((endpoint_reduce<4, false>), grid, blk, 0, stream, M_out, M_in, T, N, H, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);
    } else {
        if (T) hipLaunchKernelGGL((endpoint_reduce<1, true>), grid, blk, 0, stream, M_out, M_in, T, N, H, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);
        else   hipLaunchKernelGGL((endpoint_reduce<1, false>), grid, blk, 0, stream, M_out, M_in, T, N, H, in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid, sign_out, sign_in, out);
    }
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_endpoint_reduce_pair(const float* dfeat, const float* dv, const float* codes, int64_t N, int64_t H, int64_t nnz, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst,
                             const int32_t* out_eid, float* out_codes, float* out_U, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && H >= 0 && H % 4 == 0 && N < (int64_t(1) << 31), SGS_EINVAL, "sgs_endpoint_reduce_pair: needs H %% 4 == 0");
    if (N == 0 || H == 0) return SGS_OK;
    SGS_REQUIRE(dfeat && dv && codes && in_ptr && out_ptr && out_codes && out_U, SGS_EINVAL, "sgs_endpoint_reduce_pair: null pointer");
    const dim3 grid(static_cast<unsigned>(N));
    if (nnz >= 64 * N)
        hipLaunchKernelGGL((endpoint_reduce_pair_rowblock<16>), grid, dim3(1024), 0, stream, dfeat, dv, codes, N, H, in_ptr, in_src, in_eid, out_ptr,
                           out_dst, out_eid, out_codes, out_U);
    else
        hipLaunchKernelGGL((endpoint_reduce_pair_rowblock<4>), grid, dim3(256), 0, stream, dfeat, dv, codes, N, H, in_ptr, in_src, in_eid, out_ptr,
                           out_dst, out_eid, out_codes, out_U);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

static int bwd_prep_impl(const float* codes, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, const int64_t* active_eid,
                         int64_t n_active, const float* grad_p, const float* p, const uint32_t* maskbits, float* dz, uint32_t* dvbits,
                         float* feat, int32_t* sd, hipStream_t stream) {
    SGS_REQUIRE(N > 0 && H > 0 && H % 32 == 0 && E >= 0 && n_active >= 0 && (active_eid || n_active == E), SGS_EINVAL,
                "sgs_edge_score_bwd_prep: bad arguments (H %% 32 == 0; n_active == E when active_eid is NULL)");
    if (n_active == 0) return SGS_OK;
    SGS_REQUIRE(codes && edge_index && grad_p && p && maskbits && dz && dvbits && (feat || sd), SGS_EINVAL, "sgs_edge_score_bwd_prep: null pointer");
    hipLaunchKernelGGL(scorer_bwd_prep, dim3(static_cast<unsigned>(cdiv(cdiv(n_active, 4) * 64, kT))), dim3(kT), 0, stream, codes, H, edge_index,
                       edge_index + E, active_eid, n_active, grad_p, p, maskbits, dz, dvbits, feat, sd);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_edge_score_bwd_prep(const float* codes, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, const int64_t* active_eid,
                            int64_t n_active, const float* grad_p, const float* p, const uint32_t* maskbits, float* dz, uint32_t* dvbits,
                            float* feat, sgs_stream_t stream_) {
    SGS_REQUIRE(feat || n_active == 0, SGS_EINVAL, "sgs_edge_score_bwd_prep: null pointer");
    return bwd_prep_impl(codes, N, H, edge_index, E, active_eid, n_active, grad_p, p, maskbits, dz, dvbits, feat, nullptr,
                         static_cast<hipStream_t>(stream_));
}

/* The same pass for the FUSED backward: no feat (its consumers gather the code rows themselves); the endpoints of every active row
 * as int32 pairs instead (sd [n, 2]). */
int sgs_edge_score_bwd_prep_sd(const float* codes, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, const int64_t* active_eid,
                               int64_t n_active, const float* grad_p, const float* p, const uint32_t* maskbits, float* dz, uint32_t* dvbits,
                               int32_t* sd, sgs_stream_t stream_) {
    SGS_REQUIRE(sd || n_active == 0, SGS_EINVAL, "sgs_edge_score_bwd_prep_sd: null pointer");
    return bwd_prep_impl(codes, N, H, edge_index, E, active_eid, n_active, grad_p, p, maskbits, dz, dvbits, nullptr, sd,
                         static_cast<hipStream_t>(stream_));
}

/* MODE 5 of the bf16x6 loop (see the kernel): G [n, H] = dfeat * codes[src], opart [cdiv(n, 32) + N, H] = run-end partial sums of
 * dfeat * codes[dst].  The active rows must be sorted by source. */
size_t sgs_edge_score_bwd_fused_opart_rows(int64_t n, int64_t N) { return static_cast<size_t>(cdiv(n < 0 ? 0 : n, 32) + (N < 0 ? 0 : N)); }

int sgs_edge_score_bwd_dfeat_fused(const uint32_t* dvbits, const float* dz, const int32_t* sd, const float* codes, int64_t n, int64_t N, int64_t H,
                                   const float* W1, const float* w2, float p_drop, float* G, float* opart, void* ws, size_t ws_bytes,
                                   sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n >= 0 && N > 0 && H > 0 && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL, "sgs_edge_score_bwd_dfeat_fused: bad arguments");
    SGS_REQUIRE(sgs_edge_score_bwd_bits_supported(H), SGS_EINVAL, "sgs_edge_score_bwd_dfeat_fused: H=%lld unsupported (128 or 256)", (long long)H);
    if (n == 0) return SGS_OK;
    SGS_REQUIRE(dvbits && dz && sd && codes && W1 && w2 && G && opart, SGS_EINVAL, "sgs_edge_score_bwd_dfeat_fused: null pointer");
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_workspace_bytes(0, H, 0), SGS_EWORKSPACE, "sgs_edge_score_bwd_dfeat_fused: workspace too small");
    Carver cv(ws);
    cv.take<float>(static_cast<size_t>(H) * H);
    cv.take<float>(0);
    cv.take<float>(0);
    cv.take<unsigned int>(64);
    uint4* Wp16 = cv.take<uint4>(static_cast<size_t>(H) * H * 6 / 16);
    hipLaunchKernelGGL(pack_w1a_bf16x3<true>, dim3(static_cast<unsigned>(cdiv((H / 16) * (H / 32) * 64, kT))), dim3(kT), 0, stream, W1,
                       static_cast<int>(H), Wp16, w2, 1.0f / (1.0f - p_drop));
    ScoreArgs a{};
    a.inbits = dvbits; a.indz = dz; a.n = n; a.H = static_cast<int>(H); a.feat = G; a.codes = codes; a.sd = sd; a.opart = opart;
    const dim3 grid(static_cast<unsigned>(cdiv(n, 128))), blk(256);
    bf16x6_launch_knobs(a, 5, H);
    if (H == 256) hipLaunchKernelGGL((edge_score_bf16x6_kernel<8, 4, 5>), grid, blk, 0, stream, a, Wp16);
    else          hipLaunchKernelGGL((edge_score_bf16x6_kernel<4, 4, 5>), grid, blk, 0, stream, a, Wp16);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* The reductions that follow it: out_codes, out_U (and out_U_raw, optional) as sgs_endpoint_reduce_pair_bits produces them. */
int sgs_edge_score_bwd_reduce_fused(const float* G, const float* opart, const uint32_t* dvbits, const float* dz, const float* w2, float p_drop,
                                    int64_t N, int64_t H, int64_t nnz, const int32_t* in_ptr, const int32_t* in_eid, const int32_t* out_ptr,
                                    float* out_codes, float* out_U, float* out_U_raw, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && H >= 0 && H % 32 == 0 && N < (int64_t(1) << 31) && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL,
                "sgs_edge_score_bwd_reduce_fused: needs H %% 32 == 0");
    if (N == 0 || H == 0) return SGS_OK;
    SGS_REQUIRE(G && opart && dvbits && dz && w2 && in_ptr && in_eid && out_ptr && out_codes && out_U, SGS_EINVAL,
                "sgs_edge_score_bwd_reduce_fused: null pointer");
    const dim3 grid(static_cast<unsigned>(N));
    const float scale = 1.0f / (1.0f - p_drop);
    if (nnz >= 64 * N)
        hipLaunchKernelGGL((scorer_bwd_reduce<16>), grid, dim3(1024), 0, stream, G, opart, dvbits, dz, w2, scale, N, H, in_ptr, in_eid, out_ptr,
                           out_codes, out_U, out_U_raw);
    else
        hipLaunchKernelGGL((scorer_bwd_reduce<4>), grid, dim3(256), 0, stream, G, opart, dvbits, dz, w2, scale, N, H, in_ptr, in_eid, out_ptr,
                           out_codes, out_U, out_U_raw);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_edge_score_dw2_from_parts(const float* W1, const float* T_raw, const float* U, const float* R_raw, const float* b1, const float* c_raw,
                                  int64_t N, int64_t H, float p_drop, float* dw2, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N > 0 && H > 0 && p_drop >= 0.f && p_drop < 1.f && W1 && T_raw && U && R_raw && b1 && c_raw && dw2, SGS_EINVAL,
                "sgs_edge_score_dw2_from_parts: bad arguments");
    hipLaunchKernelGGL(scorer_dw2_from_parts, dim3(static_cast<unsigned>(H)), dim3(kT), 0, stream, W1, T_raw, U, R_raw, b1, c_raw, N,
                       static_cast<int>(H), 1.0f / (1.0f - p_drop), dw2);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

int sgs_endpoint_reduce_pair_bits(const float* dfeat, const uint32_t* dvbits, const float* dz, const float* w2, float p_drop, const float* codes,
                                  int64_t N, int64_t H, int64_t nnz, const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_eid,
                                  const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* out_codes, float* out_U,
                                  float* out_U_raw, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && H >= 0 && H % 32 == 0 && N < (int64_t(1) << 31) && p_drop >= 0.f && p_drop < 1.f, SGS_EINVAL,
                "sgs_endpoint_reduce_pair_bits: needs H %% 32 == 0");
    if (N == 0 || H == 0) return SGS_OK;
    SGS_REQUIRE(dfeat && dvbits && dz && w2 && codes && in_ptr && out_ptr && out_codes && out_U, SGS_EINVAL,
                "sgs_endpoint_reduce_pair_bits: null pointer");
    const dim3 grid(static_cast<unsigned>(N));
    const float scale = 1.0f / (1.0f - p_drop);
    const float* nodv = nullptr;
    if (nnz >= 64 * N)
        hipLaunchKernelGGL((endpoint_reduce_pair_rowblock<16, true>), grid, dim3(1024), 0, stream, dfeat, nodv, codes, N, H, in_ptr, in_src, in_eid,
                           out_ptr, out_dst, out_eid, out_codes, out_U, dvbits, dz, w2, scale, out_U_raw);
    else
        hipLaunchKernelGGL((endpoint_reduce_pair_rowblock<4, true>), grid, dim3(256), 0, stream, dfeat, nodv, codes, N, H, in_ptr, in_src, in_eid,
                           out_ptr, out_dst, out_eid, out_codes, out_U, dvbits, dz, w2, scale, out_U_raw);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* ---- endpoint-dropout scorer (EdgeProbMLP with dropout > 0, model.py:16-45): see the EPD notes at edge_score_kernel ---- */
size_t sgs_edge_score_epd_workspace_bytes(int64_t H) { return carve_bytes(2 * static_cast<size_t>(H < 0 ? 0 : H) * (H < 0 ? 0 : H), 4) + 256; }

static int epd_args(ScoreArgs& a, const char* who, const float* A, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, int64_t edge_id_offset,
                    const float* W1, const float* b1, const float* w2, const float* b2, float p_hidden, uint64_t seed, uint32_t site, float p_ep,
                    uint64_t seed_x, uint32_t site_x, uint64_t seed_y, uint32_t site_y, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (int rc = check_common(who, N, H, E, p_hidden)) return rc;
    SGS_REQUIRE(H % 16 == 0, SGS_EINVAL, "%s: the endpoint-dropout scorer needs H %% 16 == 0 (H=%lld)", who, (long long)H);
    SGS_REQUIRE(p_ep >= 0.f && p_ep < 1.f, SGS_EINVAL, "%s: bad endpoint dropout probability", who);
    SGS_REQUIRE(A && edge_index && W1 && b1 && w2 && b2, SGS_EINVAL, "%s: null pointer", who);
    SGS_REQUIRE(ws && ws_bytes >= sgs_edge_score_epd_workspace_bytes(H), SGS_EWORKSPACE, "%s: workspace too small", who);
    Carver cv(ws);
    float* WT = cv.take<float>(2 * static_cast<size_t>(H) * H);
    hipLaunchKernelGGL(transpose_w1_full, dim3(cdiv(2 * H, 32), cdiv(H, 32)), dim3(kT), 0, stream, W1, static_cast<int>(H), WT);
    a.codes = A; a.U = nullptr; a.src = edge_index; a.dst = edge_index + E; a.H = static_cast<int>(H); a.row_offset = edge_id_offset;
    a.WaT = WT; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.drop_scale = 1.0f / (1.0f - p_hidden); a.drop_thresh = dropout_thresh(p_hidden); a.seed = seed; a.epoch = epoch_ptr(); a.site = site;
    a.use_drop = p_hidden > 0.f;
    a.epd = p_ep > 0.f; a.seed_x = seed_x; a.seed_y = seed_y; a.site_x = site_x; a.site_y = site_y;
    a.ep_scale = 1.0f / (1.0f - p_ep); a.ep_thresh = dropout_thresh(p_ep);
    return SGS_OK;
}

int sgs_edge_score_epd_fwd(const float* A, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, int64_t edge_id_offset, const float* W1,
                           const float* b1, const float* w2, const float* b2, float p_hidden, uint64_t seed, uint32_t site, float p_ep,
                           uint64_t seed_x, uint32_t site_x, uint64_t seed_y, uint32_t site_y, float* p_out, void* ws, size_t ws_bytes,
                           sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E == 0) return SGS_OK;
    SGS_REQUIRE(p_out, SGS_EINVAL, "sgs_edge_score_epd_fwd: null pointer");
    ScoreArgs a{};
    if (int rc = epd_args(a, "sgs_edge_score_epd_fwd", A, N, H, edge_index, E, edge_id_offset, W1, b1, w2, b2, p_hidden, seed, site, p_ep, seed_x, site_x,
                          seed_y, site_y, ws, ws_bytes, stream)) return rc;
    a.active = nullptr; a.n = E; a.p_out = p_out;
    return launch_score<false, true>(a, stream);
}

int sgs_edge_score_epd_bwd_core(const float* A, int64_t N, int64_t H, const int64_t* edge_index, int64_t E, int64_t edge_id_offset,
                                const int64_t* active_eid, int64_t n_active, const float* grad_p, const float* W1, const float* b1, const float* w2,
                                const float* b2, float p_hidden, uint64_t seed, uint32_t site, float p_ep, uint64_t seed_x, uint32_t site_x,
                                uint64_t seed_y, uint32_t site_y, float* dv, float* hdz_part, float* dz, float* feat2, void* ws, size_t ws_bytes,
                                sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(n_active >= 0 && (active_eid || n_active == E), SGS_EINVAL, "sgs_edge_score_epd_bwd_core: n_active must equal E when active_eid is NULL");
    if (n_active == 0) return SGS_OK;
    SGS_REQUIRE(grad_p && dv && hdz_part && dz && feat2, SGS_EINVAL, "sgs_edge_score_epd_bwd_core: null pointer");
    ScoreArgs a{};
    if (int rc = epd_args(a, "sgs_edge_score_epd_bwd_core", A, N, H, edge_index, E, edge_id_offset, W1, b1, w2, b2, p_hidden, seed, site, p_ep, seed_x,
                          site_x, seed_y, site_y, ws, ws_bytes, stream)) return rc;
    a.active = active_eid; a.n = n_active; a.gp = grad_p; a.dv = dv; a.hdz = hdz_part; a.dz = dz; a.feat = feat2;
    return launch_score<true, true>(a, stream);
}

int sgs_edge_score_epd_reduce(const float* dfeat2, const float* A, int64_t N, int64_t H, const int32_t* in_ptr, const int32_t* in_src,
                              const int32_t* in_eid, const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                              const int64_t* active_eid, int64_t edge_id_offset, float p_ep, uint64_t seed_x, uint32_t site_x, uint64_t seed_y,
                              uint32_t site_y, float* dA, sgs_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    SGS_REQUIRE(N >= 0 && H > 0 && H % 4 == 0 && N < (int64_t(1) << 31) && p_ep >= 0.f && p_ep < 1.f, SGS_EINVAL, "sgs_edge_score_epd_reduce: bad arguments");
    if (N == 0) return SGS_OK;
    SGS_REQUIRE(dfeat2 && A && in_ptr && in_src && in_eid && out_ptr && out_dst && out_eid && dA, SGS_EINVAL, "sgs_edge_score_epd_reduce: null pointer");
    hipLaunchKernelGGL(epd_endpoint_reduce, dim3(static_cast<unsigned>(N)), dim3(256), 0, stream, dfeat2, A, N, H, in_ptr, in_src, in_eid, out_ptr, out_dst,
                       out_eid, active_eid, edge_id_offset, seed_x, site_x, seed_y, site_y, epoch_ptr(), dropout_thresh(p_ep), 1.0f / (1.0f - p_ep),
                       p_ep > 0.f ? 1 : 0, dA);
    SGS_LAUNCH_OK();
    return SGS_OK;
}

/* probe knobs of the forward kernel (tools/stagger_probe.py; not part of the product path: both default to 0) */
int sgs_edge_score_probe_trace(unsigned long long* buf) {
    g_probe_trace = buf;
    return SGS_OK;
}
int sgs_edge_score_probe_set(int stagger, int prio, uint32_t mode_mask) {
    g_probe_stagger = stagger;
    g_probe_prio = prio;
    g_stagger_modes = mode_mask;
    return SGS_OK;
}

}  // extern "C"
