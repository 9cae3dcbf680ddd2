// libinrfit - MI355X (gfx950 / CDNA4) kernels for the per-image INR fit hot path.  C ABI: include/inrfit.h.
//
// What runs here (reference: jp-schneider/awesome, paths relative to its checkout):
//   icnn_step_kernel   fused  coords -> z0 = relu(W_in x + b_in) -> a1 = W1 z0 + b1 + S1 x (MFMA) -> z1 = relu(a1)
//                      -> y = w_o.z1 + b_o + s_o.x -> sigmoid -> SE/BCE data term -> backward (MFMA) -> per-workgroup
//                      gradient slab.  Replaces ConvexNextNet.forward + criterion + loss.backward()
//                      (awesome/model/convex_net.py:205-214, awesome/measures/weighted_loss.py:67-92,
//                      awesome/model/path_connected_net.py:941-948).
//   icnn_update_kernel fixed-order slab reduction + Adam/Adamax + enforce_convexity clamp + ReduceLROnPlateau
//                      (torch.optim.Adam/Adamax; convex_net.py:151-154,216-220; path_connected_net.py:949-951).
//
// Design (DESIGN.md has the full derivation):
//   * one workgroup = 4 waves (one per SIMD, up to 512 VGPRs each); a wave owns 16 points of a 64-point chunk;
//   * points live on the MFMA *column* (lane & 15), hidden units on the accumulator rows, so the D tile of one
//     v_mfma_f32_16x16x4_f32 is already the B operand of the next layer / of the backward product: activations never
//     leave registers between layers; weights are the A operand, read from one LDS image used by forward
//     (ds_read_b128 along a row) and backward (ds_read_b32 down a column);
//   * bias and the skip term ride in the GEMM as extra "ext" input rows (1, x, y[, t]) placed in the padding slots of
//     the last k-group, so b1/S1 gradients fall out of the dW product for free;
//   * dW1 = dZ1^T Z0ext contracts over points: the two operands are staged once through LDS (point-major rows,
//     float4 writes, conflict-free strides), the 9x9 output tiles are split over the 4 waves;
//   * fp32 everywhere (exact-f32 MFMA == fmaf chain), fixed-order reductions, no atomics: results are reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "inrfit.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int WG_THREADS = 256;
constexpr int SP = 64;  // points staged per chunk (4 waves x 16)

constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }
// smallest s >= v with s % 8 == 4 (LDS strides: float4-aligned rows whose 4-row step lands 16 banks away)
constexpr int stride_4mod8(int v) {
    int s = round_up(v, 4);
    while (s % 8 != 4) s += 4;
    return s;
}

// Static geometry of the ICNN(h, C) kernel.  "Position" = padded index of a hidden unit or ext input inside the
// 16-row MFMA tiles: hidden unit u sits at position u; the ext inputs (1, x_0..x_{C-1}) sit in free slots.
// A k-step of the 16x16x4 MFMA is (tile tk, r): lane group g = lane>>4 supplies position 16*tk + 4*g + r.
template <int H, int C>
struct Cfg {
    static constexpr int TL = H / 16;                       // full hidden tiles
    static constexpr int REM = H % 16;                      // hidden units in the partial tile
    static constexpr int MT = (H + 15) / 16;                // tiles over hidden units (M of forward, K of backward)
    static constexpr int NEXT = C + 1;                      // ext inputs: 1, x_0..x_{C-1}
    static constexpr int R_HID = REM == 0 ? 0 : (REM < 4 ? REM : 4);  // k-steps the partial tile needs for hidden units

    // position of ext input e
    static constexpr int ext_pos(int e) {
        int n = 0;
        if (REM > 0) {
            // free slots of the partial tile inside the k-steps the hidden units already use: g ascending, r ascending
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < R_HID; ++r)
                    if (4 * g + r >= REM) {
                        if (n == e) return 16 * TL + 4 * g + r;
                        ++n;
                    }
            // then new k-steps of the partial tile
            for (int r = R_HID; r < 4; ++r)
                for (int g = 0; g < 4; ++g)
                    if (4 * g + r >= REM) {
                        if (n == e) return 16 * TL + 4 * g + r;
                        ++n;
                    }
            return -1;
        }
        // hidden tiles are full: open a new tile, fill k-step r = 0 first
        return 16 * MT + 4 * (e % 4) + e / 4;
    }
    static constexpr int pos_max() {
        int m = H - 1;
        for (int e = 0; e < NEXT; ++e) m = ext_pos(e) > m ? ext_pos(e) : m;
        return m;
    }
    static constexpr int POS_MAX = pos_max();
    static_assert(ext_pos(NEXT - 1) >= 0, "no slot for the ext inputs: unsupported n_hidden % 16");
    static constexpr int MTB = POS_MAX / 16 + 1;            // tiles over positions (K of forward, N of the dW product)
    static constexpr int PT = MTB * 16;                     // padded table length
    // valid k-steps (r = 0..nr-1) of k-group tk
    static constexpr int nr_in(int tk) {                    // forward: hidden + ext inputs
        int nr = 0;
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                const int pos = 16 * tk + 4 * g + r;
                bool used = pos < H;
                for (int e = 0; e < NEXT; ++e) used = used || pos == ext_pos(e);
                if (used && r + 1 > nr) nr = r + 1;
            }
        return nr;
    }
    static constexpr int nr_out(int tk) {                   // backward: hidden outputs only
        return tk < TL ? 4 : R_HID;
    }
    static constexpr int S = stride_4mod8(POS_MAX + 1);     // weight image row stride (floats)
    static constexpr int SA = stride_4mod8(16 * TL + round_up(REM, 4));  // stage A (dz1) row stride
    static constexpr int SB = stride_4mod8(POS_MAX + 1);    // stage B (z0ext) row stride
    static constexpr int A_G_MAX = REM == 0 ? 4 : (REM + 3) / 4;         // lane groups that write tile TL of stage A
    static constexpr int B_G_MAX = (POS_MAX % 16) / 4 + 1;               // lane groups that write the last tile of stage B

    // LDS carve (floats).  [0, IMG_FLOATS) is the "parameter image": it is kept in global memory in exactly this
    // layout (written by pack_image_kernel / icnn_update_kernel) and copied verbatim at kernel start.
    static constexpr int OFF_W = 0;                           // W1ext [H+1][S] (+16 zero floats: tail reads of the zero row)
    static constexpr int OFF_WIN = OFF_W + (H + 1) * S + 16;  // [C][PT]
    static constexpr int OFF_BIN = OFF_WIN + C * PT;
    static constexpr int OFF_FLOOR = OFF_BIN + PT;
    static constexpr int OFF_WO = OFF_FLOOR + PT;
    static constexpr int OFF_SC = OFF_WO + PT;                // scalars: b_o, s_o[0..C-1]
    static constexpr int IMG_FLOATS = OFF_SC + 8;
    static_assert(IMG_FLOATS % 4 == 0, "image must be float4-copyable");
    static constexpr int OFF_STA = IMG_FLOATS;
    static constexpr int OFF_STB = OFF_STA + SP * SA + 16;
    static constexpr int LDS_FLOATS = OFF_STB + SP * SB + 16;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;

    // flat parameter offsets (L = 1), include/inrfit.h
    static constexpr int P_WIN = 0;
    static constexpr int P_BIN = H * C;
    static constexpr int P_W1 = P_BIN + H;
    static constexpr int P_B1 = P_W1 + H * H;
    static constexpr int P_S1 = P_B1 + H;
    static constexpr int P_WO = P_S1 + H * C;
    static constexpr int P_BO = P_WO + H;
    static constexpr int P_SO = P_BO + 1;
    static constexpr int P = P_SO + C;

    // dW product: MT x MTB output tiles, contiguous runs of the row-major enumeration per wave
    static constexpr int NTILES = MT * MTB;
    static constexpr int TPW = (NTILES + 3) / 4;
    static constexpr int tile_begin(int w) { return w * TPW < NTILES ? w * TPW : NTILES; }
    static constexpr int tile_end(int w) { return (w + 1) * TPW < NTILES ? (w + 1) * TPW : NTILES; }
};

// run-time description of the parameter image (same numbers as Cfg<H,C>), for the kernels that are not templated
struct ImgMap {
    int H, C, S, PT, floats;
    int off_win, off_bin, off_floor, off_wo, off_sc;
    int ext[4];
    int p_bin, p_w1, p_b1, p_s1, p_wo, p_bo, p_so, P;
};

// image offset of flat parameter j
__device__ __forceinline__ int image_offset(const ImgMap& m, int j) {
    if (j < m.p_bin) {                       // input.weight [H][C]
        const int i = j / m.C, c = j - i * m.C;
        return m.off_win + c * m.PT + i;
    }
    if (j < m.p_w1) return m.off_bin + (j - m.p_bin);
    if (j < m.p_b1) {                        // skip.0.ln.weight [H][H]
        const int q = j - m.p_w1;
        const int o = q / m.H, i = q - o * m.H;
        return o * m.S + i;
    }
    if (j < m.p_s1) return (j - m.p_b1) * m.S + m.ext[0];
    if (j < m.p_wo) {                        // skip.0.skp.weight [H][C]
        const int q = j - m.p_s1;
        const int o = q / m.C, c = q - o * m.C;
        return o * m.S + m.ext[1 + c];
    }
    if (j < m.p_bo) return m.off_wo + (j - m.p_wo);
    return m.off_sc + (j - m.p_bo);          // b_o, s_o[c]
}

struct StepArgs {
    const float* wimg;     // [n_images][IMG_FLOATS] parameter images
    const float* targets;  // [n_images][N]            (TRAIN)
    const float* coef;     // [n_images][2] c_fg, c_bg (TRAIN)
    float* slabs;          // [n_images][wgs][PS]      (TRAIN)
    float* logits;         // [n_images][N] or null
    InrGridDesc grid;
    long long N;
    int n_images, wgs, PS, loss_kind;
};

__device__ __forceinline__ float wave16_sum(float v) {  // sum over the 16 lanes that share lane>>4
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

template <int H, int C, int WAVE>
struct DwPhase {
    using G = Cfg<H, C>;
    static constexpr int T0 = G::tile_begin(WAVE), T1 = G::tile_end(WAVE);
    static constexpr int A0 = T0 / G::MTB, A1 = T1 > T0 ? (T1 - 1) / G::MTB : A0;  // row tiles touched
    static constexpr bool b_used(int b) {
        for (int t = T0; t < T1; ++t)
            if (t % G::MTB == b) return true;
        return false;
    }

    // dW[k] += sum over the 64 staged points of A-tile(a_k)^T . B-tile(b_k)
    static __device__ __forceinline__ void run(f32x4 (&dW)[G::TPW], const float* __restrict__ stA,
                                               const float* __restrict__ stB, int l15, int g) {
        for (int s = 0; s < SP / 4; ++s) {
            // k-step s: lane group g supplies staged point 16*(s/4) + (s%4) + 4*g
            const int pt = 16 * (s >> 2) + (s & 3) + 4 * g;
            const float* pa = stA + pt * G::SA + l15;
            const float* pb = stB + pt * G::SB + l15;
            float af[A1 - A0 + 1];
            float bf[G::MTB];
#pragma unroll
            for (int a = A0; a <= A1; ++a) af[a - A0] = pa[16 * a];
#pragma unroll
            for (int b = 0; b < G::MTB; ++b)
                if (b_used(b)) bf[b] = pb[16 * b];
#pragma unroll
            for (int t = T0; t < T1; ++t) dW[t - T0] = MFMA16(af[t / G::MTB - A0], bf[t % G::MTB], dW[t - T0]);
        }
    }

    // layer-0 gradients: dL0[k] += sum over the staged points of dz0-tile(a)^T . z0ext-tile(b_ext), a = WAVE + 4k.
    // Column ext_pos(0) of the result is db_in, columns ext_pos(1+c) are dW_in[:, c].
    static constexpr int BE = G::ext_pos(0) / 16;
    static constexpr int NL0 = (G::MT - WAVE + 3) / 4;  // row tiles a = WAVE, WAVE+4, ... < MT
    static __device__ __forceinline__ void run_l0(f32x4 (&dL0)[3], const float* __restrict__ stA,
                                                  const float* __restrict__ stB, int l15, int g) {
        for (int s = 0; s < SP / 4; ++s) {
            const int pt = 16 * (s >> 2) + (s & 3) + 4 * g;
            const float* pa = stA + pt * G::SA + l15;
            const float bfr = stB[pt * G::SB + 16 * BE + l15];
#pragma unroll
            for (int k = 0; k < NL0; ++k) dL0[k] = MFMA16(pa[16 * (WAVE + 4 * k)], bfr, dL0[k]);
        }
    }
    static __device__ __forceinline__ void store_l0(const f32x4 (&dL0)[3], float* __restrict__ slab, int l15, int g) {
        const int pos = 16 * BE + l15;
        int off = -1, rs = 0;
        if (pos == G::ext_pos(0)) {
            off = G::P_BIN;
            rs = 1;
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (pos == G::ext_pos(1 + c)) {
                off = G::P_WIN + c;
                rs = C;
            }
#pragma unroll
        for (int k = 0; k < NL0; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (WAVE + 4 * k) + 4 * g + r;
                if (off >= 0 && i < H) slab[off + i * rs] = dL0[k][r];
            }
    }

    // write this wave's dW tiles into the gradient slab (flat parameter order)
    static __device__ __forceinline__ void store(const f32x4 (&dW)[G::TPW], float* __restrict__ slab, int l15, int g) {
#pragma unroll
        for (int t = T0; t < T1; ++t) {
            const int a = t / G::MTB, b = t % G::MTB;
            const int pos = 16 * b + l15;  // input position (column)
            int col_off = -1;              // offset of (row o = 0) for this column, stride per row in `rs`
            int rs = 0;
            if (pos < H) {
                col_off = G::P_W1 + pos;
                rs = H;
            } else if (pos == G::ext_pos(0)) {
                col_off = G::P_B1;
                rs = 1;
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (pos == G::ext_pos(1 + c)) {
                        col_off = G::P_S1 + c;
                        rs = C;
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * a + 4 * g + r;
                if (col_off >= 0 && o < H) slab[col_off + o * rs] = dW[t - T0][r];
            }
        }
    }
};

template <int H, int C, bool TRAIN>
__global__ __launch_bounds__(WG_THREADS, 1) void icnn_step_kernel(const StepArgs a) {
    using G = Cfg<H, C>;
    constexpr int MT = G::MT, MTB = G::MTB, S = G::S, PT = G::PT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Wimg = smem + G::OFF_W;
    float* const stA = smem + G::OFF_STA;
    float* const stB = smem + G::OFF_STB;
    float* const WinT = smem + G::OFF_WIN;
    float* const binT = smem + G::OFF_BIN;
    float* const floorT = smem + G::OFF_FLOOR;
    float* const woT = smem + G::OFF_WO;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int img = blockIdx.x / a.wgs;
    const int wg = blockIdx.x - img * a.wgs;
    const long long N = a.N;

    // ---- copy the parameter image into LDS (all loads in flight at once) ------------------------------------------
    {
        const f32x4* __restrict__ src = (const f32x4*)(a.wimg + (size_t)img * G::IMG_FLOATS);
        constexpr int NV4 = G::IMG_FLOATS / 4;
        constexpr int NIT = (NV4 + WG_THREADS - 1) / WG_THREADS;
        f32x4 tmp[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * WG_THREADS;
            if (i < NV4) tmp[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * WG_THREADS;
            if (i < NV4) ((f32x4*)smem)[i] = tmp[k];
        }
    }
    float cfg_ = 0.f, cbg_ = 0.f;
    if (TRAIN) {
        cfg_ = a.coef[2 * img];
        cbg_ = a.coef[2 * img + 1];
    }
    __syncthreads();
    const float b_o = smem[G::OFF_SC];
    float s_o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s_o[c] = smem[G::OFF_SC + 1 + c];

    // per-lane LDS row offsets of the weight image: forward reads row (16t + l15), clamped to the zero row H
    int frow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int o = 16 * t + l15;
        frow[t] = (o < H ? o : H) * S + 4 * g;
    }

    // persistent per-lane gradient accumulators (TRAIN)
    f32x4 dW[G::TPW];   // this wave's tiles of dW1ext (MFMA accumulators)
    f32x4 dL0[3];       // this wave's tiles of the layer-0 gradient product
    f32x4 dwo[MT];      // dw_o partial sums of this lane's points (VALU)
    float loss_acc = 0.f, dbo = 0.f, dso[C];
    if (TRAIN) {
#pragma unroll
        for (int k = 0; k < G::TPW; ++k) dW[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) dL0[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < MT; ++t) dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) dso[c] = 0.f;
    }

    const long long n_chunks = (N + SP - 1) / SP;
    for (long long chunk = wg; chunk < n_chunks; chunk += a.wgs) {
        // ---- coordinates of this lane's point ---------------------------------------------------------------
        const long long p = chunk * SP + wave * 16 + l15;
        const bool valid = p < N;
        const long long pc = valid ? p : N - 1;
        float x[C];
        if (a.grid.mode == INR_GRID_SEPARABLE) {
            const int row = (int)(pc / a.grid.width);
            const int col = (int)(pc - (long long)row * a.grid.width);
            x[0] = a.grid.xs[col];
            x[1] = a.grid.ys[row];
            if (C > 2) x[C - 1] = a.grid.ts ? a.grid.ts[img] : 0.f;
        } else {
            const float* cp = a.grid.coords + (size_t)img * a.grid.coords_image_stride;
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = cp[(size_t)c * N + pc];
        }

        // ---- layer 0 (VALU): z0[pos] = max(W_in[pos].x + b_in[pos], floor[pos]) in B-operand layout ----------
        f32x4 z0[MTB];
#pragma unroll
        for (int t = 0; t < MTB; ++t) {
            const int q = 16 * t + 4 * g;
            f32x4 v = *(const f32x4*)&binT[q];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x4 w = *(const f32x4*)&WinT[c * PT + q];
                v += w * x[c];
            }
            const f32x4 fl = *(const f32x4*)&floorT[q];
#pragma unroll
            for (int r = 0; r < 4; ++r) z0[t][r] = fmaxf(v[r], fl[r]);
        }

        // ---- layer 1 (MFMA): acc[t] = W1ext . z0ext,  rows = hidden outputs, cols = points ---------------------
        f32x4 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tk = 0; tk < MTB; ++tk) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const f32x4 w = *(const f32x4*)&Wimg[frow[t] + 16 * tk];
#pragma unroll
                for (int r = 0; r < G::nr_in(tk); ++r) acc[t] = MFMA16(w[r], z0[tk][r], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of later k-groups from piling up in registers
        }

        // ---- output layer, sigmoid, data term ------------------------------------------------------------------
        float ypart = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const f32x4 wo = *(const f32x4*)&woT[16 * t + 4 * g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[t][r] = fmaxf(acc[t][r], 0.f);  // z1
                ypart = fmaf(wo[r], acc[t][r], ypart);
            }
        }
        ypart += __shfl_xor(ypart, 16);
        ypart += __shfl_xor(ypart, 32);
        float y = ypart + b_o;
#pragma unroll
        for (int c = 0; c < C; ++c) y = fmaf(s_o[c], x[c], y);
        if (a.logits != nullptr && valid && g == 0) a.logits[(size_t)img * N + p] = y;

        if (TRAIN) {
            const float tg = a.targets[(size_t)img * N + pc];
            const float pr = 1.f / (1.f + expf(-y));
            const float cw = tg < 0.5f ? cfg_ : cbg_;
            float l, dy;
            if (a.loss_kind == INR_LOSS_SE) {
                const float d = tg - pr;
                l = d * d * cw;
                dy = 2.f * (pr - tg) * pr * (1.f - pr) * cw;
            } else if (a.loss_kind == INR_LOSS_EXTERNAL) {
                l = 0.f;
                dy = tg;  // `targets` carries dL/dlogit
            } else {
                const float lp = fmaxf(logf(pr), -100.f), lq = fmaxf(logf(1.f - pr), -100.f);
                l = -(tg * lp + (1.f - tg) * lq) * cw;
                const float pq = pr * (1.f - pr);
                dy = (pr - tg) / fmaxf(pq, 1e-12f) * pq * cw;
            }
            if (!valid) {
                l = 0.f;
                dy = 0.f;
            }
            if (g == 0) {
                loss_acc += l;
                dbo += dy;
#pragma unroll
                for (int c = 0; c < C; ++c) dso[c] = fmaf(dy, x[c], dso[c]);
            }
            // dw_o += dy z1 ;  dz1 = dy w_o [z1 > 0]   (in place over acc)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const f32x4 wo = *(const f32x4*)&woT[16 * t + 4 * g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z1 = acc[t][r];
                    dwo[t][r] = fmaf(dy, z1, dwo[t][r]);
                    acc[t][r] = z1 > 0.f ? dy * wo[r] : 0.f;
                }
            }
            // ---- stage dz1 (A) and z0ext (B) point-major for the dW product ---------------------------------
            {
                const int pl = wave * 16 + l15;
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (t < G::TL || g < G::A_G_MAX) *(f32x4*)&stA[pl * G::SA + 16 * t + 4 * g] = acc[t];
#pragma unroll
                for (int t = 0; t < MTB; ++t)
                    if (t < MTB - 1 || g < G::B_G_MAX) *(f32x4*)&stB[pl * G::SB + 16 * t + 4 * g] = z0[t];
            }
            // ---- backward through layer 1 (MFMA): dz0[t] = W1^T . dz1, rows = hidden inputs ----------------------
            f32x4 dz0[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) dz0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tk = 0; tk < MT; ++tk) {
#pragma unroll
                for (int r = 0; r < G::nr_out(tk); ++r) {
                    const int o = 16 * tk + 4 * g + r;
                    const float* wr = Wimg + (tk < G::TL ? o : (o < H ? o : H)) * S + l15;
#pragma unroll
                    for (int t = 0; t < MT; ++t) dz0[t] = MFMA16(wr[16 * t], acc[tk][r], dz0[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // relu mask of layer 0
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool hid = (t < G::TL) || (4 * g + r < G::REM);
                    dz0[t][r] = (hid && z0[t][r] > 0.f) ? dz0[t][r] : 0.f;
                }
            }
            __syncthreads();
            // ---- dW1ext += dZ1^T Z0ext over the 64 staged points, output tiles split over the waves --------------
            if (wave == 0) DwPhase<H, C, 0>::run(dW, stA, stB, l15, g);
            else if (wave == 1) DwPhase<H, C, 1>::run(dW, stA, stB, l15, g);
            else if (wave == 2) DwPhase<H, C, 2>::run(dW, stA, stB, l15, g);
            else DwPhase<H, C, 3>::run(dW, stA, stB, l15, g);
            __syncthreads();
            // ---- layer-0 gradients: restage dz0 over dz1, multiply with the ext columns (1, x, ..) of stage B -----
            {
                const int pl = wave * 16 + l15;
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (t < G::TL || g < G::A_G_MAX) *(f32x4*)&stA[pl * G::SA + 16 * t + 4 * g] = dz0[t];
            }
            __syncthreads();
            if (wave == 0) DwPhase<H, C, 0>::run_l0(dL0, stA, stB, l15, g);
            else if (wave == 1) DwPhase<H, C, 1>::run_l0(dL0, stA, stB, l15, g);
            else if (wave == 2) DwPhase<H, C, 2>::run_l0(dL0, stA, stB, l15, g);
            else DwPhase<H, C, 3>::run_l0(dL0, stA, stB, l15, g);
            __syncthreads();
        }
    }

    if (TRAIN) {
        float* __restrict__ slab = a.slabs + ((size_t)img * a.wgs + wg) * a.PS;
        if (wave == 0) { DwPhase<H, C, 0>::store(dW, slab, l15, g); DwPhase<H, C, 0>::store_l0(dL0, slab, l15, g); }
        else if (wave == 1) { DwPhase<H, C, 1>::store(dW, slab, l15, g); DwPhase<H, C, 1>::store_l0(dL0, slab, l15, g); }
        else if (wave == 2) { DwPhase<H, C, 2>::store(dW, slab, l15, g); DwPhase<H, C, 2>::store_l0(dL0, slab, l15, g); }
        else { DwPhase<H, C, 3>::store(dW, slab, l15, g); DwPhase<H, C, 3>::store_l0(dL0, slab, l15, g); }

        // dw_o and the scalars: reduce over the 16 point-lanes, then over the 4 waves through LDS (fixed order)
        constexpr int VS = MT * 16;               // vector length (positions)
        constexpr int WSTR = VS + 8;
        float* const scr = stA;                   // [4 waves][WSTR]
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = wave16_sum(dwo[t][r]);
                if (l15 == 0) scr[wave * WSTR + 16 * t + 4 * g + r] = v0;
            }
        }
        {
            float sc[2 + C];
            sc[0] = loss_acc;
            sc[1] = dbo;
#pragma unroll
            for (int c = 0; c < C; ++c) sc[2 + c] = dso[c];
#pragma unroll
            for (int k = 0; k < 2 + C; ++k) {
                const float v = wave16_sum(sc[k]);  // only lane group 0 contributed
                if (lane == 0) scr[wave * WSTR + VS + k] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < H; i += WG_THREADS) {
            float s0 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s0 += scr[w * WSTR + i];
            slab[G::P_WO + i] = s0;
        }
        if (tid < 2 + C) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += scr[w * WSTR + VS + tid];
            if (tid == 0) slab[G::P] = sum;              // loss partial
            else if (tid == 1) slab[G::P_BO] = sum;
            else slab[G::P_SO + tid - 2] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// slab reduction + optimizer step
// ---------------------------------------------------------------------------------------------------------------------
struct UpdArgs {
    float* wimg;          // [n_images][img.floats] parameter images (kept in step with params)
    ImgMap img;
    float* params;        // [n_images][P]
    float* opt_state;     // [n_images][2P + HDR]
    const float* slabs;   // [n_images][wgs][PS]
    float* loss_hist;     // [n_images][steps] or null
    float* grads_out;     // mode 1: [n_images][P]
    float* loss_out;      // mode 1: [n_images]
    int32_t* status;      // [n_images] or null
    InrOptDesc opt;
    int P, PS, wgs, n_images;
    int t;                // 1-based optimizer step index (bias correction)
    int hist_idx, hist_stride;
    int mode;             // 0 = optimizer step, 1 = write reduced grads + loss only
    int clamp_lo0, clamp_hi0, clamp_lo1, clamp_hi1;  // flat ranges projected onto >= 0
};

constexpr int UPD_PARAMS = 64;   // parameters per block (one 256-B line per slab row)
constexpr int UPD_GROUPS = 16;   // slab groups summed in parallel, then combined in fixed order

__global__ __launch_bounds__(UPD_PARAMS * UPD_GROUPS) void icnn_update_kernel(const UpdArgs u) {
    const int img = blockIdx.y;
    const int jl = threadIdx.x, grp = threadIdx.y;
    const int j = blockIdx.x * UPD_PARAMS + jl;
    __shared__ float red[UPD_GROUPS][UPD_PARAMS];
    float part = 0.f;
    if (j <= u.P) {
        const float* __restrict__ sl = u.slabs + (size_t)img * u.wgs * u.PS + j;
        for (int w = grp; w < u.wgs; w += UPD_GROUPS) part += sl[(size_t)w * u.PS];
    }
    red[grp][jl] = part;
    __syncthreads();
    if (grp != 0 || j > u.P) return;
    float gsum = 0.f;
#pragma unroll
    for (int k = 0; k < UPD_GROUPS; ++k) gsum += red[k][jl];  // fixed order: reproducible

    if (u.mode == 1) {
        if (j < u.P) u.grads_out[(size_t)img * u.P + j] = gsum;
        else u.loss_out[img] = gsum;
        return;
    }
    float* __restrict__ st = u.opt_state + (size_t)img * (2 * (size_t)u.P + INR_OPT_HEADER_FLOATS);
    float* __restrict__ hdr = st + 2 * (size_t)u.P;
    const bool bad_before = u.status != nullptr && u.status[img] != INR_STATUS_OK;
    if (j == u.P) {
        // loss bookkeeping + ReduceLROnPlateau (torch semantics, mode 'min', relative threshold)
        const float loss = gsum;
        if (u.loss_hist) u.loss_hist[(size_t)img * u.hist_stride + u.hist_idx] = loss;
        float lr = hdr[u.t & 1];
        if (!isfinite(loss)) {
            if (u.status) u.status[img] = INR_STATUS_NONFINITE;
        } else if (u.opt.plateau) {
            float best = hdr[3];
            int num_bad = (int)hdr[4];
            if (loss < best * (1.f - u.opt.plateau_threshold)) {
                best = loss;
                num_bad = 0;
            } else {
                num_bad += 1;
            }
            if (num_bad > u.opt.plateau_patience) {
                const float nlr = fmaxf(lr * u.opt.plateau_factor, u.opt.plateau_min_lr);
                if (lr - nlr > u.opt.plateau_eps) lr = nlr;
                num_bad = 0;
            }
            hdr[3] = best;
            hdr[4] = (float)num_bad;
        }
        hdr[(u.t + 1) & 1] = lr;  // learning rate of the NEXT step (the scheduler steps after the optimizer)
        hdr[2] = lr;
        hdr[5] = loss;
        return;
    }
    if (bad_before || !isfinite(gsum)) return;

    const float lr = hdr[u.t & 1];
    float p = u.params[(size_t)img * u.P + j];
    float m = st[j], v = st[u.P + j];
    float grad = gsum;
    if (u.opt.weight_decay != 0.f) grad = __fadd_rn(grad, __fmul_rn(u.opt.weight_decay, p));
    const double b1 = (double)u.opt.beta1, b2 = (double)u.opt.beta2;
    const double bc1 = 1.0 - pow(b1, (double)u.t);
    const float w1 = (float)(1.0 - b1);
    // exp_avg.lerp_(grad, 1 - beta1)
    m = __fadd_rn(m, __fmul_rn(w1, __fsub_rn(grad, m)));
    if (u.opt.kind == INR_OPT_ADAM) {
        const double bc2 = 1.0 - pow(b2, (double)u.t);
        const float bc2_sqrt = (float)sqrt(bc2);
        const float step_size = (float)((double)lr / bc1);
        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        v = __fadd_rn(__fmul_rn(v, u.opt.beta2), __fmul_rn(__fmul_rn((float)(1.0 - b2), grad), grad));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), u.opt.eps);
        p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-step_size, m), denom));
    } else {
        // Adamax: exp_inf = max(beta2*exp_inf, |grad| + eps); p -= lr/bc1 * m / exp_inf
        v = fmaxf(__fmul_rn(v, u.opt.beta2), __fadd_rn(fabsf(grad), u.opt.eps));
        const float clr = (float)((double)lr / bc1);
        p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-clr, m), v));
    }
    if (u.opt.clamp && ((j >= u.clamp_lo0 && j < u.clamp_hi0) || (j >= u.clamp_lo1 && j < u.clamp_hi1))) p = fmaxf(p, 0.f);
    u.params[(size_t)img * u.P + j] = p;
    u.wimg[(size_t)img * u.img.floats + image_offset(u.img, j)] = p;
    st[j] = m;
    st[u.P + j] = v;
}

// params -> parameter image (zeros, ext-input constants, then every parameter at its image offset)
__global__ __launch_bounds__(256) void pack_image_kernel(const float* __restrict__ params, float* __restrict__ wimg,
                                                         const ImgMap m) {
    const int img = blockIdx.y;
    float* __restrict__ dst = wimg + (size_t)img * m.floats;
    const float* __restrict__ src = params + (size_t)img * m.P;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.floats) return;
    // which parameter (if any) lives at image offset i?  Invert image_offset by scanning is too slow; instead every
    // thread first writes the constant background of its slot, and parameter threads overwrite afterwards (2nd launch).
    float v = 0.f;
    if (i == m.off_bin + m.ext[0]) v = 1.f;                       // ext input "1": bias slot
    for (int c = 0; c < m.C; ++c) {
        if (i == m.off_win + c * m.PT + m.ext[1 + c]) v = 1.f;    // ext input x_c
        if (i == m.off_floor + m.ext[1 + c]) v = -INFINITY;       // no relu on ext inputs
    }
    if (i == m.off_floor + m.ext[0]) v = -INFINITY;
    dst[i] = v;
    (void)src;
}

__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ params, float* __restrict__ wimg,
                                                          const ImgMap m) {
    const int img = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m.P) return;
    wimg[(size_t)img * m.floats + image_offset(m, j)] = params[(size_t)img * m.P + j];
}

// per-image loss coefficients (c_fg, c_bg): 'mean' normalisation x UnariesWeightedLoss class weight
__global__ __launch_bounds__(256) void loss_coef_kernel(const float* __restrict__ targets, long long N, InrLossDesc loss,
                                                        float* __restrict__ coef) {
    const int img = blockIdx.x;
    __shared__ unsigned long long cnt[4];
    unsigned long long fg = 0;
    if (loss.weight_mode != INR_WEIGHT_NONE && loss.weight_mode != INR_WEIGHT_EXPLICIT) {
        const float* t = targets + (size_t)img * N;
        for (long long i = threadIdx.x; i < N; i += blockDim.x) fg += t[i] < 0.5f ? 1ull : 0ull;
        for (int o = 32; o > 0; o >>= 1) fg += __shfl_xor(fg, o);
        if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = fg;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float cfg, cbg;
        const float inv_n = 1.f / (float)N;
        if (loss.weight_mode == INR_WEIGHT_EXPLICIT) {
            cfg = loss.c_fg;
            cbg = loss.c_bg;
        } else if (loss.weight_mode == INR_WEIGHT_NONE) {
            cfg = cbg = inv_n;
        } else {
            const unsigned long long nfg = cnt[0] + cnt[1] + cnt[2] + cnt[3];
            const float cc = (float)(N - (long long)nfg) / (float)nfg;  // bg_count / fg_count
            float w;
            if (loss.weight_mode == INR_WEIGHT_EQUAL) w = cc;
            else if (loss.weight_mode == INR_WEIGHT_RATIO) w = (cc - 1.f) * loss.ratio + 1.f;
            else w = rintf(cc / 10.f) + 1.f;  // sssdms (torch.round = half-to-even)
            cfg = w * inv_n;
            cbg = inv_n;
        }
        coef[2 * img] = cfg;
        coef[2 * img + 1] = cbg;
    }
}

__global__ __launch_bounds__(256) void opt_init_kernel(float* opt_state, int P, InrOptDesc opt, int step0) {
    float* hdr = opt_state + (size_t)blockIdx.x * (2 * (size_t)P + INR_OPT_HEADER_FLOATS) + 2 * (size_t)P;
    if (threadIdx.x == 0) {
        float lr;
        if (step0 == 0) {
            lr = opt.lr;
            hdr[3] = INFINITY;
            hdr[4] = 0.f;
        } else {
            lr = hdr[2];
        }
        hdr[2] = lr;
        hdr[(step0 + 1) & 1] = lr;
    }
}

__global__ __launch_bounds__(256) void miou_kernel(const float* __restrict__ out, const float* __restrict__ tgt, long long N,
                                                   float thr_out, float thr_tgt, int invert, float* __restrict__ iou) {
    const int img = blockIdx.x;
    const float* o = out + (size_t)img * N;
    const float* t = tgt + (size_t)img * N;
    unsigned long long inter = 0, uni = 0, tpos = 0;
    for (long long i = threadIdx.x; i < N; i += blockDim.x) {
        bool ob = o[i] > thr_out, tb = t[i] > thr_tgt;
        if (invert) {
            ob = !ob;
            tb = !tb;
        }
        inter += (ob && tb) ? 1ull : 0ull;
        uni += (ob || tb) ? 1ull : 0ull;
        tpos += tb ? 1ull : 0ull;
    }
    for (int s = 32; s > 0; s >>= 1) {
        inter += __shfl_xor(inter, s);
        uni += __shfl_xor(uni, s);
        tpos += __shfl_xor(tpos, s);
    }
    __shared__ unsigned long long sm[3][4];
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = inter;
        sm[1][threadIdx.x >> 6] = uni;
        sm[2][threadIdx.x >> 6] = tpos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long I = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        const unsigned long long U = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        const unsigned long long T = sm[2][0] + sm[2][1] + sm[2][2] + sm[2][3];
        iou[img] = (T == 0 || U == 0) ? 0.f : (float)((double)I / (double)U);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct KernelEntry {
    int h, c;
    void (*train)(const StepArgs);
    void (*fwd)(const StepArgs);
    int lds_bytes;
    int P;
    int clamp_lo0, clamp_hi0, clamp_lo1, clamp_hi1;
    ImgMap img;
};

template <int H, int C>
constexpr KernelEntry make_entry() {
    using G = Cfg<H, C>;
    ImgMap m{};
    m.H = H; m.C = C; m.S = G::S; m.PT = G::PT; m.floats = G::IMG_FLOATS;
    m.off_win = G::OFF_WIN; m.off_bin = G::OFF_BIN; m.off_floor = G::OFF_FLOOR; m.off_wo = G::OFF_WO; m.off_sc = G::OFF_SC;
    for (int e = 0; e < 4; ++e) m.ext[e] = e < G::NEXT ? G::ext_pos(e) : 0;
    m.p_bin = G::P_BIN; m.p_w1 = G::P_W1; m.p_b1 = G::P_B1; m.p_s1 = G::P_S1; m.p_wo = G::P_WO; m.p_bo = G::P_BO;
    m.p_so = G::P_SO; m.P = G::P;
    return KernelEntry{H, C, icnn_step_kernel<H, C, true>, icnn_step_kernel<H, C, false>, G::LDS_BYTES, G::P,
                       G::P_W1, G::P_W1 + H * H, G::P_WO, G::P_WO + H, m};
}

const KernelEntry kEntries[] = {
    make_entry<130, 2>(), make_entry<130, 3>(), make_entry<64, 2>(), make_entry<64, 3>(),
    make_entry<32, 2>(),  make_entry<32, 3>(),
};

const KernelEntry* find_entry(const InrModelDesc* m) {
    if (!m || m->kind != INR_MODEL_ICNN || m->n_layers != 1) return nullptr;
    for (const auto& e : kEntries)
        if (e.h == m->n_hidden && e.c == m->in_features) return &e;
    return nullptr;
}

int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

int wgs_per_image(long long n_points, int n_images) {
    const long long n_chunks = (n_points + SP - 1) / SP;
    long long w = cu_count() / (n_images > 0 ? n_images : 1);
    if (w < 1) w = 1;
    if (w > n_chunks) w = n_chunks;
    return (int)w;
}

int check_grid(const InrGridDesc* g, const KernelEntry* e, int n_images) {
    if (!g || g->n_points <= 0 || n_images <= 0) return INR_EINVAL;
    if (g->mode == INR_GRID_SEPARABLE) {
        if (!g->xs || !g->ys || g->width <= 0 || g->height <= 0) return INR_EINVAL;
        if ((long long)g->width * g->height != g->n_points) return INR_EINVAL;
        if (e->c > 3) return INR_EINVAL;
    } else if (g->mode == INR_GRID_EXPLICIT) {
        if (!g->coords) return INR_EINVAL;
    } else {
        return INR_EINVAL;
    }
    return INR_OK;
}

int set_lds(const KernelEntry* e) {
    // >64 KiB of dynamic LDS needs the opt-in attribute (idempotent, cheap)
    if (hipFuncSetAttribute((const void*)e->train, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes) != hipSuccess)
        return INR_ELAUNCH;
    if (hipFuncSetAttribute((const void*)e->fwd, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes) != hipSuccess)
        return INR_ELAUNCH;
    return INR_OK;
}

struct Workspace {
    float* coef;
    float* wimg;
    float* slabs;
    int wgs, PS;
    long long bytes;
};

Workspace carve(const KernelEntry* e, long long n_points, int n_images, void* base) {
    Workspace w;
    w.wgs = wgs_per_image(n_points, n_images);
    w.PS = (e->P + 1 + 3) / 4 * 4;
    const long long coef_bytes = ((long long)n_images * 2 * 4 + 255) / 256 * 256;
    const long long img_bytes = ((long long)n_images * e->img.floats * 4 + 255) / 256 * 256;
    const long long slab_bytes = (long long)n_images * w.wgs * w.PS * 4;
    w.coef = (float*)base;
    w.wimg = (float*)((char*)base + coef_bytes);
    w.slabs = (float*)((char*)base + coef_bytes + img_bytes);
    w.bytes = coef_bytes + img_bytes + slab_bytes;
    return w;
}

}  // namespace

extern "C" {

int inrfit_query(int* abi_version, int* max_hidden, int* lds_bytes) {
    if (abi_version) *abi_version = INRFIT_ABI_VERSION;
    if (max_hidden) *max_hidden = 130;
    if (lds_bytes) *lds_bytes = Cfg<130, 2>::LDS_BYTES;
    return INR_OK;
}

int inrfit_supported(const InrModelDesc* model) { return find_entry(model) ? 1 : 0; }

int64_t inrfit_param_count(const InrModelDesc* m) {
    if (!m || m->kind != INR_MODEL_ICNN || m->n_hidden <= 0 || m->in_features <= 0 || m->n_layers < 0) return INR_EINVAL;
    const int64_t h = m->n_hidden, c = m->in_features, l = m->n_layers;
    return h * c + h + l * (h * h + h + h * c) + h + 1 + c;
}

int64_t inrfit_opt_state_floats(const InrModelDesc* m) {
    const int64_t p = inrfit_param_count(m);
    return p < 0 ? p : 2 * p + INR_OPT_HEADER_FLOATS;
}

int64_t inrfit_workspace_bytes(const InrModelDesc* model, const InrGridDesc* grid, int n_images) {
    const KernelEntry* e = find_entry(model);
    if (!e) return INR_EUNSUPPORTED;
    if (!grid || grid->n_points <= 0 || n_images <= 0) return INR_EINVAL;
    return carve(e, grid->n_points, n_images, nullptr).bytes;
}

static int launch_pack(const KernelEntry* e, const Workspace& w, const float* params, int n_images, hipStream_t s) {
    hipLaunchKernelGGL(pack_image_kernel, dim3((e->img.floats + 255) / 256, n_images), dim3(256), 0, s, params, w.wimg, e->img);
    hipLaunchKernelGGL(pack_params_kernel, dim3((e->P + 255) / 256, n_images), dim3(256), 0, s, params, w.wimg, e->img);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

static int launch_step(const KernelEntry* e, const Workspace& w, bool train, const InrGridDesc* grid, const float* targets,
                       int loss_kind, int n_images, float* logits, hipStream_t s) {
    StepArgs a{};
    a.wimg = w.wimg;
    a.targets = targets;
    a.coef = w.coef;
    a.slabs = w.slabs;
    a.logits = logits;
    a.grid = *grid;
    a.N = grid->n_points;
    a.n_images = n_images;
    a.wgs = w.wgs;
    a.PS = w.PS;
    a.loss_kind = loss_kind;
    hipLaunchKernelGGL(train ? e->train : e->fwd, dim3((unsigned)(n_images * w.wgs)), dim3(WG_THREADS), e->lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

static int check_loss(const InrLossDesc* l) {
    if (!l) return INR_EINVAL;
    if (l->kind != INR_LOSS_SE && l->kind != INR_LOSS_BCE && l->kind != INR_LOSS_EXTERNAL) return INR_EINVAL;
    if (l->weight_mode < INR_WEIGHT_NONE || l->weight_mode > INR_WEIGHT_EXPLICIT) return INR_EINVAL;
    return INR_OK;
}

// common argument checks + workspace carve
static int prepare(const InrModelDesc* model, const InrGridDesc* grid, int n_images, void* workspace, int64_t workspace_bytes,
                   const KernelEntry** e_out, Workspace* w_out) {
    const KernelEntry* e = find_entry(model);
    if (!e) return INR_EUNSUPPORTED;
    if (!workspace) return INR_EINVAL;
    int rc = check_grid(grid, e, n_images);
    if (rc) return rc;
    *w_out = carve(e, grid->n_points, n_images, workspace);
    if (workspace_bytes < w_out->bytes) return INR_EWORKSPACE;
    if ((rc = set_lds(e))) return rc;
    *e_out = e;
    return INR_OK;
}

static void launch_reduce(const KernelEntry* e, const Workspace& w, int n_images, float* grads, float* loss_out, hipStream_t s) {
    UpdArgs u{};
    u.slabs = w.slabs;
    u.grads_out = grads;
    u.loss_out = loss_out;
    u.P = e->P;
    u.PS = w.PS;
    u.wgs = w.wgs;
    u.n_images = n_images;
    u.mode = 1;
    hipLaunchKernelGGL(icnn_update_kernel, dim3((e->P + 1 + UPD_PARAMS - 1) / UPD_PARAMS, n_images),
                       dim3(UPD_PARAMS, UPD_GROUPS), 0, s, u);
}

int inrfit_forward(const InrModelDesc* model, const float* params, const InrGridDesc* grid, int n_images, float* logits,
                   void* workspace, int64_t workspace_bytes, void* stream) {
    const KernelEntry* e;
    Workspace w;
    if (!params || !logits) return INR_EINVAL;
    int rc = prepare(model, grid, n_images, workspace, workspace_bytes, &e, &w);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_pack(e, w, params, n_images, s))) return rc;
    return launch_step(e, w, false, grid, nullptr, 0, n_images, logits, s);
}

int inrfit_loss_grad(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* targets,
                     const InrLossDesc* loss, int n_images, float* loss_out, float* grads, void* workspace,
                     int64_t workspace_bytes, void* stream) {
    const KernelEntry* e;
    Workspace w;
    if (!params || !targets || !loss_out || !grads) return INR_EINVAL;
    int rc = check_loss(loss);
    if (rc) return rc;
    if ((rc = prepare(model, grid, n_images, workspace, workspace_bytes, &e, &w))) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_coef_kernel, dim3(n_images), dim3(256), 0, s, targets, (long long)grid->n_points, *loss, w.coef);
    if ((rc = launch_pack(e, w, params, n_images, s))) return rc;
    if ((rc = launch_step(e, w, true, grid, targets, loss->kind, n_images, nullptr, s))) return rc;
    launch_reduce(e, w, n_images, grads, loss_out, s);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

int inrfit_backward(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* dlogits,
                    int n_images, float* grads, void* workspace, int64_t workspace_bytes, void* stream) {
    const KernelEntry* e;
    Workspace w;
    if (!params || !dlogits || !grads) return INR_EINVAL;
    int rc = prepare(model, grid, n_images, workspace, workspace_bytes, &e, &w);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_pack(e, w, params, n_images, s))) return rc;
    if ((rc = launch_step(e, w, true, grid, dlogits, INR_LOSS_EXTERNAL, n_images, nullptr, s))) return rc;
    launch_reduce(e, w, n_images, grads, w.coef /* scratch: the loss slot is unused in this mode */, s);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

int inrfit_step_only(const InrModelDesc* model, const float* params, const InrGridDesc* grid, const float* targets,
                     const InrLossDesc* loss, int n_images, int iters, void* workspace, int64_t workspace_bytes,
                     void* stream) {
    const KernelEntry* e;
    Workspace w;
    if (!params || !targets || iters < 0) return INR_EINVAL;
    int rc = check_loss(loss);
    if (rc) return rc;
    if ((rc = prepare(model, grid, n_images, workspace, workspace_bytes, &e, &w))) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_coef_kernel, dim3(n_images), dim3(256), 0, s, targets, (long long)grid->n_points, *loss, w.coef);
    if ((rc = launch_pack(e, w, params, n_images, s))) return rc;
    for (int it = 0; it < iters; ++it)
        if ((rc = launch_step(e, w, true, grid, targets, loss->kind, n_images, nullptr, s))) return rc;
    return INR_OK;
}

int inrfit_fit(const InrModelDesc* model, float* params, float* opt_state, const InrGridDesc* grid, const float* targets,
               const InrLossDesc* loss, const InrOptDesc* opt, int n_images, int steps, int step0, float* loss_hist,
               float* final_logits, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream) {
    const KernelEntry* e;
    Workspace w;
    if (!params || !opt_state || !targets || !opt || steps < 0 || step0 < 0) return INR_EINVAL;
    if (opt->kind != INR_OPT_ADAM && opt->kind != INR_OPT_ADAMAX) return INR_EINVAL;
    int rc = check_loss(loss);
    if (rc) return rc;
    if (loss->kind == INR_LOSS_EXTERNAL) return INR_EINVAL;
    if ((rc = prepare(model, grid, n_images, workspace, workspace_bytes, &e, &w))) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_coef_kernel, dim3(n_images), dim3(256), 0, s, targets, (long long)grid->n_points, *loss, w.coef);
    hipLaunchKernelGGL(opt_init_kernel, dim3(n_images), dim3(64), 0, s, opt_state, e->P, *opt, step0);
    if (status) {
        if (hipMemsetAsync(status, 0, sizeof(int32_t) * n_images, s) != hipSuccess) return INR_ELAUNCH;
    }
    if ((rc = launch_pack(e, w, params, n_images, s))) return rc;
    UpdArgs u{};
    u.wimg = w.wimg;
    u.img = e->img;
    u.params = params;
    u.opt_state = opt_state;
    u.slabs = w.slabs;
    u.loss_hist = loss_hist;
    u.status = status;
    u.opt = *opt;
    u.P = e->P;
    u.PS = w.PS;
    u.wgs = w.wgs;
    u.n_images = n_images;
    u.hist_stride = steps;
    u.mode = 0;
    u.clamp_lo0 = e->clamp_lo0;
    u.clamp_hi0 = e->clamp_hi0;
    u.clamp_lo1 = e->clamp_lo1;
    u.clamp_hi1 = e->clamp_hi1;
    const dim3 ugrid((e->P + 1 + UPD_PARAMS - 1) / UPD_PARAMS, n_images);
    for (int it = 0; it < steps; ++it) {
        if ((rc = launch_step(e, w, true, grid, targets, loss->kind, n_images, nullptr, s))) return rc;
        u.t = step0 + it + 1;
        u.hist_idx = it;
        hipLaunchKernelGGL(icnn_update_kernel, ugrid, dim3(UPD_PARAMS, UPD_GROUPS), 0, s, u);
    }
    if (hipGetLastError() != hipSuccess) return INR_ELAUNCH;
    if (final_logits) return launch_step(e, w, false, grid, nullptr, 0, n_images, final_logits, s);
    return INR_OK;
}

int inrfit_miou(const float* out, const float* tgt, int n_images, int64_t n_points, float thr_out, float thr_tgt, int invert,
                float* iou, void* stream) {
    if (!out || !tgt || !iou || n_images <= 0 || n_points <= 0) return INR_EINVAL;
    hipLaunchKernelGGL(miou_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, out, tgt, (long long)n_points, thr_out,
                       thr_tgt, invert, iou);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

const char* inrfit_strerror(int code) {
    switch (code) {
        case INR_OK: return "ok";
        case INR_EINVAL: return "invalid argument";
        case INR_EUNSUPPORTED: return "model shape has no compiled kernel";
        case INR_EWORKSPACE: return "workspace too small";
        case INR_ELAUNCH: return "HIP launch failed";
        case INR_ENODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

}  // extern "C"
