// icnn_step2.h - fused step kernel for the ICNN with TWO hidden skip layers (ConvexNextNet(n_hidden_layers=2),
// awesome/model/convex_net.py:177-220).  Same building blocks as icnn_step.h (read that header first):
//   z0 = relu(W_in x + b_in); z1 = relu(W1 z0 + b1 + S1 x); z2 = relu(W2 z1 + b2 + S2 x); y = w_o.z2 + b_o + s_o.x
// LDS budget: two 72.8 KB weight images + tables = 151 KB, so there is no room for separate stage buffers.  The two dW
// products only run when no layer product needs W2, so the stage ALIASES the W2 image: after the backward pass the
// waves overwrite it with (dz2 | z1ext), multiply, overwrite it with (dz1 | z0ext), multiply, and then W2 is
// re-fetched from HBM/L2 straight into LDS by LDS-DMA (global_load_lds_dwordx4, no registers) while the next chunk's
// layer-1 product - which only needs W1 - is already running.
#pragma once
#include "icnn_step.h"


namespace {

template <int H, int C>
struct Cfg2 {
    using G1 = Cfg<H, C>;
    static constexpr int TM = G1::TM, HM = G1::HM, HR = G1::HR, NEXT = G1::NEXT, NRL = G1::NRL, KG = G1::KG, PT = G1::PT;
    static constexpr int S = G1::S, SA = G1::SA, SB = G1::SB, RPW = G1::RPW;
    static constexpr int ext_pos(int e) { return G1::ext_pos(e); }
    static constexpr int nr_in(int tk) { return G1::nr_in(tk); }
    // ---- LDS / image carve (floats) --------------------------------------------------------------------------------
    static constexpr int OFF_SC = 0;
    static constexpr int OFF_WINE = OFF_SC + 8;                  // [4][PT]
    static constexpr int OFF_WIN = OFF_WINE + 4 * PT;            // [C][16]
    static constexpr int OFF_BIN = OFF_WIN + C * 16;
    static constexpr int OFF_FLOOR = OFF_BIN + 16;
    static constexpr int OFF_WO = OFF_FLOOR + 16;                // [PT]
    static constexpr int OFF_WCT0 = OFF_WO + PT;                 // [HR][PT] W1[:, HM+u]
    static constexpr int OFF_WCT1 = OFF_WCT0 + HR * PT;          // [HR][PT] W2[:, HM+u]
    static constexpr int WREG = round_up(H * S + 16, 256);       // one weight image region (multiple of a 1 KB DMA piece)
    static constexpr int OFF_W0 = round_up(OFF_WCT1 + HR * PT, 4);
    static constexpr int OFF_W1 = OFF_W0 + WREG;
    static constexpr int IMG_FLOATS = OFF_W1 + WREG;
    static constexpr int STAGE_FLOATS = SP * SA + 16 + SP * SB + 16;
    static constexpr bool ALIAS = STAGE_FLOATS <= WREG;         // h = 130: the stage aliases the W2 image (re-fetched per chunk)
    static constexpr int OFF_STA = ALIAS ? OFF_W1 : IMG_FLOATS; // small h: LDS has room for a separate stage
    static constexpr int OFF_STB = OFF_STA + SP * SA + 16;
    static constexpr int OFF_ACC = ALIAS ? IMG_FLOATS : IMG_FLOATS + STAGE_FLOATS;   // [4 waves][PT] d w_o, accumulated per chunk
    static constexpr int LDS_FLOATS = OFF_ACC + 4 * PT;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
    static_assert(OFF_W0 % 4 == 0 && IMG_FLOATS % 4 == 0, "alignment");
    // ---- flat parameter offsets (L = 2) -------------------------------------------------------------------------------
    static constexpr int P_WIN = 0;
    static constexpr int P_BIN = H * C;
    static constexpr int P_W1 = P_BIN + H;
    static constexpr int P_B1 = P_W1 + H * H;
    static constexpr int P_S1 = P_B1 + H;
    static constexpr int P_W2 = P_S1 + H * C;
    static constexpr int P_B2 = P_W2 + H * H;
    static constexpr int P_S2 = P_B2 + H;
    static constexpr int P_WO = P_S2 + H * C;
    static constexpr int P_BO = P_WO + H;
    static constexpr int P_SO = P_BO + 1;
    static constexpr int P = P_SO + C;
    // ---- gradient slab (icnn_step.h, Cfg: gradient slab - here with two layers of tiles) ---------------------------------
    static constexpr int SL_TILE = TM * KG * 256;                 // per layer
    static constexpr int SL_IN = 2 * SL_TILE;                     // W_in, b_in at their parameter offsets
    static constexpr int SL_LS = HR * (H + 1 + C);                // leftover block of one layer: W rows | b | S rows
    static constexpr int SL_L1 = SL_IN + P_W1;
    static constexpr int SL_L2 = SL_L1 + SL_LS;
    static constexpr int SL_WO = SL_L2 + SL_LS;                   // w_o, b_o, s_o, loss
    static constexpr int SL_LOSS = SL_WO + (P - P_WO);
    static constexpr int SL_COLS = SL_LOSS + 1;
};

#if INR_STAMPS
__device__ unsigned long long g_stamps2[16];   // per-phase cycle sums of workgroup 0 / wave 0 of the last L = 2 launch (tools/stamps2.py)
#endif

template <int H, int C, bool TRAIN, bool DX = false, int ACT0 = INR_ACT_RELU>
__global__ __launch_bounds__(WG_THREADS, 1) void icnn2_step_kernel(const StepArgs a) {
    static_assert(!DX || TRAIN, "coordinate gradients are a by-product of the backward pass");
    using G = Cfg2<H, C>;
    constexpr int TM = G::TM, KG = G::KG, HM = G::HM, HR = G::HR, S = G::S, PT = G::PT, RPW = G::RPW, NEXT = G::NEXT;
    constexpr int HRA = HR > 0 ? HR : 1;
    static_assert(TM % RPW == 0, "row tiles must split evenly over the waves that own rows");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const W0 = smem + G::OFF_W0;
    float* const W1 = smem + G::OFF_W1;
    float* const WinE = smem + G::OFF_WINE;
    float* const WinT = smem + G::OFF_WIN;
    float* const binT = smem + G::OFF_BIN;
    float* const floorT = smem + G::OFF_FLOOR;
    float* const woT = smem + G::OFF_WO;
    float* const stA = smem + G::OFF_STA;
    float* const stB = smem + G::OFF_STB;
    float* const accW = smem + G::OFF_ACC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int img = blockIdx.x / a.wgs;
    const int wg = blockIdx.x - img * a.wgs;
    const long long N = a.N;
    const float* __restrict__ gimg = a.wimg + (size_t)img * G::IMG_FLOATS;

    {   // whole image -> LDS
        const f32x4* __restrict__ src = (const f32x4*)gimg;
        constexpr int NV4 = G::IMG_FLOATS / 4;
        for (int i0 = 0; i0 < NV4; i0 += 8 * WG_THREADS) {
            f32x4 tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + tid + k * WG_THREADS;
                if (i < NV4) tmp[k] = src[i];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + tid + k * WG_THREADS;
                if (i < NV4) ((f32x4*)smem)[i] = tmp[k];
            }
        }
    }
    float cfg_ = 0.f, cbg_ = 0.f;
    if (TRAIN) {
        cfg_ = a.coef[2 * img];
        cbg_ = a.coef[2 * img + 1];
        for (int i = tid; i < 4 * PT; i += WG_THREADS) accW[i] = 0.f;
    }
    __syncthreads();
    const float b_o = smem[G::OFF_SC];
    float s_o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s_o[c] = smem[G::OFF_SC + 1 + c];
    float wol[HRA];
#pragma unroll
    for (int u = 0; u < HRA; ++u) wol[u] = HR > 0 ? woT[HM + u] : 0.f;
    const bool row_ok = wave * RPW < TM;
    const int arow = 16 * wave * RPW + l15;

    // persistent gradient accumulators
    f32x4 dWa[RPW][KG], dWb[RPW][KG];  // dW1ext, dW2ext tiles of this wave
    // d w_o is summed over the chunk's points right away and kept in LDS (accW): with it in registers hipcc kept ~100
    // accumulator registers in scratch memory and re-loaded them every chunk (85 -> 25 spilled registers, -6 % kernel time)
    f32x4 dL0[TM];
    float dwol[HRA], dWla[HRA][KG], dWlb[HRA][KG], dL0l[HRA][NEXT];
    float loss_acc = 0.f, dbo = 0.f, dso[C];
    if (TRAIN) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                dWa[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                dWb[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int t = 0; t < TM; ++t) dL0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < HRA; ++u) {
            dwol[u] = 0.f;
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                dWla[u][b] = 0.f;
                dWlb[u][b] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < NEXT; ++e) dL0l[u][e] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dso[c] = 0.f;
    }

    // ---- building blocks ------------------------------------------------------------------------------------------
    // forward product of one layer: acc[t] = Wl[16t.., :] . B ; la[u] = leftover rows . B (partial over this lane group)
    auto gemm_fwd = [&](const float* Wl, const f32x4 (&B)[KG], f32x4 (&acc)[TM], float (&la)[HRA]) {
        const float* const wf = Wl + l15 * S + 4 * g;
#pragma unroll
        for (int u = 0; u < HRA; ++u) la[u] = 0.f;
        f32x4 wq[2][TM], wlq[2][HRA];
#pragma unroll
        for (int t = 0; t < TM; ++t) wq[0][t] = *(const f32x4*)(wf + t * 16 * S);
#pragma unroll
        for (int u = 0; u < HR; ++u) wlq[0][u] = *(const f32x4*)(Wl + (HM + u) * S + 4 * g);
#pragma unroll
        for (int tk = 0; tk < KG; ++tk) {
            const int cur = tk & 1, nx = cur ^ 1;
            // next k-group's operand reads one after every other product, pinned (see icnn_step.h: reads in a burst stall the issue)
            auto next_read = [&](int i) {
                if (tk + 1 < KG) {
                    if (i < TM) wq[nx][i] = *(const f32x4*)(wf + i * 16 * S + 16 * (tk + 1));
                    else wlq[nx][i - TM] = *(const f32x4*)(Wl + (HM + i - TM) * S + 16 * (tk + 1) + 4 * g);
                }
            };
            constexpr int NRD = TM + HR;
            const int nmf = G::nr_in(tk) * TM;
            const int every = nmf / NRD > 0 ? nmf / NRD : 1;
#pragma unroll
            for (int r = 0; r < G::nr_in(tk); ++r) {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    acc[t] = MFMA16(wq[cur][t][r], B[tk][r], (tk == 0 && r == 0) ? (f32x4{0.f, 0.f, 0.f, 0.f}) : acc[t]);
                    const int q = r * TM + t;
                    if (q % every == every - 1 && q / every < NRD) next_read(q / every);
                    OPERAND_FENCE();
                }
            }
#pragma unroll
            for (int i = 0; i < NRD; ++i)
                if (i >= nmf / every) next_read(i);
#pragma unroll
            for (int u = 0; u < HR; ++u)
#pragma unroll
                for (int r = 0; r < G::nr_in(tk); ++r) la[u] = fmaf(wlq[cur][u][r], B[tk][r], la[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // backward product of one layer.  SWAP = false: out[t] rows = hidden inputs 16t + 4g + r, columns = points (the
    // B-operand layout, feeds the next backward product); SWAP = true: transposed (rows = points), feeds the layer-0
    // gradient product.  outl[u]: leftover hidden inputs (partial over this lane group's positions).
    auto gemm_bwd = [&](auto swap_tag, const float* Wl, const float* WcTl, const f32x4 (&dz)[TM], const float (&dzl)[HRA],
                        f32x4 (&out)[TM], float (&outl)[HRA], f32x4& outx) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        float bqx[2] = {0.f, 0.f};  // DX: the columns of k-group TM (skip-path inputs) as one more output tile
        outx = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int KS = 4 * TM + HR;
        const float* const wb = Wl + l15;
#pragma unroll
        for (int u = 0; u < HRA; ++u) outl[u] = 0.f;
        float bq[2][TM];
        f32x4 wcq[2][HRA];
        auto b_row = [&](int ks) -> const float* {
            const int tk = ks >> 2, r = ks & 3;
            if (tk < TM) return wb + (16 * tk + 4 * g + r) * S;
            return wb + (g == 0 ? (HM + r) * S : 0);
        };
        {
            const float* br = b_row(0);
#pragma unroll
            for (int t = 0; t < TM; ++t) bq[0][t] = br[16 * t];
            if (DX) bqx[0] = br[16 * TM];
#pragma unroll
            for (int u = 0; u < HR; ++u) wcq[0][u] = *(const f32x4*)(WcTl + u * PT + 4 * g);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tk = ks >> 2, r = ks & 3;
            if (ks + 1 < KS) {
                const float* br = b_row(ks + 1);
                if (DX) bqx[(ks + 1) & 1] = br[16 * TM];
            }
            if (r == 0 && (tk + 1) * 4 < KS) {
#pragma unroll
                for (int u = 0; u < HR; ++u) wcq[(tk + 1) & 1][u] = *(const f32x4*)(WcTl + u * PT + 16 * (tk + 1) + 4 * g);
            }
            OPERAND_FENCE();
            const float bop = tk < TM ? dz[tk < TM ? tk : 0][r] : (g == 0 ? dzl[r < HRA ? r : 0] : 0.f);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const f32x4 c0 = ks == 0 ? (f32x4{0.f, 0.f, 0.f, 0.f}) : out[t];
                out[t] = SWAP ? MFMA16(bop, bq[ks & 1][t], c0) : MFMA16(bq[ks & 1][t], bop, c0);
                if (ks + 1 < KS) bq[(ks + 1) & 1][t] = b_row(ks + 1)[16 * t];   // next k-step's operands, one read per product
                OPERAND_FENCE();
            }
            if (DX) outx = SWAP ? MFMA16(bop, bqx[ks & 1], outx) : MFMA16(bqx[ks & 1], bop, outx);
            MFMA_STEP_FENCE();
#pragma unroll
            for (int u = 0; u < HR; ++u) {  // leftover hidden inputs, this k-step's share
                if (tk < TM) outl[u] = fmaf(wcq[tk & 1][u][r], dz[tk < TM ? tk : 0][r], outl[u]);
                else if (g == 0) outl[u] = fmaf(wcq[tk & 1][u][r], dzl[r < HRA ? r : 0], outl[u]);
            }
        }
    };
    const int pl = wave * 16 + l15;
    float* const sa = stA + pl * G::SA + 4 * g;
    float* const sb = stB + pl * G::SB + 4 * g;
    // stage (dz | leftover dz) as A rows and (zprev main tiles | last k-group) as B rows of this lane's point
    auto stage = [&](const f32x4 (&dz)[TM], const float (&dzl)[HRA], const f32x4 (&zp)[KG]) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            *(f32x4*)(sa + 16 * t) = dz[t];
            *(f32x4*)(sb + 16 * t) = zp[t];
        }
        if (HR > 0 && g == 0) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < HR; ++u) v[u] = dzl[u];
            *(f32x4*)(sa + HM) = v;
        }
        if (g < 3) *(f32x4*)(sb + HM) = zp[TM];
    };
    // dW += A^T B over the 64 staged points (this wave's row tiles x all column tiles); leftover rows on the VALU
    auto dw_phase = [&](f32x4 (&dW)[RPW][KG], float (&dWl)[HRA][KG]) {
        auto stage_pt = [&](int it) { const int s = (it + wave) & 15; return 16 * (s >> 2) + (s & 3) + 4 * g; };
        float af[2][RPW], bf[2][KG];
        {
            const int pt = stage_pt(0);
#pragma unroll
            for (int j = 0; j < RPW; ++j) af[0][j] = stA[pt * G::SA + arow + 16 * j];
#pragma unroll
            for (int b = 0; b < KG; ++b) bf[0][b] = stB[pt * G::SB + 16 * b + l15];
        }
#pragma unroll
        for (int it = 0; it < SP / 4; ++it) {
            const int ptc = stage_pt(it);
            if (it + 1 < SP / 4) {
                const int pt = stage_pt(it + 1);
#pragma unroll
                for (int j = 0; j < RPW; ++j) af[(it + 1) & 1][j] = stA[pt * G::SA + arow + 16 * j];
#pragma unroll
                for (int b = 0; b < KG; ++b) bf[(it + 1) & 1][b] = stB[pt * G::SB + 16 * b + l15];
            }
            if (HR > 0 && (it & 3) == 0) {
                const f32x4 dl = *(const f32x4*)(stA + ptc * G::SA + HM);
#pragma unroll
                for (int u = 0; u < HR; ++u)
#pragma unroll
                    for (int b = 0; b < KG; ++b) dWl[u][b] = fmaf(dl[u], bf[it & 1][b], dWl[u][b]);
            }
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < RPW; ++j)
#pragma unroll
                    for (int b = 0; b < KG; ++b) dW[j][b] = MFMA16(af[it & 1][j], bf[it & 1][b], dW[j][b]);
                MFMA_STEP_FENCE();
            }
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                SGB(SG_DS_READ, 2);
                SGB(SG_VALU, 2);
                SGB(SG_MFMA, RPW);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int n_chunks = (int)((N + SP - 1) / SP);
    struct PointIn {
        float x[C];
        float tg;
    };
    auto load_point = [&](int chunk) -> PointIn {
        PointIn q;
        int pc = chunk * SP + wave * 16 + l15;
        pc = pc < (int)N ? pc : (int)N - 1;
        if (a.grid.mode == INR_GRID_SEPARABLE) {
            const int row = pc / a.grid.width;
            const int col = pc - row * a.grid.width;
            q.x[0] = a.grid.xs[col];
            q.x[1] = a.grid.ys[row];
            if (C > 2) q.x[C - 1] = a.grid.ts ? a.grid.ts[img] : 0.f;
        } else {
            const float* cp = a.grid.coords + (size_t)img * a.grid.coords_image_stride;
#pragma unroll
            for (int c = 0; c < C; ++c) q.x[c] = cp[(size_t)c * N + pc];
        }
        q.tg = TRAIN ? a.targets[(size_t)img * N + pc] : 0.f;
        return q;
    };
    PointIn nxt = load_point(wg);
#if INR_STAMPS
    unsigned long long st_sum[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
    const unsigned long long st_begin = st_prev;
#endif
    for (int chunk = wg; chunk < n_chunks; chunk += a.wgs) {
        STAMP(0);
        const int p = chunk * SP + wave * 16 + l15;
        const bool valid = p < (int)N;
        const PointIn cur = nxt;
        {
            const int cn = chunk + a.wgs;
            nxt = load_point(cn < n_chunks ? cn : chunk);
        }
        float x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = cur.x[c];
        const float tg = cur.tg;

        // ---- layer 0: all z0ext tiles up front (TM products + relu on the matrix pipe, last k-group on the VALU) --------
        const float xe = g < C ? x[g < C ? g : 0] : (g == C ? 1.f : 0.f);
        // z0 main tiles are cheap to rebuild (TM single-k-step products + relu), so they are NOT kept across the backward
        // pass: they are recomputed when layer 1's dW operands are staged.
        auto z0_main = [&](f32x4 (&z)[KG]) {
            float wie[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) wie[t] = WinE[g * PT + 16 * t + l15];
            OPERAND_FENCE();
#pragma unroll
            for (int t = 0; t < TM; ++t) z[t] = MFMA16(wie[t], xe, (f32x4{0.f, 0.f, 0.f, 0.f}));
            MFMA_STEP_FENCE();
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[t][r] = act0_f<ACT0>(z[t][r], a.act_omega);
        };
        f32x4 z0last;  // last k-group of z0ext: leftover hidden units (lane group 0) + ext inputs (lane groups 1-2)
        f32x4 z0lpre;  // ... and its pre-activations (layer-0 activations other than relu need them in the backward pass)
        {
            const int q = 4 * g;
            f32x4 v = *(const f32x4*)&binT[q];
#pragma unroll
            for (int c = 0; c < C; ++c) v += *(const f32x4*)&WinT[c * 16 + q] * x[c];
            const f32x4 fl = *(const f32x4*)&floorT[q];
            z0lpre = v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (ACT0 == INR_ACT_RELU) {
                    z0last[r] = fmaxf(v[r], fl[r]);
                } else {   // hidden leftovers: the activation; ext inputs (floor -inf): identity; unused slots stay 0
                    const bool hid = g == 0 && r < HR;
                    z0last[r] = hid ? act0_f<ACT0>(v[r], a.act_omega) : (fl[r] < 0.f ? v[r] : 0.f);
                }
            }
        }
        // ---- layer 1 (needs only the W1 image; the W2 re-fetch of the previous chunk may still be in flight) -------------
        f32x4 z1[KG];
        float la[HRA], z1l[HRA];
        {
            f32x4 z0[KG], acc[TM];
            z0_main(z0);
            z0[TM] = z0last;
            gemm_fwd(W0, z0, acc, la);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) z1[t][r] = relu0(acc[t][r]);
        }
#pragma unroll
        for (int u = 0; u < HRA; ++u) z1l[u] = 0.f;
#pragma unroll
        for (int u = 0; u < HR; ++u) z1l[u] = relu0(sum_over_groups(la[u]));
        // last k-group of z1ext: lane group 0 = leftover units, lane groups 1-2 = the same ext inputs as in z0ext
#pragma unroll
        for (int r = 0; r < 4; ++r) z1[TM][r] = g == 0 ? (r < HR ? z1l[r < HRA ? r : 0] : 0.f) : z0last[r];
        STAMP(1);
        __syncthreads();  // (A) W2 image complete in LDS (re-fetch waited for by every wave's vmcnt(0) at this barrier)
        STAMP(2);

        // ---- layer 2 ---------------------------------------------------------------------------------------------------
        f32x4 acc2[TM];
        float z2l[HRA];
        gemm_fwd(W1, z1, acc2, la);
        f32x4 wo[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) wo[t] = *(const f32x4*)&woT[16 * t + 4 * g];
        float ypart = 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc2[t][r] = relu0(acc2[t][r]);  // z2
                ypart = fmaf(wo[t][r], acc2[t][r], ypart);
            }
        ypart = sum_over_groups(ypart);
#pragma unroll
        for (int u = 0; u < HRA; ++u) z2l[u] = 0.f;
#pragma unroll
        for (int u = 0; u < HR; ++u) {
            z2l[u] = relu0(sum_over_groups(la[u]));
            ypart = fmaf(wol[u], z2l[u], ypart);
        }
        float y = ypart + b_o;
#pragma unroll
        for (int c = 0; c < C; ++c) y = fmaf(s_o[c], x[c], y);
        if (a.logits != nullptr && valid && g == 0) a.logits[(size_t)img * N + p] = y;

        STAMP(3);
        if (TRAIN) {
            const float pr = 1.f / (1.f + expf(-y));
            const float cw = tg < 0.5f ? cfg_ : cbg_;
            float l, dy;
            if (a.loss_kind == INR_LOSS_SE) {
                const float d = tg - pr;
                l = d * d * cw;
                dy = 2.f * (pr - tg) * pr * (1.f - pr) * cw;
            } else if (a.loss_kind == INR_LOSS_EXTERNAL) {
                l = 0.f;
                dy = tg;
            } else {
                const float lp = bce_log(pr), lq = bce_log(1.f - pr);   // clamped at -100, NaN kept (torch.nn.BCELoss)
                l = -(tg * lp + (1.f - tg) * lq) * cw;
                const float pq = pr * (1.f - pr);
                dy = (pr - tg) / fmaxf(pq, 1e-12f) * pq * cw;
            }
            if (!valid) {
                l = 0.f;
                dy = 0.f;
            }
            float dz2l[HRA];
#pragma unroll
            for (int u = 0; u < HRA; ++u) dz2l[u] = 0.f;
#pragma unroll
            for (int u = 0; u < HR; ++u) dz2l[u] = z2l[u] > 0.f ? dy * wol[u] : 0.f;
            if (g == 0) {
                loss_acc += l;
                dbo += dy;
#pragma unroll
                for (int c = 0; c < C; ++c) dso[c] = fmaf(dy, x[c], dso[c]);
#pragma unroll
                for (int u = 0; u < HR; ++u) dwol[u] = fmaf(dy, z2l[u], dwol[u]);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z2 = acc2[t][r];
                    const float dw = sum_over_points(dy * z2);   // over this wave's 16 points; one owner lane per row
                    // ds_add_f32 without return: fire and forget (a read-modify-write here costs an LDS round trip per row);
                    // the array is private to the wave and each row has one owner lane, so the sum order stays fixed
                    if (l15 == 0) __hip_atomic_fetch_add(&accW[wave * PT + 16 * t + 4 * g + r], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    acc2[t][r] = z2 > 0.f ? dy * wo[t][r] : 0.f;  // dz2
                }
            // ---- backward through layer 2 (W2 image): dz1 in the B-operand layout, masked by z1 -----------------------------
            f32x4 dz1[TM];
            float dz1l[HRA];
            f32x4 dzx2;  // DX: skip-path input gradient of layer 2 (rows = slots of k-group TM, columns = points)
            gemm_bwd(std::false_type{}, W1, smem + G::OFF_WCT1, acc2, dz2l, dz1, dz1l, dzx2);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dz1[t][r] = z1[t][r] > 0.f ? dz1[t][r] : 0.f;
#pragma unroll
            for (int u = 0; u < HR; ++u) {
                const float d = sum_over_groups(dz1l[u]);
                dz1l[u] = z1l[u] > 0.f ? d : 0.f;
            }
            STAMP(4);
            __syncthreads();  // (B) every wave is done with the W2 image: its region becomes the stage
            STAMP(5);
            stage(acc2, dz2l, z1);  // staged early so that dz2 and z1 leave the register file before the next product
            STAMP(6);
            // ---- backward through layer 1 (W1 image), transposed output; layer-0 gradient wave-locally ----------------------
            {
                f32x4 dz0[TM];
                float dz0l[HRA];
                f32x4 dzx1;  // DX: skip-path input gradient of layer 1 (rows = points, columns = slots of k-group TM)
                gemm_bwd(std::true_type{}, W0, smem + G::OFF_WCT0, dz1, dz1l, dz0, dz0l, dzx1);
                float wie[TM];
#pragma unroll
                for (int t = 0; t < TM; ++t) wie[t] = WinE[g * PT + 16 * t + l15];
                OPERAND_FENCE();
                f32x4 z0p[TM];
#pragma unroll
                for (int t = 0; t < TM; ++t) z0p[t] = MFMA16(xe, wie[t], (f32x4{0.f, 0.f, 0.f, 0.f}));
                MFMA_STEP_FENCE();
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (ACT0 == INR_ACT_RELU) dz0[t][r] = z0p[t][r] > 0.f ? dz0[t][r] : 0.f;
                        else dz0[t][r] *= dact0_f<ACT0>(z0p[t][r], a.act_omega);
                    }
                // ext columns of this wave's own points: (1, x_c) by slot of k-group TM - built in registers (no stage yet)
                float bfe[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int src_lane = 4 * g + r;  // the lane (in lane group 0) that owns point 4g + r
                    float v = 0.f;
                    if (l15 == G::ext_pos(0) - HM) v = 1.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float xc = __shfl(x[c], src_lane);
                        if (l15 == G::ext_pos(1 + c) - HM) v = xc;
                    }
                    bfe[r] = v;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) dL0[t] = MFMA16(dz0[t][r], bfe[r], dL0[t]);
                    MFMA_STEP_FENCE();
                }
                float hx[C];  // DX: per-point terms on lane l15: s_o dy + layer-2 skip path + W_in[HM+u] dz0l[u]
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    hx[c] = s_o[c] * dy;
                    if (DX) {  // layer-2 skip term: row ext_pos(1+c) - HM = 4*gc + rc of the (rows = slots) tile dzx2
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int slot = G::ext_pos(1 + c) - HM;
                        hx[c] += sum_over_groups(g == (slot >> 2) ? dzx2[slot & 3] : 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < HR; ++u) {
                    const float d = sum_over_groups(dz0l[u]);
                    const float z0u = DX ? __shfl(z0last[u], l15) : z0last[u];
                    float dm;
                    if constexpr (ACT0 == INR_ACT_RELU) dm = z0u > 0.f ? d : 0.f;
                    else dm = d * dact0_f<ACT0>(z0lpre[u], a.act_omega);
                    if (DX) {
#pragma unroll
                        for (int c = 0; c < C; ++c) hx[c] = fmaf(WinT[c * 16 + u], dm, hx[c]);
                    }
                    if (g == 0) {
                        dL0l[u][0] += dm;
#pragma unroll
                        for (int c = 0; c < C; ++c) dL0l[u][1 + c] = fmaf(dm, x[c], dL0l[u][1 + c]);
                    }
                }
                if (DX) {
                    float part[4][C];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < C; ++c) part[r][c] = (l15 == G::ext_pos(1 + c) - HM) ? dzx1[r] : 0.f;
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float w = WinE[c * PT + 16 * t + l15];
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[r][c] = fmaf(w, dz0[t][r], part[r][c]);
                        }
                    // lane (g, l15 = r) keeps the value of point 4g + r: C coalesced stores per wave instead of 4C masked ones
                    float dxv[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) dxv[c] = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float v = sum_over_points(part[r][c]);   // in every lane of the lane group
                            dxv[c] = l15 == r ? v : dxv[c];
                        }
                    const int pp = chunk * SP + wave * 16 + 4 * g + l15;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float v = dxv[c] + __shfl(hx[c], 4 * g + l15);   // hx lives on lane (0, point)
                        if (l15 < 4 && pp < (int)N) a.dcoords[((size_t)img * C + c) * N + pp] = v;
                    }
                }
            }
            STAMP(7);
            __syncthreads();  // (C)
            STAMP(8);
            dw_phase(dWb, dWlb);
            STAMP(9);
            __syncthreads();  // (D)
            STAMP(10);
            {
                f32x4 z0[KG];
                z0_main(z0);
                z0[TM] = z0last;
                stage(dz1, dz1l, z0);
            }
            STAMP(11);
            __syncthreads();  // (E)
            dw_phase(dWa, dWla);
            STAMP(12);
            __syncthreads();  // (F)
            STAMP(13);
            if (G::ALIAS && chunk + a.wgs < n_chunks) {
                // re-fetch the W2 image straight into LDS (LDS-DMA: wave-uniform LDS base + lane*16, no registers);
                // completion is awaited at barrier (A) of the next chunk, after that chunk's layer-1 product.
                constexpr int PIECES = G::WREG / 256;  // 1 KB pieces
                for (int pc = wave; pc < PIECES; pc += 4) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(gimg + G::OFF_W1 + pc * 256 + lane * 4),
                        (__attribute__((address_space(3))) void*)(W1 + pc * 256), 16, 0, 0);
                }
            }
        }
    }

#if INR_STAMPS
    if (TRAIN && blockIdx.x == 0 && tid == 0) {
        unsigned long long st_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_end)::"memory");
        for (int k = 0; k < 14; ++k) g_stamps2[k] = st_sum[k];
        g_stamps2[14] = st_end - st_begin;
    }
#endif
    if (TRAIN) {
        __syncthreads();
        float* __restrict__ slab = a.slabs + ((size_t)img * a.wgs + wg) * a.PS;
        // dW tiles of both layers: one 16-byte non-temporal store per lane and tile, in accumulator order (icnn_step.h, gradient slab)
        auto store_dw = [&](const f32x4 (&dW)[RPW][KG], int base) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
#pragma unroll
                for (int b = 0; b < KG; ++b) {
                    bool used = true;   // last column tile: leftover units and ext inputs only
                    if (b == KG - 1) {
                        used = l15 < HR;
#pragma unroll
                        for (int e = 0; e < NEXT; ++e) used = used || (HM + l15 == G::ext_pos(e));
                    }
                    if (used)
                        __builtin_nontemporal_store(dW[j][b], (f32x4*)(slab + base + ((((wave * RPW + j) * KG + b) * 64 + lane) << 2)));
                }
            }
        };
        if (row_ok) {
            store_dw(dWa, 0);
            store_dw(dWb, G::SL_TILE);
        }
        constexpr int SC_DWO = 0;
        constexpr int SC_DWLA = SC_DWO + PT;
        constexpr int SC_DWLB = SC_DWLA + HRA * PT;
        constexpr int SC_L0L = SC_DWLB + HRA * PT;
        constexpr int SC_SC = SC_L0L + HRA * 4;
        constexpr int SC_L0 = SC_SC + 8;
        constexpr int WSTR = SC_L0 + HM * 4;
        static_assert(4 * WSTR <= SP * G::SA + SP * G::SB, "reduction scratch must fit the stage");
        float* const scr = stA + wave * WSTR;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (l15 == 0) scr[SC_DWO + 16 * t + 4 * g + r] = accW[wave * PT + 16 * t + 4 * g + r];
#pragma unroll
        for (int u = 0; u < HR; ++u) {
            const float v = sum_over_points(dwol[u]);
            if (lane == 0) scr[SC_DWO + HM + u] = v;
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                const float wa = sum_over_groups(dWla[u][b]);
                const float wbv = sum_over_groups(dWlb[u][b]);
                if (g == 0) {
                    scr[SC_DWLA + u * PT + 16 * b + l15] = wa;
                    scr[SC_DWLB + u * PT + 16 * b + l15] = wbv;
                }
            }
#pragma unroll
            for (int e = 0; e < NEXT; ++e) {
                const float w = sum_over_points(dL0l[u][e]);
                if (lane == 0) scr[SC_L0L + u * 4 + e] = w;
            }
        }
        {
            int e = -1;
#pragma unroll
            for (int k = 0; k < NEXT; ++k)
                if (HM + l15 == G::ext_pos(k)) e = k;
            if (e >= 0) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[SC_L0 + (16 * t + 4 * g + r) * 4 + e] = dL0[t][r];
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
                const float v = sum_over_points(sc[k]);
                if (lane == 0) scr[SC_SC + k] = v;
            }
        }
        __syncthreads();
        auto wsum = [&](int i) { return ((stA[i] + stA[WSTR + i]) + stA[2 * WSTR + i]) + stA[3 * WSTR + i]; };
        for (int i = tid; i < H; i += WG_THREADS) slab[G::SL_WO + i] = wsum(SC_DWO + i);
        for (int i = tid; i < 2 * HR * PT; i += WG_THREADS) {
            const int layer = i / (HRA * PT), q = i - layer * (HRA * PT);  // (loop is empty when HR == 0)
            const int u = q / PT, pos = q - u * PT;
            const float v = wsum((layer ? SC_DWLB : SC_DWLA) + q);
            const int lb = layer ? G::SL_L2 : G::SL_L1;   // leftover block: W rows | b | S rows
            if (pos < H) slab[lb + u * H + pos] = v;
            else if (pos == G::ext_pos(0)) slab[lb + HR * H + u] = v;
            else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (pos == G::ext_pos(1 + c)) slab[lb + HR * H + HR + u * C + c] = v;
            }
        }
        for (int i = tid; i < HM * NEXT; i += WG_THREADS) {
            const int row = i / NEXT, e = i - row * NEXT;
            const float v = wsum(SC_L0 + row * 4 + e);
            if (e == 0) slab[G::SL_IN + G::P_BIN + row] = v;
            else slab[G::SL_IN + G::P_WIN + row * C + (e - 1)] = v;
        }
        if (tid < HR * NEXT) {
            const int u = tid / NEXT, e = tid - u * NEXT;
            const float v = wsum(SC_L0L + u * 4 + e);
            if (e == 0) slab[G::SL_IN + G::P_BIN + HM + u] = v;
            else slab[G::SL_IN + G::P_WIN + (HM + u) * C + (e - 1)] = v;
        }
        if (tid < 2 + C) {
            const float v = wsum(SC_SC + tid);
            if (tid == 0) slab[G::SL_LOSS] = v;
            else if (tid == 1) slab[G::SL_WO + (G::P_BO - G::P_WO)] = v;
            else slab[G::SL_WO + (G::P_SO - G::P_WO) + tid - 2] = v;
        }
    }
}

}  // namespace
