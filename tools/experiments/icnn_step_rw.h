// icnn_step_rw.h - role-split variant of the fused L = 1 step kernel (gfx950 / CDNA4): two waves per SIMD with different jobs.
//
// icnn_step_kernel runs one wave per SIMD; every VALU-only stretch (output layer, data term, masks, reductions, phase starts and
// drains) leaves that SIMD's matrix pipe idle.  Here a workgroup has 8 waves - waves w and w + 4 share a SIMD (checked with
// tools/micro/wave_simd.hip) - and the work of a chunk is split by ROLE, not by points:
//   front waves 0..3: inputs, layer 0, forward product, data term, backward product, layer-0 gradient of 16 points each; they
//                     stage dz1 (and the point's inputs) for the back waves;
//   back  waves 4..7: the weight-gradient product dW1ext += dZ1^T Z0ext of the PREVIOUS chunk (2 row tiles x 9 column tiles each).
// One s_barrier per chunk.  The back waves own the dW accumulators, so both roles fit 256 registers; only dz1 is staged (two
// buffers), because the back waves rebuild z0ext themselves: the layer-0 product with swapped operands gives a D tile with rows =
// points 4g + r and columns = units, which is, register r by register r, the B operand of the dW k-step over the points
// {r, 4 + r, 8 + r, 12 + r} of a 16-point group (8 more MFMAs per group).  Same arithmetic per element as icnn_step_kernel
// except for the order in which points are summed into d w_o and dW1ext (deterministic, fixed).
//
// STATUS: correct (bit-reproducible, passes the parity tests), NOT the default.  Measured at 1024x1024 (48 chunks per workgroup):
// 16.4 us per chunk against 15.0 us for icnn_step_kernel; front waves alone 12.8 us (592 MFMAs: 38 spilled registers at the
// 256-register budget and the per-chunk DPP reduction of d w_o cost ~4 k cycles), back waves alone 5.6 us (320 MFMAs; the z0 rebuild
// of each 16-point group is one serial LDS -> MFMA -> VALU -> LDS chain), together only 2 us less than their sum: the two streams
// alternate on the matrix pipe instead of one filling the other's gaps; s_setprio on the front waves changes nothing and
// throttling the back waves with s_sleep (to spread their products over the whole chunk) makes it slower.  With 4
// chunks per workgroup (one 256x256 image) the extra pipeline stage costs what the overlap gains (76.0 vs 70.5 us per launch).
// Selected with INRFIT_RW=1 in the environment.
#pragma once
#include "icnn_step.h"

namespace {

constexpr int RW_THREADS = 512;


template <int H, int C>
struct CfgRW {
    using G = Cfg<H, C>;
    static constexpr int SE = 20;                                   // row stride of the per-wave ext-column tile
    static constexpr int STA_FLOATS = SP * G::SA + 16;              // one dz1 stage: [64 points][SA]
    static constexpr int OFF_STA = G::IMG_FLOATS;                   // two of them
    static constexpr int OFF_STX = OFF_STA + 2 * STA_FLOATS;        // [2][64][4]  (x_0.., 1, 0) per staged point
    static constexpr int OFF_STE = OFF_STX + 2 * SP * 4;            // [4 front waves][16 points][SE] last k-group of z0ext
    static constexpr int OFF_ACC = OFF_STE + 4 * 16 * SE;           // [4 front waves][PT] d w_o
    static constexpr int LDS_FLOATS = OFF_ACC + 4 * G::PT;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
};

template <int H, int C>
__global__ __launch_bounds__(RW_THREADS, 1) void icnn_step_rw_kernel(const StepArgs a) {
    constexpr bool TRAIN = true, DX = false;
    using G = Cfg<H, C>;
    using R = CfgRW<H, C>;
    constexpr int TM = G::TM, KG = G::KG, HM = G::HM, HR = G::HR, S = G::S, PT = G::PT, RPW = G::RPW, NEXT = G::NEXT, SE = R::SE;
    constexpr int HRA = HR > 0 ? HR : 1;
    static_assert(4 * RPW == TM, "every back wave owns RPW row tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Wimg = smem + G::OFF_W;
    float* const WcT = smem + G::OFF_WCT;
    float* const WinE = smem + G::OFF_WINE;
    float* const WinT = smem + G::OFF_WIN;
    float* const binT = smem + G::OFF_BIN;
    float* const floorT = smem + G::OFF_FLOOR;
    float* const woT = smem + G::OFF_WO;
    float* const stA0 = smem + R::OFF_STA;
    float* const stX0 = smem + R::OFF_STX;
    float* const stE = smem + R::OFF_STE;
    float* const accW = smem + R::OFF_ACC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;   // 0: front, 1: back
    const int q = wave & 3;       // point group of a front wave / row-tile group of a back wave (= the SIMD they share)
    const int l15 = lane & 15, g = lane >> 4;
    const int img = blockIdx.x / a.wgs;
    const int wg = blockIdx.x - img * a.wgs;
    const long long N = a.N;

    // ---- parameter image into LDS, accumulators in LDS to zero ----------------------------------------------------------
    {
        const f32x4* __restrict__ src = (const f32x4*)(a.wimg + (size_t)img * G::IMG_FLOATS);
        constexpr int NV4 = G::IMG_FLOATS / 4;
        constexpr int NIT = (NV4 + RW_THREADS - 1) / RW_THREADS;
        f32x4 tmp[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * RW_THREADS;
            if (i < NV4) tmp[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * RW_THREADS;
            if (i < NV4) ((f32x4*)smem)[i] = tmp[k];
        }
        for (int i = tid; i < 4 * PT; i += RW_THREADS) accW[i] = 0.f;
    }
    const float cfg_ = a.coef[2 * img], cbg_ = a.coef[2 * img + 1];
    __syncthreads();

    const int n_chunks = (int)((N + SP - 1) / SP);
    const int n_my = wg < n_chunks ? (n_chunks - wg - 1) / a.wgs + 1 : 0;   // chunks of this workgroup
    float* __restrict__ slab = a.slabs + ((size_t)img * a.wgs + wg) * a.PS;

    // reduction scratch (aliases the dz1 stages after the last barrier of the chunk loop)
    constexpr int SC_DWO = 0;                    // [PT]            dw_o by position
    constexpr int SC_DWL = SC_DWO + PT;          // [HRA][PT]       leftover rows of dW1ext by column position
    constexpr int SC_L0L = SC_DWL + HRA * PT;    // [HRA][4]        leftover rows of the layer-0 gradient
    constexpr int SC_SC = SC_L0L + HRA * 4;      // [8]             loss, db_o, ds_o
    constexpr int SC_L0 = SC_SC + 8;             // [HM][4]         layer-0 gradient of the main units by ext input
    constexpr int WSTR = SC_L0 + HM * 4;
    static_assert(4 * WSTR <= 2 * R::STA_FLOATS, "reduction scratch must fit the stages");
    float* const scr = stA0 + q * WSTR;

    if (role == 0) {
        // =========================================== front waves ==========================================================
        const float b_o = smem[G::OFF_SC];
        float s_o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s_o[c] = smem[G::OFF_SC + 1 + c];
        float wol[HRA];  // w_o of the leftover units
#pragma unroll
        for (int u = 0; u < HRA; ++u) wol[u] = HR > 0 ? woT[HM + u] : 0.f;
        const float* const wf = Wimg + l15 * S + 4 * g;  // forward A operand: row 16t + l15, columns 16tk + 4g ..+3
        const float* const wb = Wimg + l15;              // backward operand: row o, column 16t + l15

        f32x4 dL0[TM];         // layer-0 gradient of this wave's own points (MFMA)
        float dwol[HRA];       // d w_o of the leftover units (lane group 0 only)
        float dL0l[HRA][NEXT]; // leftover rows of the layer-0 gradient (lane group 0 only)
        float loss_acc = 0.f, dbo = 0.f, dso[C];
#pragma unroll
        for (int t = 0; t < TM; ++t) dL0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < HRA; ++u) {
            dwol[u] = 0.f;
#pragma unroll
            for (int e = 0; e < NEXT; ++e) dL0l[u][e] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dso[c] = 0.f;

        // coordinates (and target) of this lane's point of a chunk; invalid points are clamped to the last valid one
        struct PointIn {
            float x[C];
            float tg;
        };
        const bool fast_div = N <= (1ll << 24);
        const float inv_width = a.grid.mode == INR_GRID_SEPARABLE ? 1.f / (float)a.grid.width : 0.f;
        auto load_point = [&](int chunk) -> PointIn {
            PointIn pin;
            int pc = chunk * SP + q * 16 + l15;   // points per image < 2^31 (checked on the host)
            pc = pc < (int)N ? pc : (int)N - 1;
            if (a.grid.mode == INR_GRID_SEPARABLE) {
                int row, col;
                if (fast_div) {   // pc < 2^24 is exact in fp32 and the quotient estimate is off by at most one
                    row = (int)((float)pc * inv_width);
                    col = pc - row * a.grid.width;
                    if (col < 0) {
                        row -= 1;
                        col += a.grid.width;
                    } else if (col >= a.grid.width) {
                        row += 1;
                        col -= a.grid.width;
                    }
                } else {
                    row = pc / a.grid.width;
                    col = pc - row * a.grid.width;
                }
                pin.x[0] = a.grid.xs[col];
                pin.x[1] = a.grid.ys[row];
                if (C > 2) pin.x[C - 1] = a.grid.ts ? a.grid.ts[img] : 0.f;
            } else {
                const float* cp = a.grid.coords + (size_t)img * a.grid.coords_image_stride;
    #pragma unroll
                for (int c = 0; c < C; ++c) pin.x[c] = cp[(size_t)c * N + pc];
            }
            pin.tg = TRAIN ? a.targets[(size_t)img * N + pc] : 0.f;
            return pin;
        };
        PointIn nxt = load_point(wg);

        for (int i = 0; i <= n_my; ++i) {
            if (i < n_my) {
                const int chunk = wg + i * a.wgs;
                float* const stA = stA0 + (i & 1) * R::STA_FLOATS;
                float* const stX = stX0 + (i & 1) * SP * 4;
                const int p = chunk * SP + q * 16 + l15;
                const bool valid = p < (int)N;
                const PointIn cur = nxt;
                float x[C];
#pragma unroll
                for (int c = 0; c < C; ++c) x[c] = cur.x[c];
                const float tg = cur.tg;

                // layer 0.  Main tiles on the matrix pipe: z0pre = [W_in | b_in] . (x, 1) is one k-step per tile (A rows from the
                // WinE table, B = this point's (x_0.., 1) by lane group); its D tile is already the B operand layout.  The last
                // k-group (leftover hidden units + ext inputs 1, x_c) is 16 positions of VALU work with per-position tables.
                f32x4 z0[KG];
                const float xe = g < C ? x[g < C ? g : 0] : (g == C ? 1.f : 0.f);
                auto z0_tile = [&](int tk) -> f32x4 {
                    f32x4 z = MFMA16(WinE[g * PT + 16 * tk + l15], xe, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = relu0(z[r]);
                    return z;
                };
                {
                    const int q = 4 * g;
                    f32x4 v = *(const f32x4*)&binT[q];
#pragma unroll
                    for (int c = 0; c < C; ++c) v += *(const f32x4*)&WinT[c * 16 + q] * x[c];
                    const f32x4 fl = *(const f32x4*)&floorT[q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) z0[TM][r] = fmaxf(v[r], fl[r]);
                }

                // ---- layer 1 (MFMA, software pipelined): acc[t] = W1ext . z0ext ----------------------------------------
                // Every LDS operand of k-group tk+1 is requested while k-group tk multiplies (double-buffered registers), so no
                // wait sits between a read and its use inside the stream of MFMAs.
                f32x4 acc[TM];
                float la[HRA];  // leftover units' pre-activation, partial over this lane group's positions
#pragma unroll
                for (int u = 0; u < HRA; ++u) la[u] = 0.f;
                // Operand reads of k-group tk+1 are spread over the products of k-group tk, one ds_read after every third MFMA and
                // pinned there (OPERAND_FENCE: MFMAs and LDS reads keep their program order, VALU work may still move).  A wave
                // issues in order and an LDS read holds the issue port for tens of cycles: reads issued back to back let the matrix
                // pipe run dry (tools/micro/mfma_rate.hip: 40.7 cycles per MFMA with 8 b128 reads in a burst, 33.6 spread out).
                f32x4 wq[2][TM];    // A operands of the main units: rows 16t + l15, 4 k-steps each
                f32x4 wlq[2][HRA];  // same columns of the leftover units' rows
                float winq[2];      // layer-0 A operand of the next z0 tile
                f32x4 wo[TM];       // w_o of this lane's positions (requested during the last k-group)
#pragma unroll
                for (int t = 0; t < TM; ++t) wq[0][t] = *(const f32x4*)(wf + t * 16 * S);
#pragma unroll
                for (int u = 0; u < HR; ++u) wlq[0][u] = *(const f32x4*)(Wimg + (HM + u) * S + 4 * g);
                z0[0] = z0_tile(0);
                winq[1] = WinE[g * PT + 16 * (TM > 1 ? 1 : 0) + l15];
#pragma unroll
                for (int tk = 0; tk < KG; ++tk) {
                    const int cur = tk & 1, nx = cur ^ 1;
                    constexpr int NRD = TM + HR + 1;  // reads per k-group
                    auto next_read = [&](int i) {     // i-th operand read for k-group tk+1 (last k-group: w_o for the output layer)
                        if (tk + 1 < KG) {
                            if (i < TM) wq[nx][i] = *(const f32x4*)(wf + i * 16 * S + 16 * (tk + 1));
                            else if (i < TM + HR) wlq[nx][i - TM] = *(const f32x4*)(Wimg + (HM + i - TM) * S + 16 * (tk + 1) + 4 * g);
                            else if (i == TM + HR && tk + 2 < TM) winq[cur] = WinE[g * PT + 16 * (tk + 2) + l15];
                        } else if (i < TM) {
                            wo[i] = *(const f32x4*)&woT[16 * i + 4 * g];
                        }
                    };
                    if (tk == (KG > 2 ? 1 : 0)) {   // next chunk's inputs: address arithmetic in the shadow of these products, latency under the chunk
                        const int cn = chunk + a.wgs;
                        nxt = load_point(cn < n_chunks ? cn : chunk);
                    }
                    f32x4 zn = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (tk + 1 < TM) zn = MFMA16(winq[nx], xe, (f32x4{0.f, 0.f, 0.f, 0.f}));  // pre-activation of z0[tk+1]
                    const int nmf = G::nr_in(tk) * TM;                    // products of this k-group
                    const int every = nmf / NRD > 0 ? nmf / NRD : 1;      // one read after every `every`-th product
#pragma unroll
                    for (int r = 0; r < G::nr_in(tk); ++r) {
#pragma unroll
                        for (int t = 0; t < TM; ++t) {
                            acc[t] = MFMA16(wq[cur][t][r], z0[tk][r], (tk == 0 && r == 0) ? (f32x4{0.f, 0.f, 0.f, 0.f}) : acc[t]);
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
                        for (int r = 0; r < G::nr_in(tk); ++r) la[u] = fmaf(wlq[cur][u][r], z0[tk][r], la[u]);
                    if (tk + 1 < TM) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) z0[tk + 1][r] = relu0(zn[r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }

                // relu mask of layer 0 in the transposed layout of the backward product (rows = points): z0^T is the layer-0 product
                // with swapped operands, TM more MFMAs - issued here, where the matrix pipe would otherwise idle under the VALU work
                // of the output layer and the data term.
                f32x4 z0p[TM];
                if (TRAIN) {
                    float wie[TM];
#pragma unroll
                    for (int t = 0; t < TM; ++t) wie[t] = WinE[g * PT + 16 * t + l15];
                    OPERAND_FENCE();
#pragma unroll
                    for (int t = 0; t < TM; ++t) z0p[t] = MFMA16(xe, wie[t], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    MFMA_STEP_FENCE();
                }
                // ---- output layer, sigmoid, data term ------------------------------------------------------------------
                float ypart = 0.f;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[t][r] = relu0(acc[t][r]);  // z1
                        ypart = fmaf(wo[t][r], acc[t][r], ypart);
                    }
                }
                ypart = sum_over_groups(ypart);
                float z1l[HRA];
#pragma unroll
                for (int u = 0; u < HR; ++u) {
                    z1l[u] = relu0(sum_over_groups(la[u]));
                    ypart = fmaf(wol[u], z1l[u], ypart);
                }
                float y = ypart + b_o;
#pragma unroll
                for (int c = 0; c < C; ++c) y = fmaf(s_o[c], x[c], y);
                if (a.logits != nullptr && valid && g == 0) a.logits[(size_t)img * N + p] = y;

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
                    float dzl[HRA];  // dz1 of the leftover units (same value in all 4 lane groups)
#pragma unroll
                    for (int u = 0; u < HRA; ++u) dzl[u] = 0.f;
#pragma unroll
                    for (int u = 0; u < HR; ++u) dzl[u] = z1l[u] > 0.f ? dy * wol[u] : 0.f;
                    if (g == 0) {
                        loss_acc += l;
                        dbo += dy;
#pragma unroll
                        for (int c = 0; c < C; ++c) dso[c] = fmaf(dy, x[c], dso[c]);
#pragma unroll
                        for (int u = 0; u < HR; ++u) dwol[u] = fmaf(dy, z1l[u], dwol[u]);
                    }
                    const int pl = q * 16 + l15;  // this lane's row in the stage
                    float* const sa = stA + pl * G::SA + 4 * g;
                    // dz1 of tile t (in place over acc), dw_o accumulation, staging of dz1 (A) and z0ext (B)
                    auto dz1_tile = [&](int t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float z1 = acc[t][r];
                            // d w_o: summed over this wave's 16 points right away and accumulated in LDS (one owner lane per row, fixed order)
                            const float dw = sum_over_points(dy * z1);
                            if (l15 == 0) __hip_atomic_fetch_add(&accW[q * PT + 16 * t + 4 * g + r], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            acc[t][r] = z1 > 0.f ? dy * wo[t][r] : 0.f;
                        }
                        *(f32x4*)(sa + 16 * t) = acc[t];
                    };
                    if (HR > 0 && g == 0) {
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < HR; ++u) v[u] = dzl[u];
                        *(f32x4*)(sa + HM) = v;
                    }
                    *(f32x4*)(stE + (q * 16 + l15) * SE + 4 * g) = z0[TM];   // last k-group of this wave's points: B operand of the layer-0 gradient
                    if (g == 0) {   // (x_0.., 1, 0) of this point: the back waves rebuild z0 from it
                        f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < C; ++c) xv[c] = x[c];
                        xv[C] = 1.f;
                        *(f32x4*)(stX + pl * 4) = xv;
                    }

                    // ---- backward through layer 1 (MFMA, pipelined): dZ0 = dZ1 . W1 ----------------------------------------
                    // Operands swapped w.r.t. the forward product (same registers): the D tile comes out transposed - rows =
                    // this wave's points 4g+r, columns = hidden unit 16t + l15 - which is the A operand of the layer-0
                    // gradient product dW_in = dZ0^T . (1, x), so that product needs no staging and no barrier.
                    f32x4 dz0[TM];
                    f32x4 dzx = f32x4{0.f, 0.f, 0.f, 0.f};  // DX: same product for the columns of k-group TM (skip-path inputs)
                    float bqx[2] = {0.f, 0.f};
                    float dz0l[HRA];
#pragma unroll
                    for (int u = 0; u < HRA; ++u) dz0l[u] = 0.f;
                    constexpr int KS = 4 * TM + HR;  // k-steps over the hidden outputs
                    float bq[2][TM];      // B operands (weights): row o of this k-step, columns 16t + l15
                    f32x4 wcq[2][HRA];    // leftover input columns W1[o][HM+u] at this lane's positions of a tile
                    auto b_row = [&](int ks) -> const float* {  // LDS row of the weight operand for k-step ks
                        const int tk = ks >> 2, r = ks & 3;
                        if (tk < TM) return wb + (16 * tk + 4 * g + r) * S;
                        return wb + (g == 0 ? (HM + r) * S : 0);  // leftover outputs live in lane group 0 (others: A = 0)
                    };
                    dz1_tile(0);
                    {
                        const float* br = b_row(0);
#pragma unroll
                        for (int t = 0; t < TM; ++t) bq[0][t] = br[16 * t];
                        if (DX) bqx[0] = br[16 * TM];
#pragma unroll
                        for (int u = 0; u < HR; ++u) wcq[0][u] = *(const f32x4*)(WcT + u * PT + 4 * g);
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
                            for (int u = 0; u < HR; ++u) wcq[(tk + 1) & 1][u] = *(const f32x4*)(WcT + u * PT + 16 * (tk + 1) + 4 * g);
                        }
                        OPERAND_FENCE();
                        const float bop = tk < TM ? acc[tk < TM ? tk : 0][r] : (g == 0 ? dzl[r < HRA ? r : 0] : 0.f);
#pragma unroll
                        for (int t = 0; t < TM; ++t) {  // D = dZ0 with POINTS on the rows; next k-step's operand reads one per product
                            dz0[t] = MFMA16(bop, bq[ks & 1][t], ks == 0 ? (f32x4{0.f, 0.f, 0.f, 0.f}) : dz0[t]);
                            if (ks + 1 < KS) bq[(ks + 1) & 1][t] = b_row(ks + 1)[16 * t];
                            OPERAND_FENCE();
                        }
                        if (DX) dzx = MFMA16(bop, bqx[ks & 1], dzx);
                        MFMA_STEP_FENCE();
                        if (r == 1 && tk + 1 < TM) dz1_tile(tk + 1);  // next tile's dz1 + staging, in the shadow of the MFMAs
                        // leftover hidden inputs: dz0l[u] += W1[:, HM+u] . dz1 - this k-step's share (HR FMAs per MFMA block, not 4 HR
                        // in one gap every fourth block)
#pragma unroll
                        for (int u = 0; u < HR; ++u) {
                            if (tk < TM) dz0l[u] = fmaf(wcq[tk & 1][u][r], acc[tk < TM ? tk : 0][r], dz0l[u]);
                            else if (g == 0) dz0l[u] = fmaf(wcq[tk & 1][u][r], dzl[r < HRA ? r : 0], dz0l[u]);
                        }
                        MFMA_STEP_FENCE();
                    }
                    // relu mask of layer 0 (z0p, computed before the output layer); then dL0[t] += dZ0[:, tile t]^T . ext columns of
                    // this wave's own stage-B rows.
                    {
                        float bfe[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) bfe[r] = stE[(q * 16 + 4 * g + r) * SE + l15];
#pragma unroll
                        for (int t = 0; t < TM; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dz0[t][r] = z0p[t][r] > 0.f ? dz0[t][r] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int t = 0; t < TM; ++t) dL0[t] = MFMA16(dz0[t][r], bfe[r], dL0[t]);
                            MFMA_STEP_FENCE();
                        }
                    }
                    // leftover rows of the layer-0 gradient (lane group 0, VALU)
                    float hx[C];  // DX: the contributions that live per point on lane l15: s_o dy + W_in[HM+u] dz0l[u]
#pragma unroll
                    for (int c = 0; c < C; ++c) hx[c] = s_o[c] * dy;
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const float d = sum_over_groups(dz0l[u]);
                        // position HM + u lives in lane group 0, k-step u (DX needs it in every lane group)
                        const float z0u = DX ? __shfl(z0[TM][u], l15) : z0[TM][u];
                        const float dm = z0u > 0.f ? d : 0.f;
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
                        // dL/dx_c of point 4g + r: sum over hidden units (lanes l15, tiles t) of W_in[.,c] dz0, plus the skip
                        // path (column ext_pos(1+c) of the transposed product dzx), plus the per-point terms of that point.
                        float part[4][C];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < C; ++c) part[r][c] = (l15 == G::ext_pos(1 + c) - HM) ? dzx[r] : 0.f;
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
                }   // TRAIN

            }
            __syncthreads();   // chunk i is staged; the back waves are done with chunk i - 1
        }

        // ---- per-lane sums of the front waves -> scratch row q ------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (l15 == 0) scr[SC_DWO + 16 * t + 4 * g + r] = accW[q * PT + 16 * t + 4 * g + r];
#pragma unroll
        for (int u = 0; u < HR; ++u) {
            const float v = sum_over_points(dwol[u]);  // lane group 0 only
            if (lane == 0) scr[SC_DWO + HM + u] = v;
#pragma unroll
            for (int e = 0; e < NEXT; ++e) {
                const float w = sum_over_points(dL0l[u][e]);  // lane group 0 only
                if (lane == 0) scr[SC_L0L + u * 4 + e] = w;
            }
        }
        {   // dL0 tiles: row 16t + 4g + r = hidden unit, column l15 = slot of k-group TM; keep the ext-input columns
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
                const float v = sum_over_points(sc[k]);  // lane group 0 only
                if (lane == 0) scr[SC_SC + k] = v;
            }
        }
    } else {
        // =========================================== back waves ===========================================================
        const int arow = 16 * q * RPW + l15;   // first dW row tile of this wave (+ lane column)
        constexpr bool row_ok = true;
        f32x4 dW[RPW][KG];     // dW1ext tiles: rows 16 (q RPW + j).., columns 16 b..
        float dWl[HRA][KG];    // leftover rows of dW1ext: column 16b + l15, partial over this lane group's points
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int b = 0; b < KG; ++b) dW[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < HRA; ++u)
#pragma unroll
            for (int b = 0; b < KG; ++b) dWl[u][b] = 0.f;
        // tables of the last column tile (leftover units + ext inputs) at this lane's column
        float winl[C];
#pragma unroll
        for (int c = 0; c < C; ++c) winl[c] = WinT[c * 16 + l15];
        const float binl = binT[l15], floorl = floorT[l15];

        for (int i = 0; i <= n_my; ++i) {
            if (i >= 1) {
                const float* const stA = stA0 + ((i - 1) & 1) * R::STA_FLOATS;
                const float* const stX = stX0 + ((i - 1) & 1) * SP * 4;
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int grp = (gi + q) & 3;   // 16 staged points; the waves start on different groups
                    // ---- z0ext^T of these points: rows = points 16 grp + 4g + r, columns = positions ----------------------
                    f32x4 zT[KG];
                    {
                        const float xe = stX[(16 * grp + l15) * 4 + g];   // A operand: row = point l15, k = input slot g
                        float wie[TM];
#pragma unroll
                        for (int b = 0; b < TM; ++b) wie[b] = WinE[g * PT + 16 * b + l15];
                        f32x4 xr[4];   // inputs of the points 4g + r (last column tile, VALU)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xr[r] = *(const f32x4*)(stX + (16 * grp + 4 * g + r) * 4);
                        OPERAND_FENCE();
#pragma unroll
                        for (int b = 0; b < TM; ++b) zT[b] = MFMA16(xe, wie[b], (f32x4{0.f, 0.f, 0.f, 0.f}));
                        MFMA_STEP_FENCE();
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = binl;
#pragma unroll
                            for (int c = 0; c < C; ++c) v = fmaf(winl[c], xr[r][c], v);
                            zT[TM][r] = fmaxf(v, floorl);
                        }
#pragma unroll
                        for (int b = 0; b < TM; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r) zT[b][r] = relu0(zT[b][r]);
                    }
                    // ---- 4 k-steps: points 16 grp + 4g + r ------------------------------------------------------------------
                    float af[4][RPW];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < RPW; ++j) af[r][j] = stA[(16 * grp + 4 * g + r) * G::SA + arow + 16 * j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (HR > 0 && r == q) {   // every back wave takes a quarter of the k-steps for the leftover rows
                            const f32x4 dl = *(const f32x4*)(stA + (16 * grp + 4 * g + r) * G::SA + HM);
#pragma unroll
                            for (int u = 0; u < HR; ++u)
#pragma unroll
                                for (int b = 0; b < KG; ++b) dWl[u][b] = fmaf(dl[u], zT[b][r], dWl[u][b]);
                        }
#pragma unroll
                        for (int j = 0; j < RPW; ++j)
#pragma unroll
                            for (int b = 0; b < KG; ++b) dW[j][b] = MFMA16(af[r][j], zT[b][r], dW[j][b]);
                        MFMA_STEP_FENCE();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
            // ---- dW1ext / layer-0 tiles of this wave ---------------------------------------------------------------------
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    const int o0 = 16 * (q * RPW + j) + 4 * g;  // first of this lane's 4 rows
#pragma unroll
                    for (int b = 0; b < KG; ++b) {
                        const int pos = 16 * b + l15;
                        int off = -1, rs = 0;
                        if (pos < H) {
                            off = G::P_W1 + pos;
                            rs = H;
                        } else if (pos == G::ext_pos(0)) {
                            off = G::P_B1;
                            rs = 1;
                        } else {
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                if (pos == G::ext_pos(1 + c)) {
                                    off = G::P_S1 + c;
                                    rs = C;
                                }
                        }
                        if (off >= 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) SLAB_ST(&slab[off + (o0 + r) * rs], dW[j][b][r]);
                        }
                    }
                }
            }
#pragma unroll
        for (int u = 0; u < HR; ++u)
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                const float w = sum_over_groups(dWl[u][b]);
                if (g == 0) scr[SC_DWL + u * PT + 16 * b + l15] = w;
            }
    }
    __syncthreads();
    float* const stA = stA0;
    auto wsum = [&](int i) { return ((stA[i] + stA[WSTR + i]) + stA[2 * WSTR + i]) + stA[3 * WSTR + i]; };
    for (int i = tid; i < H; i += RW_THREADS) slab[G::P_WO + i] = wsum(SC_DWO + i);
    for (int i = tid; i < HR * PT; i += RW_THREADS) {
        const int u = i / PT, pos = i - u * PT;
        const float v = wsum(SC_DWL + i);
        if (pos < H) slab[G::P_W1 + (HM + u) * H + pos] = v;
        else if (pos == G::ext_pos(0)) slab[G::P_B1 + HM + u] = v;
        else {
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (pos == G::ext_pos(1 + c)) slab[G::P_S1 + (HM + u) * C + c] = v;
        }
    }
    for (int i = tid; i < HM * NEXT; i += RW_THREADS) {
        const int row = i / NEXT, e = i - row * NEXT;
        const float v = wsum(SC_L0 + row * 4 + e);
        if (e == 0) slab[G::P_BIN + row] = v;
        else slab[G::P_WIN + row * C + (e - 1)] = v;
    }
    if (tid < HR * NEXT) {
        const int u = tid / NEXT, e = tid - u * NEXT;
        const float v = wsum(SC_L0L + u * 4 + e);
        if (e == 0) slab[G::P_BIN + HM + u] = v;
        else slab[G::P_WIN + (HM + u) * C + (e - 1)] = v;
    }
    if (tid < 2 + C) {
        const float v = wsum(SC_SC + tid);
        if (tid == 0) slab[G::P] = v;  // loss partial
        else if (tid == 1) slab[G::P_BO] = v;
        else slab[G::P_SO + tid - 2] = v;
    }
}

}  // namespace
