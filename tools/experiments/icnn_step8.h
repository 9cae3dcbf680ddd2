// icnn_step8.h - 8-wave variant of the L = 1 training step kernel (icnn_step.h): two waves per SIMD.
//
// Same arithmetic, same LDS image, same stage layout, same slab format as icnn_step_kernel<H,C,true>.  What changes is
// the division of labour: a workgroup is 8 waves = 4 point groups (16 points each) x 2 HALVES of the hidden units.
// Both waves of a point group compute the (cheap) layer 0 for all units, then each multiplies only its half of the
// output tiles of the forward product, of the backward product and of the layer-0 gradient; they exchange the two
// per-point partial logits through LDS (one barrier) and read each other's dz1 tiles back from the stage (which has to
// be written anyway).  Per-wave state halves, so a wave fits in 256 registers and two waves share a SIMD: while one wave
// sits in its VALU epilogue, waits for LDS or for a barrier, the other one keeps the matrix pipe busy.
#pragma once
#include "icnn_step.h"

namespace {

constexpr int WG8_THREADS = 512;

template <int H, int C>
__global__ __launch_bounds__(WG8_THREADS, 2) void icnn_step8_kernel(const StepArgs a) {
    using G = Cfg<H, C>;
    constexpr int TM = G::TM, KG = G::KG, HM = G::HM, HR = G::HR, S = G::S, PT = G::PT, NEXT = G::NEXT;
    constexpr int TH = TM / 2;  // output tiles per half
    static_assert(TM == 8, "one dW row tile per wave: 8 waves <-> 8 row tiles");
    static_assert(HR <= 2, "each half owns at most one leftover unit");
    constexpr int HRA = HR > 0 ? HR : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Wimg = smem + G::OFF_W;
    float* const WcT = smem + G::OFF_WCT;
    float* const WinE = smem + G::OFF_WINE;
    float* const WinT = smem + G::OFF_WIN;
    float* const binT = smem + G::OFF_BIN;
    float* const floorT = smem + G::OFF_FLOOR;
    float* const woT = smem + G::OFF_WO;
    float* const stA = smem + G::OFF_STA;
    float* const stB = smem + G::OFF_STB;
    float* const exch = smem + G::LDS_FLOATS;  // [4 point groups][2 halves][16 points][2]: partial logit, own leftover z1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = w8 & 3, hf = w8 >> 2;  // waves w and w+4 share a SIMD: the two halves of one point group
    const int l15 = lane & 15, g = lane >> 4;
    const int img = blockIdx.x / a.wgs;
    const int wg = blockIdx.x - img * a.wgs;
    const long long N = a.N;
    const bool own_left = hf < HR;  // this half owns leftover unit `hf`

    {   // parameter image -> LDS
        const f32x4* __restrict__ src = (const f32x4*)(a.wimg + (size_t)img * G::IMG_FLOATS);
        constexpr int NV4 = G::IMG_FLOATS / 4;
        constexpr int NIT = (NV4 + WG8_THREADS - 1) / WG8_THREADS;
        f32x4 tmp[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * WG8_THREADS;
            if (i < NV4) tmp[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * WG8_THREADS;
            if (i < NV4) ((f32x4*)smem)[i] = tmp[k];
        }
    }
    const float cfg_ = a.coef[2 * img], cbg_ = a.coef[2 * img + 1];
    __syncthreads();
    const float b_o = smem[G::OFF_SC];
    float s_o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s_o[c] = smem[G::OFF_SC + 1 + c];
    float wol[HRA];
#pragma unroll
    for (int u = 0; u < HRA; ++u) wol[u] = HR > 0 ? woT[HM + u] : 0.f;

    const int t0 = TH * hf;                                    // first output tile of this half
    const float* const wf = Wimg + (16 * t0 + l15) * S + 4 * g;  // forward A operand rows of this half
    const float* const wb = Wimg + 16 * t0 + l15;              // backward weight operand: row o, columns of this half
    const float* const wlrow = Wimg + (HM + (own_left ? hf : 0)) * S + 4 * g;   // own leftover unit's row
    const float* const wcol = WcT + (own_left ? hf : 0) * PT + 4 * g;           // own leftover unit's input column

    // persistent gradient accumulators
    f32x4 dW[KG];       // row tile w8 of dW1ext, all column tiles
    f32x4 dL0[TH];      // layer-0 gradient of this wave's points, own hidden tiles
    f32x4 dwo[TH];      // dw_o partial sums, own tiles
    float dwol = 0.f;   // ... own leftover unit (lane group 0)
    float dWl[HRA][KG]; // leftover rows of dW1ext (this wave's share of the k-steps)
    float dL0l[NEXT];   // own leftover row of the layer-0 gradient (lane group 0)
    float loss_acc = 0.f, dbo = 0.f, dso[C];
#pragma unroll
    for (int b = 0; b < KG; ++b) dW[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dL0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < HRA; ++u)
#pragma unroll
        for (int b = 0; b < KG; ++b) dWl[u][b] = 0.f;
#pragma unroll
    for (int e = 0; e < NEXT; ++e) dL0l[e] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) dso[c] = 0.f;

    const int n_chunks = (int)((N + SP - 1) / SP);
    struct PointIn {
        float x[C];
        float tg;
    };
    auto load_point = [&](int chunk) -> PointIn {
        PointIn q;
        int pc = chunk * SP + pg * 16 + l15;
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
        q.tg = a.targets[(size_t)img * N + pc];
        return q;
    };
    PointIn nxt = load_point(wg);
    const int pl = pg * 16 + l15;  // this lane's row in the stages
    float* const sa = stA + pl * G::SA + 4 * g;
    float* const sb = stB + pl * G::SB + 4 * g;
    float* const ex_own = exch + ((pg * 2 + hf) * 16 + l15) * 2;
    const float* const ex_oth = exch + ((pg * 2 + (hf ^ 1)) * 16 + l15) * 2;

    for (int chunk = wg; chunk < n_chunks; chunk += a.wgs) {
        const int p = chunk * SP + pg * 16 + l15;
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
        const float xe = g < C ? x[g < C ? g : 0] : (g == C ? 1.f : 0.f);

        // ---- layer 0 (all units; both halves need every k-group as the B operand) ---------------------------------------
        f32x4 z0[KG];
        {
            float wie[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) wie[t] = WinE[g * PT + 16 * t + l15];
            const int q = 4 * g;
            f32x4 v = *(const f32x4*)&binT[q];
#pragma unroll
            for (int c = 0; c < C; ++c) v += *(const f32x4*)&WinT[c * 16 + q] * x[c];
            const f32x4 fl = *(const f32x4*)&floorT[q];
            OPERAND_FENCE();
#pragma unroll
            for (int t = 0; t < TM; ++t) z0[t] = MFMA16(wie[t], xe, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) z0[TM][r] = fmaxf(v[r], fl[r]);
            MFMA_STEP_FENCE();
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) z0[t][r] = fmaxf(z0[t][r], 0.f);
        }

        // ---- layer 1, own half of the output tiles ---------------------------------------------------------------------
        f32x4 acc[TH];
        float la = 0.f;  // own leftover unit's pre-activation (partial over this lane group's positions)
        f32x4 wo[TH];
        {
            f32x4 wq[2][TH], wlq[2];
#pragma unroll
            for (int t = 0; t < TH; ++t) wq[0][t] = *(const f32x4*)(wf + t * 16 * S);
            wlq[0] = *(const f32x4*)(wlrow);
#pragma unroll
            for (int tk = 0; tk < KG; ++tk) {
                const int cb = tk & 1, nb = cb ^ 1;
                if (tk + 1 < KG) {
#pragma unroll
                    for (int t = 0; t < TH; ++t) wq[nb][t] = *(const f32x4*)(wf + t * 16 * S + 16 * (tk + 1));
                    wlq[nb] = *(const f32x4*)(wlrow + 16 * (tk + 1));
                } else {
#pragma unroll
                    for (int t = 0; t < TH; ++t) wo[t] = *(const f32x4*)&woT[16 * (t0 + t) + 4 * g];
                }
                OPERAND_FENCE();
#pragma unroll
                for (int r = 0; r < G::nr_in(tk); ++r) {
#pragma unroll
                    for (int t = 0; t < TH; ++t)
                        acc[t] = MFMA16(wq[cb][t][r], z0[tk][r], (tk == 0 && r == 0) ? (f32x4{0.f, 0.f, 0.f, 0.f}) : acc[t]);
                    MFMA_STEP_FENCE();
                }
#pragma unroll
                for (int r = 0; r < G::nr_in(tk); ++r) la = fmaf(wlq[cb][r], z0[tk][r], la);
            }
        }
        // ---- partial logit of this half; exchange with the other half ----------------------------------------------------
        float ypart = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[t][r] = fmaxf(acc[t][r], 0.f);  // z1
                ypart = fmaf(wo[t][r], acc[t][r], ypart);
            }
        ypart = sum_over_groups(ypart);
        const float z1l_own = own_left ? fmaxf(sum_over_groups(la), 0.f) : 0.f;
        if (own_left) ypart = fmaf(wol[hf < HRA ? hf : 0], z1l_own, ypart);
        if (g == 0) {
            ex_own[0] = ypart;
            ex_own[1] = z1l_own;
        }
        __syncthreads();  // (1)
        float z1l[HRA];
        float y;
        {
            const float yo = ex_oth[0], zo = ex_oth[1];
            y = (hf == 0 ? ypart + yo : yo + ypart) + b_o;  // same summation order in both halves
#pragma unroll
            for (int u = 0; u < HRA; ++u) z1l[u] = (u == hf) ? z1l_own : zo;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) y = fmaf(s_o[c], x[c], y);

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
            const float lp = fmaxf(logf(pr), -100.f), lq = fmaxf(logf(1.f - pr), -100.f);
            l = -(tg * lp + (1.f - tg) * lq) * cw;
            const float pq = pr * (1.f - pr);
            dy = (pr - tg) / fmaxf(pq, 1e-12f) * pq * cw;
        }
        if (!valid) {
            l = 0.f;
            dy = 0.f;
        }
        float dzl[HRA];
#pragma unroll
        for (int u = 0; u < HRA; ++u) dzl[u] = (u < HR && z1l[u] > 0.f) ? dy * wol[u] : 0.f;
        if (g == 0) {
            if (hf == 0) {
                loss_acc += l;
                dbo += dy;
#pragma unroll
                for (int c = 0; c < C; ++c) dso[c] = fmaf(dy, x[c], dso[c]);
            }
            if (own_left) dwol = fmaf(dy, z1l_own, dwol);
        }
        // dz1 of the own tiles (in place), dw_o, staging: own dz1 tiles, own half of the z0ext tiles
#pragma unroll
        for (int t = 0; t < TH; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z1 = acc[t][r];
                dwo[t][r] = fmaf(dy, z1, dwo[t][r]);
                acc[t][r] = z1 > 0.f ? dy * wo[t][r] : 0.f;
            }
            *(f32x4*)(sa + 16 * (t0 + t)) = acc[t];
            *(f32x4*)(sb + 16 * (t0 + t)) = z0[t0 + t];
        }
        if (hf == 0) {
            if (HR > 0 && g == 0) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < HR; ++u) v[u] = dzl[u];
                *(f32x4*)(sa + HM) = v;
            }
            if (g < 3) *(f32x4*)(sb + HM) = z0[TM];
        }
        const float z0l_own = z0[TM][hf < 4 ? hf : 0];  // position HM + hf lives in lane group 0, k-step hf
        __syncthreads();  // (2) stage complete (also what the dW phase needs)

        // ---- backward product, own half of the hidden inputs (transposed output) ------------------------------------------
        f32x4 dzp[TH];  // the other half's dz1 tiles of this lane's point (B-operand layout), read back from the stage
#pragma unroll
        for (int t = 0; t < TH; ++t) dzp[t] = *(const f32x4*)(sa + 16 * (TH * (hf ^ 1) + t));
        f32x4 dz0[TH];
        float dz0l = 0.f;
        {
            constexpr int KS = 4 * TM + HR;
            float bq[2][TH];
            f32x4 wcq[2];
            auto b_row = [&](int ks) -> const float* {
                const int tk = ks >> 2, r = ks & 3;
                if (tk < TM) return wb + (16 * tk + 4 * g + r) * S;
                return wb + (g == 0 ? (HM + r) * S : 0);
            };
            {
                const float* br = b_row(0);
#pragma unroll
                for (int t = 0; t < TH; ++t) bq[0][t] = br[16 * t];
                wcq[0] = *(const f32x4*)(wcol);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int tk = ks >> 2, r = ks & 3;
                if (ks + 1 < KS) {
                    const float* br = b_row(ks + 1);
#pragma unroll
                    for (int t = 0; t < TH; ++t) bq[(ks + 1) & 1][t] = br[16 * t];
                }
                if (r == 0 && (tk + 1) * 4 < KS) wcq[(tk + 1) & 1] = *(const f32x4*)(wcol + 16 * (tk + 1));
                OPERAND_FENCE();
                // dz1 of k-group tk: own registers, the partner's (read back above), or the leftover units
                float bop;
                if (tk < TM) {
                    const int th = tk < TM ? tk / TH : 0, tl = tk < TM ? tk % TH : 0;  // static after unrolling
                    const float own_v = acc[tl][r], oth_v = dzp[tl][r];
                    bop = (th == hf) ? own_v : oth_v;
                } else {
                    bop = g == 0 ? dzl[r < HRA ? r : 0] : 0.f;
                }
#pragma unroll
                for (int t = 0; t < TH; ++t)
                    dz0[t] = MFMA16(bop, bq[ks & 1][t], ks == 0 ? (f32x4{0.f, 0.f, 0.f, 0.f}) : dz0[t]);
                MFMA_STEP_FENCE();
                if (r == 3 || ks == KS - 1) {  // own leftover hidden input: dz0l += W1[:, HM+hf] . dz1 over this k-group
                    if (tk < TM) {
                        const int th = tk / TH, tl = tk % TH;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) dz0l = fmaf(wcq[tk & 1][rr], (th == hf) ? acc[tl][rr] : dzp[tl][rr], dz0l);
                    } else if (g == 0) {
#pragma unroll
                        for (int rr = 0; rr < HR; ++rr) dz0l = fmaf(wcq[tk & 1][rr], dzl[rr], dz0l);
                    }
                }
            }
        }
        // relu mask (transposed layer-0 product), layer-0 gradient of the own tiles
        {
            float bfe[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bfe[r] = stB[(pg * 16 + 4 * g + r) * G::SB + HM + l15];
            float wie[TH];
#pragma unroll
            for (int t = 0; t < TH; ++t) wie[t] = WinE[g * PT + 16 * (t0 + t) + l15];
            OPERAND_FENCE();
            f32x4 z0p[TH];
#pragma unroll
            for (int t = 0; t < TH; ++t) z0p[t] = MFMA16(xe, wie[t], (f32x4{0.f, 0.f, 0.f, 0.f}));
            MFMA_STEP_FENCE();
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dz0[t][r] = z0p[t][r] > 0.f ? dz0[t][r] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int t = 0; t < TH; ++t) dL0[t] = MFMA16(dz0[t][r], bfe[r], dL0[t]);
                MFMA_STEP_FENCE();
            }
        }
        if (own_left) {
            const float d = sum_over_groups(dz0l);
            if (g == 0) {
                const float dm = z0l_own > 0.f ? d : 0.f;
                dL0l[0] += dm;
#pragma unroll
                for (int c = 0; c < C; ++c) dL0l[1 + c] = fmaf(dm, x[c], dL0l[1 + c]);
            }
        }

        // ---- dW1ext: row tile w8 x all column tiles over the 64 staged points; leftover rows on the VALU ----------------
        {
            auto stage_pt = [&](int it) { const int s = (it + 2 * w8) & 15; return 16 * (s >> 2) + (s & 3) + 4 * g; };
            float af[2], bf[2][KG];
            {
                const int pt = stage_pt(0);
                af[0] = stA[pt * G::SA + 16 * w8 + l15];
#pragma unroll
                for (int b = 0; b < KG; ++b) bf[0][b] = stB[pt * G::SB + 16 * b + l15];
            }
#pragma unroll
            for (int it = 0; it < SP / 4; ++it) {
                const int ptc = stage_pt(it);
                if (it + 1 < SP / 4) {
                    const int pt = stage_pt(it + 1);
                    af[(it + 1) & 1] = stA[pt * G::SA + 16 * w8 + l15];
#pragma unroll
                    for (int b = 0; b < KG; ++b) bf[(it + 1) & 1][b] = stB[pt * G::SB + 16 * b + l15];
                }
                if (HR > 0 && it < 2) {  // each of the 8 waves takes 2 of the 16 k-steps for the leftover rows
                    const f32x4 dl = *(const f32x4*)(stA + ptc * G::SA + HM);
#pragma unroll
                    for (int u = 0; u < HR; ++u)
#pragma unroll
                        for (int b = 0; b < KG; ++b) dWl[u][b] = fmaf(dl[u], bf[it & 1][b], dWl[u][b]);
                }
                OPERAND_FENCE();
#pragma unroll
                for (int b = 0; b < KG; ++b) dW[b] = MFMA16(af[it & 1], bf[it & 1][b], dW[b]);
                MFMA_STEP_FENCE();
            }
        }
        __syncthreads();  // (3) everybody is done with the stage
    }

    // ---- epilogue: slab of this workgroup -------------------------------------------------------------------------------
    float* __restrict__ slab = a.slabs + ((size_t)img * a.wgs + wg) * a.PS;
    {
        const int o0 = 16 * w8 + 4 * g;
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
                for (int r = 0; r < 4; ++r) slab[off + (o0 + r) * rs] = dW[b][r];
            }
        }
    }
    constexpr int SC_DWO = 0;                    // [PT]        dw_o by position (each position: the 4 waves of its half)
    constexpr int SC_DWL = SC_DWO + PT;          // [HRA][PT]   leftover rows of dW1ext (all 8 waves)
    constexpr int SC_L0L = SC_DWL + HRA * PT;    // [HRA][4]    leftover rows of the layer-0 gradient
    constexpr int SC_SC = SC_L0L + HRA * 4;      // [8]         loss, db_o, ds_o (waves of half 0)
    constexpr int SC_L0 = SC_SC + 8;             // [HM][4]     layer-0 gradient of the main units by ext input
    constexpr int WSTR = SC_L0 + HM * 4;
    static_assert(8 * WSTR <= SP * G::SA + SP * G::SB + 32, "reduction scratch must fit the stages");
    float* const scr = stA + w8 * WSTR;
    for (int i = lane; i < WSTR; i += 64) scr[i] = 0.f;  // every wave only fills its own part
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = sum_over_points(dwo[t][r]);
            if (l15 == 0) scr[SC_DWO + 16 * (t0 + t) + 4 * g + r] = v;
        }
    if (own_left) {
        const float v = sum_over_points(dwol);
        if (lane == 0) scr[SC_DWO + HM + hf] = v;
#pragma unroll
        for (int e = 0; e < NEXT; ++e) {
            const float w = sum_over_points(dL0l[e]);
            if (lane == 0) scr[SC_L0L + hf * 4 + e] = w;
        }
    }
#pragma unroll
    for (int u = 0; u < HR; ++u)
#pragma unroll
        for (int b = 0; b < KG; ++b) {
            const float w = sum_over_groups(dWl[u][b]);
            if (g == 0) scr[SC_DWL + u * PT + 16 * b + l15] = w;
        }
    {
        int e = -1;
#pragma unroll
        for (int k = 0; k < NEXT; ++k)
            if (HM + l15 == G::ext_pos(k)) e = k;
        if (e >= 0) {
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[SC_L0 + (16 * (t0 + t) + 4 * g + r) * 4 + e] = dL0[t][r];
        }
    }
    if (hf == 0) {
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
    auto wsum = [&](int i) {  // fixed order over the 8 waves (waves that do not own an entry contributed 0)
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += stA[w * WSTR + i];
        return s;
    };
    for (int i = tid; i < H; i += WG8_THREADS) slab[G::P_WO + i] = wsum(SC_DWO + i);
    for (int i = tid; i < HR * PT; i += WG8_THREADS) {
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
    for (int i = tid; i < HM * NEXT; i += WG8_THREADS) {
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
        if (tid == 0) slab[G::P] = v;
        else if (tid == 1) slab[G::P_BO] = v;
        else slab[G::P_SO + tid - 2] = v;
    }
}

}  // namespace
