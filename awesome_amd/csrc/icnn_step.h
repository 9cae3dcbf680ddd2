// icnn_step.h - the fused forward + data term + backward kernel of the ICNN fit (gfx950 / CDNA4).
// Included by inrfit.hip (single translation unit).  Design notes: DESIGN.md §4.1.
//
// Reference arithmetic being computed (jp-schneider/awesome, awesome/model/convex_net.py:205-214):
//   z0 = relu(W_in x + b_in);  z1 = relu(W1 z0 + b1 + S1 x);  y = w_o.z1 + b_o + s_o.x
// plus sigmoid, SE/BCE data term (awesome/measures/se.py:21-23, weighted_loss.py:67-92) and the full backward pass.
//
// Geometry.  h = 16*TM + HR.  The 16*TM "main" hidden units go through v_mfma_f32_16x16x4_f32; the HR (<= 4) leftover
// units are handled on the VALU in the shadow of the MFMAs, so the matrix pipe does no padding work.
//   * points sit on the MFMA column (lane & 15), hidden units on the accumulator rows: the D tile of one product is the
//     B operand of the next one (k-step (tile, r) takes position 16*tile + 4*(lane>>4) + r from every lane group), so
//     activations stay in registers from layer to layer and into the backward product;
//   * the last k-group of the forward product carries the leftover hidden units (lane group 0) and the "ext" inputs
//     (1, x_0.., lane groups 1-2): bias and skip weights are columns of the same LDS weight image, and their gradients
//     fall out of the dW product;
//   * dW = dZ1^T Z0ext contracts over points: both operands are staged once per 64-point chunk through LDS (point-major
//     rows, float4 writes), every wave then owns TM/4 row tiles x all column tiles - the same code for all waves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include <type_traits>

#include "inrfit.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// relu as ONE instruction: v_max_i32 on the bit pattern (negative floats are negative integers; -0.0 and negative NaNs
// become +0).  fmaxf / fmed3 cost two (hipcc first quiets a possible signalling NaN with v_max x, x), and inline asm is not an
// option: the hazard recogniser does not insert the MFMA-write -> VALU-read wait states in front of an asm statement.
__device__ __forceinline__ float relu0(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// Layer-0 activation = the encode stage (include/inrfit.h INR_ACT_*).  sin / cos: one v_fract_f32 range reduction on the
// argument in revolutions, then v_sin_f32 / v_cos_f32 (which take revolutions).
constexpr float INV_2PI = 0.15915494309189535f;
__device__ __forceinline__ float hw_sin(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * INV_2PI)); }
__device__ __forceinline__ float hw_cos(float x) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * INV_2PI)); }
template <int ACT>
__device__ __forceinline__ float act0_f(float pre, float omega) {
    if constexpr (ACT == INR_ACT_COS) return hw_cos(pre);
    else if constexpr (ACT == INR_ACT_SIN) return hw_sin(omega * pre);
    else return relu0(pre);
}
template <int ACT>
__device__ __forceinline__ float dact0_f(float pre, float omega) {   // d act / d pre
    if constexpr (ACT == INR_ACT_COS) return -hw_sin(pre);
    else if constexpr (ACT == INR_ACT_SIN) return omega * hw_cos(omega * pre);
    else return pre > 0.f ? 1.f : 0.f;
}
// LLVM SchedGroupMask bits for __builtin_amdgcn_sched_group_barrier
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_DS_READ 0x100
#define SG_DS_WRITE 0x200

namespace {

constexpr int WG_THREADS = 256;
constexpr int SP = 64;  // points per chunk (4 waves x 16)

constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }
// smallest s >= v with s % 8 == 4: float4-aligned rows whose 4-row step lands 16 banks away (conflict-free b32 column
// reads by the two 16-lane groups of a half-wave, conflict-free b128 row writes by 8 consecutive rows)
constexpr int stride_4mod8(int v) {
    int s = round_up(v, 4);
    while (s % 8 != 4) s += 4;
    return s;
}

template <int H, int C>
struct Cfg {
    static constexpr int TM = H / 16;            // hidden tiles on the matrix pipe
    static constexpr int HM = 16 * TM;           // hidden units on the matrix pipe
    static constexpr int HR = H % 16;            // leftover hidden units (VALU)
    static_assert(HR <= 4, "leftover hidden units must fit lane group 0 of one k-group (n_hidden % 16 <= 4)");
    static_assert(TM >= 1 && C >= 1 && C <= 3, "unsupported shape");
    static constexpr int NEXT = C + 1;           // ext inputs: 1, x_0..x_{C-1}
    // position (padded index) of: hidden unit u -> u;  ext input e -> lane group 1 + e/2, k-step e%2 of k-group TM
    static constexpr int ext_pos(int e) { return HM + 4 * (1 + e / 2) + (e % 2); }
    static constexpr int NRL = HR > 2 ? HR : 2;  // k-steps of the last forward k-group
    static constexpr int KG = TM + 1;            // forward k-groups (= column tiles of the dW product)
    static constexpr int nr_in(int tk) { return tk < TM ? 4 : NRL; }
    static constexpr int nr_out(int tk) { return tk < TM ? 4 : HR; }
    static constexpr int PT = 16 * KG;           // padded table length
    static constexpr int S = stride_4mod8(HM + 10);   // weight image row stride
    static constexpr int SA = stride_4mod8(HM + 4);   // stage A row stride: dz1[0..HM) | leftover dz1 (HR) | pad
    static constexpr int SB = stride_4mod8(HM + 10);  // stage B row stride: z0ext positions
    static constexpr int RPW = (TM + 3) / 4;     // row tiles of the dW product per wave

    // ---- LDS carve (floats).  [0, IMG_FLOATS) is the parameter image, kept in HBM in exactly this layout. ----------
    // Small tables first (reachable with immediate offsets from one base), then the big weight image.
    static constexpr int OFF_SC = 0;                             // b_o, s_o[0..C-1]
    static constexpr int OFF_WINE = OFF_SC + 8;                  // [4][PT] layer-0 A operand: rows W_in[:,0..C-1], b_in, (0)
    static constexpr int OFF_WIN = OFF_WINE + 4 * PT;            // [C][16] W_in of k-group TM's positions (leftovers + ext)
    static constexpr int OFF_BIN = OFF_WIN + C * 16;             // [16]
    static constexpr int OFF_FLOOR = OFF_BIN + 16;               // [16] relu floor: 0 for hidden, -inf for ext inputs
    static constexpr int OFF_WO = OFF_FLOOR + 16;                // [PT]
    static constexpr int OFF_WCT = OFF_WO + PT;                  // [HR][PT]: W1[:, HM+u] transposed (leftover inputs)
    static constexpr int OFF_W = OFF_WCT + HR * PT;              // W1ext [H][S]: row o, columns = positions
    static constexpr int IMG_FLOATS = OFF_W + H * S + 16;        // +16 zero floats: tail reads past the last row
    static_assert(IMG_FLOATS % 4 == 0 && OFF_W % 4 == 0, "image must be float4-copyable / aligned");
    static constexpr int OFF_STA = IMG_FLOATS;
    static constexpr int OFF_STB = OFF_STA + SP * SA + 16;
    static constexpr int LDS_FLOATS = OFF_STB + SP * SB + 16;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");

    // ---- flat parameter offsets (L = 1), include/inrfit.h ---------------------------------------------------------
    static constexpr int P_WIN = 0;
    static constexpr int P_BIN = H * C;
    static constexpr int P_W1 = P_BIN + H;
    static constexpr int P_B1 = P_W1 + H * H;
    static constexpr int P_S1 = P_B1 + H;
    static constexpr int P_WO = P_S1 + H * C;
    static constexpr int P_BO = P_WO + H;
    static constexpr int P_SO = P_BO + 1;
    static constexpr int P = P_SO + C;

    // ---- gradient slab (one per workgroup and step; read by icnn_update_kernel through slab_param_of_col) ------------
    // [0, SL_TILE): the dW1ext accumulator tiles exactly as the matrix pipe leaves them - tile (row tile rt, column tile b), lane, register
    // r <-> row 16 rt + 4 (lane >> 4) + r, column position 16 b + (lane & 15) - so every lane stores ONE 16-byte vector per tile (72
    // global_store_dwordx4 per lane instead of 288 dword stores: the 17 MB of slabs per launch leave 1.1 us sooner).
    // Then every parameter the tiles do not carry, in parameter order (W_in, b_in | leftover rows of W1, b1, S1 | w_o, b_o, s_o), then the
    // loss partial.  The ORDER in which slabs are summed is untouched, so results are bit-identical to the parameter-ordered layout.
    static constexpr int SL_TILE = TM * KG * 256;
    static constexpr int SL_W1L = SL_TILE + P_W1;
    static constexpr int SL_B1L = SL_W1L + HR * H;
    static constexpr int SL_S1L = SL_B1L + HR;
    static constexpr int SL_WO = SL_S1L + HR * C;
    static constexpr int SL_LOSS = SL_WO + (P - P_WO);
    static constexpr int SL_COLS = SL_LOSS + 1;
    // parameter j (or P = the loss) -> slab column, for everything outside the tiles
    __host__ __device__ static constexpr int slab_col(int j) {
        return j < P_W1 ? SL_TILE + j
             : j < P_B1 ? SL_W1L + (j - P_W1 - HM * H)
             : j < P_S1 ? SL_B1L + (j - P_B1 - HM)
             : j < P_WO ? SL_S1L + (j - P_S1 - HM * C)
                        : SL_WO + (j - P_WO);
    }
};

// run-time description of the parameter image (same numbers as Cfg<H,C> / Cfg2<H,C>) for the untemplated kernels
struct ImgMap {
    int H, C, L, HM, S, PT, floats;
    int off_sc, off_wine, off_win, off_bin, off_floor, off_wo;
    int off_wct[2], off_w[2];   // per hidden layer
    int ext[4];
    int p_bin, p_wo, p_bo, p_so, P;
    int p_w[2], p_b[2], p_s[2]; // flat offsets of skip.k.ln.weight / ln.bias / skp.weight
    int sl_tile;                // gradient slab: floats of ONE layer's tile region (Cfg::SL_TILE); L of them lead the slab
    int sl_cols;                // gradient slab: columns in use (Cfg::SL_COLS, or P + 1)
    int KG, kg_magic;           // column tiles per row tile; 65536 / KG + 1 (tile index / KG without a division: exact below 2^13)
    int identity;               // 1: the "slab" is a plain gradient vector [P | loss] (the layer-by-layer path, wide.h): column = parameter
};

// Gradient slab column -> flat parameter index (P = the loss partial, -1 = padding); the inverse of the step kernels' stores:
//   [layer 0 tiles (sl_tile floats)] ... [layer L-1 tiles] [W_in, b_in] [per layer: leftover rows of W, of b, of S] [w_o, b_o, s_o] [loss]
// Straight-line selects on fields read with constant indices: it sits in front of the update kernel's first loads.
__device__ __forceinline__ int slab_param_of_col(const ImgMap& m, int col) {
    if (m.identity) return col < m.sl_cols ? col : -1;
    const int H = m.H, HM = m.HM, C = m.C, KG = m.KG, kgm = m.kg_magic, T = m.sl_tile, two = m.L > 1;
    const int e0 = m.ext[0], e1 = m.ext[1], e2 = m.ext[2], e3 = m.ext[3], pwo = m.p_wo, ncols = m.sl_cols;
    // tile regions
    const int tl = (two && col >= T) ? 1 : 0, ct = col - tl * T;
    const int pw = tl ? m.p_w[1] : m.p_w[0], pb = tl ? m.p_b[1] : m.p_b[0], ps = tl ? m.p_s[1] : m.p_s[0];
    const int r = ct & 3, lane = (ct >> 2) & 63, tb = ct >> 8;
    const int rt = (tb * kgm) >> 16, b = tb - rt * KG;
    const int o = 16 * rt + 4 * (lane >> 4) + r, pos = 16 * b + (lane & 15);
    int tile = -1;
    tile = (pos == e3 && C > 2) ? ps + o * C + 2 : tile;
    tile = (pos == e2 && C > 1) ? ps + o * C + 1 : tile;
    tile = pos == e1 ? ps + o * C : tile;
    tile = pos == e0 ? pb + o : tile;
    tile = pos < H ? pw + o * H + pos : tile;
    // rest region
    const int HR = H - HM, LS = HR * (H + 1 + C), NT = two ? 2 * T : T;
    const int c = col - NT, c1 = c - m.p_w[0];
    const int rl = (two && c1 >= LS) ? 1 : 0, cl = c1 - rl * LS;           // leftover block of layer rl
    const int qw = rl ? m.p_w[1] : m.p_w[0], qb = rl ? m.p_b[1] : m.p_b[0], qs = rl ? m.p_s[1] : m.p_s[0];
    const int c2 = cl - HR * H, c3 = c2 - HR;
    int rest = qs + HM * C + c3;
    rest = c3 < 0 ? qb + HM + c2 : rest;
    rest = c2 < 0 ? qw + HM * H + cl : rest;
    const int ct2 = c1 - (two ? 2 * LS : LS);
    rest = ct2 >= 0 ? pwo + ct2 : rest;                                   // w_o, b_o, s_o, loss
    rest = c1 < 0 ? c : rest;                                             // W_in, b_in
    const int res = col < NT ? tile : rest;
    return col < ncols ? res : -1;
}

// Image slots of flat parameter j: every parameter has a primary slot; W_in/b_in of the leftover units and
// W_k[:, HM+u] are mirrored into a second table.  Returns the number of slots (1 or 2).
__device__ __forceinline__ int image_slots(const ImgMap& m, int j, int (&slot)[2]) {
    if (j < m.p_bin) {  // input.weight [H][C] -> layer-0 A operand rows 0..C-1 (+ k-group TM table for leftovers)
        const int i = j / m.C, c = j - i * m.C;
        slot[0] = m.off_wine + c * m.PT + i;
        if (i >= m.HM) {
            slot[1] = m.off_win + c * 16 + (i - m.HM);
            return 2;
        }
        return 1;
    }
    if (j < m.p_w[0]) {  // input.bias -> layer-0 A operand row C
        const int i = j - m.p_bin;
        slot[0] = m.off_wine + m.C * m.PT + i;
        if (i >= m.HM) {
            slot[1] = m.off_bin + (i - m.HM);
            return 2;
        }
        return 1;
    }
    for (int k = 0; k < m.L; ++k) {
        if (j < m.p_b[k]) {  // skip.k.ln.weight [H][H]
            const int q = j - m.p_w[k];
            const int o = q / m.H, i = q - o * m.H;
            slot[0] = m.off_w[k] + o * m.S + i;
            if (i >= m.HM) {
                slot[1] = m.off_wct[k] + (i - m.HM) * m.PT + o;
                return 2;
            }
            return 1;
        }
        if (j < m.p_s[k]) {  // skip.k.ln.bias
            slot[0] = m.off_w[k] + (j - m.p_b[k]) * m.S + m.ext[0];
            return 1;
        }
        if (j < m.p_s[k] + m.H * m.C) {  // skip.k.skp.weight [H][C]
            const int q = j - m.p_s[k];
            const int o = q / m.C, c = q - o * m.C;
            slot[0] = m.off_w[k] + o * m.S + m.ext[1 + c];
            return 1;
        }
    }
    if (j < m.p_bo) {
        slot[0] = m.off_wo + (j - m.p_wo);
        return 1;
    }
    slot[0] = m.off_sc + (j - m.p_bo);  // b_o, s_o[c]
    return 1;
}

struct StepArgs {
    const float* wimg;     // [n_images][IMG_FLOATS] parameter images
    const float* targets;  // [n_images][N]            (TRAIN)
    const float* coef;     // [n_images][2] c_fg, c_bg (TRAIN)
    float* slabs;          // [n_images][wgs][PS]      (TRAIN)
    float* logits;         // [n_images][N] or null
    float* dcoords;        // [n_images][C][N] dL/dcoords (DX kernels only)
    InrGridDesc grid;
    long long N;
    int n_images, wgs, PS, loss_kind;
    float act_omega;       // INR_ACT_SIN kernels
};

// Cross-lane sums on the VALU (DPP / permlane swaps): no LDS round trip, no s_waitcnt.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// log(x) clamped at -100 like torch.nn.BCELoss - with a compare + select, NOT fmaxf: fmaxf(NaN, -100) = -100 would turn a NaN
// probability into a finite loss, while torch's clamp keeps the NaN and the reference then raises "Loss is nan or inf!".
__device__ __forceinline__ float bce_log(float x) {
    const float l = logf(x);
    return l < -100.f ? -100.f : l;
}

__device__ __forceinline__ float sum_over_points(float v) {  // the 16 lanes sharing lane>>4 (one DPP row)
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float sum_over_groups(float v) {  // the 4 lanes sharing lane&15
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows 0+1, rows 2+3
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // halves
    }
    return v;
}

#ifndef INR_SCHED_HINTS
#define INR_SCHED_HINTS 1
#endif
#if INR_SCHED_HINTS
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#else
#define SGB(mask, n)
#endif
// Keep successive k-steps of MFMAs in program order (everything else may still move across): without it hipcc regroups
// the products per accumulator, i.e. into dependent chains that pay the 40-cycle latency instead of the 32-cycle issue.
#ifndef INR_BWD_B128
#define INR_BWD_B128 0   // 1: the backward product reads the weight operand of a k-step as TM / 4 16-byte LDS reads (column-permuted tiles) instead of TM 4-byte ones: 343 fewer instructions per chunk, bit-identical, NOT faster (profiles/r04_issue_slot_budget.md section 5)
#endif
#ifndef INR_STAGE_Z0_EARLY
#define INR_STAGE_Z0_EARLY 1
#endif
#ifndef INR_SLAB_NT
#define INR_SLAB_NT 1   // the slab tiles are written once and read once by another kernel: non-temporal stores (A/B in DESIGN.md 6)
#endif
#ifndef INR_MFMA_ORDER
#define INR_MFMA_ORDER 1
#endif
#if INR_MFMA_ORDER
#define MFMA_STEP_FENCE() __builtin_amdgcn_sched_barrier(0x7F6)
// ... and one that LDS reads may not cross either: operand reads written ahead of a block of MFMAs stay ahead of it
// (hipcc otherwise sinks each read to just before its use and waits for it there, one LDS latency per pair of MFMAs).
#define OPERAND_FENCE() __builtin_amdgcn_sched_barrier(0x676)
#else
#define MFMA_STEP_FENCE()
#define OPERAND_FENCE()
#endif

// Diagnostic build only (-DINR_STAMPS=1): per-phase cycle sums of workgroup 0 / wave 0, read back by tools/stamps.py.
// Never enabled in the shipped library (stamps fence the schedule; read their SHARES, not the run time).
#ifndef INR_STAMPS
#define INR_STAMPS 0
#endif
#if INR_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_wgtimes[1024][4];   // per workgroup, s_memrealtime (100 MHz): entry, loop start, loop end, stores done
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_sum[k] += t_ - st_prev;                                                                 \
        st_prev = t_;                                                                              \
    } while (0)
#else
#define STAMP(k)
#endif

// DX: additionally write dL/d(coordinates) (needed when the ICNN sits behind a learned deformation of the grid).
template <int H, int C, bool TRAIN, bool DX = false, int ACT0 = INR_ACT_RELU>
__global__ __launch_bounds__(WG_THREADS, 1) void icnn_step_kernel(const StepArgs a) {
    static_assert(!DX || TRAIN, "coordinate gradients are a by-product of the backward pass");
    static_assert(!DX || ACT0 == INR_ACT_RELU, "the coordinate-gradient kernels are built for relu networks only");
    using G = Cfg<H, C>;
    constexpr int TM = G::TM, KG = G::KG, HM = G::HM, HR = G::HR, S = G::S, PT = G::PT, RPW = G::RPW, NEXT = G::NEXT;
    // Backward product dZ0 = dZ1 . W1: with B128 a lane reads the weight row of a k-step as 16-byte vectors, columns 4 l15 .. 4 l15 + 3 of
    // every 64-column block - element j of block u is the operand of tile 4 u + j, which therefore holds hidden unit 64 u + 4 l15 + j where
    // the 4-byte reads' tile t holds 16 t + l15.  Everything dZ0 meets afterwards (the layer-0 relu mask, the layer-0 gradient's rows,
    // W_in for the coordinate gradients) is indexed through bcol; each output element is the same sum in the same order.
    constexpr bool B128 = INR_BWD_B128 && (TM % 4 == 0);
    auto bcol = [](int t, int l) { return B128 ? 64 * (t >> 2) + 4 * l + (t & 3) : 16 * t + l; };
    constexpr int HRA = HR > 0 ? HR : 1;  // array extent (zero-length arrays are not allowed)
    static_assert(TM % RPW == 0, "row tiles must split evenly over the waves that own rows");
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

#if INR_STAMPS
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
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
    float wol[HRA];  // w_o of the leftover units
#pragma unroll
    for (int u = 0; u < HRA; ++u) wol[u] = HR > 0 ? woT[HM + u] : 0.f;

    // per-lane LDS addresses
    const float* const wf = Wimg + l15 * S + 4 * g;  // forward A operand: row 16t + l15, columns 16tk + 4g ..+3
    const float* const wb = Wimg + l15;              // backward A operand: row o, column 16t + l15
    const bool row_ok = (4 * RPW == TM) || wave * RPW < TM;             // this wave owns row tiles of the dW product
    const int arow = 16 * wave * RPW + l15;          // first dW row tile of this wave (+ lane column)

    // persistent gradient accumulators (TRAIN)
    f32x4 dW[RPW][KG];     // dW1ext tiles: rows 16*(wave*RPW+j).., columns 16*b..             (MFMA)
    f32x4 dL0[TM];         // layer-0 gradient of this wave's own points: rows 16t.., columns = slots of k-group TM (MFMA)
    f32x4 dwo[TM];         // dw_o partial sums over this lane's points                          (VALU)
    float dwol[HRA];       // ... leftover units (lane group 0 only)
    float dWl[HRA][KG];    // leftover rows of dW1ext: column 16b + l15, partial over this lane group's points
    float dL0l[HRA][NEXT]; // leftover rows of the layer-0 gradient (lane group 0 only)
    float loss_acc = 0.f, dbo = 0.f, dso[C];
    if (TRAIN) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int b = 0; b < KG; ++b) dW[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            dwo[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            dL0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < HRA; ++u) {
            dwol[u] = 0.f;
#pragma unroll
            for (int b = 0; b < KG; ++b) dWl[u][b] = 0.f;
#pragma unroll
            for (int e = 0; e < NEXT; ++e) dL0l[u][e] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) dso[c] = 0.f;
    }

    const int n_chunks = (int)((N + SP - 1) / SP);
    // coordinates (and target) of this lane's point of a chunk; invalid points are clamped to the last valid one
    struct PointIn {
        float x[C];
        float tg;
    };
    const bool fast_div = N <= (1ll << 24);
    const float inv_width = a.grid.mode == INR_GRID_SEPARABLE ? 1.f / (float)a.grid.width : 0.f;
    auto load_point = [&](int chunk) -> PointIn {
        PointIn q;
        int pc = chunk * SP + wave * 16 + l15;   // points per image < 2^31 (checked on the host)
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
    unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
    const unsigned long long st_begin = st_prev;
    unsigned long long st_f0 = 0, st_fsum = 0;
    const unsigned long long rt_loop = __builtin_amdgcn_s_memrealtime();
#endif
    for (int chunk = wg; chunk < n_chunks; chunk += a.wgs) {
        STAMP(0);
        const int p = chunk * SP + wave * 16 + l15;
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
            for (int r = 0; r < 4; ++r) z[r] = act0_f<ACT0>(z[r], a.act_omega);
            return z;
        };
        f32x4 z0lpre;   // pre-activations of the last k-group (its leftover hidden units: lane group 0, k-step u)
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
                    z0[TM][r] = fmaxf(v[r], fl[r]);
                } else {   // hidden leftovers: the activation; ext inputs (floor -inf): identity; unused slots stay 0 (cos(0) is not)
                    const bool hid = g == 0 && r < HR;
                    z0[TM][r] = hid ? act0_f<ACT0>(v[r], a.act_omega) : (fl[r] < 0.f ? v[r] : 0.f);
                }
            }
        }

        STAMP(1);
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
#if INR_STAMPS
            if (tk == 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_f0)::"memory"); __builtin_amdgcn_sched_barrier(0); }
            if (tk == 7) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); st_fsum += t_ - st_f0; }
#endif
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
                for (int r = 0; r < 4; ++r) z0[tk + 1][r] = act0_f<ACT0>(zn[r], a.act_omega);
            }
#if INR_STAGE_Z0_EARLY
            // z0 tile tk has served as this k-group's operand; all that is left for it is to be the dW product's staged operand
            if (TRAIN && tk < TM) *(f32x4*)(stB + (wave * 16 + l15) * G::SB + 4 * g + 16 * tk) = z0[tk];
#endif
            __builtin_amdgcn_sched_barrier(0);
        }

        STAMP(2);
        // relu mask of layer 0 in the transposed layout of the backward product (rows = points): z0^T is the layer-0 product
        // with swapped operands, TM more MFMAs - issued here, where the matrix pipe would otherwise idle under the VALU work
        // of the output layer and the data term.
        f32x4 z0p[TM];
        if (TRAIN) {
            float wie[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) wie[t] = WinE[g * PT + bcol(t, l15)];   // (the columns the backward product's tile t holds)
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
#if !(defined(INR_EXP_NOVALU) && (INR_EXP_NOVALU & 1))   // (timing experiment: wrong results)
                acc[t][r] = relu0(acc[t][r]);  // z1
#endif
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
                const float lp = bce_log(pr), lq = bce_log(1.f - pr);   // clamped at -100, NaN kept (torch.nn.BCELoss)
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
            const int pl = wave * 16 + l15;  // this lane's row in the stages
            float* const sa = stA + pl * G::SA + 4 * g;
            float* const sb = stB + pl * G::SB + 4 * g;
            // dz1 of tile t (in place over acc), dw_o accumulation, staging of dz1 (A) and z0ext (B)
            auto dz1_tile = [&](int t) {
                const f32x4 wot = wo[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z1 = acc[t][r];
                    dwo[t][r] = fmaf(dy, z1, dwo[t][r]);
#if defined(INR_EXP_NOVALU) && (INR_EXP_NOVALU & 2)
                    acc[t][r] = dy * wot[r];
#else
                    acc[t][r] = z1 > 0.f ? dy * wot[r] : 0.f;
#endif
                }
                *(f32x4*)(sa + 16 * t) = acc[t];
#if !INR_STAGE_Z0_EARLY
                *(f32x4*)(sb + 16 * t) = z0[t];
#endif
            };
            if (HR > 0 && g == 0) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < HR; ++u) v[u] = dzl[u];
                *(f32x4*)(sa + HM) = v;
            }
            if (g < 3) *(f32x4*)(sb + HM) = z0[TM];

            STAMP(3);
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
            float bq[2][B128 ? 1 : TM];          // B operands (weights): row o of this k-step, columns 16t + l15
            f32x4 bq4[2][B128 ? TM / 4 : 1];     // B128: the same row as 16-byte reads, columns 64 u + 4 l15 .. + 3
            f32x4 wcq[2][HRA];    // leftover input columns W1[o][HM+u] at this lane's positions of a tile
            auto b_row = [&](int ks) -> const float* {  // LDS row of the weight operand for k-step ks
                const int tk = ks >> 2, r = ks & 3;
                if (tk < TM) return wb + (16 * tk + 4 * g + r) * S;
                return wb + (g == 0 ? (HM + r) * S : 0);  // leftover outputs live in lane group 0 (others: A = 0)
            };
            auto b_vec = [&](int ks, int u) { return *(const f32x4*)(b_row(ks) + 3 * l15 + 64 * u); };   // (wb = Wimg + l15)
            dz1_tile(0);
            {
                const float* br = b_row(0);
                if constexpr (B128) {
#pragma unroll
                    for (int u = 0; u < TM / 4; ++u) bq4[0][u] = b_vec(0, u);
                } else {
#pragma unroll
                    for (int t = 0; t < TM; ++t) bq[0][t] = br[16 * t];
                }
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
                    if constexpr (B128) {
                        dz0[t] = MFMA16(bop, bq4[ks & 1][t >> 2][t & 3], ks == 0 ? (f32x4{0.f, 0.f, 0.f, 0.f}) : dz0[t]);
                        if (ks + 1 < KS && (t & 3) == 0) bq4[(ks + 1) & 1][t >> 2] = b_vec(ks + 1, t >> 2);
                    } else {
                        dz0[t] = MFMA16(bop, bq[ks & 1][t], ks == 0 ? (f32x4{0.f, 0.f, 0.f, 0.f}) : dz0[t]);
                        if (ks + 1 < KS) bq[(ks + 1) & 1][t] = b_row(ks + 1)[16 * t];
                    }
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
            STAMP(4);
            // relu mask of layer 0 (z0p, computed before the output layer); then dL0[t] += dZ0[:, tile t]^T . ext columns of
            // this wave's own stage-B rows.
            {
                float bfe[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bfe[r] = stB[(wave * 16 + 4 * g + r) * G::SB + HM + l15];
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#if defined(INR_EXP_NOVALU) && (INR_EXP_NOVALU & 4)
                        asm volatile("" ::"v"(z0p[t][r]));
#else
                        if constexpr (ACT0 == INR_ACT_RELU) dz0[t][r] = z0p[t][r] > 0.f ? dz0[t][r] : 0.f;
                        else dz0[t][r] *= dact0_f<ACT0>(z0p[t][r], a.act_omega);
#endif
                    }
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
                float dm;
                if constexpr (ACT0 == INR_ACT_RELU) dm = z0u > 0.f ? d : 0.f;
                else dm = d * dact0_f<ACT0>(z0lpre[u], a.act_omega);   // (used in lane group 0 only, where z0lpre[u] is unit HM + u)
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
                        const float w = WinE[c * PT + bcol(t, l15)];
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
            STAMP(5);
            __syncthreads();
            STAMP(6);

            // ---- dW1ext += dZ1^T Z0ext over the 64 staged points (MFMA, pipelined); leftover rows on the VALU ------
            {
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
                    const int cur = it & 1, nx = cur ^ 1;
                    const int ptc = stage_pt(it), ptn = stage_pt(it + 1 < SP / 4 ? it + 1 : it);
                    // next k-step's operands, one ds_read after each of the first RPW + KG products (never two in a row: an LDS read
                    // holds the issue port, a burst lets the matrix pipe run dry)
                    auto next_read = [&](int i) {
                        if (it + 1 < SP / 4) {
                            if (i < RPW) af[nx][i] = stA[ptn * G::SA + arow + 16 * i];
                            else if (i < RPW + KG) bf[nx][i - RPW] = stB[ptn * G::SB + 16 * (i - RPW) + l15];
                        }
                    };
                    if (HR > 0 && (it & 3) == 0) {  // every wave takes a quarter of the k-steps for the leftover rows
                        const f32x4 dl = *(const f32x4*)(stA + ptc * G::SA + HM);
#pragma unroll
                        for (int u = 0; u < HR; ++u)
#pragma unroll
                            for (int b = 0; b < KG; ++b) dWl[u][b] = fmaf(dl[u], bf[cur][b], dWl[u][b]);
                    }
                    if (row_ok) {
#pragma unroll
                        for (int j = 0; j < RPW; ++j)
#pragma unroll
                            for (int b = 0; b < KG; ++b) {
                                dW[j][b] = MFMA16(af[cur][j], bf[cur][b], dW[j][b]);
                                next_read(j * KG + b);
                                OPERAND_FENCE();
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < RPW + KG; ++i) next_read(i);
                    }
                    if (RPW * KG < RPW + KG) {
#pragma unroll
                        for (int i = RPW * KG; i < RPW + KG; ++i) next_read(i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            STAMP(7);
            __syncthreads();
            STAMP(8);
        }
    }
#if INR_STAMPS
    unsigned long long st_loop_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_loop_end)::"memory");
    const unsigned long long rt_loop_end = __builtin_amdgcn_s_memrealtime();
#endif

    if (TRAIN) {
        float* __restrict__ slab = a.slabs + ((size_t)img * a.wgs + wg) * a.PS;
        // ---- dW1ext tiles of this wave: one 16-byte store per lane and tile, in accumulator order (Cfg: gradient slab) --------
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
#pragma unroll
                for (int b = 0; b < KG; ++b) {
                    bool used = true;   // the last column tile holds the leftover units and the ext inputs only: skip its padding
                    if (b == KG - 1) {
                        used = l15 < HR;
#pragma unroll
                        for (int e = 0; e < NEXT; ++e) used = used || (HM + l15 == G::ext_pos(e));
                    }
                    if (used) {
                        f32x4* dst = (f32x4*)(slab + ((((wave * RPW + j) * KG + b) * 64 + lane) << 2));
#if INR_SLAB_NT
                        __builtin_nontemporal_store(dW[j][b], dst);
#else
                        *dst = dW[j][b];
#endif
                    }
                }
            }
        }
        // ---- everything that was summed per lane: reduce within the wave, then over the waves through LDS -------------
        constexpr int SC_DWO = 0;                    // [PT]            dw_o by position
        constexpr int SC_DWL = SC_DWO + PT;          // [HRA][PT]       leftover rows of dW1ext by column position
        constexpr int SC_L0L = SC_DWL + HRA * PT;    // [HRA][4]        leftover rows of the layer-0 gradient
        constexpr int SC_SC = SC_L0L + HRA * 4;      // [8]             loss, db_o, ds_o
        constexpr int SC_L0 = SC_SC + 8;             // [HM][4]         layer-0 gradient of the main units by ext input
        constexpr int WSTR = SC_L0 + HM * 4;
        static_assert(4 * WSTR <= SP * G::SA, "reduction scratch must fit stage A");
        float* const scr = stA + wave * WSTR;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = sum_over_points(dwo[t][r]);
                if (l15 == 0) scr[SC_DWO + 16 * t + 4 * g + r] = v;
            }
#pragma unroll
        for (int u = 0; u < HR; ++u) {
            const float v = sum_over_points(dwol[u]);  // lane group 0 only
            if (lane == 0) scr[SC_DWO + HM + u] = v;
#pragma unroll
            for (int b = 0; b < KG; ++b) {
                const float w = sum_over_groups(dWl[u][b]);
                if (g == 0) scr[SC_DWL + u * PT + 16 * b + l15] = w;
            }
#pragma unroll
            for (int e = 0; e < NEXT; ++e) {
                const float w = sum_over_points(dL0l[u][e]);  // lane group 0 only
                if (lane == 0) scr[SC_L0L + u * 4 + e] = w;
            }
        }
        {   // dL0 tiles: row 4g + r of tile t = hidden unit bcol(t, 4g + r), column l15 = slot of k-group TM; keep the ext-input columns
            int e = -1;
#pragma unroll
            for (int k = 0; k < NEXT; ++k)
                if (HM + l15 == G::ext_pos(k)) e = k;
            if (e >= 0) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[SC_L0 + bcol(t, 4 * g + r) * 4 + e] = dL0[t][r];
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
        __syncthreads();
        auto wsum = [&](int i) { return ((stA[i] + stA[WSTR + i]) + stA[2 * WSTR + i]) + stA[3 * WSTR + i]; };
        for (int i = tid; i < H; i += WG_THREADS) slab[G::SL_WO + i] = wsum(SC_DWO + i);
        for (int i = tid; i < HR * PT; i += WG_THREADS) {
            const int u = i / PT, pos = i - u * PT;
            const float v = wsum(SC_DWL + i);
            if (pos < H) slab[G::SL_W1L + u * H + pos] = v;
            else if (pos == G::ext_pos(0)) slab[G::SL_B1L + u] = v;
            else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (pos == G::ext_pos(1 + c)) slab[G::SL_S1L + u * C + c] = v;
            }
        }
        for (int i = tid; i < HM * NEXT; i += WG_THREADS) {
            const int row = i / NEXT, e = i - row * NEXT;
            const float v = wsum(SC_L0 + row * 4 + e);
            if (e == 0) slab[G::SL_TILE + G::P_BIN + row] = v;
            else slab[G::SL_TILE + G::P_WIN + row * C + (e - 1)] = v;
        }
        if (tid < HR * NEXT) {
            const int u = tid / NEXT, e = tid - u * NEXT;
            const float v = wsum(SC_L0L + u * 4 + e);
            if (e == 0) slab[G::SL_TILE + G::P_BIN + HM + u] = v;
            else slab[G::SL_TILE + G::P_WIN + (HM + u) * C + (e - 1)] = v;
        }
        if (tid < 2 + C) {
            const float v = wsum(SC_SC + tid);
            if (tid == 0) slab[G::SL_LOSS] = v;  // loss partial
            else if (tid == 1) slab[G::slab_col(G::P_BO)] = v;
            else slab[G::slab_col(G::P_SO) + tid - 2] = v;
        }
    }
#if INR_STAMPS
    if (TRAIN && (tid & 63) == 0 && blockIdx.x < 1024) {
        __builtin_amdgcn_s_waitcnt(0);   // this wave's slab stores have left
        const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            g_wgtimes[blockIdx.x][0] = rt_entry;
            g_wgtimes[blockIdx.x][1] = rt_loop;
            g_wgtimes[blockIdx.x][2] = rt_loop_end;
        }
        atomicMax(&g_wgtimes[blockIdx.x][3], rt_end);
    }
    if (TRAIN && blockIdx.x == 0 && tid == 0) {
        unsigned long long t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
        for (int k = 0; k < 9; ++k) g_stamps[k] = st_sum[k];
        g_stamps[9] = st_begin;     // (absolute) loop start
        g_stamps[10] = st_loop_end - st_begin;
        g_stamps[11] = t_end - st_loop_end;   // epilogue
        g_stamps[12] = st_fsum;               // forward k-groups 2..6 (5 x 33 MFMAs per chunk)
    }
#endif
}

}  // namespace
