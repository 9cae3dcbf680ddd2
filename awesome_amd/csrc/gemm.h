// gemm.h - hand-written fp32 MFMA GEMM for the layer-by-layer path (wide.h) and the star prior's minibatch products (star.h).
//
//     C [M x N] = op(A) . op(B)        op(A) = A [M][K] or A^T with A stored [K][M];   op(B) = B^T with B stored [N][K], or B [K][N]
//
// row-major operands with arbitrary row strides, any M, N, K (at the edges addresses are clamped into the matrix and what lies outside
// is replaced by 0 on the way into LDS - nothing is padded in memory and no load sits behind a branch), optional split of the
// contraction over blockIdx.z (every z writes its own [M x N] partial: the weight gradients contract over the 65 536 points of an
// image and are added up in chunk order by a second kernel - no atomics, reproducible), and an epilogue functor applied to every
// output element in registers (bias + skip + relu of a hidden layer; the relu / periodic-activation mask of the backward pass), so
// the activations make one trip to HBM per layer and direction instead of three.  The NN product can also leave the column sums of its
// output tile against the points' (1, x) (`extsum`: the bias / skip-weight gradients of the layer below), the HIDDEN epilogue copies the
// ext columns of the activation rows along.
//
// Shape of the kernel (gfx950): 256 threads = 4 waves, a 128 x 128 output tile per workgroup, 64 x 64 per wave = 4 x 4 accumulator
// tiles of v_mfma_f32_16x16x4_f32 (64 accumulator registers), K in steps of 16 through LDS, global loads of step s + 1 in flight
// while step s multiplies (register-staged double buffer: two 20 KB LDS buffers, one barrier per step; <= 128 VGPRs: four workgroups
// per CU).  Three instantiations per operand form: the general one picks 16- / 8- / 4-byte loads per operand and tile (odd strides,
// unaligned bases: star.h's shapes); V4 loads 16 bytes everywhere from clamped addresses; the buffer mode (GemmArgs::buf, what wide.h
// launches) reads the operands through buffer descriptors - the hardware's range check is the only bound, no clamp and no select is
// left in the k-loop (64 MFMAs, 8 LDS reads, 4 loads, 4 LDS writes, ~15 others).  Workgroups are
// dealt to tiles XCD by XCD (consecutive tiles share an L2).  Measured (h = 256, N = 65 536, one launch): the MFMAs alone 58 us (54.6 at
// the pipe's peak), the k-loop 71 - 79 and 83 - 99 with epilogue for V4, 77 - 94 with epilogue in buffer mode; the ablation table is in profiles/NOTES.md.
//
// How the operands reach the matrix pipe with ONE ds_read_b128 per operand tile and 16 k (8 LDS reads per 64 MFMAs):
//   * an operand that is contiguous along k (A [M][K], B [N][K]) sits in LDS as [row][16 k] (row stride 20 floats: 16-byte aligned,
//     two-way conflicts in three of sixteen slots); lane (g, l15) reads row 16 i + l15, k = 4 g .. 4 g + 3.  MFMA number kk of the step takes element kk of that
//     vector from every lane group, i.e. it contracts k in {kk, 4 + kk, 8 + kk, 12 + kk} - any partition of the 16 k will do as long
//     as both operands use the same one;
//   * an operand that is contiguous along its free index (A^T stored [K][M], B [K][N]) sits in LDS as [k][128 (+4)]; lane (g, l15)
//     reads row k = 4 g + kk, columns 4 l15 .. 4 l15 + 3 - element i of the vector is its value for accumulator tile i, i.e. tile i
//     holds the rows / columns 4 q + i instead of 16 i + q.  The permutation is undone where the tile is stored (for the columns it
//     even helps: a lane's four tiles are four consecutive floats of C, one 16-byte store).  A B operand stored [N][K] gets the same
//     column order by a row permutation on its way INTO LDS (tile row 4 a + b of a 64-row half is written to LDS row 16 b + a): every operand
//     form stores 16 bytes per lane and row.
// The reference arithmetic these products serve: awesome/model/convex_net.py:205-214 (z_{k+1} = relu(W_k z_k + b_k + S_k x)) and its
// backward pass; star.ipynb cell 2 / 3 for the star prior.
#pragma once
#include "icnn_step.h"

namespace {

#ifndef GM_WAVES
#define GM_WAVES 4   // waves per SIMD the V4 kernels are compiled for (<= 128 VGPRs)
#endif
constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 16;
constexpr int GM_LDK = 20;            // [row][k] layout: floats per row
constexpr int GM_LDF = GM_BM + 4;     // [k][free] layout: floats per k-row
constexpr int GM_STAGE = GM_BM * GM_LDK > GM_BK * GM_LDF ? GM_BM * GM_LDK : GM_BK * GM_LDF;   // floats per operand and buffer

enum { GEMM_EPI_STORE = 0, GEMM_EPI_HIDDEN = 1, GEMM_EPI_MASK = 2, GEMM_EPI_MASKP = 3 };   // (MASKP: kernel-side name of MASK with a periodic mask_act)

struct GemmArgs {
    const float* A;        // TA ? [K][lda] : [M][lda]
    const float* B;        // TB ? [N][ldb] : [K][ldb]      (TB: C = A . B^T)
    float* C;              // [M][ldc] (+ z * c_split_stride)
    int M, N, K, lda, ldb, ldc;
    int k_per_split;       // multiple of GM_BK; gridDim.z = ceil(K / k_per_split)
    long long c_split_stride;
    // epilogue
    int epi;
    const float* bias;     // HIDDEN: b [N]
    const float* skip;     // HIDDEN: S [N][C]
    const float* ext;      // HIDDEN: x_c of row m = ext[m * ext_ld + 1 + c] (the ext columns (1, x) of the previous layer's activations)
    int ext_ld, C_in;
    int ext_copy;          // HIDDEN: also C[m][N .. N + ext_copy) = ext[m][0 .. ext_copy) (the ext columns and padding of the activation rows travel along)
    const float* mask;     // MASK: multiply by [mask[m * mask_ld + n] > 0] (mask_act == relu) or by the activation's derivative at it
    int mask_ld, mask_act;
    float omega;
    int vecA, vecB;        // widest aligned load of the operand: 4 floats (base and row stride multiples of 16 bytes), 2, or 1
    int buf;               // V4 operands may be read through buffer descriptors: rows past an operand's last row read as 0 in hardware, and
                           // whatever else lies outside op(A) / op(B) (a k past K inside a padded row, a free index past M / N) is FINITE
                           // and either multiplied by a zero of the other operand or lands in rows / columns of C that are not stored -
                           // no address clamp, no select (wide.h: zero-padded weight copy, zeroed padding columns)
    int c_zero_to;         // MASK: columns [N, c_zero_to) of every stored row get zeros (the padding of a [.][ldc] activation buffer)
    float* extsum;         // NN products only, optional: [row tiles][N][1 + C_in] = sum over the tile's 128 rows m of C[m][n] (1, x_m) (x from `ext`)
    int padA, padB;        // the operand's rows may be READ up to the next multiple of 4 floats past their logical end (finite values
                           // there: the padded activation rows of wide.h); what is read there is replaced by 0
};

// One thread's share of an operand tile: two chunks of 4 consecutive floats along the operand's contiguous dimension, each at (r, c ..
// c + 3) of the stored matrix - r below `rmax`, c below `cmax`; everything out of range is to read as 0.  No branch around any load and
// no use of a loaded value here: addresses are clamped into the matrix and the LDS store (store_tiles) replaces what was out of range
// by 0 - so all loads of a k-step stay in flight behind the MFMAs of the step before
// (a guarded load, or a select next to the load, costs a wait for the data in front of the MFMAs).
//   V = 4 / 2: vector loads; a vector that straddles cmax is loaded whole (the caller guarantees it may: `pad`, or cmax a multiple of V)
//   V = 1:     four clamped scalar loads
// Which chunks a thread owns is chosen for coalescing - neighbouring lanes read neighbouring 16 bytes:
//   operand contiguous along k, tile [128 rows][16 k]: chunk h = row (tid >> 2) + 64 h, k offset 4 (tid & 3)  (4 lanes = a row's 64 bytes)
//   operand contiguous along its free index, tile [16 k][128]: chunk h = k row tid >> 4, offset 4 (tid & 15) + 64 h  (16 lanes = 256 bytes)
// (round 4 measured the k-loop of the three products at 73 / 84 / 72.5 us with 8 consecutive floats per thread - neighbouring lanes 32
// bytes apart, in two instructions - against 62 - 63 us without any global load.)
#if defined(GM_EXP) && (GM_EXP & 16)   // timing experiment: no LDS reads in the k-loop
#define GM_LDS_READ(ptr, alt) (alt)
#else
#define GM_LDS_READ(ptr, alt) (*(const f32x4*)(ptr))
#endif

template <int V>
__device__ __forceinline__ void gemm_load4(const float* __restrict__ base, int ld, int r, int rmax, int c, int cmax, f32x4& v) {
    const bool rok = r < rmax;
    const float* row = base + (size_t)(rok ? r : rmax - 1) * ld;
    if (V == 4) {
        v = *(const f32x4*)(row + (c < cmax ? c : 0));
    } else if (V == 2) {
        const f32x2 lo = *(const f32x2*)(row + (c < cmax ? c : 0)), hi = *(const f32x2*)(row + (c + 2 < cmax ? c + 2 : 0));
        v = f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = row[c + e < cmax ? c + e : cmax - 1];
    }
}
// the load flavour of a tile whose contiguous range is [c0, c0 + len) of cmax: the operand's vector width where no vector straddles
// cmax (or may be read past it), scalar loads otherwise.  Uniform over the workgroup.
__device__ __forceinline__ int gemm_flavour(int vec, int pad, int c0, int len, int cmax) {
    if (vec == 1) return 1;
    const bool straddle = (cmax % vec) != 0 && c0 < cmax && cmax < c0 + len;
    return (straddle && !pad) ? 1 : vec;
}
__device__ __forceinline__ void gemm_load4_any(int flavour, const float* __restrict__ base, int ld, int r, int rmax, int c, int cmax, f32x4& v) {
    if (flavour == 4) gemm_load4<4>(base, ld, r, rmax, c, cmax, v);
    else if (flavour == 2) gemm_load4<2>(base, ld, r, rmax, c, cmax, v);
    else gemm_load4<1>(base, ld, r, rmax, c, cmax, v);
}

// V4: both operands take 16-byte loads everywhere (aligned bases and row strides; no vector straddles the end of a row, or it may be
// read: gemm_all_vec4) - the k-loop then has no branch in it.  The other instantiation picks a load flavour per operand and tile.
template <bool TA, bool TB, int MODE, int EPI>   // MODE 0: general, 1: V4, 2: V4 through buffer loads (GemmArgs::buf); EPI: the epilogue (GemmArgs::epi)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE ? GM_WAVES : 2, GM_WAVES))) void gemm_kernel(const GemmArgs a) {
    constexpr bool V4 = MODE >= 1, BUF = MODE == 2, ISMASK = EPI == GEMM_EPI_MASK || EPI == GEMM_EPI_MASKP;
    __shared__ __attribute__((aligned(16))) float As[2][GM_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // workgroup -> tile.  n fastest, so that the column tiles of one row tile run together and share its A rows in a cache - and the
    // hardware deals consecutive workgroups to the 8 XCDs (each with its own L2) in turn, so the launch order is first re-dealt such
    // that consecutive tiles land on ONE XCD (wg % 8 picks the XCD, wg / 8 the slot on it).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gxy = gx * gridDim.y, total = gxy * gridDim.z;
        if ((total & 7) == 0 && gxy > 1) {
            const int wg = bz * gxy + by * gx + bx, t = (wg & 7) * (total >> 3) + (wg >> 3);
            bz = t / gxy;
            by = (t - bz * gxy) / gx;
            bx = t - bz * gxy - by * gx;
        }
    }
    const int m0 = by * GM_BM, n0 = bx * GM_BN;
    const int k_lo = bz * a.k_per_split;
    const int k_hi = min(a.K, k_lo + a.k_per_split);
#if defined(GM_EXP) && (GM_EXP & 2)   // timing experiment: the epilogue alone
    const int steps = 0;
#else
    const int steps = (k_hi - k_lo + GM_BK - 1) / GM_BK;
#endif

    // ---- global -> registers -> LDS: every thread owns two 16-byte chunks of each operand tile (gemm_load4) ----------------------------
    // flavours of the operands' interior k-steps (a contiguous-k operand's last, partial step may need another: see load_tiles)
    const int flA = V4 ? 4 : (TA ? gemm_flavour(a.vecA, a.padA, m0, GM_BM, a.M) : a.vecA);
    const int flB = V4 ? 4 : (TB ? a.vecB : gemm_flavour(a.vecB, a.padB, n0, GM_BN, a.N));
    // chunk h of this thread in operand tiles at k = kbase: (row, column) of the stored matrix, and their bounds
    const int rmaxA = TA ? k_hi : a.M, cmaxA = TA ? a.M : k_hi, rmaxB = TB ? a.N : k_hi, cmaxB = TB ? k_hi : a.N;
    auto chunkA = [&](int kbase, int h, int& r, int& c) {
        r = TA ? kbase + (tid >> 4) : m0 + (tid >> 2) + 64 * h;
        c = TA ? m0 + (tid & 15) * 4 + 64 * h : kbase + (tid & 3) * 4;
    };
    auto chunkB = [&](int kbase, int h, int& r, int& c) {
        r = TB ? n0 + (tid >> 2) + 64 * h : kbase + (tid >> 4);
        c = TB ? kbase + (tid & 3) * 4 : n0 + (tid & 15) * 4 + 64 * h;
    };
    // BUF: raw buffer descriptors over the stored operands (stride 0, num_records = rows x row stride in bytes: an offset past it reads 0)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, BUF ? (TA ? a.K : a.M) * a.lda * 4 : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, BUF ? (TB ? a.N : a.K) * a.ldb * 4 : 0, 0x00020000);
    auto load_tiles = [&](int kbase, f32x4 (&ra)[2], f32x4 (&rb)[2]) {
        const bool tail = kbase + GM_BK > k_hi;
        const int fa = V4 ? 4 : (!TA && tail ? gemm_flavour(a.vecA, a.padA, kbase, GM_BK, k_hi) : flA);
        const int fb = V4 ? 4 : (TB && tail ? gemm_flavour(a.vecB, a.padB, kbase, GM_BK, k_hi) : flB);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rA, cA, rB, cB;
            chunkA(kbase, h, rA, cA);
            chunkB(kbase, h, rB, cB);
            if (BUF) {   // one 32-bit offset per chunk; the descriptor's range check is the only bound
                // (a chunk past the free extent of an operand stored along its free index is sent out of range too: it reads 0 without
                // touching the next row's cache lines - at h = 350 a ninth of the column chunks)
                const unsigned oa = TA && cA >= cmaxA ? 0xfffffff0u : (unsigned)(rA * a.lda + cA) * 4u;
                const unsigned ob = !TB && cB >= cmaxB ? 0xfffffff0u : (unsigned)(rB * a.ldb + cB) * 4u;
                ra[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0));
                rb[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ob, 0, 0));
            } else if (V4) {
#if !(defined(GM_EXP) && (GM_EXP & 256))   // (timing experiment: the A tiles of the k-loop are not loaded)
                gemm_load4<4>(a.A, a.lda, rA, rmaxA, cA, cmaxA, ra[h]);
#endif
#if !(defined(GM_EXP) && (GM_EXP & 512))   // (... the B tiles)
                gemm_load4<4>(a.B, a.ldb, rB, rmaxB, cB, cmaxB, rb[h]);
#endif
            } else {
                gemm_load4_any(fa, a.A, a.lda, rA, rmaxA, cA, cmaxA, ra[h]);
                gemm_load4_any(fb, a.B, a.ldb, rB, rmaxB, cB, cmaxB, rb[h]);
            }
        }
    };
    // registers -> LDS, what lies outside the matrix replaced by 0 (element e of a chunk is valid iff its row is and c + e < cmax)
    auto store_tiles = [&](int kbase, float* sA, float* sB, const f32x4 (&ra)[2], const f32x4 (&rb)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rA, cA, rB, cB;
            chunkA(kbase, h, rA, cA);
            chunkB(kbase, h, rB, cB);
            const int la = BUF ? 4 : (rA < rmaxA ? cmaxA - cA : 0), lb = BUF ? 4 : (rB < rmaxB ? cmaxB - cB : 0);
            float* dA = !TA ? sA + ((tid >> 2) + 64 * h) * GM_LDK + (tid & 3) * 4 : sA + (tid >> 4) * GM_LDF + (tid & 15) * 4 + 64 * h;
            // (B stored [N][K]: tile row 4 a + b of a 64-row half goes to LDS row 16 b + a, so that the reader's row 16 j + l15 is column
            // 4 l15 + j of C - a lane's four tiles are four consecutive columns in every operand form: 16-byte stores)
            float* dB = TB ? sB + (64 * h + 16 * ((tid >> 2) & 3) + (tid >> 4)) * GM_LDK + (tid & 3) * 4 : sB + (tid >> 4) * GM_LDF + (tid & 15) * 4 + 64 * h;
            f32x4 oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa[e] = e < la ? ra[h][e] : 0.f;
                ob[e] = e < lb ? rb[h][e] : 0.f;
            }
            *(f32x4*)dA = oa;
            *(f32x4*)dB = ob;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[2], rb[2];
    if (steps > 0) {
        load_tiles(k_lo, ra, rb);
        store_tiles(k_lo, As[0], Bs[0], ra, rb);
    }
    __syncthreads();
    // one k-step of MFMAs on the tiles in LDS buffer `cur`
    auto multiply = [&](int cur) {
        const float* sa = As[cur];
        const float* sb = Bs[cur];
        if (!TA && TB) {          // both contiguous along k: one read per tile; the B tiles in two halves (24 operand registers, not 32)
            f32x4 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = GM_LDS_READ(sa + (wm * 64 + 16 * i + l15) * GM_LDK + 4 * g, ra[0]);
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                f32x4 bv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = GM_LDS_READ(sb + (wn * 64 + 16 * (2 * jh + j) + l15) * GM_LDK + 4 * g, rb[0]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][2 * jh + j] = MFMA16(av[i][kk], bv[j][kk], acc[i][2 * jh + j]);
            }
        } else {
            f32x4 av[4], bv[4];   // k-contiguous operand: [tile]; free-contiguous operand: read per kk
            if (!TA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = GM_LDS_READ(sa + (wm * 64 + 16 * i + l15) * GM_LDK + 4 * g, ra[0]);
            }
            if (TB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = GM_LDS_READ(sb + (wn * 64 + 16 * j + l15) * GM_LDK + 4 * g, rb[0]);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 af, bf;
                if (TA) af = GM_LDS_READ(sa + (4 * g + kk) * GM_LDF + wm * 64 + 4 * l15, ra[0]);
                if (!TB) bf = GM_LDS_READ(sb + (4 * g + kk) * GM_LDF + wn * 64 + 4 * l15, rb[0]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = MFMA16(TA ? af[i] : av[i][kk], TB ? bv[j][kk] : bf[j], acc[i][j]);
            }
        }
    };
    // every step but the last: the next step's tiles are requested, this step multiplies, the tiles are written to the other buffer.
    // (The last step is peeled so that the loop body has no condition in it: hipcc's wait-count insertion does not correlate "no
    // loads were issued" with "no stores follow" and puts a vmcnt(0) between the A and the B loads of a conditional version.)
    for (int s = 0; s + 1 < steps; ++s) {
        const int cur = s & 1;
#if !(defined(GM_EXP) && (GM_EXP & 4))
        load_tiles(k_lo + (s + 1) * GM_BK, ra, rb);
#endif
        __builtin_amdgcn_sched_barrier(0);   // the loads are issued in front of the step's MFMAs and waited for behind them (hipcc's
        multiply(cur);                       // scheduler would move both into the MFMA sequence, half a step apart)
        __builtin_amdgcn_sched_barrier(0);
#if !(defined(GM_EXP) && (GM_EXP & 4))
        store_tiles(k_lo + (s + 1) * GM_BK, As[cur ^ 1], Bs[cur ^ 1], ra, rb);
#endif
#if !(defined(GM_EXP) && (GM_EXP & 8))
        __syncthreads();
#endif
    }
    if (steps > 0) multiply((steps - 1) & 1);

#if defined(GM_EXP) && (GM_EXP & 1)   // timing experiment: the k-loop alone (one store keeps it alive)
    {
        float t = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 4; ++r) t += acc[i][j][r];
        if (t != 12345.678f) return;
    }
#endif
    // ---- epilogue: acc[i][j][r] = C[m][n] with m = m0 + 64 wm + mrow(i, 4 g + r), n = n0 + 64 wn + ncol(j, l15) ---------------------
    // Whatever the epilogue reads per output row (the point's coordinates; the mask values) is requested for FOUR rows at a time from
    // clamped addresses, ahead of the arithmetic that uses it: one wait per four rows instead of one per element.
    float* __restrict__ Cz = a.C + (size_t)bz * a.c_split_stride;
    auto ncol = [&](int j) { return n0 + wn * 64 + 4 * l15 + j; };
    auto mrow = [&](int i, int r) { const int q = 4 * g + r; return m0 + wm * 64 + (TA ? 4 * q + i : 16 * i + q); };
    // HIDDEN: the per-column constants (bias, skip weights) of this lane's four columns once
    float bn[4] = {0.f, 0.f, 0.f, 0.f}, sn[4][3] = {};
    if (EPI == GEMM_EPI_HIDDEN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(ncol(j), a.N - 1);
            bn[j] = a.bias[n];
#pragma unroll
            for (int c = 0; c < 3; ++c) sn[j][c] = c < a.C_in ? a.skip[n * a.C_in + c] : 0.f;
        }
    }
    // a lane's four columns are consecutive (operand B contiguous along n) and its row of C / of the mask may be accessed as 16 bytes
    const bool vec_c = ncol(0) + 3 < a.N && (a.ldc & 3) == 0 && (((size_t)Cz) & 15) == 0;
    const bool vec_mask = !TB && ISMASK && ncol(0) + 3 < a.N && (a.mask_ld & 3) == 0 && (((size_t)a.mask) & 15) == 0;
    float es[4][4] = {};   // NN + extsum: this lane's share of sum_m C[m][n] (1, x_m) for its four columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float xm[4][3] = {};
        f32x4 zm[4];
        if (EPI == GEMM_EPI_HIDDEN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* xp = a.ext + (size_t)min(mrow(i, r), a.M - 1) * a.ext_ld + 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) xm[r][c] = xp[c < a.C_in ? c : 0];
            }
        } else if (ISMASK) {
            if (!TA && !TB && a.extsum) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* xp = a.ext + (size_t)min(mrow(i, r), a.M - 1) * a.ext_ld + 1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) xm[r][c] = xp[c < a.C_in ? c : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* mp = a.mask + (size_t)min(mrow(i, r), a.M - 1) * a.mask_ld;
                if (vec_mask) {
                    zm[r] = *(const f32x4*)(mp + ncol(0));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) zm[r][j] = mp[min(ncol(j), a.N - 1)];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow(i, r);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[i][j][r];
                if (EPI == GEMM_EPI_HIDDEN) {
                    v += bn[j];
#pragma unroll
                    for (int c = 0; c < 3; ++c) v = fmaf(sn[j][c], xm[r][c], v);   // (unused channels: 0 * x)
                    v = fmaxf(v, 0.f);
                } else if (ISMASK) {
                    const float z = zm[r][j];
                    if (EPI == GEMM_EPI_MASKP) v *= a.mask_act == INR_ACT_COS ? -hw_sin(z) : a.omega * hw_cos(a.omega * z);
                    else v = z > 0.f ? v : 0.f;
                }
                o[j] = v;
            }
            if (!TA && !TB && a.extsum) {   // (rows past M are products of zero-filled operand rows: 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    es[j][0] += o[j];
#pragma unroll
                    for (int c = 0; c < 3; ++c) es[j][1 + c] = fmaf(o[j], xm[r][c], es[j][1 + c]);
                }
            }
            if (m >= a.M) continue;
            if (vec_c) {
                *(f32x4*)(Cz + (size_t)m * a.ldc + ncol(0)) = o;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = ncol(j);
                    if (n < a.N) Cz[(size_t)m * a.ldc + n] = o[j];
                    else if (ISMASK && n < a.c_zero_to) Cz[(size_t)m * a.ldc + n] = 0.f;
                }
            }
        }
    }
    if (EPI == GEMM_EPI_HIDDEN && a.ext_copy > 0 && bx == (int)gridDim.x - 1 && tid < GM_BM && m0 + tid < a.M) {
        const float* src = a.ext + (size_t)(m0 + tid) * a.ext_ld;
        float* dst = Cz + (size_t)(m0 + tid) * a.ldc + a.N;
        for (int c = 0; c < a.ext_copy; ++c) dst[c] = src[c];
    }
    if (!TA && !TB && a.extsum) {
        // the column sums of the tile: lanes that share a column (4 row groups g, 2 waves wm) through LDS, added in a fixed order
        __syncthreads();                      // every wave is past its last read of the operand tiles
        float* red = &As[0][0];               // [wave][g][l15][j][4]
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(f32x4*)(red + ((((wave * 4 + g) * 16 + l15) * 4 + j) << 2)) = f32x4{es[j][0], es[j][1], es[j][2], es[j][3]};
        __syncthreads();
        if (tid < 128) {
            const int wn_ = tid >> 6, l = (tid & 63) >> 2, j = tid & 3, n = n0 + wn_ * 64 + 4 * l + j;
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wm_ = 0; wm_ < 2; ++wm_)
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) t += *(const f32x4*)(red + (((((wm_ * 2 + wn_) * 4 + g_) * 16 + l) * 4 + j) << 2));
            if (n < a.N)
                for (int c = 0; c <= a.C_in; ++c) a.extsum[((size_t)by * a.N + n) * (1 + a.C_in) + c] = t[c];
        }
    }
}

inline int gemm_vec_width(const float* p, int ld) {
    if ((((size_t)p) & 15) == 0 && (ld & 3) == 0) return 4;
    if ((((size_t)p) & 7) == 0 && (ld & 1) == 0) return 2;
    return 1;
}

// row-major C[M x N] = op(A) op(B); `splits` > 1 cuts K into that many z-slices of k_per_split (multiple of 16) writing
// C + z * c_split_stride.  Returns INR_OK / INR_ELAUNCH.
inline int gemm_launch(hipStream_t s, bool tA, bool tB, GemmArgs g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return INR_EINVAL;
    if (g.k_per_split <= 0) g.k_per_split = (g.K + GM_BK - 1) / GM_BK * GM_BK;
    const int splits = (g.K + g.k_per_split - 1) / g.k_per_split;
    g.vecA = gemm_vec_width(g.A, g.lda);
    g.vecB = gemm_vec_width(g.B, g.ldb);
    const dim3 grid((g.N + GM_BN - 1) / GM_BN, (g.M + GM_BM - 1) / GM_BM, splits);
    // 16-byte loads throughout: aligned operands whose contiguous extent is a multiple of 4 floats or may be read up to the next one
    // (a split contraction ends inside the rows of an operand contiguous along k: always readable)
    const int contigA = tA ? g.M : g.K, contigB = tB ? g.K : g.N;
    if (g.buf && ((long long)(tA ? g.K : g.M) * g.lda * 4 >= (1ll << 31) || (long long)(tB ? g.N : g.K) * g.ldb * 4 >= (1ll << 31))) g.buf = 0;   // (32-bit byte offsets)
    const bool v4 = g.vecA == 4 && g.vecB == 4 && ((contigA & 3) == 0 || g.padA) && ((contigB & 3) == 0 || g.padB) && (g.k_per_split & 3) == 0;
    // epilogues other than the plain store exist for the products that use them: HIDDEN on A . B^T (a layer's forward), MASK on A . B (its
    // backward); each kernel holds the code of ONE epilogue
    const int mode = v4 && g.buf ? 2 : (v4 ? 1 : 0);
    if (g.epi == GEMM_EPI_HIDDEN && !(!tA && tB)) return INR_EINVAL;
    if (g.epi == GEMM_EPI_MASK && !(!tA && !tB)) return INR_EINVAL;
#define GEMM_GO3(TA_, TB_, EPI_)                                                                          \
    do {                                                                                                  \
        if (mode == 2) hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 2, EPI_>), grid, dim3(256), 0, s, g);    \
        else if (mode == 1) hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 1, EPI_>), grid, dim3(256), 0, s, g); \
        else hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 0, EPI_>), grid, dim3(256), 0, s, g);              \
    } while (0)
    if (g.epi == GEMM_EPI_HIDDEN) GEMM_GO3(false, true, GEMM_EPI_HIDDEN);
    else if (g.epi == GEMM_EPI_MASK && g.mask_act == INR_ACT_RELU) GEMM_GO3(false, false, GEMM_EPI_MASK);
    else if (g.epi == GEMM_EPI_MASK) GEMM_GO3(false, false, GEMM_EPI_MASKP);
    else if (!tA && tB) GEMM_GO3(false, true, GEMM_EPI_STORE);
    else if (!tA && !tB) GEMM_GO3(false, false, GEMM_EPI_STORE);
    else if (tA && !tB) GEMM_GO3(true, false, GEMM_EPI_STORE);
    else GEMM_GO3(true, true, GEMM_EPI_STORE);
#undef GEMM_GO3
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

// the plain product (no epilogue, no split): what wide.h / star.h used to hand to a BLAS library.
// tA: A is stored [K][lda] (C = A^T ...);  tB: B is stored [N][ldb] (C = ... B^T)
inline int gemm_rm(hipStream_t s, bool tA, bool tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.epi = GEMM_EPI_STORE;
    return gemm_launch(s, tA, tB, g);
}

// C [M x N] = sum over the z-slices of part [splits][M x N], slices added in order (the fixed-order second half of a split-K product)
__global__ __launch_bounds__(256) void gemm_splitk_sum_kernel(const float* __restrict__ part, int splits, long long mn, float* __restrict__ C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= mn) return;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += part[(size_t)z * mn + e];
    C[e] = v;
}

constexpr int GEMM_SPLITK_MAX = 16;
// k_per_split of a contraction of length K cut into at most GEMM_SPLITK_MAX slices of at least 128
inline int gemm_splitk_len(int K) {
    int len = (K + GEMM_SPLITK_MAX - 1) / GEMM_SPLITK_MAX;
    len = (len + GM_BK - 1) / GM_BK * GM_BK;
    return len < 128 ? 128 : len;
}
// the plain product with the contraction cut into slices (a long K with a small M x N: a weight gradient over a minibatch):
// `part` holds GEMM_SPLITK_MAX x M x N floats; C is dense [M][N]
inline int gemm_rm_splitk(hipStream_t s, bool tA, bool tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          float* part) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = part;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
    g.k_per_split = gemm_splitk_len(K);
    g.c_split_stride = (long long)M * N;
    g.epi = GEMM_EPI_STORE;
    const int splits = (K + g.k_per_split - 1) / g.k_per_split;
    int rc = gemm_launch(s, tA, tB, g);
    if (rc) return rc;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(gemm_splitk_sum_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, (const float*)part, splits, mn, C);
    return hipGetLastError() == hipSuccess ? INR_OK : INR_ELAUNCH;
}

}  // namespace
