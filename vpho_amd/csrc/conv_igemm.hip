// Implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin.  NHWC activations make every (r,s) slice of an im2col
// row a contiguous channel run, so the A tile is gathered with 16-byte loads; weights are pre-packed [Cout][K].
// Block = WMxWN waves; block tile BMxBN (128x128 and 128x64 with 8 waves, 64x64 with 4 waves), BK = 32.  Both LDS tiles
// are K-contiguous with a 4-float pad (row stride 36 floats): each lane fetches 4 consecutive k of its row with one
// conflict-free ds_read_b128 and feeds them to 4 MFMAs (lanes 0-31 supply k = 4h+q of one half, lanes 32-63 the other;
// A and B use the same k permutation so the sum is unchanged).  Global->register prefetch of tile k+1 overlaps the
// MFMAs of tile k (double-buffered LDS, one barrier per K-step).  Blocks are renumbered so that the 8 round-robin
// dispatched XCDs each walk consecutive N-tiles of the same M-tile (A tile reuse in that XCD's L2).
//
// Roofline: fp32 MFMA, 2*M*N*K flop per launch against 157.3 TFLOP/s.
#include "common.h"
#include "../../include/vpho_hip.h"
#include <cstdlib>
#include <algorithm>

VPHO_STAMP_DECL(conv)

namespace {

#ifndef CONV_ABLATE
#define CONV_ABLATE 0                          // timing experiments only (scripts/kernel_ablate.sh conv_igemm CONV_ABLATE 1 2 4; direct-to-LDS kernel: 32 a quarter of the fragment reads, 64 no stage fills after the second, 128 no stage barrier, 256 no epilogue): 1 no global loads, 2 no in-loop
#endif                                         // barriers, 4 no LDS stores in conv_igemm_kernel -- wrong results, never in the product build
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;
constexpr int GLDS_PRE_MAX = 512;     // most input channels of a pre-activation 1x1 convolution on the direct-to-LDS kernel

struct Geo {
    vpho_conv_desc d;
    int M, K, tiles_m, tiles_n, ntiles;
    int w_ld;                       // floats between consecutive weight rows (K unless the launch reduces a K-slice)
    long long x_zs, w_zs, y_zs;     // per blockIdx.y advance of x / w / y (split reductions, groups; 0 for ordinary launches)
    long long b_zs, r_zs, x2_zs, pre_zs, ru_zs;   // ... and of bias / res / x2 / in_scale + in_shift / res_up (groups only)
    int y_linear, r_linear, vec_epilogue;
    int uni;   // Cin % 32 == 0: wave-uniform taps (direct-to-LDS kernel)
    int dbg;   // A/B switches whose results are bit-identical (VPHO_CONV_DBG, read per call): 8 = residual tile requested in the epilogue
               // (round 3) instead of in front of the last k stage, 16 = second stage requested after the first has landed (round 3).
               // The timing ablations that produce WRONG results (skip global loads / barriers / LDS stores) are compile-time only:
               // -DCONV_ABLATE=<mask> in a diagnostic build (scripts/kernel_ablate.sh), never in the product library
};

// blockIdx.y = slice of a split reduction or group of a grouped launch: the same problem on advanced operands
__device__ __forceinline__ void advance_group(vpho_conv_desc& d, const Geo& g, const unsigned by) {
    d.x += by * g.x_zs; d.w += by * g.w_zs; d.y += by * g.y_zs;
    if (d.bias) d.bias += by * g.b_zs;
    if (d.res) d.res += by * g.r_zs;
    if (d.x2) d.x2 += by * g.x2_zs;
    if (d.in_scale) { d.in_scale += by * g.pre_zs; d.in_shift += by * g.pre_zs; }
    if (d.res_up) d.res_up += by * g.ru_zs;
}

// VEC = number of consecutive 16-byte pieces (of one tile row) a thread moves per pass: 2 halves the per-K-step address
// arithmetic and bounds checks (needs Cin % 8 == 0 so that a 32-byte piece never straddles two filter taps)
template <int BM, int BN, int WM, int WN, int VEC>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const Geo g) {
    constexpr int NT = 64 * WM * WN;                  // threads per block; waves arranged WM x WN
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int KQN = 8 / VEC;                      // threads per tile row
    constexpr int ROWS = NT / KQN;                    // tile rows covered by one pass
    constexpr int A_LD = (BM + ROWS - 1) / ROWS, B_LD = (BN + ROWS - 1) / ROWS;
    constexpr bool A_PART = BM < ROWS, B_PART = BN < ROWS;   // fewer tile rows than one pass covers: extra threads idle
    static_assert(TM >= 1 && TN >= 1, "tile too small for the wave layout");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD];

    vpho_conv_desc d = g.d;
    advance_group(d, g, blockIdx.y);
    // XCD-aware renumbering: hardware block b runs on XCD (b % 8); give each XCD a contiguous run of logical tiles
    int per_xcd = gridDim.x >> 3, ntiles = g.ntiles, M_live = g.M;
    if (d.row_map) {
        // pixel-list launch: the live row count is device data.  The grid was sized for every pixel; the live tiles are spread
        // over the 8 XCDs again (contiguous runs of the LIVE tiles), the rest of the grid exits
        M_live = min(*d.row_count, g.M);
        ntiles = (M_live + BM - 1) / BM * g.tiles_n;
        per_xcd = (ntiles + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    }
    const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lb >= ntiles) return;
    const int tile_n = lb % g.tiles_n, tile_m = lb / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int kq = tid % KQN, lrow = tid / KQN;

    // per-thread A rows (output pixels): element offset of the window origin + window origin coordinates
    long long a_off[A_LD];
    int a_iy0[A_LD], a_ix0[A_LD];
    const int ohw = d.OH * d.OW;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = m0 + lrow + ROWS * j;
        if (m < M_live && !(A_PART && lrow >= BM)) {
            if (d.row_map) m = d.row_map[m];
            int n = m / ohw, rem = m - n * ohw;
            int oy = rem / d.OW, ox = rem - oy * d.OW;
            a_iy0[j] = oy * d.stride - d.pad_y;
            a_ix0[j] = ox * d.stride - d.pad_x;
            a_off[j] = ((long long)(n * d.H + a_iy0[j]) * d.W + a_ix0[j]) * d.x_ld;
        } else {
            a_off[j] = 0; a_iy0[j] = -(1 << 28); a_ix0[j] = 0;          // never in bounds
        }
    }
    long long b_off[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int n = n0 + lrow + ROWS * j;
        b_off[j] = (n < d.Cout && !(B_PART && lrow >= BN)) ? (long long)n * g.w_ld + 4 * VEC * kq : -1;
    }
    // (r, s, c) of this thread's 16-byte piece of the NEXT tile to load; advanced by BK channels per tile without divisions
    int ld_c, ld_r, ld_s;
    {
        const int kg = 4 * VEC * kq, rs = kg / d.Cin;
        ld_c = kg - rs * d.Cin; ld_r = rs / d.KW; ld_s = rs - ld_r * d.KW;
    }

    // two register stages: tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one step earlier)
    // is written to LDS, so every global load has two K-steps of MFMA time to land
    f32x4 ra0[A_LD * VEC], rb0[B_LD * VEC], ra1[A_LD * VEC], rb1[B_LD * VEC];
    auto load_next = [&](f32x4 (&ra)[A_LD * VEC], f32x4 (&rb)[B_LD * VEC]) {
        const bool kin = ld_r < d.KH;
        const int tap = (ld_r * d.W + ld_s) * d.x_ld + ld_c;
        f32x4 sc[VEC], sh[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) { sc[u] = f32x4{1.f, 1.f, 1.f, 1.f}; sh[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (d.in_scale != nullptr && kin) {
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                sc[u] = *reinterpret_cast<const f32x4*>(d.in_scale + ld_c + 4 * u);
                sh[u] = *reinterpret_cast<const f32x4*>(d.in_shift + ld_c + 4 * u);
            }
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const unsigned iy = (unsigned)(a_iy0[j] + ld_r), ix = (unsigned)(a_ix0[j] + ld_s);
            const bool inb = kin && iy < (unsigned)d.H && ix < (unsigned)d.W;
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (inb) {
                    v = *reinterpret_cast<const f32x4*>(d.x + a_off[j] + tap + 4 * u);
                    if (d.in_scale != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = v[e] * sc[u][e] + sh[u][e];
                            v[e] = t > 0.f ? t : t * d.in_slope;
                        }
                    }
                }
                ra[j * VEC + u] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const bool inb = kin && b_off[j] >= 0;
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (inb) v = *reinterpret_cast<const f32x4*>(d.w + b_off[j] + 4 * u);
                rb[j * VEC + u] = v;
            }
            b_off[j] += b_off[j] >= 0 ? BK : 0;
        }
        ld_c += BK;
        while (ld_c >= d.Cin) { ld_c -= d.Cin; if (++ld_s == d.KW) { ld_s = 0; ++ld_r; } }
    };
    auto store_tiles = [&](int buf, const f32x4 (&ra)[A_LD * VEC], const f32x4 (&rb)[B_LD * VEC]) {
        float* As = smem + buf * (BM + BN) * LDS_LD;
        float* Bs = As + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            if (A_PART && lrow >= BM) continue;
#pragma unroll
            for (int u = 0; u < VEC; ++u) *reinterpret_cast<f32x4*>(As + (lrow + ROWS * j) * LDS_LD + 4 * (VEC * kq + u)) = ra[j * VEC + u];
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            if (B_PART && lrow >= BN) continue;
#pragma unroll
            for (int u = 0; u < VEC; ++u) *reinterpret_cast<f32x4*>(Bs + (lrow + ROWS * j) * LDS_LD + 4 * (VEC * kq + u)) = rb[j * VEC + u];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    auto step = [&](int kt, f32x4 (&ld_a)[A_LD * VEC], f32x4 (&ld_b)[B_LD * VEC], const f32x4 (&st_a)[A_LD * VEC], const f32x4 (&st_b)[B_LD * VEC]) {
        const int buf = kt & 1;
        if (kt + 2 < nk && !(CONV_ABLATE & 1)) load_next(ld_a, ld_b);
        const float* As = smem + buf * (BM + BN) * LDS_LD + (wm * (BM / WM) + li) * LDS_LD + 4 * lh;
        const float* Bs = smem + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn * (BN / WN) + li) * LDS_LD + 4 * lh;
        // fragment double buffer: the LDS reads of k-group kk+1 are in flight while the MFMAs of group kk issue
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + (kk + 1) * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i][q], b[kk & 1][j][q], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk && !(CONV_ABLATE & 4)) store_tiles(buf ^ 1, st_a, st_b);
        if (!(CONV_ABLATE & 2)) __syncthreads();
    };
    load_next(ra0, rb0);
    store_tiles(0, ra0, rb0);
    if (nk > 1) load_next(ra1, rb1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, ra0, rb0, ra1, rb1);                       // loads tile kt+2 -> stage 0, stores stage 1 (tile kt+1)
        if (kt + 1 < nk) step(kt + 1, ra1, rb1, ra0, rb0);  // loads tile kt+3 -> stage 1, stores stage 0 (tile kt+2)
    }

    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    auto offsets = [&](int row, long long& yo, long long& ro) {
        if (d.row_map && d.rows_scatter) row = d.row_map[row];          // pixel list, results stored at the pixels' own positions
        if (g.y_linear && g.r_linear) {
            yo = (long long)row * d.y_sx;
            ro = (long long)row * d.r_sx;
        } else {
            int n = row / ohw, rem = row - n * ohw;
            int oy = rem / d.OW, ox = rem - oy * d.OW;
            yo = n * d.y_sn + oy * d.y_sy + ox * d.y_sx;
            ro = n * d.r_sn + oy * d.r_sy + ox * d.r_sx;
        }
    };
    if (g.vec_epilogue) {
        // Stage the accumulator tile through the (now idle) LDS so that every lane moves 16 contiguous bytes of one
        // output pixel: residual read, bias, activation and store all run on 16-B accesses (512 B per 32 lanes).
        constexpr int C_LD = BN + 4;
        float* Cs = smem;                                    // BM x (BN+4) floats <= 2*(BM+BN)*LDS_LD
        static_assert(BM * C_LD <= 2 * (BM + BN) * LDS_LD, "epilogue tile does not fit the staging LDS");
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    Cs[r * C_LD + wn * (BN / WN) + j * 32 + li] = acc[i][j][e];
                }
        __syncthreads();
        constexpr int V_PER_ROW = BN / 4, ITERS = BM * V_PER_ROW / NT;
        f32x4 v[ITERS], rv[ITERS];
        long long yo[ITERS];
        bool ok[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NT;
            const int r = idx / V_PER_ROW, c4 = idx - r * V_PER_ROW;
            const int row = m0 + r, col = n0 + 4 * c4;
            ok[it] = row < M_live && col < d.Cout;
            v[it] = *reinterpret_cast<const f32x4*>(Cs + r * C_LD + 4 * c4);
            long long ro = 0;
            yo[it] = 0;
            if (ok[it]) { offsets(row, yo[it], ro); yo[it] += col; ro += col; }
            rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (d.res && ok[it]) rv[it] = *reinterpret_cast<const f32x4*>(d.res + ro);
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (!ok[it]) continue;
            const int idx = tid + it * NT;
            const int c4 = idx % V_PER_ROW;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n0 + 4 * c4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = v[it][k] + bv[k] + rv[it][k]; o[k] = t > 0.f ? t : t * d.out_slope; }
            if (d.gate) {                                            // backward of a LeakyReLU whose output is `gate` (same layout as y)
                const f32x4 gt = *reinterpret_cast<const f32x4*>(d.gate + yo[it]);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = gt[k] > 0.f ? o[k] : o[k] * d.gate_slope;
            }
            *reinterpret_cast<f32x4*>(d.y + yo[it]) = o;
        }
        return;
    }
    const int row_base = m0 + wm * (BM / WM) + 4 * lh, col_base = n0 + wn * (BN / WN) + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + j * 32;
        if (col >= d.Cout) continue;
        const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_base + i * 32 + (e & 3) + 8 * (e >> 2);
                if (row >= M_live) continue;
                long long yo, ro;
                offsets(row, yo, ro);
                float v = acc[i][j][e] + bv;
                if (d.res) v += d.res[ro + col];
                v = v > 0.f ? v : v * d.out_slope;
                if (d.gate) v = d.gate[yo + col] > 0.f ? v : v * d.gate_slope;
                d.y[yo + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (no prologue affine): tiles are filled by `buffer_load_dwordx4 ... lds` (16 B per lane, no VGPR round
// trip, no ds_write).  One wave instruction writes 64 x 16 B = 8 tile rows of 128 B linearly, so the LDS rows are
// unpadded; bank conflicts are avoided by XOR-swizzling the 16-byte chunk index of row r with (r >> 1) & 7 -- applied to
// the per-lane SOURCE address (the LDS destination is lane-linear by construction) and again when the MFMA fragments are
// read.  Tiles are addressed through buffer resources: out-of-image taps / K and M tails carry an out-of-range offset, for which
// the hardware writes zeros -- no masking, no 64-bit address arithmetic in the loop.

// Geo::uni (Cin % 32 == 0): the 32 k of a stage lie in ONE tap: the tap is wave-uniform, the k / tap advance rides in the buffer
// instruction's scalar offset and the per-lane offsets are loop constants (only the padding test of a 3x3 stays per stage; for 1x1
// unpadded convolutions the loop has no per-lane address work at all).
// x summed over the 8 lanes that differ in lane bits 3, 4, 5 (the 8 rows of an epilogue pass), on the vector ALU alone: a DPP rotate
// inside the 16-lane row, then the gfx950 row / half swaps (v_permlane16_swap: odd rows of the first register <-> even rows of the second;
// v_permlane32_swap: upper half of the first <-> lower half of the second; both registers start as x, so their sum is the exchange sum).
// Every lane of the group ends with the same bits (a + b is commutative).  Inline assembly, both registers in / out: the builtins of this
// hipcc return their second result equal to the first (scripts/microbench/lane_sum.hip prints what the hardware does).
__device__ __forceinline__ float sum_lane_bit_4(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_lane_bit_5(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_lane_bits_345(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));       // row_ror:8
    return sum_lane_bit_5(sum_lane_bit_4(x));
}
template <int BM, int BN, int WM, int WN, bool PRE = false>
__global__ __launch_bounds__(64 * WM * WN, 4) void conv_igemm_glds_kernel(const Geo g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int ROWS = NT / 8;                      // tile rows per pass: every wave fills 8 consecutive rows
    constexpr int A_LD = BM / ROWS, B_LD = BN / ROWS;
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1 && ROWS % 16 == 0, "tile/wave layout");
    constexpr int TILE = (BM + BN) * BK;              // floats per stage (unpadded)
    // PRE instantiations keep the pre-activation scale | shift table behind the two stages, in the SAME LDS object: with a second
    // __shared__ array the compiler no longer separates the fills from the fragment reads and waits for the next stage's fill
    // before it reads the current one
    __shared__ __attribute__((aligned(1024))) float smem[2 * TILE + (PRE ? 2 * GLDS_PRE_MAX : 0)];
    float* const pre_tab = smem + 2 * TILE;
    VPHO_STAMP_INIT();

    vpho_conv_desc d = g.d;
    advance_group(d, g, blockIdx.y);
    int per_xcd = gridDim.x >> 3, ntiles = g.ntiles, M_live = g.M;
    if (d.row_map) {
        // pixel-list launch: the live row count is device data.  The grid was sized for every pixel; the live tiles are spread
        // over the 8 XCDs again (contiguous runs of the LIVE tiles), the rest of the grid exits
        M_live = min(*d.row_count, g.M);
        ntiles = (M_live + BM - 1) / BM * g.tiles_n;
        per_xcd = (ntiles + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    }
    const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lb >= ntiles) return;
    const int tile_n = lb % g.tiles_n, tile_m = lb / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps the LDS destinations on the scalar unit
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = wave * 8 + (lane >> 3);                        // tile row filled by this lane (per pass: + ROWS*j)
    const int kq = (lane & 7) ^ ((lrow >> 1) & 7);                  // logical 16-B chunk this lane must FETCH (swizzle)

    // Buffer addressing: 32-bit byte offsets into x / w (the host checks both extents are < 4 GB); a lane whose tap falls outside
    // the image, whose pixel / weight row is a tail, or whose k is beyond K gets offset 0xFFFFFFFF -- out of the buffer's range,
    // for which `buffer_load ... lds` writes zeros (scripts/microbench/buffer_lds_oob.hip checks this on the hardware).
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, 0xFFFFFFF0u, 0x00020000);
    unsigned a_off[A_LD];
    int a_iy0[A_LD], a_ix0[A_LD];
    const int ohw = d.OH * d.OW;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = m0 + lrow + ROWS * j;
        if (m < M_live) {
            if (d.row_map) m = d.row_map[m];
            int n = m / ohw, rem = m - n * ohw;
            int oy = rem / d.OW, ox = rem - oy * d.OW;
            a_iy0[j] = oy * d.stride - d.pad_y;
            a_ix0[j] = ox * d.stride - d.pad_x;
            a_off[j] = (unsigned)(((long long)(n * d.H + a_iy0[j]) * d.W + a_ix0[j]) * d.x_ld * 4);   // mod 2^32: exact once a valid tap is added
        } else {
            a_off[j] = 0; a_iy0[j] = -(1 << 28); a_ix0[j] = 0;
        }
    }
    unsigned b_off[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int n = n0 + lrow + ROWS * j;
        b_ok[j] = n < d.Cout;
        b_off[j] = (unsigned)(((long long)n * g.w_ld + 4 * kq) * 4);
    }
    int ld_c, ld_r, ld_s;
    {
        const int kg = 4 * kq, rs = kg / d.Cin;
        ld_c = kg - rs * d.Cin; ld_r = rs / d.KW; ld_s = rs - ld_r * d.KW;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto fill = [&](int buf) {
        float* As = smem + buf * TILE + wave * 8 * BK;              // wave-uniform destination of this wave's 8 rows
        float* Bs = smem + buf * TILE + BM * BK + wave * 8 * BK;
        const bool kin = ld_r < d.KH;
        const unsigned tap = (unsigned)(((ld_r * d.W + ld_s) * d.x_ld + ld_c) * 4);
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const unsigned iy = (unsigned)(a_iy0[j] + ld_r), ix = (unsigned)(a_ix0[j] + ld_s);
            const unsigned off = (kin && iy < (unsigned)d.H && ix < (unsigned)d.W) ? a_off[j] + tap : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(As + ROWS * j * BK), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const unsigned off = (kin && b_ok[j]) ? b_off[j] : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Bs + ROWS * j * BK), 16, (int)off, 0, 0, 0);
            b_off[j] += BK * 4;
        }
        ld_c += BK;
        while (ld_c >= d.Cin) { ld_c -= d.Cin; if (++ld_s == d.KW) { ld_s = 0; ++ld_r; } }
    };

    // ---- UNI path state: offsets relative to a base shifted by the padding, so that every per-lane offset and the scalar tap
    // offset are non-negative (the hardware adds them as unsigned 32-bit numbers)
    const long long pad_shift = ((long long)d.pad_y * d.W + d.pad_x) * d.x_ld;
    const __amdgpu_buffer_rsrc_t xu = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(d.x) - (uintptr_t)(pad_shift * 4)), 0, 0xFFFFFFF0u, 0x00020000);
    int u_voff[A_LD], u_boff[B_LD];
    int u_r = 0, u_s = 0, u_c = 0;                                  // wave-uniform tap of the stage being filled
    const bool simple = d.KH == 1 && d.KW == 1 && d.pad_y == 0 && d.pad_x == 0;
    const bool UNI = g.uni != 0;
    if (UNI) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const bool live = a_iy0[j] > -(1 << 27);
            u_voff[j] = live ? (int)(a_off[j] + (unsigned)(pad_shift * 4) + 16u * (unsigned)kq) : -1;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) u_boff[j] = b_ok[j] ? (int)b_off[j] : -1;
    }
    // second input (vpho_conv_desc.x2: the projection shortcut merged into a 1x1 convolution): the stages with k >= Cin read it -- a
    // whole 32-k stage lies in one source, so the switch is wave-uniform: another buffer resource, another set of per-lane offsets
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x2 ? d.x2 : d.x), 0, 0xFFFFFFF0u, 0x00020000);
    int u_voff2[A_LD];
    if (d.x2) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int m = m0 + lrow + ROWS * j;
            if (m < M_live) {
                const int n = m / ohw, rem = m - n * ohw;
                const int oy = rem / d.OW, ox = rem - oy * d.OW;
                u_voff2[j] = (int)((((long long)(n * d.H2 + oy * d.stride2) * d.W2 + ox * d.stride2) * d.x2_ld) * 4 + 16 * kq);
            } else u_voff2[j] = -1;
        }
    }
    auto fill_uni = [&](int buf, int kt) {
        float* As = smem + buf * TILE + wave * 8 * BK;
        float* Bs = smem + buf * TILE + BM * BK + wave * 8 * BK;
        const int tap_s = ((u_r * d.W + u_s) * d.x_ld + u_c) * 4;
        if (d.x2 && kt * BK >= d.Cin) {
            const int c2 = (kt * BK - d.Cin) * 4;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int vo = u_voff2[j];       // (a local, like u_boff below: an array element passed straight to the builtin makes this hipcc drop the host stub)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2r, (lds_ptr)(As + ROWS * j * BK), 16, vo, c2, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            int off = u_voff[j];
            if (!simple) {
                const unsigned iy = (unsigned)(a_iy0[j] + u_r), ix = (unsigned)(a_ix0[j] + u_s);
                off = (iy < (unsigned)d.H && ix < (unsigned)d.W) ? off : -1;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xu, (lds_ptr)(As + ROWS * j * BK), 16, off, tap_s, 0, 0);
        }
        }
        const int koff = kt * BK * 4;
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int bo = u_boff[j];          // (a local: passing the array element straight to the builtin makes this hipcc drop the host stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Bs + ROWS * j * BK), 16, bo, koff, 0, 0);
        }
        u_c += BK;
        if (u_c >= d.Cin && !d.x2) { u_c = 0; if (++u_s == d.KW) { u_s = 0; ++u_r; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    const int sw = (li >> 1) & 7;                                   // (row >> 1) & 7 of every fragment row of this lane
    // Pre-activation BN + LeakyReLU of the INPUT (encoding.Residual): the tiles travel global -> LDS untouched, the affine and
    // the activation are applied to the A fragments as they are read (1x1 convolutions: k is the channel, no padding to keep
    // zero).  The table is staged before the first fill: an ordinary load consumed in the loop would drain the fill queue.
    if constexpr (PRE) {
        for (int c = tid; c < d.Cin; c += NT) { pre_tab[c] = d.in_scale[c]; pre_tab[GLDS_PRE_MAX + c] = d.in_shift[c]; }
        __syncthreads();
    }
    // ---- epilogue items of this thread (16-byte epilogue): tile row / channel quad, destination offset, residual.  The residual is
    // REQUESTED in front of the last k stage's matrix work, not after it: a 1x1 expansion with a residual (conv3 of every bottleneck:
    // K = 64 ... 512) has 2 ... 16 stages, and its tile then spent an HBM round trip of the residual tile on top of them with both
    // workgroups of the CU marching in step (44 / 65 / 94 TF/s on the K = 64 / 128 / 256 layers).  Nothing else is in flight
    // during the last stage (no next fill), so the stage-end wait costs nothing extra.
    constexpr int V_PER_ROW = BN / 4, ITERS = BM * V_PER_ROW / NT;
    f32x4 rv[ITERS];                                                // the only epilogue state that lives through the last stage
    auto offsets = [&](int row, long long& yoff, long long& ro) {
        if (d.row_map && d.rows_scatter) row = d.row_map[row];          // pixel list, results stored at the pixels' own positions
        if (g.y_linear && g.r_linear) {
            yoff = (long long)row * d.y_sx;
            ro = (long long)row * d.r_sx;
        } else {
            int n = row / ohw, rem = row - n * ohw;
            int oy = rem / d.OW, ox = rem - oy * d.OW;
            yoff = n * d.y_sn + oy * d.y_sy + ox * d.y_sx;
            ro = n * d.r_sn + oy * d.r_sy + ox * d.r_sx;
        }
    };
    auto item = [&](int it, long long& yoff, long long& ro) -> bool {
        const int idx = tid + it * NT;
        const int r = idx / V_PER_ROW, c4 = idx - r * V_PER_ROW;
        const int row = m0 + r, col = n0 + 4 * c4;
        const bool live = row < M_live && col < d.Cout;
        yoff = ro = 0;
        if (live) { offsets(row, yoff, ro); yoff += col; ro += col; }
        return live;
    };
    auto load_res = [&]() {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            long long yoff, ro;
            const bool live = item(it, yoff, ro);
            rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (d.res && live) rv[it] = *reinterpret_cast<const f32x4*>(d.res + ro);
        }
    };
    // residual = a coarser map, bilinearly up-sampled to this output pixel (the FPN's top-down add): the expression, operand order
    // included, of resize_bilinear_nhwc_kernel (misc.hip), so that conv + fused add == conv, then the accumulate pass, bit for bit
    auto load_res_up = [&]() {
        const float sy = (float)d.ru_H / (float)d.OH, sx = (float)d.ru_W / (float)d.OW;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NT;
            const int r = idx / V_PER_ROW, c4 = idx - r * V_PER_ROW;
            int row = m0 + r;
            const int col = n0 + 4 * c4;
            rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < M_live && col < d.Cout) {
                if (d.row_map) row = d.row_map[row];
                const int n = row / ohw, rem = row - n * ohw;
                const int oy = rem / d.OW, ox = rem - oy * d.OW;
                int y0, y1, x0, x1; float ly, lx;
                lin_src(oy, sy, d.ru_H, y0, y1, ly);
                lin_src(ox, sx, d.ru_W, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float* b = d.res_up + (long long)n * d.ru_H * d.ru_W * d.ru_ld + col;
                const f32x4 a00 = *reinterpret_cast<const f32x4*>(b + ((long long)y0 * d.ru_W + x0) * d.ru_ld);
                const f32x4 a01 = *reinterpret_cast<const f32x4*>(b + ((long long)y0 * d.ru_W + x1) * d.ru_ld);
                const f32x4 a10 = *reinterpret_cast<const f32x4*>(b + ((long long)y1 * d.ru_W + x0) * d.ru_ld);
                const f32x4 a11 = *reinterpret_cast<const f32x4*>(b + ((long long)y1 * d.ru_W + x1) * d.ru_ld);
#pragma unroll
                for (int u = 0; u < 4; ++u) rv[it][u] = hy * (hx * a00[u] + lx * a01[u]) + ly * (hx * a10[u] + lx * a11[u]);
            }
        }
    };
    const bool res_early = g.vec_epilogue && (d.res != nullptr || d.res_up != nullptr) && !(g.dbg & 8);
    // the first TWO stages are requested back to back and only the first is waited for (counted vmcnt: fills complete in issue order):
    // a short-K tile (K = 64: two stages in all) pays one memory round trip at its start, not one and a half
    if (UNI) fill_uni(0, 0); else fill(0);
    VPHO_STAMP_AT(1);
    if (nk > 1 && !(g.dbg & 16)) {
        if (UNI) fill_uni(1, 1); else fill(1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD + B_LD) : "memory");
        __syncthreads();
    } else {
        VPHO_SYNC_LDS_DMA();
    }
    auto compute = [&](int kt) {
        const int buf = kt & 1;
        const float* As = smem + buf * TILE + (wm * (BM / WM) + li) * BK;
        const float* Bs = smem + buf * TILE + BM * BK + (wn * (BN / WN) + li) * BK;
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int ch = ((2 * kk + lh) ^ sw) * 4;
            if (!(CONV_ABLATE & 32) || kk == 0) {                     // timing (32): one fragment read per stage instead of four
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BK + ch);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BK + ch);
            }
            if constexpr (PRE) {
                const int c = kt * BK + (2 * kk + lh) * 4;          // channel of this lane's 4 consecutive k
                const f32x4 sc = *reinterpret_cast<const f32x4*>(pre_tab + c), sh = *reinterpret_cast<const f32x4*>(pre_tab + GLDS_PRE_MAX + c);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = a[i][e] * sc[e] + sh[e];
                        a[i][e] = t > 0.f ? t : t * d.in_slope;
                    }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
    };
    VPHO_STAMP_AT(2);
    VPHO_PRIO_MAIN();
    for (int kt = 0; kt + 1 < nk; ++kt) {
        if ((kt > 0 || (g.dbg & 16)) && !(CONV_ABLATE & 64)) { if (UNI) fill_uni((kt & 1) ^ 1, kt + 1); else fill((kt & 1) ^ 1); }     // stage 1 is already on its way
        compute(kt);
        if (!(CONV_ABLATE & 128)) VPHO_SYNC_LDS_DMA();
    }
    // last stage: no next fill; the residual tile is requested here and lands under this stage's matrix work
    if (res_early) { if (d.res_up) load_res_up(); else load_res(); __builtin_amdgcn_sched_barrier(0); }
    compute(nk - 1);
    VPHO_SYNC_LDS_DMA();
    VPHO_PRIO_REST();
    VPHO_STAMP_AT(3);
    if ((CONV_ABLATE & 256) && acc[0][0][0] != 1.2345e-30f) return;    // timing: no epilogue

    if (g.vec_epilogue) {
        constexpr int C_LD = BN;                                    // ds_write_b32 halves are separate bank groups: no pad needed
        float* Cs = smem;
        static_assert(BM * C_LD <= 2 * TILE, "epilogue tile does not fit the staging LDS");
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    Cs[r * C_LD + wn * (BN / WN) + j * 32 + li] = acc[i][j][e];
                }
        __syncthreads();
        if (!res_early) { if (d.res_up) load_res_up(); else load_res(); }
        // BatchNorm reductions in the epilogue (vpho_conv_desc.stats / bn_x): a thread's items all lie in ONE channel quad (NT % V_PER_ROW == 0),
        // so its share of the column sums stays in two registers quads until the tile is out
        static_assert(NT % V_PER_ROW == 0 && NT >= 2 * BN, "stats layout");
        f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 bn_m = st0, bn_i = st0, bn_g = st0, bn_b = st0;
        if (d.bn_x) {
            const int c = n0 + 4 * (tid % V_PER_ROW);
            if (c < d.Cout) {
                bn_m = *reinterpret_cast<const f32x4*>(d.bn_mean + c); bn_i = *reinterpret_cast<const f32x4*>(d.bn_invstd + c);
                if (!d.gate) { bn_g = *reinterpret_cast<const f32x4*>(d.bn_gamma + c); bn_b = *reinterpret_cast<const f32x4*>(d.bn_beta + c); }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            long long yoff, ro;
            if (!item(it, yoff, ro)) continue;
            const int idx = tid + it * NT;
            const int r = idx / V_PER_ROW, c4 = idx - r * V_PER_ROW;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * C_LD + 4 * c4);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n0 + 4 * c4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = v[k] + bv[k] + rv[it][k]; o[k] = t > 0.f ? t : t * d.out_slope; }
            if (d.gate) {                                            // backward of a LeakyReLU whose output is `gate` (same layout as y)
                const f32x4 gt = *reinterpret_cast<const f32x4*>(d.gate + yoff);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = gt[k] > 0.f ? o[k] : o[k] * d.gate_slope;
            }
            if (d.bn_x) {
                // the gate from the BatchNorm INPUT: t = the forward pass's own expression (bn_apply_kernel, train_score.hip: the same bits,
                // hence the same sign as the stored activation); sums of dy and dy * xhat for d beta / d gamma
                const f32x4 xv = *reinterpret_cast<const f32x4*>(d.bn_x + yoff);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - bn_m[k]) * bn_i[k];
                    if (!d.gate) {                                   // (with a stored gate -- residual blocks -- the sign came from it above)
                        const float t = xh * bn_g[k] + bn_b[k];
                        o[k] = t > 0.f ? o[k] : o[k] * d.gate_slope;
                    }
                    st0[k] += o[k]; st1[k] += o[k] * xh;
                }
            } else if (d.stats) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { st0[k] += o[k]; st1[k] += o[k] * o[k]; }
            }
            *reinterpret_cast<f32x4*>(d.y + yoff) = o;
        }
        if (d.stats) {
            // a wave's lanes with the same channel quad (lane bits >= log2 V_PER_ROW) are added on the vector ALU, the waves' sums in a fixed
            // order through LDS; one partial row per M-tile: [tile_m][2][Cout]
            static_assert(V_PER_ROW == 16 || V_PER_ROW == 32, "lane layout of the epilogue items");
            constexpr int NWAVES = NT / 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (V_PER_ROW == 16) { st0[k] = sum_lane_bit_4(st0[k]); st1[k] = sum_lane_bit_4(st1[k]); }
                st0[k] = sum_lane_bit_5(st0[k]); st1[k] = sum_lane_bit_5(st1[k]);
            }
            __syncthreads();                                        // every thread has read its items of Cs
            float* R = smem;                                        // [wave][2][BN]
            if (lane < V_PER_ROW) {
                *reinterpret_cast<f32x4*>(R + (wave * 2 + 0) * BN + 4 * lane) = st0;
                *reinterpret_cast<f32x4*>(R + (wave * 2 + 1) * BN + 4 * lane) = st1;
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int pl = tid / BN, c = tid - pl * BN;
                float t = R[pl * BN + c];
#pragma unroll
                for (int k = 1; k < NWAVES; ++k) t += R[(k * 2 + pl) * BN + c];
                if (n0 + c < d.Cout) d.stats[((long long)tile_m * 2 + pl) * d.Cout + n0 + c] = t;
            }
        }
        VPHO_STAMP_AT(4);
        VPHO_STAMP_WRITE(conv, blockIdx.x);
        return;
    }
    const int row_base = m0 + wm * (BM / WM) + 4 * lh, col_base = n0 + wn * (BN / WN) + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + j * 32;
        if (col >= d.Cout) continue;
        const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_base + i * 32 + (e & 3) + 8 * (e >> 2);
                if (row >= M_live) continue;
                long long yoff, ro;
                offsets(row, yoff, ro);
                float v = acc[i][j][e] + bv;
                if (d.res) v += d.res[ro + col];
                v = v > 0.f ? v : v * d.out_slope;
                if (d.gate) v = d.gate[yoff + col] > 0.f ? v : v * d.gate_slope;
                d.y[yoff + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent multi-tile variant of the direct-to-LDS kernel for 1x1 convolutions whose launch has MORE tiles than the chip has workgroup
// slots (round 5).  The in-kernel stamps of profiles/r05_inkernel_clock.txt say where a short-K tile's life goes: 128 -> 512 on 32 x 32
// maps (4 k stages): entry -> first fills requested 2.1 us, first fill wait 2.0, main loop 16.1, epilogue 5.9, slot turnover 1.1 -- and for
// 45 % of the launch a CU has ONE workgroup in its main loop, for 22 % none.  Here a workgroup walks its share of the tiles and
//   * requests the NEXT tile's first two k stages right behind the current tile's last stage, so they land under the epilogue
//     (both stage buffers are free then: the epilogue no longer stages through them);
//   * stages its accumulators through a WAVE-PRIVATE 1 KB LDS slice (8 rows x 32 columns per pass, LDS operations of one wave execute
//     in order): no workgroup barrier in the epilogue, whole 128-byte lines on 16-byte accesses; the bias of a tile arrives in LDS with
//     the tile's first stage (one more LDS-DMA per wave), so the epilogue consumes no register-returning load that was issued behind
//     the next tile's fills (its wait would wait for them as well);
//   * stores / residual loads go through buffer resources with an out-of-range offset for dead items: every wave issues EXACTLY NIT
//     stores per tile, so the counted wait for the next tile's first stage (vmcnt = the younger operations) is exact.
// The k order of every output element is that of conv_igemm_glds_kernel: bit-identical results (tests/test_gpu_conv.py).
// Served: Cin % 32 == 0, 1x1, unpadded, 16-byte epilogue, optional bias / residual / second input (x2); no pixel list, no up-sampled
// residual, no gate, no splits, no prologue -- everything else stays on conv_igemm_glds_kernel.
#define VPHO_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
template <int BM, int BN, int WM, int WN, bool STATS>
__device__ __forceinline__ void conv_pers_body(const Geo& g) {
    constexpr int NT = 64 * WM * WN, NW = WM * WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int ROWS = NT / 8;
    constexpr int A_LD = BM / ROWS, B_LD = BN / ROWS, NF = A_LD + B_LD;      // LDS-DMA instructions per stage and wave
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1 && ROWS % 16 == 0 && BN / WN <= 64, "tile/wave layout");
    constexpr int TILE = (BM + BN) * BK;
    constexpr int EPI = 8 * 32;                       // floats of a wave's staging slice: 8 rows x 32 columns (one pass)
    constexpr int NIT = TM * 4 * TN;                  // passes = epilogue items (= stores) per lane and tile
    // [2] stages | [NW] staging slices | [2][BN] bias of the current / the next tile
    __shared__ __attribute__((aligned(1024))) float smem[2 * TILE + NW * EPI + 2 * BN];
    VPHO_STAMP_INIT();

    vpho_conv_desc d = g.d;
    advance_group(d, g, blockIdx.y);                  // grouped launches: every group is its own persistent walk over gridDim.x slots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = wave * 8 + (lane >> 3);
    const int kq = (lane & 7) ^ ((lrow >> 1) & 7);
    const int ohw = d.OH * d.OW;
    const __amdgpu_buffer_rsrc_t xu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x2 ? d.x2 : d.x), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(d.y, 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.res ? d.res : d.y), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : d.w), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- tile schedule: XCD x (hardware workgroup b runs on XCD b % 8) owns the contiguous run [x * per_xcd, (x + 1) * per_xcd) of the
    // logical tiles (consecutive tiles = the N-tiles of one M-tile: its A rows stay in that XCD's L2); slot s of the XCD's gridDim.x / 8
    // workgroups takes tiles s, s + nslot, s + 2 nslot, ... of the run
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3, per_xcd = (g.ntiles + 7) >> 3;
    int lt = blockIdx.x >> 3;
    auto tile_at = [&](int t, int& m0, int& n0) -> bool {
        const int lb = xcd * per_xcd + t;
        if (t >= per_xcd || lb >= g.ntiles) return false;
        const int tile_m = lb / g.tiles_n, tile_n = lb - tile_m * g.tiles_n;
        m0 = tile_m * BM; n0 = tile_n * BN;
        return true;
    };
    int m0, n0;
    if (!tile_at(lt, m0, n0)) return;

    int u_voff[A_LD], u_boff[B_LD], u_voff2[A_LD], bias_off;
    auto setup = [&](int m0_, int n0_) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int m = m0_ + lrow + ROWS * j;
            u_voff[j] = -1; u_voff2[j] = -1;
            if (m < g.M) {
                const int n = m / ohw, rem = m - n * ohw;
                const int oy = rem / d.OW, ox = rem - oy * d.OW;
                u_voff[j] = (int)((((long long)(n * d.H + oy * d.stride) * d.W + ox * d.stride) * d.x_ld) * 4 + 16 * kq);
                if (d.x2) u_voff2[j] = (int)((((long long)(n * d.H2 + oy * d.stride2) * d.W2 + ox * d.stride2) * d.x2_ld) * 4 + 16 * kq);
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int n = n0_ + lrow + ROWS * j;
            u_boff[j] = n < d.Cout ? (int)(((long long)n * g.w_ld + 4 * kq) * 4) : -1;
        }
        const int bc = n0_ + wn * (BN / WN) + lane;                // this lane's bias element of the wave's BN / WN columns
        bias_off = (d.bias && lane < BN / WN && bc < d.Cout) ? bc * 4 : -1;
    };
    // stage fill; stage 0 of a tile also brings the tile's bias (one dword per lane, straight into LDS: out-of-range = zeros) --
    // NF + 1 operations; the waves of one column block write the same values to the same slots
    auto fill = [&](int buf, int kt, int tp) {
        float* As = smem + buf * TILE + wave * 8 * BK;
        float* Bs = smem + buf * TILE + BM * BK + wave * 8 * BK;
        if (d.x2 && kt * BK >= d.Cin) {
            const int c2 = (kt * BK - d.Cin) * 4;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int vo = u_voff2[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2r, (lds_ptr)(As + ROWS * j * BK), 16, vo, c2, 0, 0);
            }
        } else {
            const int c1 = kt * BK * 4;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int vo = u_voff[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xu, (lds_ptr)(As + ROWS * j * BK), 16, vo, c1, 0, 0);
            }
        }
        const int koff = kt * BK * 4;
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int bo = u_boff[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Bs + ROWS * j * BK), 16, bo, koff, 0, 0);
        }
        if (kt == 0) {
            const int bo = bias_off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (lds_ptr)(smem + 2 * TILE + NW * EPI + tp * BN + wn * (BN / WN)), 4, bo, 0, 0, 0);
        }
    };

    const int nk = (g.K + BK - 1) / BK;
    const int sw = (li >> 1) & 7;
    f32x16 acc[TM][TN];
    auto compute = [&](int kt) {
        const int buf = kt & 1;
        const float* As = smem + buf * TILE + (wm * (BM / WM) + li) * BK;
        const float* Bs = smem + buf * TILE + BM * BK + (wn * (BN / WN) + li) * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int ch = ((2 * kk + lh) ^ sw) * 4;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BK + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BK + ch);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        }
    };

    // ---- epilogue items of this lane: pass (i, gq, j) moves rows i * 32 + 8 gq .. + 7, columns 32 j .. + 31 of the wave's block; the
    // lane's 16-byte piece of a pass = row lane / 8, columns 4 (lane % 8) .. + 3
    float* Cw = smem + 2 * TILE + wave * EPI;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    auto item_off = [&](int m0_, int n0_, int i, int gq, int j, int& yo, int& ro) {
        const int row = m0_ + wm * (BM / WM) + i * 32 + 8 * gq + er, col = n0_ + wn * (BN / WN) + j * 32 + ec;
        yo = ro = -1;                                               // out of the buffer's range: loads return zeros, stores are dropped
        if (row < g.M && col < d.Cout) {
            long long y_, r_;
            if (g.y_linear && g.r_linear) { y_ = (long long)row * d.y_sx; r_ = (long long)row * d.r_sx; }
            else {
                const int n = row / ohw, rem = row - n * ohw;
                const int oy = rem / d.OW, ox = rem - oy * d.OW;
                y_ = n * d.y_sn + oy * d.y_sy + ox * d.y_sx; r_ = n * d.r_sn + oy * d.r_sy + ox * d.r_sx;
            }
            yo = (int)((y_ + col) * 4);
            if (d.res) ro = (int)((r_ + col) * 4);
        }
    };

    // ---- BatchNorm reductions (vpho_conv_desc.stats, forward form): a wave leaves the column sums of its 32 x 64 block in the first 128
    // floats of its staging slice at the end of the epilogue; the four row blocks of a column are added -- fixed order -- behind the NEXT
    // tile's first barrier (or the closing one), so the epilogue stays free of workgroup barriers.  The slice is not written again before
    // the stage-end barrier of that next tile.  (The store is issued behind the counted wait and is older than everything the next one counts.)
    // (STATS is a template parameter, conv_igemm_pers_bn_kernel: as a run-time switch its 16 registers cost the inference kernel 7 spills)
    constexpr bool stats_on = STATS;
    static_assert(2 * (BN / WN) <= EPI && 2 * BN <= NT, "stats layout");
    auto flush_stats = [&](int m0_, int n0_) {
        if (tid < 2 * BN) {
            const int pl = tid / BN, c = tid - pl * BN;
            const int wn_c = c / (BN / WN), cc = c - wn_c * (BN / WN);
            const float* q = smem + 2 * TILE + wn_c * EPI + pl * (BN / WN) + cc;
            float t = q[0];
#pragma unroll
            for (int w = 1; w < WM; ++w) t += q[w * WN * EPI];
            if (n0_ + c < d.Cout) d.stats[((long long)(m0_ / BM) * 2 + pl) * d.Cout + n0_ + c] = t;
        }
    };
    int pm0 = 0, pn0 = 0;                                           // the tile whose sums wait in the slices

    int tp = 0;                                                     // bias slot of the current tile
    setup(m0, n0);
    fill(0, 0, tp);
    if (nk > 1) fill(1, 1, tp);
    VPHO_STAMP_AT(1);
    bool first = true;
    while (true) {
        // ---- the tile's first stage (and its bias) has landed?  Operations complete in issue order; younger than stage 0 are stage 1
        // (NF) and, from the second tile on, the previous tile's NIT stores
        if (nk > 1) { if (first) VPHO_WAIT_VM(NF); else VPHO_WAIT_VM(NF + NIT); }
        else        { if (first) VPHO_WAIT_VM(0);  else VPHO_WAIT_VM(NIT); }
        VPHO_BARRIER_LDS_ONLY();                                    // (not __syncthreads(): its fence would wait for the previous tile's stores)
        if (stats_on && !first) flush_stats(pm0, pn0);
        if (first) VPHO_STAMP_AT(2);
        VPHO_PRIO_MAIN();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            if (kt > 0) fill((kt & 1) ^ 1, kt + 1, tp);             // stage 1 is already on its way
            compute(kt);
            VPHO_SYNC_LDS_DMA();
        }
        // ---- last stage: the residual tile is requested here and lands under this stage's matrix work
        f32x4 rv[NIT];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    int yo, ro;
                    item_off(m0, n0, i, gq, j, yo, ro);
                    rv[(i * 4 + gq) * TN + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 0, 0));
                }
        __builtin_amdgcn_sched_barrier(0);
        compute(nk - 1);
        VPHO_SYNC_LDS_DMA();                                        // every wave is done with both stage buffers; the residual has landed
        // (the compiler does not see the wait inside the macro: a use of the residual registers HERE makes it place its own wait for them
        // in front of the next tile's fills instead of in the epilogue, where it would wait for those fills too)
#pragma unroll
        for (int k = 0; k < NIT; ++k) asm volatile("" :: "v"(rv[k]));
        VPHO_PRIO_REST();
        if (first) VPHO_STAMP_AT(3);
        // ---- next tile: its first two stages are requested BEFORE this tile's epilogue
        const int em0 = m0, en0 = n0, etp = tp;
        lt += nslot;
        const bool more = tile_at(lt, m0, n0);
        if (more) {
            tp ^= 1;
            setup(m0, n0);
            fill(0, 0, tp);
            if (nk > 1) fill(1, 1, tp);
        }
        // ---- epilogue through the wave's own LDS slice: no workgroup barrier (the LDS operations of one wave execute in order)
        const float* Bq = smem + 2 * TILE + NW * EPI + etp * BN + wn * (BN / WN) + ec;
        f32x4 st0[TN], st1[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { st0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; st1[j] = st0[j]; }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) Cw[(e4 + 4 * lh) * 32 + li] = acc[i][j][4 * gq + e4];
                    int yo, ro;
                    item_off(em0, en0, i, gq, j, yo, ro);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Cw + er * 32 + ec);
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(Bq + j * 32);
                    const f32x4 r = rv[(i * 4 + gq) * TN + j];
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float t = v[k] + bq[k] + r[k]; o[k] = t > 0.f ? t : t * d.out_slope; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), yr, yo, 0, 0);
                    if (stats_on) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const float u = yo != -1 ? o[k] : 0.f; st0[j][k] += u; st1[j][k] += u * u; }     // dead rows / columns: nothing
                    }
                }
        if (stats_on) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { st0[j][k] = sum_lane_bits_345(st0[j][k]); st1[j][k] = sum_lane_bits_345(st1[j][k]); }
                if (lane < 8) {                                     // behind the last pass's read of the slice (one wave's LDS operations execute in order)
                    *reinterpret_cast<f32x4*>(Cw + j * 32 + ec) = st0[j];
                    *reinterpret_cast<f32x4*>(Cw + (BN / WN) + j * 32 + ec) = st1[j];
                }
            }
            pm0 = em0; pn0 = en0;
        }
        if (first) { VPHO_STAMP_AT(4); }
        first = false;
        if (!more) break;
    }
    if (stats_on) { VPHO_BARRIER_LDS_ONLY(); flush_stats(pm0, pn0); }
    VPHO_STAMP_WRITE(conv, blockIdx.x);
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 4) void conv_igemm_pers_kernel(const Geo g) { conv_pers_body<BM, BN, WM, WN, false>(g); }
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 4) void conv_igemm_pers_bn_kernel(const Geo g) { conv_pers_body<BM, BN, WM, WN, true>(g); }

// ---------------------------------------------------------------------------------------------------------------------
// Opt-in split-bf16 variant of the direct-to-LDS kernel (vpho_conv_desc.w_planes / plane_terms; VPHO_CONV_MFMA=bf16x6|bf16x9 in the
// Python plan; NOT the default -- see csrc/score_ode.hip::head_tile_split for the arithmetic and tests/test_gpu_split_head.py /
// test_gpu_conv.py for the error study).  Weights arrive as three bf16 planes [3][Cout][K] that sum to the fp32 weights exactly;
// activations stay fp32 in HBM and LDS and are split into their three bf16 pieces as the fragments are read.  Stage = 16 k
// (one v_mfma_f32_32x32x16_bf16 step): A rows of 64 B, B rows of 32 B per plane; register double buffer: the fragments of stage
// kt+1 are read and split in the shadow of the MFMAs of stage kt.  Shapes: Cin % 16 == 0 (every stage inside one filter tap), no
// prologue affine; everything else takes the fp32 kernels.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
constexpr int SBK = 16;

__device__ __forceinline__ void conv_split3(const f32x4& x0, const f32x4& x1, cbf16x8& h, cbf16x8& m, cbf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        const __bf16 hb = (__bf16)x;
        const float r1 = x - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;
        h[e] = hb; m[e] = mb; l[e] = (__bf16)r2;
    }
}

template <int BM, int BN, int WM, int WN, int TERMS>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_split_kernel(const Geo g) {
    constexpr int NT = 64 * WM * WN, NW = WM * WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_ROWS = NW * 16;                   // A tile rows per pass (a wave instruction = 16 rows of 64 B)
    constexpr int A_LD = BM / A_ROWS;
    constexpr int B_ROWS = NW * 32;                   // plane rows per pass (a wave instruction = 32 rows of 32 B)
    constexpr int B_PASSES = (3 * BN + B_ROWS - 1) / B_ROWS;
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && BM % A_ROWS == 0, "tile/wave layout");
    constexpr int A_FLOATS = BM * SBK, PLANE_FLOATS = BN * SBK / 2;
    constexpr int TILE = A_FLOATS + 3 * PLANE_FLOATS;                 // floats per stage
    constexpr int EPI_FLOATS = BM * BN;                               // the epilogue stages the accumulator tile through LDS
    __shared__ __attribute__((aligned(1024))) float smem[(2 * TILE > EPI_FLOATS) ? 2 * TILE : EPI_FLOATS];

    vpho_conv_desc d = g.d;
    int per_xcd = gridDim.x >> 3, ntiles = g.ntiles, M_live = g.M;
    if (d.row_map) {
        M_live = min(*d.row_count, g.M);
        ntiles = (M_live + BM - 1) / BM * g.tiles_n;
        per_xcd = (ntiles + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    }
    const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lb >= ntiles) return;
    const int tile_n = lb % g.tiles_n, tile_m = lb / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ohw = d.OH * d.OW;

    // ---- A fill: lane -> (row = wave*16 + lane/4 [+ A_ROWS*j], physical 16-B chunk lane%4), fetching logical chunk ^ (row>>2)&3
    const long long pad_shift = ((long long)d.pad_y * d.W + d.pad_x) * d.x_ld;
    const __amdgpu_buffer_rsrc_t xu = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(d.x) - (uintptr_t)(pad_shift * 4)), 0, 0xFFFFFFF0u, 0x00020000);
    const int arow = wave * 16 + (lane >> 2);
    int a_iy0[A_LD], a_ix0[A_LD], u_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int lr = arow + A_ROWS * j;
        int m = m0 + lr;
        const int kq = (lane & 3) ^ ((lr >> 2) & 3);
        if (m < M_live) {
            if (d.row_map) m = d.row_map[m];
            const int n = m / ohw, rem = m - n * ohw;
            const int oy = rem / d.OW, ox = rem - oy * d.OW;
            a_iy0[j] = oy * d.stride - d.pad_y;
            a_ix0[j] = ox * d.stride - d.pad_x;
            const unsigned off = (unsigned)(((long long)(n * d.H + a_iy0[j]) * d.W + a_ix0[j]) * d.x_ld * 4);
            u_voff[j] = (int)(off + (unsigned)(pad_shift * 4) + 16u * (unsigned)kq);
        } else {
            a_iy0[j] = -(1 << 28); a_ix0[j] = 0; u_voff[j] = -1;
        }
    }
    // ---- B fill: flat row index over the three planes: fr = wave*32 + lane/2 + B_ROWS*pass -> (plane fr / BN, row fr % BN)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w_planes), 0, 0xFFFFFFF0u, 0x00020000);
    int b_voff[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
        const int fr = wave * 32 + (lane >> 1) + B_ROWS * ps;
        const int plane = fr / BN, r = fr - plane * BN, n = n0 + r;
        const int ch = (lane & 1) ^ ((r >> 3) & 1);
        b_voff[ps] = (plane < 3 && n < d.Cout) ? (int)((((long long)plane * d.Cout + n) * g.w_ld + ch * 8) * 2) : -1;
    }
    const bool simple = d.KH == 1 && d.KW == 1 && d.pad_y == 0 && d.pad_x == 0;
    int u_r = 0, u_s = 0, u_c = 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto fill = [&](int buf, int kt) {
        float* As = smem + buf * TILE + wave * 16 * SBK;
        float* Bs = smem + buf * TILE + A_FLOATS + wave * 32 * (SBK / 2);
        const int tap_s = ((u_r * d.W + u_s) * d.x_ld + u_c) * 4;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            int off = u_voff[j];
            if (!simple) {
                const unsigned iy = (unsigned)(a_iy0[j] + u_r), ix = (unsigned)(a_ix0[j] + u_s);
                off = (iy < (unsigned)d.H && ix < (unsigned)d.W) ? off : -1;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xu, (lds_ptr)(As + A_ROWS * j * SBK), 16, off, tap_s, 0, 0);
        }
        const int koff = kt * SBK * 2;
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            if (wave * 32 + B_ROWS * ps < 3 * BN) {          // wave-uniform: this wave's 32 rows of the pass exist
                const int bo = b_voff[ps];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Bs + B_ROWS * ps * (SBK / 2)), 16, bo, koff, 0, 0);
            }
        }
        u_c += SBK;
        if (u_c >= d.Cin) { u_c = 0; if (++u_s == d.KW) { u_s = 0; ++u_r; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frag { cbf16x8 ah[TM], am[TM], al[TM]; cbf16x8 b[TN][3]; };
    const int a_sw = (li >> 2) & 3, b_sw = (li >> 3) & 1;
    auto read_frags = [&](int buf, Frag& f) {
        const float* As = smem + buf * TILE + (wm * (BM / WM) + li) * SBK;
        const __bf16* Bp = reinterpret_cast<const __bf16*>(smem + buf * TILE + A_FLOATS) + (wn * (BN / WN) + li) * SBK + ((lh ^ b_sw) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) f.b[j][p] = *reinterpret_cast<const cbf16x8*>(Bp + (p * BN + j * 32) * SBK);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(As + i * 32 * SBK + ((2 * lh) ^ a_sw) * 4);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(As + i * 32 * SBK + ((2 * lh + 1) ^ a_sw) * 4);
            conv_split3(x0, x1, f.ah[i], f.am[i], f.al[i]);
        }
    };
    auto products = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (TERMS == 9) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am[i], f.b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.b[j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.b[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am[i], f.b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am[i], f.b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.b[j][0], acc[i][j], 0, 0, 0);
            }
    };

    const int nk = g.K / SBK;                                          // the host guarantees K % 16 == 0
    Frag fr[2];
    fill(0, 0);
    if (nk > 1) fill(1, 1);
    VPHO_SYNC_LDS_DMA();
    read_frags(0, fr[0]);
    for (int kt = 0; kt < nk; kt += 2) {
        VPHO_SYNC_LDS_DMA();                                           // stage kt is in registers everywhere; fill(kt+1) has landed
        if (kt + 2 < nk) fill(0, kt + 2);
        if (kt + 1 < nk) read_frags(1, fr[1]);
        products(fr[0]);
        if (kt + 1 < nk) {
            VPHO_SYNC_LDS_DMA();
            if (kt + 3 < nk) fill(1, kt + 3);
            if (kt + 2 < nk) read_frags(0, fr[0]);
            products(fr[1]);
        }
    }
    __syncthreads();

    // ---- epilogue (same as conv_igemm_glds_kernel): accumulator tile through LDS, 16-byte rows out
    auto offsets = [&](int row, long long& yo, long long& ro) {
        if (d.row_map && d.rows_scatter) row = d.row_map[row];
        if (g.y_linear && g.r_linear) {
            yo = (long long)row * d.y_sx;
            ro = (long long)row * d.r_sx;
        } else {
            int n = row / ohw, rem = row - n * ohw;
            int oy = rem / d.OW, ox = rem - oy * d.OW;
            yo = n * d.y_sn + oy * d.y_sy + ox * d.y_sx;
            ro = n * d.r_sn + oy * d.r_sy + ox * d.r_sx;
        }
    };
    constexpr int C_LD = BN;
    float* Cs = smem;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                Cs[r * C_LD + wn * (BN / WN) + j * 32 + li] = acc[i][j][e];
            }
    __syncthreads();
    constexpr int V_PER_ROW = BN / 4, ITERS = BM * V_PER_ROW / NT;
    f32x4 v[ITERS], rv[ITERS];
    long long yo[ITERS];
    bool ok[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        const int r = idx / V_PER_ROW, c4 = idx - r * V_PER_ROW;
        const int row = m0 + r, col = n0 + 4 * c4;
        ok[it] = row < M_live && col < d.Cout;
        v[it] = *reinterpret_cast<const f32x4*>(Cs + r * C_LD + 4 * c4);
        long long ro = 0;
        yo[it] = 0;
        if (ok[it]) { offsets(row, yo[it], ro); yo[it] += col; ro += col; }
        rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.res && ok[it]) rv[it] = *reinterpret_cast<const f32x4*>(d.res + ro);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if (!ok[it]) continue;
        const int idx = tid + it * NT;
        const int c4 = idx % V_PER_ROW;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n0 + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = v[it][k] + bv[k] + rv[it][k]; o[k] = t > 0.f ? t : t * d.out_slope; }
        *reinterpret_cast<f32x4*>(d.y + yo[it]) = o;
    }
}

}  // namespace

extern "C" int vpho_conv2d_nhwc_f32(const vpho_conv_desc* dp, void* stream) {
    VPHO_REQUIRE(dp != nullptr, "vpho_conv2d_nhwc_f32: null descriptor");
    const vpho_conv_desc& d = *dp;
    VPHO_REQUIRE(d.x && d.w && d.y, "vpho_conv2d_nhwc_f32: null tensor");
    VPHO_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.KH > 0 && d.KW > 0 && d.stride > 0,
                 "vpho_conv2d_nhwc_f32: non-positive dimension");
    VPHO_REQUIRE(d.Cin % 4 == 0 && d.x_ld % 4 == 0 && d.x_ld >= d.Cin, "vpho_conv2d_nhwc_f32: Cin=%d x_ld=%d must be multiples of 4, x_ld>=Cin", d.Cin, d.x_ld);
    VPHO_REQUIRE(((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.w & 15) == 0, "vpho_conv2d_nhwc_f32: x/w must be 16-byte aligned");
    VPHO_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "vpho_conv2d_nhwc_f32: in_scale/in_shift must come together");
    VPHO_REQUIRE(d.in_scale == nullptr || (((uintptr_t)d.in_scale & 15) == 0 && ((uintptr_t)d.in_shift & 15) == 0), "vpho_conv2d_nhwc_f32: in_scale/in_shift alignment");
    VPHO_REQUIRE(d.OH > 0 && d.OW > 0, "vpho_conv2d_nhwc_f32: empty output");
    // the last tap of the last output pixel may not start beyond the padded input by more than the pad
    VPHO_REQUIRE((d.OH - 1) * d.stride - d.pad_y < d.H && (d.OW - 1) * d.stride - d.pad_x < d.W, "vpho_conv2d_nhwc_f32: output larger than input allows");
    long long M = (long long)d.N * d.OH * d.OW;
    VPHO_REQUIRE(M < (1ll << 31), "vpho_conv2d_nhwc_f32: too many output pixels");
    Geo g;
    g.d = d;
    g.M = (int)M;
    g.K = d.KH * d.KW * d.Cin + (d.x2 ? d.Cin2 : 0);
    const int splits = d.splits > 1 ? d.splits : 1;
    g.w_ld = d.w_ld > 0 ? d.w_ld : g.K;
    g.x_zs = splits > 1 ? d.x_split : 0; g.w_zs = splits > 1 ? d.w_split : 0; g.y_zs = splits > 1 ? d.y_split : 0;
    g.b_zs = g.r_zs = g.x2_zs = g.pre_zs = g.ru_zs = 0;
    // grouped launch (ABI 11): blockIdx.y = group; the twin hand / object branches of the feature path (same shapes, different weights,
    // backbone_FPN_HFL.py:79-109) as ONE launch -- twice the tiles, half the launches, every output's k order unchanged
    const int groups = d.groups > 1 ? d.groups : 1;
    if (groups > 1) {
        VPHO_REQUIRE(splits == 1 && !d.row_map && !d.gate && !d.w_planes, "vpho_conv2d_nhwc_f32: grouped launches take no splits / pixel list / gate / bf16 planes");
        VPHO_REQUIRE(d.x_group >= 0 && d.w_group > 0 && d.y_group > 0 && d.x_group % 4 == 0 && d.w_group % 4 == 0 && d.y_group % 4 == 0 && d.bias_group % 4 == 0 &&
                     d.res_group % 4 == 0 && d.x2_group % 4 == 0 && d.pre_group % 4 == 0 && d.ru_group % 4 == 0,
                     "vpho_conv2d_nhwc_f32: group strides must be non-negative multiples of 4 floats (x_group may be 0: a shared input)");
        g.x_zs = d.x_group; g.w_zs = d.w_group; g.y_zs = d.y_group; g.b_zs = d.bias_group; g.r_zs = d.res_group; g.x2_zs = d.x2_group;
        g.pre_zs = d.pre_group; g.ru_zs = d.ru_group;
    }
    const int ny = splits * groups;                               // grid.y (at most one of the two is > 1)
    VPHO_REQUIRE(g.w_ld >= g.K && g.w_ld % 4 == 0 && g.x_zs % 4 == 0 && g.w_zs % 4 == 0, "vpho_conv2d_nhwc_f32: w_ld / split strides must be multiples of 4, w_ld >= K");
    VPHO_REQUIRE(splits == 1 || (!d.res && !d.bias && !d.in_scale && !d.gate), "vpho_conv2d_nhwc_f32: split launches produce plain partial sums (no bias / residual / prologue / gate)");
    g.y_linear = (d.y_sy == d.y_sx * d.OW && d.y_sn == d.y_sy * d.OH) ? 1 : 0;
    g.r_linear = (d.res == nullptr) || (d.r_sy == d.r_sx * d.OW && d.r_sn == d.r_sy * d.OH) ? 1 : 0;
    VPHO_REQUIRE((d.row_map == nullptr) == (d.row_count == nullptr), "vpho_conv2d_nhwc_f32: row_map / row_count must come together");
    if (d.row_map) {
        VPHO_REQUIRE(splits == 1 && !d.gate, "vpho_conv2d_nhwc_f32: pixel-list launches take no splits / gate");
        if (!d.rows_scatter) g.y_linear = g.r_linear = 1;   // compact output: row r of y (and res) is the r-th listed pixel
    }
    // 16-byte epilogue when every output pixel's channel run (and the residual's) is 16-byte addressable
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    g.vec_epilogue = (d.Cout % 4 == 0 && al16(d.y) && d.y_sx % 4 == 0 && d.y_sy % 4 == 0 && d.y_sn % 4 == 0 &&
                      (!d.bias || al16(d.bias)) &&
                      (!d.gate || al16(d.gate)) &&
                      (!d.res || (al16(d.res) && d.r_sx % 4 == 0 && d.r_sy % 4 == 0 && d.r_sn % 4 == 0))) ? 1 : 0;
    // (round 5: the direct-to-LDS kernel's own ablations on the one-round layers, scripts/conv_oneround.py, 3.36 ms per step over nine shapes: stage
    // fills 6 %, stage barrier 1.5 %, epilogue 8.6 %, matrix instructions + launch shape the rest; 4-byte stores straight from the accumulators
    // instead of this 16-byte epilogue: +2.5 %)
    const char* dbg_env = getenv("VPHO_CONV_DBG");
    g.dbg = dbg_env ? (atoi(dbg_env) & (8 | 16)) : 0;          // only the bit-identical A/B orders exist at run time
    if (d.x2) {
        // second input concatenated along the channels (projection shortcut merged into conv3): direct-to-LDS kernels, wave-uniform taps
        VPHO_REQUIRE(d.KH == 1 && d.KW == 1 && d.pad_y == 0 && d.pad_x == 0 && d.Cin % BK == 0 && d.Cin2 > 0 && d.Cin2 % BK == 0 && d.x2_ld % 4 == 0 &&
                     d.x2_ld >= d.Cin2 && d.stride2 > 0 && al16(d.x2) && !d.row_map && !d.in_scale && splits == 1 && !d.w_planes,
                     "vpho_conv2d_nhwc_f32: x2 needs a 1x1 unpadded convolution, Cin and Cin2 multiples of 32, no pixel list / prologue / splits / planes");
        VPHO_REQUIRE((d.OH - 1) * d.stride2 < d.H2 && (d.OW - 1) * d.stride2 < d.W2 && 4.0 * d.N * d.H2 * d.W2 * (double)d.x2_ld < 3.9e9,
                     "vpho_conv2d_nhwc_f32: x2 of %d x %d pixels read at stride %d does not cover the %d x %d output (or exceeds 3.9 GB)", d.H2, d.W2, d.stride2, d.OH, d.OW);
    }
    if (d.res_up) {
        // up-sampled residual (the FPN's top-down add): served by the direct-to-LDS kernels' 16-byte epilogue only
        VPHO_REQUIRE(!d.res && !d.gate && splits == 1 && d.ru_H > 0 && d.ru_W > 0 && d.ru_ld >= d.Cout && d.ru_ld % 4 == 0 && al16(d.res_up),
                     "vpho_conv2d_nhwc_f32: res_up needs no res / gate / splits, a (N, ru_H, ru_W, ru_ld >= Cout) map with ru_ld %% 4 == 0, 16-byte aligned");
        VPHO_REQUIRE(g.vec_epilogue && d.in_scale == nullptr && !d.w_planes && (!d.row_map || d.rows_scatter),
                     "vpho_conv2d_nhwc_f32: res_up needs 16-byte addressable output rows, no prologue affine, no split-bf16 planes, scattered pixel lists");
        VPHO_REQUIRE(4.0 * d.N * d.ru_H * d.ru_W * (double)d.ru_ld < 3.9e9, "vpho_conv2d_nhwc_f32: res_up map too large");
    }
    hipStream_t s = (hipStream_t)stream;
    // BatchNorm reductions in the epilogue (ABI 12): the direct-to-LDS kernels' 16-byte epilogue, one partial row per M-tile
    if (d.stats_rows) *d.stats_rows = 0;
    const bool stats_shape = g.vec_epilogue && splits == 1 && groups == 1 && !d.row_map && !d.w_planes && (d.in_scale == nullptr);
    if (d.bn_x) {
        VPHO_REQUIRE(d.stats && d.stats_rows && d.bn_mean && d.bn_invstd && (d.gate || (d.bn_gamma && d.bn_beta)),
                     "vpho_conv2d_nhwc_f32: bn_x needs stats, stats_rows, mean / invstd and -- without a stored gate -- gamma / beta");
        VPHO_REQUIRE(stats_shape && al16(d.bn_x) && al16(d.bn_mean) && al16(d.bn_invstd) && (d.gate || (al16(d.bn_gamma) && al16(d.bn_beta))),
                     "vpho_conv2d_nhwc_f32: bn_x is served by the direct-to-LDS 16-byte epilogue only (Cout %% 4 == 0, aligned, no splits / groups / pixel list / prologue / planes)");
    }
    VPHO_REQUIRE(!d.stats || (d.stats_rows && d.stats_cap > 0 && al16(d.stats)), "vpho_conv2d_nhwc_f32: stats needs stats_rows (host) and stats_cap");
    const long long big_tiles = ((M + 127) / 128) * ((d.Cout + 127) / 128) * ny;
    const double m_acc = (d.row_map && d.rows_hint > 0) ? (double)d.rows_hint : (double)M;   // rows the launch really computes
    const double flops = 2.0 * m_acc * d.Cout * g.K * ny;
    // algorithmic HBM bytes: input, packed weights and output once each (+ residual, + bias)
    const double bytes = groups * 4.0 * ((double)d.N * d.H * d.W * d.Cin + (d.x2 ? (double)d.N * d.OH * d.OW * d.Cin2 : 0.0) + (double)d.Cout * g.K + m_acc * d.Cout * (d.res ? 2 : 1) + d.Cout + (d.res_up ? (double)d.N * d.ru_H * d.ru_W * d.Cout : 0.0));
    static const int force_tile = getenv("VPHO_CONV_TILE") ? atoi(getenv("VPHO_CONV_TILE")) : 0;   // tuning aid
    // tile choice (measured on MI355X, scripts/conv_tune.py): 8-wave 128x128 when it still gives >= 2 tiles per CU,
    // 8-wave 128x64 when that gives >= 1 tile per CU, else the 4-wave 64x64 tile (small feature maps, narrow heads)
    const long long tiles_12864 = ((M + 127) / 128) * ((d.Cout + 63) / 64) * ny;
    int variant = 64;
    if (big_tiles >= 256 && d.Cout % 128 == 0) variant = 1288;
    else if (tiles_12864 >= 256 && d.Cout >= 48) variant = 12864;
    if (force_tile) variant = force_tile;
    bool stats_kernel = false, stats_failed = false;               // stats_kernel: the switch below launches conv_igemm_glds_kernel
    auto launch = [&](auto kernel, int bm, int bn, int threads, int cls) {
        vpho::ProfScope prof(cls, s, flops, bytes);
        g.tiles_m = (int)((M + bm - 1) / bm); g.tiles_n = (d.Cout + bn - 1) / bn;
        g.ntiles = g.tiles_m * g.tiles_n;
        if (d.stats) {
            const bool on = stats_kernel && stats_shape && g.tiles_m <= d.stats_cap;
            if (on) *d.stats_rows = g.tiles_m;
            else { g.d.stats = nullptr; if (d.bn_x) { stats_failed = true; return; } }     // never drop the gate silently
        }
        hipLaunchKernelGGL(kernel, dim3((g.ntiles + 7) / 8 * 8, ny), dim3(threads), 0, s, g);
    };
    static const int no_glds = getenv("VPHO_CONV_NO_GLDS") ? atoi(getenv("VPHO_CONV_NO_GLDS")) : 0;   // tuning aid
    // the direct-to-LDS kernels address x and w by 32-bit byte offsets: both extents (all splits included) must stay below 4 GB
    // (the largest activation of the reference's configurations is 0.27 GB)
    const double x_extent = 4.0 * (((double)d.N * d.H * d.W - 1) * d.x_ld + d.Cin + (double)(ny - 1) * (double)g.x_zs);
    const double w_extent = 4.0 * (((double)d.Cout - 1) * g.w_ld + g.K + (double)(ny - 1) * (double)g.w_zs);
    VPHO_REQUIRE(x_extent < 3.9e9 && w_extent < 3.9e9 && g.x_zs >= 0 && g.w_zs >= 0,
                 "vpho_conv2d_nhwc_f32: input (%.2f GB) and weights (%.2f GB) must each stay below 3.9 GB", x_extent * 1e-9, w_extent * 1e-9);
    static const int no_uni = getenv("VPHO_CONV_NO_UNI") ? atoi(getenv("VPHO_CONV_NO_UNI")) : 0;          // tuning aid
    g.uni = (d.Cin % BK == 0 && d.pad_y >= 0 && d.pad_x >= 0 && !no_uni) ? 1 : 0;
    // a pre-activation prologue rides on the direct-to-LDS kernel when the fragment's k is a plain channel index (1x1, unpadded,
    // Cin a multiple of 32 and within the LDS table); everything else with a prologue takes the register-staged kernel
    const bool pre_on_read = d.in_scale != nullptr && g.uni && d.KH == 1 && d.KW == 1 && d.pad_y == 0 && d.pad_x == 0 && d.Cin <= GLDS_PRE_MAX;
    const bool glds = (d.in_scale == nullptr || pre_on_read) && (!no_glds || d.res_up != nullptr || d.x2 != nullptr || d.bn_x != nullptr);
    stats_kernel = glds && variant != 128;
    VPHO_REQUIRE(!d.x2 || (g.uni && variant != 128), "vpho_conv2d_nhwc_f32: x2 is served by the direct-to-LDS kernels with wave-uniform taps only");
    VPHO_REQUIRE(!d.res_up || variant != 128, "vpho_conv2d_nhwc_f32: res_up is not served by the forced register-staged tile");
    // opt-in split-bf16 products (never the default): shapes the split kernel takes, everything else stays on the fp32 kernels
    const bool split_ok = d.w_planes != nullptr && (d.plane_terms == 6 || d.plane_terms == 9) && d.in_scale == nullptr && splits == 1 && !d.gate &&
                          d.Cin % SBK == 0 && g.vec_epilogue && d.pad_y >= 0 && d.pad_x >= 0 && g.w_ld == g.K &&
                          ((uintptr_t)d.w_planes & 15) == 0 && 6.0 * d.Cout * g.K < 3.9e9;
    if (split_ok) {
        const bool x9 = d.plane_terms == 9;
        switch (variant) {
            case 128: case 1288:
                if (x9) launch(conv_igemm_split_kernel<128, 128, 4, 2, 9>, 128, 128, 512, vpho::PROF_CONV128);
                else    launch(conv_igemm_split_kernel<128, 128, 4, 2, 6>, 128, 128, 512, vpho::PROF_CONV128);
                break;
            case 12864:
                if (x9) launch(conv_igemm_split_kernel<128, 64, 4, 2, 9>, 128, 64, 512, vpho::PROF_CONV128x64);
                else    launch(conv_igemm_split_kernel<128, 64, 4, 2, 6>, 128, 64, 512, vpho::PROF_CONV128x64);
                break;
            default:
                if (x9) launch(conv_igemm_split_kernel<64, 64, 2, 2, 9>, 64, 64, 256, vpho::PROF_CONV64);
                else    launch(conv_igemm_split_kernel<64, 64, 2, 2, 6>, 64, 64, 256, vpho::PROF_CONV64);
                break;
        }
        return vpho::check_launch("conv_igemm_split_kernel");
    }
    // persistent multi-tile kernel (round 5): 1x1 launches of the 128x128 class with more tiles than workgroup slots (two 80-KB workgroups
    // per CU).  VPHO_CONV_PERS: 0 = never (round 4's kernel, A/B aid -- same bits), 1 = default, 2 = also launches of at most one round
    {
        const char* pe = getenv("VPHO_CONV_PERS");
        const int pers = pe ? atoi(pe) : 1;
        static int slots_of[64] = {0};
        int dev = 0;
        VPHO_HIP(hipGetDevice(&dev));
        int& slots = slots_of[dev & 63];
        if (!slots) {
            hipDeviceProp_t prop;
            VPHO_HIP(hipGetDeviceProperties(&prop, dev));
            slots = 2 * prop.multiProcessorCount;
        }
        const double y_extent = 4.0 * ((double)(d.N - 1) * d.y_sn + (double)(d.OH - 1) * d.y_sy + (double)(d.OW - 1) * d.y_sx + d.Cout);
        const double r_extent = d.res ? 4.0 * ((double)(d.N - 1) * d.r_sn + (double)(d.OH - 1) * d.r_sy + (double)(d.OW - 1) * d.r_sx + d.Cout) : 0.0;
        const bool simple = d.KH == 1 && d.KW == 1 && d.pad_y == 0 && d.pad_x == 0;
        const bool ok = pers && variant == 1288 && glds && !pre_on_read && d.in_scale == nullptr && g.uni && simple && g.vec_epilogue && !d.row_map && !d.res_up &&
                        !d.gate && !d.bn_x && splits == 1 && y_extent < 3.9e9 && r_extent < 3.9e9 && d.y_sn >= 0 && d.y_sy >= 0 && d.y_sx >= 0 &&
                        (!d.x2 || 4.0 * d.N * d.H2 * d.W2 * (double)d.x2_ld < 3.9e9);
        if (ok && (big_tiles > slots || pers >= 2)) {
            vpho::ProfScope prof(vpho::PROF_CONV128, s, flops, bytes);
            g.tiles_m = (int)((M + 127) / 128); g.tiles_n = (d.Cout + 127) / 128;
            g.ntiles = g.tiles_m * g.tiles_n;
            if (d.stats) {
                if (stats_shape && g.tiles_m <= d.stats_cap) *d.stats_rows = g.tiles_m; else g.d.stats = nullptr;
            }
            const int grid = (int)std::min<long long>((g.ntiles + 7) / 8 * 8, std::max(8, slots / groups / 8 * 8));      // the groups share the slots
            if (g.d.stats) hipLaunchKernelGGL((conv_igemm_pers_bn_kernel<128, 128, 4, 2>), dim3(grid, groups), dim3(512), 0, s, g);
            else hipLaunchKernelGGL((conv_igemm_pers_kernel<128, 128, 4, 2>), dim3(grid, groups), dim3(512), 0, s, g);
            return vpho::check_launch("conv_igemm_pers_kernel");
        }
    }
    switch (variant) {
        case 128:  launch(conv_igemm_kernel<128, 128, 2, 2, 1>, 128, 128, 256, vpho::PROF_CONV128); break;
        case 1288:
            if (glds && pre_on_read) launch(conv_igemm_glds_kernel<128, 128, 4, 2, true>, 128, 128, 512, vpho::PROF_CONV128);
            else if (glds) launch(conv_igemm_glds_kernel<128, 128, 4, 2>, 128, 128, 512, vpho::PROF_CONV128);
            else      launch(conv_igemm_kernel<128, 128, 4, 2, 1>, 128, 128, 512, vpho::PROF_CONV128);
            break;
        case 12864:
            if (glds && pre_on_read) launch(conv_igemm_glds_kernel<128, 64, 4, 2, true>, 128, 64, 512, vpho::PROF_CONV128x64);
            else if (glds) launch(conv_igemm_glds_kernel<128, 64, 4, 2>, 128, 64, 512, vpho::PROF_CONV128x64);
            else      launch(conv_igemm_kernel<128, 64, 4, 2, 1>, 128, 64, 512, vpho::PROF_CONV128x64);
            break;
        default:
            if (glds && pre_on_read) launch(conv_igemm_glds_kernel<64, 64, 2, 2, true>, 64, 64, 256, vpho::PROF_CONV64);
            else if (glds) launch(conv_igemm_glds_kernel<64, 64, 2, 2>, 64, 64, 256, vpho::PROF_CONV64);
            else      launch(conv_igemm_kernel<64, 64, 2, 2, 1>, 64, 64, 256, vpho::PROF_CONV64);
            break;
    }
    if (stats_failed)
        return vpho::fail("vpho_conv2d_nhwc_f32: bn_x: the launch needs %d partial rows (stats_cap %d) or runs a kernel without the BatchNorm epilogue (tile %d)",
                          g.tiles_m, d.stats_cap, variant);
    return vpho::check_launch("conv_igemm_kernel");
}
