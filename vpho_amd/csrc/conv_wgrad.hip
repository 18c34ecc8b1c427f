// Weight gradient of the NHWC convolutions as ONE implicit "TN" GEMM on the fp32 matrix cores:
//     dW[co][(r,s),ci] = sum over output pixels p of dY[p][co] * x[n(p), oy(p)*stride + r - pad_y, ox(p)*stride + s - pad_x, ci]
// -- torch.nn.functional.conv2d's weight gradient, what loss.backward() computes for every nn.Conv2d of
// lib/model/backbone_FPN_HFL.py, encoding.py and head_inplane.py (lib/engine/train_diff_hand_obj.py:181-182).
//
// The reduction runs over the pixel index, which is the SLOW index of both NHWC operands, so the tiles are staged k-major:
// a stage is 32 pixels x (BM output channels of dY | BN input channels of the tap-shifted x), every pixel row one contiguous
// 16-byte-chunked channel run moved by `buffer_load_dwordx4 ... lds` (out-of-image taps, pixel and channel tails carry an
// out-of-range offset: the hardware writes zeros).  The MFMA fragments are then single dwords at [k][lane & 31]: half-wave-contiguous ds_read_b32, conflict-free
// without padding or swizzle.  A workgroup owns one (Cout tile, tile of the flattened (tap, ci) axis) for one slice of the pixel
// range (split-K; see the kernel for which workgroup takes which slice); a 16-byte chunk never straddles taps (Cin % 4 == 0), so the tap is a per-lane constant and
// narrow inputs (the 4-channel stem: 16 taps per 64-column tile) take the same path.  Slices are summed in a fixed order by a
// second kernel.  No im2col / transposed copies are materialised.
#include "common.h"
#include "../../include/vpho_hip.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace {

constexpr int BK = 32;
#ifndef WGRAD_RING
#define WGRAD_RING 2                           // LDS stages of the 64 x 64 tile (A/B builds: 3)
#endif
#ifndef WGRAD_ABLATE
#define WGRAD_ABLATE 0                         // timing experiments (scripts/kernel_ablate.sh conv_wgrad WGRAD_ABLATE ...): 1 no stage fills after the first, 2 no fragment
#endif                                         // reads after the first stage, 4 no stage barrier -- wrong results, never in the product build

struct WgArgs {
    const float* x; const float* dy; float* out;       // out: [splits][Cout][K]
    int N, H, W, Cin, x_ld, OH, OW, Cout, dy_ld, KH, KW, stride, pad_y, pad_x;
    int M, K, tiles_n, ntiles, m_chunk, splits;
    // optional: the reduction runs over the LISTED groups of BK consecutive pixels only (ascending; vpho_window_groups_i32) -- dY is
    // known to be zero everywhere else (the gradient of a map that is read through RoIAlign lives in the RoI windows).  Needs
    // OW % BK == 0 (a group never leaves its pixel row).  The slices then cut the LIST, m_chunk / BK entries each.
    const int* glist; const int* gcount;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_wgrad_tn_kernel(const WgArgs a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_INSTR = BK * BM / 256, B_INSTR = BK * BN / 256;         // 1024-byte wave instructions per stage
    constexpr int A_LD = A_INSTR / NW, B_LD = B_INSTR / NW;
    constexpr int A_CPR = BM / 4, B_CPR = BN / 4;                           // 16-byte chunks per tile row
    static_assert(TM >= 1 && TN >= 1 && A_INSTR % NW == 0 && B_INSTR % NW == 0 && A_LD >= 1 && B_LD >= 1, "tile/wave layout");
    static_assert(64 % A_CPR == 0 && 64 % B_CPR == 0, "a wave instruction must cover whole tile rows");
    constexpr int TILE = (BM + BN) * BK;
    // stages in LDS: 2, or WGRAD_RING for the 64 x 64 tile (a fill then has RING - 1 stage times to land; 3 x 16 KB still leaves 3 workgroups per CU)
    constexpr int NB = (BM == 64 && WGRAD_RING > 2) ? WGRAD_RING : 2;
    constexpr int FI = A_LD + B_LD;                                        // direct-to-LDS instructions of one stage fill, per wave
    __shared__ __attribute__((aligned(1024))) float smem[NB * TILE];

    // Workgroup -> (pixel slice, tile).  Workgroups go to the 8 XCDs round-robin, each XCD has its own L2: XCD x takes the slices
    // s = x, x + 8, ... and runs ALL tiles of a slice one after the other, so both operands of a slice (its dY rows, its x rows) are
    // fetched from HBM by one XCD once and shared through its L2 by the slice's tiles.  (Round 3 gave every XCD a range of tiles
    // for all slices: an operand shared by tiles of different ranges was fetched once per XCD -- 64 -> 256 channels on 64 x 64 x 64
    // pixels moved 536 MB instead of 335 and ran at 37 TF/s.)
    // With fewer than 8 slices that would leave XCDs idle: then every XCD takes a range of tiles of every slice (consecutive column
    // tiles of one Cout tile on one XCD share dY), the order of round 3.
    int slice, lb;
    if (a.splits >= 8) {
        const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
        slice = (seq / a.ntiles) * 8 + xcd; lb = seq % a.ntiles;
    } else {
        const int gx = (a.ntiles + 7) / 8 * 8, bx = blockIdx.x % gx;
        slice = blockIdx.x / gx; lb = (bx & 7) * (gx >> 3) + (bx >> 3);
    }
    if (slice >= a.splits || lb >= a.ntiles) return;
    const int tile_n = lb % a.tiles_n, tile_m = lb / a.tiles_n;
    const int co0 = tile_m * BM, c0 = tile_n * BN;                          // c0: column of the flattened (tap, ci) axis
    const bool listed = a.glist != nullptr;
    const int s_per = a.m_chunk / BK;                                       // listed: list entries per slice
    const int s_begin = slice * s_per;
    int s_end = 0;
    if (listed) { const int total = *a.gcount; s_end = s_begin + s_per < total ? s_begin + s_per : total; }
    const int m_begin = listed ? 0 : slice * a.m_chunk;
    const int m_end = listed ? a.M : (m_begin + a.m_chunk < a.M ? m_begin + a.m_chunk : a.M);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ohw = a.OH * a.OW;

    // fill roles of this lane: A_LD rows of the dY tile, B_LD rows of the x tile (one 16-byte chunk of each)
    int a_row[A_LD], b_row[B_LD];
    const int a_col = co0 + 4 * (lane % A_CPR), b_col = c0 + 4 * (lane % B_CPR);
    const bool a_ok = a_col < a.Cout, b_ok = b_col < a.K;
    const int tap = b_col / a.Cin, b_ci = b_col - tap * a.Cin;              // this lane's tap and first input channel
    const int tr = tap / a.KW, ts = tap - tr * a.KW;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_row[j] = (wave + NW * j) * (64 / A_CPR) + lane / A_CPR;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) b_row[j] = (wave + NW * j) * (64 / B_CPR) + lane / B_CPR;
    // pixel (n, oy, ox) of each x row of the FIRST stage (stages are filled in order kt = 0, 1, ...: fill() advances them)
    int b_n[B_LD], b_oy[B_LD], b_ox[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int m = m_begin + b_row[j];
        b_n[j] = m / ohw;
        const int rem = m - b_n[j] * ohw;
        b_oy[j] = rem / a.OW; b_ox[j] = rem - b_oy[j] * a.OW;
    }
    // buffer addressing (32-bit byte offsets, extents < 4 GB checked by the host): offset 0xFFFFFFFF is out of range and makes the
    // hardware write zeros -- out-of-image taps, pixel and channel tails (see conv_igemm.hip)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;

    auto fill = [&](int buf, int kt) {
        float* As = smem + buf * TILE;
        float* Bs = As + BM * BK;
        const int mb = listed ? a.glist[s_begin + kt] * BK : m_begin + kt * BK;
        if (listed) {                                      // the group's pixel row from its first pixel (uniform: scalar divisions)
            const int q = mb / a.OW, ox0 = mb - q * a.OW, n = q / a.OH, oy = q - n * a.OH;
#pragma unroll
            for (int j = 0; j < B_LD; ++j) { b_n[j] = n; b_oy[j] = oy; b_ox[j] = ox0 + b_row[j]; }
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int m = mb + a_row[j];
            const unsigned off = (a_ok && m < m_end) ? ((unsigned)m * (unsigned)a.dy_ld + (unsigned)a_col) * 4u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr)(As + (wave + NW * j) * 256), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int m = mb + b_row[j];
            unsigned off = 0xFFFFFFFFu;
            if (b_ok && m < m_end) {
                const unsigned iy = (unsigned)(b_oy[j] * a.stride + tr - a.pad_y), ix = (unsigned)(b_ox[j] * a.stride + ts - a.pad_x);
                if (iy < (unsigned)a.H && ix < (unsigned)a.W) off = (((unsigned)(b_n[j] * a.H + (int)iy) * (unsigned)a.W + ix) * (unsigned)a.x_ld + (unsigned)b_ci) * 4u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(Bs + (wave + NW * j) * 256), 16, (int)off, 0, 0, 0);
            // this row's pixel for the next stage: BK pixels further, (n, oy, ox) advanced without a division
            if (listed) continue;
            if (a.OW >= 8) {
                b_ox[j] += BK;
                while (b_ox[j] >= a.OW) { b_ox[j] -= a.OW; if (++b_oy[j] == a.OH) { b_oy[j] = 0; ++b_n[j]; } }
            } else {                                       // narrow maps / rows as 1x1 images: the wrap loop would run up to BK times
                const int m2 = m + BK;
                b_n[j] = m2 / ohw;
                const int rem = m2 - b_n[j] * ohw;
                b_oy[j] = rem / a.OW; b_ox[j] = rem - b_oy[j] * a.OW;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = listed ? (s_end > s_begin ? s_end - s_begin : 0) : (m_end - m_begin + BK - 1) / BK;
    if (nk > 0) {
#pragma unroll
        for (int s0 = 0; s0 < NB - 1; ++s0) if (s0 < nk) fill(s0, s0);
        VPHO_SYNC_LDS_DMA();
    }
    constexpr int HK = BK / 4;
    float av[2][HK][TM], bv[2][HK][TN];
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NB;
        const bool filled = kt + NB - 1 < nk;
        if (filled && !(WGRAD_ABLATE & 1)) fill((kt + NB - 1) % NB, kt + NB - 1);
        const float* As = smem + buf * TILE + wm * (BM / WM) + li;
        const float* Bs = smem + buf * TILE + BM * BK + wn * (BN / WN) + li;
        // the stage's fragments are read in two halves of 8 k-steps; the second half's ds_reads are in flight while the first
        // half's MFMAs run (scheduling barriers keep the compiler from sinking the reads back next to their uses)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((WGRAD_ABLATE & 2) && kt > 0) break;                 // timing: the first stage's fragments for every stage
#pragma unroll
            for (int kk = 0; kk < HK; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[h][kk][i] = As[(2 * (h * HK + kk) + lh) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[h][kk][j] = Bs[(2 * (h * HK + kk) + lh) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int kk = 0; kk < HK; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[h][kk][i], bv[h][kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(WGRAD_ABLATE & 4)) {
            // stage kt + 1 has landed once at most the fills of the NB - 2 stages behind it are outstanding (in order); in the tail, where no
            // fill was issued this stage, everything
            // (the fence-free barrier: __syncthreads() would make the compiler drain the fills in flight, common.h)
            if (NB > 2 && filled) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NB - 2) * FI) : "memory"); VPHO_BARRIER_LDS_ONLY(); }
            else VPHO_SYNC_LDS_DMA();
        }
    }

    float* out = a.out + (long long)slice * a.Cout * a.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = c0 + wn * (BN / WN) + j * 32 + li;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < a.Cout) out[(long long)row * a.K + col] = acc[i][j][e];
            }
    }
}

// out[i] = sum_s part[s][i], s ascending (n multiple of 4, 16-byte aligned)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int splits, long long n4, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
    for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(part)[(long long)k * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = s;
}
// The same sum for a SMALL dW cut into MANY slices (1x1 64 -> 256 on 64 x 64 maps: 4 096 quads x 256 slices -- the kernel above is 16
// workgroups whose threads walk 256 dependent-latency loads): G threads share a quad, thread g sums the slices g, g + G, ... in
// ascending order, the G sums are combined in ascending g.  A fixed order for a given (slices, G), hence deterministic; not the order
// of the kernel above (which order a shape takes is decided by wgrad_reduce_groups from its sizes alone).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const float* __restrict__ part, int splits, long long n4, float* __restrict__ out) {
    constexpr int Q = 256 / G;
    __shared__ f32x4 red[G][Q];
    const int q = threadIdx.x % Q, g = threadIdx.x / Q;
    const long long i = (long long)blockIdx.x * Q + q;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) for (int k = g; k < splits; k += G) s += reinterpret_cast<const f32x4*>(part)[(long long)k * n4 + i];
    red[g][q] = s;
    __syncthreads();
    if (g != 0 || i >= n4) return;
    for (int k = 1; k < G; ++k) s += red[k][q];
    reinterpret_cast<f32x4*>(out)[i] = s;
}
// threads per quad: 1 (the plain kernel) while it already fills the chip or the slices are few
inline int wgrad_reduce_groups(int splits, long long n4) {
    if (splits < 8 || n4 >= 256ll * 1024) return 1;
    return splits >= 32 && n4 < 64ll * 1024 ? 16 : 4;
}

// ordered list of the live groups of BK consecutive pixels of an (N, H, W) map: group g is live when one of its pixels lies in the
// window (y0, x0, w, h) of its image (vpho_roi_windows_i32's table).  One workgroup: every thread counts its run of consecutive
// groups, an exclusive scan of the counts gives each run its place -- ascending order, no atomics.
__global__ __launch_bounds__(1024) void window_groups_kernel(const int* __restrict__ wins, int N, int H, int W, int* __restrict__ list,
                                                              int* __restrict__ count) {
    __shared__ int s_cnt[1024];
    const int G = N * H * W / BK, per = (G + 1023) / 1024, g0 = threadIdx.x * per;
    auto live = [&](int g) {
        const int m = g * BK, n = m / (H * W), rem = m - n * H * W;
        const int* w = wins + 5 * n;
        for (int k = 0; k < BK; ++k) {
            const int y = (rem + k) / W, x = (rem + k) - y * W;
            if (y >= w[1] && y < w[1] + w[4] && x >= w[2] && x < w[2] + w[3]) return true;
        }
        return false;
    };
    int c = 0;
    for (int g = g0; g < g0 + per && g < G; ++g) c += live(g) ? 1 : 0;
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                  // inclusive scan
        const int v = threadIdx.x >= o ? s_cnt[threadIdx.x - o] : 0;
        __syncthreads();
        s_cnt[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = s_cnt[threadIdx.x] - c;
    for (int g = g0; g < g0 + per && g < G; ++g) if (live(g)) list[pos++] = g;
    if (threadIdx.x == 1023) *count = s_cnt[1023];
}

struct WgPlan { int bm, bn, tiles_m, tiles_n, splits, m_chunk; };

WgPlan plan_wgrad(long long M, int Cin, int Cout, int taps) {
    WgPlan p;
    const int K = Cin * taps;
    static const int tile_env = getenv("VPHO_WGRAD_TILE") ? atoi(getenv("VPHO_WGRAD_TILE")) : 0;          // tuning aid
    // 128x128 tiles pay off only for the largest products (measured: 3x3 256->256 on 64x64x64 pixels 100 vs 87 TF/s); everywhere else
    // the 64x64 tile wins or ties (3x3 128->128: 79 vs 53 TF/s) -- more tiles, hence fewer, longer pixel slices and less partial-sum traffic
    // (round 5, re-swept per shape with the grouped slice sum, scripts/wgrad_layers.py under VPHO_WGRAD_TILE=64 / 128: the 128 x 128 tile now
    // wins or ties from M K Cout ~ 4e9 on -- 1x1 1024 -> 256 on 16 x 16 maps 101 -> 92 us, 3x3 128 -> 128 on 32 x 32 217 -> 206 -- and loses below
    // (3x3 128 -> 128 on 16 x 16: 74 against 88): 19.6 -> 18.7 ms of weight gradients per training step)
    static const double big_min = getenv("VPHO_WGRAD_BIG_MIN") ? atof(getenv("VPHO_WGRAD_BIG_MIN")) : 4e9;      // tuning aid (round 4's rule: 5e10)
    const bool big = tile_env ? tile_env == 128 : (K >= 128 && Cout >= 128 && (double)M * K * Cout >= big_min);
    p.bm = p.bn = big ? 128 : 64;
    // Three tiles (re-swept per shape, round 5, scripts/wgrad_layers.py under VPHO_WGRAD_TILE = 64 / 128 / 12864: 19.6 / 20.8 / 18.5 ms of weight
    // gradients per training step, this rule 18.4, the best tile per shape 18.3): 128 output channels x 64 columns of the (tap, ci) axis --
    // 0.047 bytes of LDS fill per flop against 0.0625 for 64 x 64, three workgroups per CU -- wherever Cout >= 128 and the product is not tiny
    // (it also serves K = 64: 64 -> 256 on 64 x 64 maps 93 -> 88 us; 3x3 256 -> 256 on 16 x 16 202 -> 190); 128 x 128 for the 1x1 layers with
    // K a multiple of 128 from M K Cout = 4e9 on (1024 -> 256 on 16 x 16: 101 -> 92) and for the largest products; 64 x 64 for the rest.
    if (!tile_env) {
        const double prod = (double)M * K * Cout;
        const bool big128 = K >= 128 && Cout >= 128 && ((taps == 1 && K % 128 == 0 && prod >= big_min) || prod >= 5e10);
        if (big128) p.bm = p.bn = 128;
        else if (Cout >= 128 && K >= 64 && prod >= 1e9) { p.bm = 128; p.bn = 64; }
        else p.bm = p.bn = 64;
    } else if (tile_env == 12864 && Cout >= 128) { p.bm = 128; p.bn = 64; }
    p.tiles_m = (Cout + p.bm - 1) / p.bm;
    p.tiles_n = (K + p.bn - 1) / p.bn;
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    // Workgroups for 256 CUs (2 x 128x128 or 4 x 64x64 tiles fit a CU's LDS): about two rounds of them, but no slices shorter than
    // 16 stages of 32 pixels (8 on the smallest maps, where nothing else fills the chip) -- a slice pays its prologue, its 16 KB
    // (64 KB) of partial sums and their reduction whatever its length.  Swept over the training step's 47 shapes
    // (scripts/r04_wgrad_want.sh, r04_wgrad_stages.sh): the 3x3 layers want ~2 000 workgroups, the 1x1 layers on 32 x 32 maps ~500
    // (8 tiles x 64 slices of 1 024 pixels); a fixed target of 1 536 with 8-stage slices cost 1.5 ms per step more.
    static const int want_env = getenv("VPHO_WGRAD_WANT") ? atoi(getenv("VPHO_WGRAD_WANT")) : 0;      // tuning aid
    static const int stages_env = getenv("VPHO_WGRAD_STAGES") ? atoi(getenv("VPHO_WGRAD_STAGES")) : 0;
    const long long want = want_env > 0 ? (big ? want_env : 2 * want_env) : (big ? 1024 : 2048);
    const long long min_stages = stages_env > 0 ? stages_env : (M < 8192 ? 8 : 16);
    long long splits = std::max<long long>(1, (want + tiles - 1) / tiles);
    splits = std::min<long long>(splits, std::max<long long>(1, M / (min_stages * BK)));
    splits = std::min<long long>(splits, 256);
    if (splits >= 8) splits = std::max<long long>(8, (splits + 4) / 8 * 8);          // slices go to the 8 XCDs round-robin (see the kernel)
    long long chunk = (M + splits - 1) / splits;
    chunk = (chunk + BK - 1) / BK * BK;
    p.m_chunk = (int)chunk;
    p.splits = (int)((M + chunk - 1) / chunk);
    return p;
}

}  // namespace

extern "C" long long vpho_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW) {
    if (N <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return -1;
    const WgPlan p = plan_wgrad((long long)N * OH * OW, Cin, Cout, KH * KW);
    return p.splits > 1 ? (long long)p.splits * Cout * KH * KW * Cin * 4 : 0;
}

extern "C" int vpho_window_groups_i32(const int* wins, int N, int H, int W, int* group_list, int* group_count, void* stream) {
    VPHO_REQUIRE(wins && group_list && group_count && N > 0 && H > 0 && W > 0, "vpho_window_groups_i32: bad argument");
    VPHO_REQUIRE(W % BK == 0 && (long long)N * H * W < (1ll << 31), "vpho_window_groups_i32: the map width must be a multiple of %d", BK);
    hipLaunchKernelGGL(window_groups_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, wins, N, H, W, group_list, group_count);
    return vpho::check_launch("window_groups_kernel");
}

static int wgrad_launch(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout, int dy_ld,
                        int KH, int KW, int stride, int pad_y, int pad_x, float* dw, void* workspace, void* stream, const int* glist, const int* gcount);

extern "C" int vpho_conv2d_wgrad_nhwc_f32(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout, int dy_ld,
                                          int KH, int KW, int stride, int pad_y, int pad_x, float* dw, void* workspace, void* stream) {
    return wgrad_launch(x, N, H, W, Cin, x_ld, dy, OH, OW, Cout, dy_ld, KH, KW, stride, pad_y, pad_x, dw, workspace, stream, nullptr, nullptr);
}

extern "C" int vpho_conv2d_wgrad_groups_nhwc_f32(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout,
                                                 int dy_ld, int KH, int KW, int stride, int pad_y, int pad_x, const int* group_list,
                                                 const int* group_count, float* dw, void* workspace, void* stream) {
    VPHO_REQUIRE(group_list && group_count && OW % BK == 0, "vpho_conv2d_wgrad_groups_nhwc_f32: needs the group list and an output width that is a multiple of %d", BK);
    return wgrad_launch(x, N, H, W, Cin, x_ld, dy, OH, OW, Cout, dy_ld, KH, KW, stride, pad_y, pad_x, dw, workspace, stream, group_list, group_count);
}

static int wgrad_launch(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout, int dy_ld,
                        int KH, int KW, int stride, int pad_y, int pad_x, float* dw, void* workspace, void* stream, const int* glist, const int* gcount) {
    VPHO_REQUIRE(x && dy && dw, "vpho_conv2d_wgrad_nhwc_f32: null tensor");
    VPHO_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && OH > 0 && OW > 0, "vpho_conv2d_wgrad_nhwc_f32: non-positive dimension");
    VPHO_REQUIRE(Cin % 4 == 0 && x_ld % 4 == 0 && x_ld >= Cin && Cout % 4 == 0 && dy_ld % 4 == 0 && dy_ld >= Cout,
                 "vpho_conv2d_wgrad_nhwc_f32: Cin=%d x_ld=%d Cout=%d dy_ld=%d must be multiples of 4 (ld >= channels)", Cin, x_ld, Cout, dy_ld);
    VPHO_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dw & 15) == 0, "vpho_conv2d_wgrad_nhwc_f32: x/dy/dw must be 16-byte aligned");
    VPHO_REQUIRE((OH - 1) * stride - pad_y < H && (OW - 1) * stride - pad_x < W, "vpho_conv2d_wgrad_nhwc_f32: output larger than input allows");
    const long long M = (long long)N * OH * OW;
    VPHO_REQUIRE(M < (1ll << 31) && (long long)N * H * W < (1ll << 31), "vpho_conv2d_wgrad_nhwc_f32: too many pixels");
    VPHO_REQUIRE((double)N * H * W * x_ld * 4.0 < 4.0e9 && (double)M * dy_ld * 4.0 < 4.0e9, "vpho_conv2d_wgrad_nhwc_f32: x and dy must each stay below 4 GB (32-bit buffer offsets)");
    const int taps = KH * KW;
    const WgPlan p = plan_wgrad(M, Cin, Cout, taps);
    VPHO_REQUIRE(p.splits == 1 || (workspace && ((uintptr_t)workspace & 15) == 0), "vpho_conv2d_wgrad_nhwc_f32: %d pixel slices need the workspace", p.splits);
    WgArgs a;
    a.x = x; a.dy = dy; a.out = p.splits > 1 ? (float*)workspace : dw;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.x_ld = x_ld; a.OH = OH; a.OW = OW; a.Cout = Cout; a.dy_ld = dy_ld;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad_y = pad_y; a.pad_x = pad_x;
    a.M = (int)M; a.K = taps * Cin; a.tiles_n = p.tiles_n; a.ntiles = p.tiles_m * p.tiles_n; a.m_chunk = p.m_chunk;
    a.glist = glist; a.gcount = gcount; a.splits = p.splits;
    hipStream_t s = (hipStream_t)stream;
    // see the kernel: XCD = blockIdx.x & 7
    const dim3 grid(p.splits >= 8 ? 8u * (unsigned)a.ntiles * (unsigned)((p.splits + 7) / 8) : (unsigned)((a.ntiles + 7) / 8 * 8) * (unsigned)p.splits);
    // dW[Cout][taps * Cin] = dY^T (M x Cout) . im2col(x) (M x taps * Cin): 2 M Cout K flop; operands read once + dW written once
    // (a launch on a pixel-group list reduces only over the live groups -- device data --: it is timed with its full-size count and
    // reported as an upper bound by its own class id would mislead, so group launches are left out of the classes)
    vpho::ProfScope prof(glist ? -1 : (p.bm == 128 ? vpho::PROF_WGRAD128 : vpho::PROF_WGRAD64), s, 2.0 * (double)M * Cout * (double)a.K,
                         4.0 * ((double)N * H * W * Cin + (double)M * Cout + (double)Cout * a.K));
    if (p.bm == 128 && p.bn == 64) hipLaunchKernelGGL((conv_wgrad_tn_kernel<128, 64, 4, 2>), grid, dim3(512), 0, s, a);
    else if (p.bm == 128) hipLaunchKernelGGL((conv_wgrad_tn_kernel<128, 128, 4, 2>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_tn_kernel<64, 64, 2, 2>), grid, dim3(256), 0, s, a);
    if (p.splits > 1) {
        const long long n4 = (long long)Cout * a.K / 4;
        static const int g_env = getenv("VPHO_WGRAD_REDUCE_GROUPS") ? atoi(getenv("VPHO_WGRAD_REDUCE_GROUPS")) : 0;      // tuning aid: 1 / 4 / 16
        const int G = g_env == 1 || g_env == 4 || g_env == 16 ? g_env : wgrad_reduce_groups(p.splits, n4);
        if (G == 16) hipLaunchKernelGGL(wgrad_reduce_grouped_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, (const float*)workspace, p.splits, n4, dw);
        else if (G == 4) hipLaunchKernelGGL(wgrad_reduce_grouped_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, s, (const float*)workspace, p.splits, n4, dw);
        else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, p.splits, n4, dw);
    }
    return vpho::check_launch("conv_wgrad_tn_kernel");
}
