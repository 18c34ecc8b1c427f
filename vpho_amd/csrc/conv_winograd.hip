// Winograd F(2x2, 3x3) convolution (stride 1, padding 1) on the gfx950 fp32 matrix cores: the default for the stride-1 3x3 convolutions of
// the inference plan and of the training step (VPHO_WINOGRAD=0 / VPHO_TRAIN_WINOGRAD=0 select the direct kernels).
//
// Y(2x2) = A^T [ sum_c (G g_c G^T) o (B^T d_c B) ] A: the 16 element-wise products are 16 independent GEMMs
//   M_f[tile][cout] = sum_cin V_f[tile][cin] * U_f[cout][cin]      (f = 4*fy + fx)
// with 2.25 x fewer multiply-adds than the direct 3x3.  U = G g G^T is computed once on the host (model/pack.py); V = B^T d B is
// formed by the workgroup from the NHWC input (additions only) and written straight into the LDS stage; the output transform runs on
// the accumulators in registers.
//
// Workgroup = 4 waves = 64 tiles x 64 output channels x all 16 frequencies; wave = 32 tiles x 32 channels x 16 frequencies: 16
// independent 32x32 accumulator tiles (256 registers), so a (tile, channel) pair has all its frequencies in ONE lane and the output
// transform needs no exchange.  One wave per SIMD: latency is hidden by the 16 independent MFMA chains per k step.
// Stage = 8 input channels: V [16][64][8] and U [16][64][8] fp32 (32-byte rows, 16-byte chunks exchanged on odd 8-row groups),
// double buffered (128 KB).
#include "common.h"
#include <type_traits>
#include "../../include/vpho_hip.h"
#include <cstdlib>
#include <algorithm>

VPHO_STAMP_DECL(wino)

namespace {

#ifndef WINO8_ABLATE
#define WINO8_ABLATE 0                         // the same for conv_winograd8_kernel (scripts/kernel_ablate.sh conv_winograd WINO8_ABLATE ...): 1 no transform / V
#endif                                         // stores, 2 no patch loads, 4 no U fill, 16 no stage barrier -- wrong results, compile-time only
#ifndef WINO_ABLATE
#define WINO_ABLATE 0                          // timing experiments only (scripts/wino_ablate.sh; 512: no output stores; 32 staged kernel: no input DMA): 1 no patch loads, 2 no U fill, 4 no transform /
#endif                                         // V stores, 8 no stage barrier, 16 no fragment reads -- wrong results, never in the product build
constexpr int WK = 8;                          // input channels per stage
constexpr int W_TB = 64, W_CB = 64;            // tiles / output channels per workgroup
constexpr int W_STAGE = 2 * 16 * 64 * WK;      // floats per stage: V | U
constexpr int W_IN_PIXELS = 480;               // STAGED kernel: pixels (64 bytes each) of the input region IN behind the two stages: 128 + 30 KB of the 160
// STAGED geometry: a 64-tile block = whole tile rows (TW | 64) of one image (NR = 64 / TW <= TH, TH % NR == 0) or all rows of NR / TH
// images (TH | NR), and the region's pixels fit IN
inline bool wino_staged_ok(int TH, int TW, int W) {
    if (TW < 4 || TW > 32 || (W_TB % TW) != 0) return false;
    const int NR = W_TB / TW;
    if (NR <= TH ? (TH % NR) != 0 : (NR % TH) != 0) return false;
    const int rpp = NR < TH ? NR : TH, ipb = NR / rpp;
    return ipb * (2 * rpp + 2) * (W + 2) <= W_IN_PIXELS;
}

struct WinoArgs {
    const float* x; const float* u; const float* bias; float* y;
    int N, H, W, Cin, x_ld, Cout, y_ld;
    int TH, TW, T;                             // tiles per column / row / in total
    float out_slope;
    // RoI windows (vpho_roi_windows_i32): only the 2 x 2 tiles -- on the image's even grid, so a pixel is computed from the same 4 x 4
    // patch as in the full map: bit-identical -- that touch an image's window; y = the COMPACT (rows, Cout) matrix of the window pixels
    const float* gate; float gate_slope;       // optional, laid out like y: y = gate > 0 ? y : gate_slope * y (LeakyReLU backward given its output)
    const int* wins;                           // [N][5] = (first row, y0, x0, w, h) or NULL
    const int* tile_base;                      // [N + 1] first tile of every image; [N] = live tiles (vpho_winograd_window_tiles_i32)
    int scatter;                               // with wins: 1 = y is the ordinary (N,H,W,y_ld) map, window pixels written in place, the rest untouched
    // grouped launch (blockIdx.y = group: the twin hand / object branches): the same problem on x + g*x_gs, u + g*u_gs, bias + g*b_gs, y + g*y_gs
    long long x_gs, u_gs, b_gs, y_gs;
    // conv_winograd_bn_kernel only (ABI 12): BatchNorm reductions in the epilogue, see vpho_conv_desc.stats / bn_x
    float* stats; const float* bn_x; const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
};
__device__ __forceinline__ WinoArgs wino_group(WinoArgs a, const unsigned g) {
    a.x += g * a.x_gs; a.u += g * a.u_gs; a.y += g * a.y_gs;
    if (a.bias) a.bias += g * a.b_gs;
    return a;
}

struct WinoBn { float* stats; int cap; int* rows; const float* x; const float* mean; const float* invstd; const float* gamma; const float* beta; };

// tile t -> image n, tile coordinates (ty, tx) on the image's even grid; false past the last live tile
__device__ inline bool wino_tile(const WinoArgs& a, int t, int& n, int& ty, int& tx) {
    if (!a.wins) {
        if (t >= a.T) return false;
        n = t / (a.TH * a.TW);
        const int rem = t - n * a.TH * a.TW;
        ty = rem / a.TW; tx = rem - ty * a.TW;
        return true;
    }
    if (t >= a.tile_base[a.N]) return false;
    int lo = 0, hi = a.N - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.tile_base[mid] <= t) lo = mid; else hi = mid - 1; }
    n = lo;
    const int* w = a.wins + 5 * n;
    const int tx0 = w[2] >> 1, ntx = ((w[2] + w[3] - 1) >> 1) - tx0 + 1;
    const int lt = t - a.tile_base[n];
    ty = (w[1] >> 1) + lt / ntx; tx = tx0 + lt % ntx;
    return true;
}

// STAGED (round 5; full maps whose 64-tile blocks are whole tile rows: wino_staged_ok): the input patches do not come through vector
// registers.  Ablations of the kernel (scripts/wino_ablate.sh): the 16 x 16-byte patch loads of a lane cost 245 us of a 1 290-us launch --
// three times what the U fills of the same volume cost, whatever their addresses: it is the loads' return path into the register file,
// beside the matrix instructions, that is dear.  Here the block's UNIQUE input pixels of a super-stage (a tile's 4 x 4 patch overlaps its
// neighbours': 1 024 pixel loads for ~340-400 distinct pixels) are moved by `buffer_load ... lds` into a third LDS region IN (<= 30 KB,
// single: filled in the even stage, read in the odd one, the stage barriers in between), and a lane reads its patch from there with 16
// ds_read_b128 just before it needs it -- so one 64-register patch set is enough (the second one was there to cover HBM latency).
// MODE 0: patches through registers; 1: staged, regular blocks (full maps); 2: RoI-window launches -- a block of the live-tile list is staged
// when its 64 tiles are live, lie in one image and their bounding pixel region fits IN (decided per workgroup: both loops are in the kernel).
// BN (training): the reductions of the BatchNorm next to the convolution in the epilogue -- WinoArgs.stats / bn_x, the same contract as
// vpho_conv_desc.stats / bn_x (include/vpho_hip.h).  BN = 1 (conv_winograd_bn_kernel): the forward form (sum v | sum v^2: two registers);
// BN = 2 (conv_winograd_bnb_kernel): the backward form (gate recomputed from bn_x, sum v | sum v * xhat: six registers and a pointer -- kept
// out of the forward instantiation, whose output transform already runs with 256 + 256 registers in use); 0: the inference kernels.
template <int MODE, int BN>
__device__ __forceinline__ void wino_body(const WinoArgs& a_) {
    const WinoArgs a = wino_group(a_, blockIdx.y);
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    VPHO_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int wt = wave & 1, wc = wave >> 1;
    // XCD-aware block order: workgroup i runs on XCD i % 8, and every XCD has its own L2.  All tile blocks that share an output-channel
    // block share its 16 x 64 x Cin slice of u (1 MB at Cin = 256), so XCD x takes ONE channel block (x % ncb) and every (8 / ncb)-th
    // tile block: its L2 holds that slice instead of all of u, and tile blocks beyond the last live one (RoI windows) are spread evenly.
    const int ncb = a.Cout / W_CB;
    int tb, cb;
    if (ncb <= 8 && (8 % ncb) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, G = 8 / ncb;
        cb = xcd % ncb; tb = slot * G + xcd / ncb;
    } else { tb = blockIdx.x / ncb; cb = blockIdx.x % ncb; }
    if (tb * W_TB >= a.T) return;                        // grid rounded up to whole XCD rounds
    const int t0 = tb * W_TB, c0 = cb * W_CB;
    if (a.wins && t0 >= a.tile_base[a.N]) return;      // the grid is sized for every tile of the full maps; uniform per workgroup
    // destination of every tile of this block (output transform): row of its top-left pixel in y, row pitch, which of the 4 pixels exist
    __shared__ int s_row[W_TB], s_pitch[W_TB];
    __shared__ float s_bias[W_CB];                       // the block's 64 biases: loaded here, read in the epilogue (no global-load latency there)
    if (tid >= 64 && tid < 64 + W_CB) s_bias[tid - 64] = (a.bias && c0 + tid - 64 < a.Cout) ? a.bias[c0 + tid - 64] : 0.f;
    if (tid < W_TB) {
        int n, ty, tx, row = 0, pitch = (a.W << 4);
        if (wino_tile(a, t0 + tid, n, ty, tx)) {
            if (!a.wins) { row = (n * a.H + 2 * ty) * a.W + 2 * tx; pitch = (a.W << 4) | 0xF; }
            else {
                const int* w = a.wins + 5 * n;
                const int y = 2 * ty - w[1], x = 2 * tx - w[2];              // window coordinates of the tile's top-left pixel (may be -1)
                const int my = (y >= 0 ? 1 : 0) | (y + 1 < w[4] ? 2 : 0), mx = (x >= 0 ? 1 : 0) | (x + 1 < w[3] ? 2 : 0);
                row = a.scatter ? (n * a.H + 2 * ty) * a.W + 2 * tx : w[0] + y * w[3] + x;
                pitch = ((a.scatter ? a.W : w[3]) << 4) | ((my & 1) && (mx & 1) ? 1 : 0) | ((my & 1) && (mx & 2) ? 2 : 0) | ((my & 2) && (mx & 1) ? 4 : 0) | ((my & 2) && (mx & 2) ? 8 : 0);
            }
        }
        s_row[tid] = row; s_pitch[tid] = pitch;
    }

    // ---- V producer: thread = (tile tl, channel pair cp): its 4 x 4 input patch, two channels (8-byte loads)
    // the four channel pairs of a pixel sit on four neighbouring lanes: one wave instruction reads 16 pixels x 32 contiguous bytes
    const int tl = wave * 16 + (lane >> 2), cp = lane & 3;
    int pn = 0, py0 = 0, px0 = 0;
    bool tile_live;
    {
        int ty = 0, tx = 0;
        tile_live = wino_tile(a, t0 + tl, pn, ty, tx);
        py0 = 2 * ty - 1; px0 = 2 * tx - 1;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // patch loads through a buffer resource: the 16 byte offsets are loop constants, a pixel outside the image (or of a dead tile) carries
    // an out-of-range offset and the hardware returns zeros -- no branch per pixel, the channel advance rides in the scalar offset
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0xFFFFFFF0u, 0x00020000);
    int poff[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int iy = py0 + (p >> 2), ix = px0 + (p & 3);
        const bool in = tile_live && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        poff[p] = in ? (int)((((long long)(pn * a.H + iy) * a.W + ix) * a.x_ld + 4 * cp) * 4) : -1;
        if ((WINO_ABLATE & 128) && p > 0) poff[p] = poff[0];          // every patch load hits the same line: instruction count kept, line requests / 16
        if ((WINO_ABLATE & 256)) poff[p] = (int)(((long long)(tl * 16 + p) * 4 + cp) * 16);   // dense, conflict-free addresses: 64 lanes x 16 B contiguous
    }
    // A load is 16 bytes = 4 channels of a 16-channel SUPER-STAGE (half the vector-memory instructions of 8-byte loads, and a wave
    // instruction asks for 64 contiguous bytes of a pixel instead of 32).  A k stage takes one channel PAIR of every lane: stage e
    // (0 / 1) of super-stage ss multiplies channels 16 ss + {0,1,4,5,8,9,12,13} + 2 e -- any 8 channels will do as long as u holds the
    // same ones in the same slots (pack.winograd_weights).  pc: the super-stage being consumed, pnx: the next one, in flight.
    // ---- STAGED: the block's pixel region.  A block is NR = 64 / TW whole tile rows: of one image (NR <= TH), or all TH rows of
    // ipb = NR / TH consecutive images.  Region of a part: RR = 2 rpp + 2 pixel rows (the tile rows' pixels + one halo row above and below)
    // x RC = W + 2 columns; pixel (k, rr, rc) = image n0 + k, row 2 ty0 - 1 + rr, column rc - 1.  IN holds 16 channels = 64 bytes per pixel.
    float* INb = smem + 2 * W_STAGE;
    int doff[8];                                                   // this wave's DMA instructions j = wave, wave + 4, ...: 16 pixels each
    int in_ni = 0;                                                 // DMA instructions of the block
    int rowb[4];                                                   // LDS float offset of this lane's patch rows
    bool stg = MODE == 1;                                          // workgroup-uniform
    if constexpr (MODE == 2) {
        int na = 0, tya = 0, txa = 0, nb = 0, tyb = 0, txb = 0;
        const bool la = wino_tile(a, t0, na, tya, txa), lb = wino_tile(a, t0 + W_TB - 1, nb, tyb, txb);
        if (la && lb && na == nb) {
            const int* w = a.wins + 5 * na;
            const int tx0 = w[2] >> 1, ntx = ((w[2] + w[3] - 1) >> 1) - tx0 + 1;
            const int RR = 2 * (tyb - tya + 1) + 2, RC = 2 * ntx + 2, P = RR * RC;     // the tile rows tya .. tyb of the window, one halo pixel around
            if (P <= W_IN_PIXELS) {
                stg = true;
                in_ni = (P + 15) >> 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = 16 * (wave + 4 * j) + (lane >> 2);
                    const int rr = q / RC, rc = q - rr * RC;
                    const int iy = 2 * tya - 1 + rr, ix = 2 * tx0 - 1 + rc;
                    const bool in = q < P && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    doff[j] = in ? (int)((((long long)(na * a.H + iy) * a.W + ix) * a.x_ld + 4 * (lane & 3)) * 4) : -1;
                }
                const int ty = (py0 + 1) >> 1, tx = (px0 + 1) >> 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) rowb[i] = ((2 * (ty - tya) + i) * RC + 2 * (tx - tx0)) * 16 + 4 * cp;
            }
        }
    }
    if constexpr (MODE == 1) {
        const int NR = W_TB / a.TW, rpp = NR < a.TH ? NR : a.TH, ipb = NR / rpp;
        const int RR = 2 * rpp + 2, RC = a.W + 2, P = ipb * RR * RC;
        const int n0 = t0 / (a.TH * a.TW), ty0 = (t0 - n0 * a.TH * a.TW) / a.TW;
        in_ni = (P + 15) >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 16 * (wave + 4 * j) + (lane >> 2);
            const int k = q / (RR * RC), r2 = q - k * RR * RC, rr = r2 / RC, rc = r2 - rr * RC;
            const int n = n0 + k, iy = 2 * ty0 - 1 + rr, ix = rc - 1;
            const bool in = q < P && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            doff[j] = in ? (int)((((long long)(n * a.H + iy) * a.W + ix) * a.x_ld + 4 * (lane & 3)) * 4) : -1;
        }
        const int k = tl / (rpp * a.TW), rem = tl - k * rpp * a.TW, tyl = rem / a.TW, txl = rem - tyl * a.TW;
#pragma unroll
        for (int i = 0; i < 4; ++i) rowb[i] = ((k * RR + 2 * tyl + i) * RC + 2 * txl) * 16 + 4 * cp;
    }
    // the wave's share of the region's DMA instructions [j0, j1) for super-stage ss (16 channels = 64 bytes per pixel at byte ss * 64)
    auto fill_in = [&](int ss, int j0, int j1) {
        if (WINO_ABLATE & (1 | 32)) return;
#pragma unroll
        for (int j = j0; j < j1; ++j)
            if (wave + 4 * j < in_ni)                               // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(INb + (wave + 4 * j) * 256), 16, doff[j], ss * 64, 0, 0);
    };
    f32x4 pc[16], pnx[16];                                         // pnx: the register path only (unused, hence free, in a staged loop)
    // patch rows [i0, i1) of this lane from IN
    auto read_patch = [&](f32x4 (&dst)[16], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (WINO_ABLATE & 1) { dst[4 * i + j] = f32x4{(float)(4 * i + j), 1.f, 2.f, 3.f}; continue; }
                dst[4 * i + j] = *reinterpret_cast<const f32x4*>(INb + rowb[i] + 16 * j);
            }
    };
    f32x2 r[16];
    auto load_patch = [&](f32x4 (&dst)[16], int ss, int p0 = 0, int p1 = 16) {
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            if ((WINO_ABLATE & 1) || ((WINO_ABLATE & 32) && (p & 1))) { dst[p] = f32x4{(float)p, 1.f, 2.f, 3.f}; continue; }
            dst[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, poff[p], ss * 64, 0));
            if (WINO_ABLATE & 64) { asm volatile("" :: "v"(dst[p])); dst[p] = f32x4{(float)p, 1.f, 2.f, 3.f}; }      // load issued, result unused
        }
    };
    // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows first (row_transform), then columns of one frequency row fy
    auto row_transform = [&](const f32x4 (&src)[16], int e) {
        f32x2 patch[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) patch[p] = e ? f32x2{src[p][2], src[p][3]} : f32x2{src[p][0], src[p][1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r[0 * 4 + c] = patch[0 * 4 + c] - patch[2 * 4 + c];
            r[1 * 4 + c] = patch[1 * 4 + c] + patch[2 * 4 + c];
            r[2 * 4 + c] = patch[2 * 4 + c] - patch[1 * 4 + c];
            r[3 * 4 + c] = patch[1 * 4 + c] - patch[3 * 4 + c];
        }
    };
    const int sw = (tl >> 3) & 1;
    auto store_v_row = [&](int buf, int fy) {
        if (WINO_ABLATE & 4) return;
        float* V = smem + buf * W_STAGE;
        const f32x2 v0 = r[fy * 4 + 0] - r[fy * 4 + 2];
        const f32x2 v1 = r[fy * 4 + 1] + r[fy * 4 + 2];
        const f32x2 v2 = r[fy * 4 + 2] - r[fy * 4 + 1];
        const f32x2 v3 = r[fy * 4 + 1] - r[fy * 4 + 3];
        const f32x2 vv[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
            const int f = fy * 4 + fx;
            // row (f, tl): 8 floats; channel pair cp lives in 16-byte chunk (cp >> 1) ^ sw, 8-byte half cp & 1
            *reinterpret_cast<f32x2*>(V + (f * 64 + tl) * WK + (((cp >> 1) ^ sw) * 4) + (cp & 1) * 2) = vv[fx];
        }
    };
    // ---- U fill: direct to LDS, 32-byte rows: one wave instruction = 32 rows; 16 f x 64 rows = 32 instructions per stage, 8 per wave
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int ustage = 16 * a.Cout * WK * 4;                        // bytes of one k stage of u (stage-tiled: [Cin/8][16][Cout][8])
    int uoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int fr = (wave * 8 + j) * 32 + (lane >> 1);          // flat row over (f, cout-in-block): f = fr / 64, r = fr % 64
        const int f = fr >> 6, r_ = fr & 63;
        const int ch = (lane & 1) ^ ((r_ >> 3) & 1);
        const int co = c0 + r_;
        uoff[j] = co < a.Cout ? (int)((((long long)f * a.Cout + co) * WK + ch * 4) * 4) : -1;       // within the stage's block of u
    }
    auto fill_u = [&](int buf, int kc, int j0 = 0, int j1 = 8) {
        if (WINO_ABLATE & 2) return;
        float* U = smem + buf * W_STAGE + 16 * 64 * WK + wave * 8 * 32 * WK;
#pragma unroll
        for (int j = j0; j < j1; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (lds_ptr)(U + j * 32 * WK), 16, uoff[j], kc * ustage, 0, 0);
    };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;

    const int nss = a.Cin / (2 * WK);                              // super-stages of 16 channels = 2 k stages
    const int fsw = (li >> 3) & 1;
    auto run = [&](auto staged_tag) __attribute__((always_inline)) {
    constexpr bool STAGED = decltype(staged_tag)::value;
    if constexpr (STAGED) {
        fill_in(0, 0, 8);
        fill_u(0, 0);
        VPHO_SYNC_LDS_DMA();
        read_patch(pc, 0, 4);
    } else {
        load_patch(pc, 0);
        fill_u(0, 0);
    }
    VPHO_STAMP_AT(1);
    row_transform(pc, 0);
#pragma unroll
    for (int fy = 0; fy < 4; ++fy) store_v_row(0, fy);
    VPHO_SYNC_LDS_DMA();
    // One wave per SIMD: nothing hides a latency unless the instruction stream does.  A stage is 16 frequency groups of 4 MFMAs (64
    // cycles each); the A / B fragments of a group are requested two groups ahead (register double buffer), and the next stage's work
    // rides in the groups' shadows, ONE memory instruction per MFMA (a burst of vector-memory instructions stalls the wave's issue --
    // and with it the matrix pipe -- for longer than the MFMAs in flight last): in both stages the 8 U fills (groups 0-1), in the EVEN
    // stage of a super-stage the 16 patch loads of the next super-stage (groups 2-5), the row transform (group 11) and one frequency row
    // of the column transform + its 4 LDS writes (groups 12-15).  Branch-free -- the last stages prefetch the last super-stage once
    // more into registers / the idle buffer -- so that the loop body is ONE basic block the scheduler can interleave.
    // (Round 5, measured and not kept: requesting the next stage's first fragments right behind the stage barrier, in FRONT of the current
    // stage's last four matrix instructions -- whose operands are in registers --, so that no LDS round trip is exposed at the boundary:
    // same-box A/B over six layers, 3 interleaved runs: +0.5 ... +1.5 % on five of them (1 306-1 313 against 1 290-1 295 us at 256 -> 256 on
    // 64 x 64), -2 % at 256 -> 256 on 16 x 16.  The boundary's cost is the barrier's skew, not the fragment latency behind it.)
    // (Round 5, the ablations re-run on this kernel -- scripts/wino_ablate.sh, 256 -> 256 on 64 x 64 x 64 images, 1 290 us: without the fragment
    // reads 1 285 (they cost nothing any more), without the stage barrier 1 203, without the U fills 1 219, without the patch loads AND the
    // transforms they feed 1 048, the matrix instructions alone ~1 010.  Of the patch loads' 245 us, 171 go when all 16 loads of a lane hit one
    // line and 74 when the addresses are dense: it is the gather's way through the address / L1 path -- 16 half-lines per wave instruction --,
    // not DRAM latency: requesting the lines of the super-stage after next one super-stage early (four 4-byte `buffer_load ... lds` per even
    // stage into a scratch kilobyte) changed nothing, 1 294-1 299 against 1 272-1 292, and was dropped; so was taking patch columns 2, 3 from
    // the right neighbour's lanes (v_mov_dpp row_shl:4 for 3 tiles of 4, 3/8 fewer line requests): 1 273 against 1 287-1 296, the other layers +-0.)
    auto stage = [&](int kc, int e, int ssn) {
        const int buf = e;                                          // stage kc = 2 ss + e computes from buffer e
        const int kn = kc + 1 < 2 * nss ? kc + 1 : kc;              // the stage whose V / U this one prepares (into buffer e ^ 1)
        const float* V = smem + buf * W_STAGE + (wt * 32 + li) * WK + ((lh ^ fsw) * 4);
        const float* U = smem + buf * W_STAGE + 16 * 64 * WK + (wc * 32 + li) * WK + ((lh ^ fsw) * 4);
        f32x4 av[2], bv[2];
        av[0] = *reinterpret_cast<const f32x4*>(V);
        bv[0] = *reinterpret_cast<const f32x4*>(U);
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if (f < 15 && !(WINO_ABLATE & 16)) {
                av[(f + 1) & 1] = *reinterpret_cast<const f32x4*>(V + (f + 1) * 64 * WK);
                bv[(f + 1) & 1] = *reinterpret_cast<const f32x4*>(U + (f + 1) * 64 * WK);
            }
            // the U fills FIRST, then the patch loads: vector-memory operations complete in order, so the end-of-stage wait can let the 16
            // patch loads stay in flight (vmcnt(16)) and still know the fills have landed -- the patch has until the row transform of the
            // NEXT stage to arrive (with vmcnt(0) here it had to beat the barrier of its own stage: 12 % of the kernel's time)
            if (f < 2) fill_u(buf ^ 1, kn, 4 * f, 4 * f + 4);
            if constexpr (STAGED) {
                // even stage: the next super-stage's pixels into IN (every wave has read the previous ones: the barrier behind the odd stage);
                // odd stage: this lane's patch from IN (they landed before the barrier behind the even stage), a patch row per two groups
                if (e == 0 && f >= 2 && f < 6) fill_in(ssn, 2 * (f - 2), 2 * (f - 2) + 2);
                if (e == 1 && f >= 2 && f < 10 && !(f & 1)) read_patch(pc, (f - 2) >> 1, ((f - 2) >> 1) + 1);
                if (f == 11) row_transform(pc, e == 0 ? 1 : 0);
            } else {
                if (e == 0 && f >= 2 && f < 6) load_patch(pnx, ssn, 4 * (f - 2), 4 * (f - 2) + 4);
                if (f == 11) { if (e == 0) row_transform(pc, 1); else row_transform(pnx, 0); }
            }
            if (f >= 12) store_v_row(buf ^ 1, f - 12);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[f & 1][q], bv[f & 1][q], acc[f], 0, 0, 0);
            // order inside the group: the next fragments first, then each MFMA followed by a share of the group's other work
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(WINO_ABLATE & 8)) {
            if (!STAGED && e == 0 && !(WINO_ABLATE & 1)) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); __syncthreads(); }
            else VPHO_SYNC_LDS_DMA();
        }
    };
    VPHO_STAMP_AT(2);
    VPHO_PRIO_MAIN();
    for (int ss = 0; ss < nss; ++ss) {
        const int ssn = ss + 1 < nss ? ss + 1 : ss;
        stage(2 * ss, 0, ssn);
        stage(2 * ss + 1, 1, ssn);
        if constexpr (!STAGED) {
#pragma unroll
            for (int p = 0; p < 16; ++p) pc[p] = pnx[p];
        }
    }
    };
    if constexpr (MODE == 0) run(std::false_type{});
    else if constexpr (MODE == 1) run(std::true_type{});
    else { if (stg) run(std::true_type{}); else run(std::false_type{}); }
    VPHO_PRIO_REST();
    VPHO_STAMP_AT(3);

    // ---- output transform on the accumulators: A^T = [1 1 1 0; 0 1 -1 -1]; row e -> tile, lane -> output channel
    const int co = c0 + wc * 32 + li;
    const float bias = s_bias[wc * 32 + li];
    // the 16 destination records of this lane's tile rows in ONE LDS round trip (the patch registers are dead by now): read one by one
    // inside the loop each was a dependent LDS latency in front of its stores
    // (the records of the lane's tile rows are fetched in two batches of eight: sixteen at once held 32 registers on top of the 256
    // accumulators being read out, and the one dword that did not fit went through scratch -- a kernel with a private segment pays for it
    // at every wave launch, not only at the spill)
    int pm_e[16], row_e[16];
    auto fetch_records = [&](int e0) {
#pragma unroll
        for (int e = e0; e < e0 + 8; ++e) {
            const int trow = wt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            pm_e[e] = s_pitch[trow]; row_e[e] = s_row[trow];
        }
    };
    fetch_records(0);
    // BN: a lane owns one output channel and 16 tiles x 4 pixels of it -- its share of the column sums is two registers
    float st0 = 0.f, st1 = 0.f, bn_m = 0.f, bn_i = 0.f, bn_g = 0.f, bn_b = 0.f;
    constexpr int BNX_FLOATS = W_TB * 4 * W_CB;                      // the block's BatchNorm-input tile: 64 tiles x 4 pixels x 64 channels = 64 KB
    if constexpr (BN == 2) {
        bn_m = a.bn_mean[co]; bn_i = a.bn_invstd[co]; bn_g = a.bn_gamma[co]; bn_b = a.bn_beta[co];
        // The BatchNorm input of the block's 256 output pixels comes into the (idle) stage buffers by LDS-DMA, 64 instructions of 1 KB
        // (one tile = 4 pixels x 64 channels each), one memory round trip for the workgroup.  Read element by element through registers
        // instead -- 64 dependent 4-byte loads per lane with 256 + 256 registers in use, a handful in flight at a time -- the same data cost
        // the gated input-gradient launches +40 % (3 x 3 64 -> 64 on 64 x 64 maps: 182 us against 128 for the forward convolution).
        static_assert(BNX_FLOATS <= 2 * W_STAGE, "BatchNorm-input tile does not fit the stage buffers");
        const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bn_x), 0, 0xFFFFFFF0u, 0x00020000);
        const int pl = lane >> 4, cq = (lane & 15) * 4;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int t = wave * 16 + i;
            const int pmv = s_pitch[t], rw = s_row[t];
            const int pitch = pmv >> 4;
            const int poff = pl == 0 ? 0 : pl == 1 ? 1 : pl == 2 ? pitch : pitch + 1;
            const bool live = ((pmv >> pl) & 1) && c0 + cq < a.Cout;
            const int off = live ? (int)(unsigned)((((long long)rw + poff) * a.y_ld + c0 + cq) * 4) : -1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr2, (__attribute__((address_space(3))) void*)(smem + t * 256), 16, off, 0, 0, 0);
        }
        VPHO_SYNC_LDS_DMA();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (e == 4) fetch_records(8);
        float s0[4], s1[4];
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
            s0[fx] = acc[0 * 4 + fx][e] + acc[1 * 4 + fx][e] + acc[2 * 4 + fx][e];
            s1[fx] = acc[1 * 4 + fx][e] - acc[2 * 4 + fx][e] - acc[3 * 4 + fx][e];
        }
        const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
        const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
        const int pm = pm_e[e];
        if ((pm & 0xF) && co < a.Cout) {
            const int pitch = pm >> 4;
            float* yp = a.y + (long long)row_e[e] * a.y_ld + co;
            const float o[4] = {y00 + bias, y01 + bias, y10 + bias, y11 + bias};
            const long long offs[4] = {0, (long long)a.y_ld, (long long)pitch * a.y_ld, (long long)(pitch + 1) * a.y_ld};
            const float* gp = a.gate ? a.gate + (long long)row_e[e] * a.y_ld + co : nullptr;
            const float* xp = nullptr;                               // BN == 2: this lane's channel of the tile's four pixels, in LDS
            if constexpr (BN == 2) xp = smem + (wt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 256 + wc * 32 + li;
#pragma unroll
            for (int p = 0; p < 4; ++p) if ((pm >> p) & 1) {
                float v = o[p];
                if (gp) v = gp[offs[p]] > 0.f ? v : v * a.gate_slope;
                if ((WINO_ABLATE & 512) && v != 1.2345e-30f) continue;          // timing: the output transform without its stores
                if constexpr (BN == 2) {
                    // gate recomputed from the BatchNorm input: the forward pass's own expression (bn_apply_kernel), hence its sign
                    const float xh = (xp[p * W_CB] - bn_m) * bn_i;
                    const float t = xh * bn_g + bn_b;
                    v = t > 0.f ? v : v * a.gate_slope;
                    st0 += v; st1 += v * xh;
                    yp[offs[p]] = v;
                    continue;
                }
                if constexpr (BN == 1) {
                    v = v > 0.f ? v : v * a.out_slope;
                    st0 += v; st1 += v * v;
                    yp[offs[p]] = v;
                    continue;
                }
                yp[offs[p]] = v > 0.f ? v : v * a.out_slope;
            }
        }
    }
    if constexpr (BN != 0) {
        // lanes li / li + 32 hold the two row halves of a channel, the waves wt = 0 / 1 the two tile halves: one partial row per tile
        // block, [tb][2][Cout], combined in a fixed order (the stage buffers are free: every wave has passed the last stage's barrier)
        st0 += __shfl_xor(st0, 32); st1 += __shfl_xor(st1, 32);
        float* S = smem + BNX_FLOATS;                                // (behind the BatchNorm-input tile other waves may still be reading)
        if (lh == 0) { S[(wt * 2 + 0) * W_CB + wc * 32 + li] = st0; S[(wt * 2 + 1) * W_CB + wc * 32 + li] = st1; }
        __syncthreads();
        if (tid < 2 * W_CB) {
            const int pl = tid / W_CB, c = tid - pl * W_CB;
            a.stats[((long long)tb * 2 + pl) * a.Cout + c0 + c] = S[pl * W_CB + c] + S[(2 + pl) * W_CB + c];
        }
    }
    VPHO_STAMP_AT(4);
    VPHO_STAMP_WRITE(wino, blockIdx.x);
}
template <int MODE>
__global__ __launch_bounds__(256) void conv_winograd_kernel(const WinoArgs a_) { wino_body<MODE, 0>(a_); }
template <int MODE>
__global__ __launch_bounds__(256) void conv_winograd_bn_kernel(const WinoArgs a_) { wino_body<MODE, 1>(a_); }
template <int MODE>
__global__ __launch_bounds__(256) void conv_winograd_bnb_kernel(const WinoArgs a_) { wino_body<MODE, 2>(a_); }


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4 experiment, used for the short-K layers only (see wino_launch): the same arithmetic with TWO waves per SIMD.  conv_winograd_kernel keeps all 16 frequencies of a 32 x 32 (tile, channel) block
// in one wave: 256 accumulator registers, ONE wave per SIMD -- whatever that wave does besides matrix instructions (patch loads, the
// input transform, V stores, waits) is time the matrix pipe idles (PMC: 0.61 busy; the floor of a pure MFMA loop is 0.79 of the kernel's
// time).  Here a workgroup is 8 waves: wave (q, fh) holds the frequencies fy in {2 fh, 2 fh + 1} (8 of 16: 128 accumulators) of the
// 32 x 32 block q = (wt, wc).  The two waves that share a SIMD alternate roles: the four waves of one frequency half PRODUCE V for the
// even 16-channel super-stages, the other four for the odd ones (patch loads two stages ahead of their transform: one 64-register patch
// buffer, no second one), so a wave's load / transform stretches meet its SIMD partner's matrix instructions.  Same stage layout in LDS,
// same U, same k order, same transform expressions: the output is bit-identical to conv_winograd_kernel.  Only the output transform
// needs an exchange, once per workgroup: A^T M A mixes all four fy, so each wave hands its partner the half of its accumulators the
// partner finishes (16 KB per wave through the idle stage buffers) and transforms the other half of the tile rows itself.
__global__ __launch_bounds__(512) void conv_winograd8_kernel(const WinoArgs a_) {
    const WinoArgs a = wino_group(a_, blockIdx.y);
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int q = wave & 3, fh = wave >> 2, wt = q & 1, wc = q >> 1;
    const int ncb = a.Cout / W_CB;
    int tb, cb;
    if (ncb <= 8 && (8 % ncb) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, G = 8 / ncb;
        cb = xcd % ncb; tb = slot * G + xcd / ncb;
    } else { tb = blockIdx.x / ncb; cb = blockIdx.x % ncb; }
    if (tb * W_TB >= a.T) return;
    const int t0 = tb * W_TB, c0 = cb * W_CB;
    if (a.wins && t0 >= a.tile_base[a.N]) return;
    __shared__ int s_row[W_TB], s_pitch[W_TB];
    if (tid < W_TB) {
        int n, ty, tx, row = 0, pitch = (a.W << 4);
        if (wino_tile(a, t0 + tid, n, ty, tx)) {
            if (!a.wins) { row = (n * a.H + 2 * ty) * a.W + 2 * tx; pitch = (a.W << 4) | 0xF; }
            else {
                const int* w = a.wins + 5 * n;
                const int y = 2 * ty - w[1], x = 2 * tx - w[2];
                const int my = (y >= 0 ? 1 : 0) | (y + 1 < w[4] ? 2 : 0), mx = (x >= 0 ? 1 : 0) | (x + 1 < w[3] ? 2 : 0);
                row = a.scatter ? (n * a.H + 2 * ty) * a.W + 2 * tx : w[0] + y * w[3] + x;
                pitch = ((a.scatter ? a.W : w[3]) << 4) | ((my & 1) && (mx & 1) ? 1 : 0) | ((my & 1) && (mx & 2) ? 2 : 0) | ((my & 2) && (mx & 1) ? 4 : 0) | ((my & 2) && (mx & 2) ? 8 : 0);
            }
        }
        s_row[tid] = row; s_pitch[tid] = pitch;
    }

    // ---- V producer (the four waves of one frequency half at a time): thread = (tile tl, channel quad cp) of a 16-channel super-stage
    const int tl = q * 16 + (lane >> 2), cp = lane & 3;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0xFFFFFFF0u, 0x00020000);
    // patch addressing without a 16-register offset table: the offset of the patch's (virtual) top-left pixel + one validity bit per pixel
    int pbase;
    unsigned pmask = 0;
    {
        int pn = 0, ty = 0, tx = 0;
        const bool tile_live = wino_tile(a, t0 + tl, pn, ty, tx);
        const int py0 = 2 * ty - 1, px0 = 2 * tx - 1;
        pbase = (int)((((long long)(pn * a.H + py0) * a.W + px0) * a.x_ld + 4 * cp) * 4);       // mod 2^32; only used where the pixel is valid
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int iy = py0 + (p >> 2), ix = px0 + (p & 3);
            if (tile_live && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) pmask |= 1u << p;
        }
    }
    const int row_b = a.W * a.x_ld * 4, pix_b = a.x_ld * 4;
    f32x4 pc[16];
    auto load_patch = [&](int ss) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int live = -(int)((pmask >> p) & 1u);                    // all ones / zero: no branch per pixel
            const int off = ((pbase + (p >> 2) * row_b + (p & 3) * pix_b) & live) | ~live;                // dead pixel: offset -1, out of range, the hardware returns zeros
            pc[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, ss * 64, 0));
        }
    };
    const int sw = (tl >> 3) & 1;
    // B^T d B of channel pair e of the patch buffer -> V of one k stage (the expressions of conv_winograd_kernel::row_transform / store_v_row)
    auto produce = [&](int e, int buf) {
        float* V = smem + buf * W_STAGE;
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) {
            f32x2 r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 p0 = e ? f32x2{pc[0 * 4 + c][2], pc[0 * 4 + c][3]} : f32x2{pc[0 * 4 + c][0], pc[0 * 4 + c][1]};
                const f32x2 p1 = e ? f32x2{pc[1 * 4 + c][2], pc[1 * 4 + c][3]} : f32x2{pc[1 * 4 + c][0], pc[1 * 4 + c][1]};
                const f32x2 p2 = e ? f32x2{pc[2 * 4 + c][2], pc[2 * 4 + c][3]} : f32x2{pc[2 * 4 + c][0], pc[2 * 4 + c][1]};
                const f32x2 p3 = e ? f32x2{pc[3 * 4 + c][2], pc[3 * 4 + c][3]} : f32x2{pc[3 * 4 + c][0], pc[3 * 4 + c][1]};
                r[c] = fy == 0 ? p0 - p2 : fy == 1 ? p1 + p2 : fy == 2 ? p2 - p1 : p1 - p3;
            }
            const f32x2 vv[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
#pragma unroll
            for (int fx = 0; fx < 4; ++fx)
                *reinterpret_cast<f32x2*>(V + ((fy * 4 + fx) * 64 + tl) * WK + (((cp >> 1) ^ sw) * 4) + (cp & 1) * 2) = vv[fx];
        }
    };
    // ---- U fill: 32 wave instructions of 1 KB per stage, 4 per wave
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int ustage = 16 * a.Cout * WK * 4;
    int uoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int fr = (wave * 4 + j) * 32 + (lane >> 1);
        const int f = fr >> 6, r_ = fr & 63;
        const int ch = (lane & 1) ^ ((r_ >> 3) & 1);
        const int co = c0 + r_;
        uoff[j] = co < a.Cout ? (int)((((long long)f * a.Cout + co) * WK + ch * 4) * 4) : -1;
    }
    auto fill_u = [&](int buf, int kc) {
        float* U = smem + buf * W_STAGE + 16 * 64 * WK + wave * 4 * 32 * WK;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (lds_ptr)(U + j * 32 * WK), 16, uoff[j], kc * ustage, 0, 0);
    };

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;

    const int nss = a.Cin / (2 * WK), nst = 2 * nss;
    const int fsw = (li >> 3) & 1;
    // prologue: group fh owns the super-stages ss = fh (mod 2).  Group 0 produces V of stage 0; group 1's first patches are on their way
    fill_u(0, 0);
    if (fh == 0 || nss > 1) load_patch(fh);
    if (fh == 0) produce(0, 0);
    VPHO_SYNC_LDS_DMA();
    for (int kc = 0; kc < nst; ++kc) {
        const int buf = kc & 1;
        const int ph = (kc + 2 * fh) & 3;               // 1: request the patches of super-stage (kc + 3) / 2; 2: they travel; 3 / 0: transform pair 0 / 1
        // transform phases FIRST consume their patches (landed two stages ago) and only then request U: vector-memory operations complete
        // in order, so a wait for a patch register behind a freshly issued fill would wait for the fill; the request phase issues the
        // fill first, so that its stage-end wait (vmcnt(16)) covers the fill and lets the 16 patch loads fly
        const bool loads = ph == 1 && (kc + 3) / 2 < nss && !(WINO8_ABLATE & 2);
        if (ph == 3) { if ((kc + 1) / 2 < nss && !(WINO8_ABLATE & 1)) produce(0, buf ^ 1); }            // V of stage kc + 1 = 2 ss
        else if (ph == 0 && !(WINO8_ABLATE & 1)) produce(1, buf ^ 1);                                    // V of stage kc + 1 = 2 ss + 1
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nst && !(WINO8_ABLATE & 4)) fill_u(buf ^ 1, kc + 1);
        if (loads) load_patch((kc + 3) / 2);
        __builtin_amdgcn_sched_barrier(0);
        const float* V = smem + buf * W_STAGE + ((fh * 8) * 64 + wt * 32 + li) * WK + ((lh ^ fsw) * 4);
        const float* U = smem + buf * W_STAGE + 16 * 64 * WK + ((fh * 8) * 64 + wc * 32 + li) * WK + ((lh ^ fsw) * 4);
        f32x4 av[2], bv[2];
        av[0] = *reinterpret_cast<const f32x4*>(V);
        bv[0] = *reinterpret_cast<const f32x4*>(U);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (f < 7) {
                av[(f + 1) & 1] = *reinterpret_cast<const f32x4*>(V + (f + 1) * 64 * WK);
                bv[(f + 1) & 1] = *reinterpret_cast<const f32x4*>(U + (f + 1) * 64 * WK);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[f & 1][k], bv[f & 1][k], acc[f], 0, 0, 0);
            // the next frequency's fragments are REQUESTED before this frequency's four matrix instructions issue (left alone the compiler
            // reads two frequencies, waits, and issues their eight MFMAs: an LDS round trip exposed every 512 cycles)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // U of the next stage must have landed; patches requested in this stage (after the fills) may stay in flight
        if (WINO8_ABLATE & 16) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (loads) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); __syncthreads(); }
        else VPHO_SYNC_LDS_DMA();
    }

    // ---- output transform.  Wave (q, fh) finishes the accumulator rows e in [8 fh, 8 fh + 8) of its 32 x 32 block; the other 8 rows of
    // its 8 frequencies go to the partner (q, 1 - fh) through LDS: [wave][f 0..7][e 0..7][lane]
    float* X = smem;
    {
        // [wave][e8 0..7][frequency quad 0..1][lane][4]: 16-byte stores and loads, lane-contiguous (conflict-free)
        f32x4* mine = reinterpret_cast<f32x4*>(X + wave * (8 * 8 * 64)) + lane;
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8)
#pragma unroll
            for (int hv = 0; hv < 2; ++hv) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fh ? acc[4 * hv + k][e8] : acc[4 * hv + k][8 + e8];
                mine[(e8 * 2 + hv) * 64] = v;
            }
    }
    __syncthreads();
    const f32x4* theirs = reinterpret_cast<const f32x4*>(X + (wave ^ 4) * (8 * 8 * 64)) + lane;
    const int co = c0 + wc * 32 + li;
    const float bias = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) {
        const int e = 8 * fh + e8;
        const int trow = wt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float m[16];                                     // all 16 frequencies of this (tile, channel): own half from registers, the other from LDS
        const f32x4 t0 = theirs[(e8 * 2 + 0) * 64], t1 = theirs[(e8 * 2 + 1) * 64];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float own = fh ? acc[f][8 + e8] : acc[f][e8];
            const float oth = f < 4 ? t0[f & 3] : t1[f & 3];
            m[fh * 8 + f] = own;
            m[(1 - fh) * 8 + f] = oth;
        }
        float s0[4], s1[4];
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
            s0[fx] = m[0 * 4 + fx] + m[1 * 4 + fx] + m[2 * 4 + fx];
            s1[fx] = m[1 * 4 + fx] - m[2 * 4 + fx] - m[3 * 4 + fx];
        }
        const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
        const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
        const int pm = s_pitch[trow];
        if ((pm & 0xF) && co < a.Cout) {
            const int pitch = pm >> 4;
            float* yp = a.y + (long long)s_row[trow] * a.y_ld + co;
            const float o[4] = {y00 + bias, y01 + bias, y10 + bias, y11 + bias};
            const long long offs[4] = {0, (long long)a.y_ld, (long long)pitch * a.y_ld, (long long)(pitch + 1) * a.y_ld};
            const float* gp = a.gate ? a.gate + (long long)s_row[trow] * a.y_ld + co : nullptr;
#pragma unroll
            for (int p = 0; p < 4; ++p) if ((pm >> p) & 1) {
                float v = o[p];
                if (gp) v = gp[offs[p]] > 0.f ? v : v * a.gate_slope;
                yp[offs[p]] = v > 0.f ? v : v * a.out_slope;
            }
        }
    }
}

}  // namespace

// tile_base[n] = first tile of image n, tile_base[N] = number of live tiles: the 2 x 2 tiles of the even grid that touch window n
__global__ void wino_tile_base_kernel(const int* __restrict__ wins, int N, int* __restrict__ tile_base) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int base = 0;
    for (int n = 0; n < N; ++n) {
        const int* w = wins + 5 * n;
        tile_base[n] = base;
        base += (((w[1] + w[4] - 1) >> 1) - (w[1] >> 1) + 1) * (((w[2] + w[3] - 1) >> 1) - (w[2] >> 1) + 1);
    }
    tile_base[N] = base;
}

static int wino_launch(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout, float out_slope,
                       const int* wins, const int* tile_base, int tiles_hint, float* y, int y_ld, void* stream,
                       const float* gate = nullptr, float gate_slope = 1.f, int scatter = 0, int groups = 1, long long x_group = 0,
                       const struct WinoBn* bn = nullptr);

// U = G g G^T on the DEVICE for weights that change every step (training): one thread per (output channel, input channel) pair of the
// convolution the result is for.  mode 0: that convolution is the forward one (w packed as [Cout][(r*3+s)*Cin + ci]); mode 1: its
// input-gradient convolution = a 3x3 convolution of dY with the spatially flipped, channel-transposed weights (Cin outputs, Cout
// inputs).  Written straight into the stage-tiled layout of pack.winograd_weights (same slots, same channel order).
__device__ __forceinline__ void wino_weights_item(const float* __restrict__ w, int Cout, int Cin, int mode, float* __restrict__ u, const long long i) {
    const int OUT = mode ? Cin : Cout, IN = mode ? Cout : Cin;
    if (i >= (long long)OUT * IN) return;
    // forward: neighbouring threads = neighbouring ci (contiguous reads); transposed: neighbouring threads = neighbouring OUT = ci as well
    const int o = mode ? (int)(i % OUT) : (int)(i / IN), c = mode ? (int)(i / OUT) : (int)(i % IN);
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            g[r][q] = mode ? (double)w[(long long)c * 9 * Cin + ((2 - r) * 3 + (2 - q)) * Cin + o] : (double)w[(long long)o * 9 * Cin + (r * 3 + q) * Cin + c];
    double t[4][3];                                             // G g
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        t[0][q] = g[0][q]; t[1][q] = 0.5 * (g[0][q] + g[1][q] + g[2][q]); t[2][q] = 0.5 * (g[0][q] - g[1][q] + g[2][q]); t[3][q] = g[2][q];
    }
    const int kc = 2 * (c >> 4) + ((c & 3) >> 1), j = 2 * ((c & 15) >> 2) + (c & 1);
#pragma unroll
    for (int fy = 0; fy < 4; ++fy) {
        const double v[4] = {t[fy][0], 0.5 * (t[fy][0] + t[fy][1] + t[fy][2]), 0.5 * (t[fy][0] - t[fy][1] + t[fy][2]), t[fy][2]};
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) u[(((long long)kc * 16 + fy * 4 + fx) * OUT + o) * 8 + j] = (float)v[fx];
    }
}
__global__ void wino_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int mode, float* __restrict__ u) {
    wino_weights_item(w, Cout, Cin, mode, u, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
// the same for a LIST of weight tensors in one launch (training: every 3x3 weight changes in the optimiser step, its forward and its
// input-gradient transform are both due -- 81 launches of ~6 us per step otherwise); a workgroup finds its record by its first block
struct WinoWSeg { const float* w; float* u; int Cout, Cin, mode, blk0; };
__global__ __launch_bounds__(256) void wino_weights_multi_kernel(const WinoWSeg* __restrict__ segs, int nseg) {
    int lo = 0, hi = nseg - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                   // last record with blk0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    const WinoWSeg sg = segs[lo];
    wino_weights_item(sg.w, sg.Cout, sg.Cin, sg.mode, sg.u, (long long)(b - sg.blk0) * 256 + threadIdx.x);
}

extern "C" int vpho_winograd_window_tiles_i32(const int* wins, int N, int* tile_base, void* stream) {
    VPHO_REQUIRE(wins && tile_base && N > 0, "vpho_winograd_window_tiles_i32: bad argument");
    hipLaunchKernelGGL(wino_tile_base_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, wins, N, tile_base);
    return vpho::check_launch("wino_tile_base_kernel");
}

extern "C" int vpho_conv3x3_winograd_rows_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld,
                                                   int Cout, float out_slope, const int* wins, const int* tile_base, int tiles_hint,
                                                   float* y_rows, int y_ld, void* stream) {
    VPHO_REQUIRE(wins && tile_base, "vpho_conv3x3_winograd_rows_nhwc_f32: bad argument");
    return wino_launch(x, u, bias, N, H, W, Cin, x_ld, Cout, out_slope, wins, tile_base, tiles_hint, y_rows, y_ld, stream);
}

extern "C" int vpho_conv3x3_winograd_scatter_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld,
                                                      int Cout, float out_slope, const int* wins, const int* tile_base, float* y, int y_ld, void* stream) {
    VPHO_REQUIRE(wins && tile_base, "vpho_conv3x3_winograd_scatter_nhwc_f32: bad argument");
    return wino_launch(x, u, bias, N, H, W, Cin, x_ld, Cout, out_slope, wins, tile_base, 0, y, y_ld, stream, nullptr, 1.f, 1);
}

extern "C" int vpho_conv3x3_winograd_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                              float out_slope, float* y, int y_ld, void* stream) {
    return wino_launch(x, u, bias, N, H, W, Cin, x_ld, Cout, out_slope, nullptr, nullptr, 0, y, y_ld, stream);
}

extern "C" int vpho_conv3x3_winograd_grouped_nhwc_f32(const float* x, long long x_group, const float* u, const float* bias, int groups, int N, int H, int W,
                                                      int Cin, int x_ld, int Cout, float out_slope, float* y, int y_ld, void* stream) {
    VPHO_REQUIRE(groups >= 1, "vpho_conv3x3_winograd_grouped_nhwc_f32: bad argument");
    return wino_launch(x, u, bias, N, H, W, Cin, x_ld, Cout, out_slope, nullptr, nullptr, 0, y, y_ld, stream, nullptr, 1.f, 0, groups, x_group);
}

extern "C" int vpho_conv3x3_winograd_gate_nhwc_f32(const float* x, const float* u, const float* gate, float gate_slope, int N, int H, int W, int Cin,
                                                   int x_ld, int Cout, float* y, int y_ld, void* stream) {
    VPHO_REQUIRE(gate, "vpho_conv3x3_winograd_gate_nhwc_f32: bad argument");
    return wino_launch(x, u, nullptr, N, H, W, Cin, x_ld, Cout, 1.f, nullptr, nullptr, 0, y, y_ld, stream, gate, gate_slope);
}

extern "C" int vpho_conv3x3_winograd_stats_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                                    float out_slope, float* y, int y_ld, float* stats, int stats_cap, int* stats_rows, const float* bn_x,
                                                    const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                                    float gate_slope, void* stream) {
    VPHO_REQUIRE(stats && stats_rows && stats_cap > 0, "vpho_conv3x3_winograd_stats_nhwc_f32: bad argument");
    *stats_rows = 0;
    const char* w8 = getenv("VPHO_WINO8");
    VPHO_REQUIRE(!(w8 && atoi(w8) != 0), "vpho_conv3x3_winograd_stats_nhwc_f32: not served by the 8-wave kernel (VPHO_WINO8)");
    const WinoBn bn{stats, stats_cap, stats_rows, bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta};
    return wino_launch(x, u, bias, N, H, W, Cin, x_ld, Cout, out_slope, nullptr, nullptr, 0, y, y_ld, stream, nullptr, gate_slope, 0, 1, 0, &bn);
}

extern "C" int vpho_winograd_weights_f32(const float* w_packed, int Cout, int Cin, int for_input_gradient, float* u, void* stream) {
    VPHO_REQUIRE(w_packed && u && Cout > 0 && Cin > 0, "vpho_winograd_weights_f32: bad argument");
    VPHO_REQUIRE((for_input_gradient ? Cout : Cin) % 16 == 0, "vpho_winograd_weights_f32: the convolution's input channels must be a multiple of 16");
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, Cout, Cin, for_input_gradient ? 1 : 0, u);
    return vpho::check_launch("wino_weights_kernel");
}

extern "C" int vpho_winograd_weights_multi_f32(const void* segments, int n_segments, long long total_blocks, void* stream) {
    VPHO_REQUIRE(segments && n_segments > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "vpho_winograd_weights_multi_f32: bad argument");
    static_assert(sizeof(WinoWSeg) == 32, "record = 2 pointers + 4 int32");
    hipLaunchKernelGGL(wino_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const WinoWSeg*)segments, n_segments);
    return vpho::check_launch("wino_weights_multi_kernel");
}

static int wino_launch(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout, float out_slope,
                       const int* wins, const int* tile_base, int tiles_hint, float* y, int y_ld, void* stream,
                       const float* gate, float gate_slope, int scatter, int groups, long long x_group, const WinoBn* bn) {
    VPHO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "vpho_conv3x3_winograd_nhwc_f32: bad argument");
    VPHO_REQUIRE(H % 2 == 0 && W % 2 == 0 && Cin % (2 * WK) == 0 && Cout % W_CB == 0 && x_ld % 4 == 0 && x_ld >= Cin && y_ld >= Cout,
                 "vpho_conv3x3_winograd_nhwc_f32: needs even H, W, Cin %% 16 == 0, Cout %% 64 == 0, x_ld %% 4 == 0");
    VPHO_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)u & 15) == 0 && 64.0 * Cout * Cin < 3.9e9 && 4.0 * N * H * W * x_ld < 3.9e9,
                 "vpho_conv3x3_winograd_nhwc_f32: alignment / size (16-byte aligned x and u, tensors below 3.9 GB: 32-bit buffer offsets)");
    WinoArgs a;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.x_ld = x_ld; a.Cout = Cout; a.y_ld = y_ld;
    a.TH = H / 2; a.TW = W / 2; a.T = N * a.TH * a.TW; a.out_slope = out_slope;
    a.wins = wins; a.tile_base = tile_base; a.gate = gate; a.gate_slope = gate_slope; a.scatter = scatter;
    // groups: N images per group; u (groups, Cin/8, 16, Cout, 8), bias (groups, Cout), y (groups * N, H, W, y_ld); x_group = 0: a shared input
    VPHO_REQUIRE(groups >= 1 && (groups == 1 || (!wins && !gate && x_group >= 0 && x_group % 4 == 0)), "vpho_conv3x3_winograd_nhwc_f32: grouped launches take no windows / gate");
    a.x_gs = groups > 1 ? x_group : 0; a.u_gs = groups > 1 ? 16ll * Cout * Cin : 0; a.b_gs = groups > 1 ? Cout : 0;
    a.y_gs = groups > 1 ? (long long)N * H * W * y_ld : 0;
    a.stats = nullptr; a.bn_x = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr;
    const size_t lds = (size_t)2 * W_STAGE * sizeof(float);
    const size_t lds_staged = lds + (size_t)W_IN_PIXELS * 64;
    VPHO_DYN_LDS(conv_winograd_kernel<0>, lds);
    VPHO_DYN_LDS(conv_winograd_kernel<1>, lds_staged);
    VPHO_DYN_LDS(conv_winograd_kernel<2>, lds_staged);
    const int tbs = (a.T + W_TB - 1) / W_TB;
    // executed flops: 16 GEMMs of T x Cout x Cin (the direct 3x3 would be 2.25 x this)
    // (with windows: the live tiles when the caller knows them -- tiles_hint, profiling passes only --, else all)
    const double tl = (wins && tiles_hint > 0) ? (double)tiles_hint : (double)a.T;
    vpho::ProfScope prof(vpho::PROF_WINOGRAD, (hipStream_t)stream, groups * 2.0 * 16.0 * tl * Cout * (double)Cin,
                         groups * 4.0 * (4.0 * tl * Cin + 16.0 * Cout * Cin + 4.0 * tl * Cout));
    const int ncb = Cout / W_CB;
    unsigned blocks = (unsigned)(tbs * ncb);
    if (ncb <= 8 && (8 % ncb) == 0) { const int G = 8 / ncb; blocks = (unsigned)((tbs + G - 1) / G) * 8u; }
    // Which kernel: measured on MI355X (scripts/wino_bench.py, profiles/r04_winograd_two_waves.txt) the 8-wave kernel wins only where a
    // tile has few k stages (Cin = 64: 124 against 131 us at 64 x 64 x 64 images) -- its per-stage barrier joins 8 unequally loaded waves
    // and costs 14 % of its time against 4 % for the 4-wave kernel -- and loses 1 ... 10 % on the long-K layers.  VPHO_WINO8 = 1 / 0 forces
    // either kernel (read per call; the two are bit-identical, tests/test_gpu_conv.py).
    // Round 5: with the halved epilogue and the input staged through LDS the 4-wave kernel wins there as well (64 -> 64 on 64 x 64 x 64
    // images: 113 us staged, 115 through registers, 124 for the 8-wave kernel), so the 8-wave kernel is only run on request.
    const char* w8 = getenv("VPHO_WINO8");
    const bool use8 = w8 ? atoi(w8) != 0 : false;
    if (!use8) {
        // input patches through LDS where the blocks are whole tile rows of full maps (see the kernel); VPHO_WINO_STAGED=0: registers (A/B aid,
        // bit-identity test; read per call)
        const char* st = getenv("VPHO_WINO_STAGED");
        const bool staged = st ? atoi(st) != 0 : true;
        if (bn) {
            // training: BatchNorm reductions in the epilogue, one partial row per tile block (full maps only)
            VPHO_REQUIRE(bn->stats && bn->rows && !wins && !gate && groups == 1 && ((uintptr_t)bn->stats & 15) == 0,
                         "vpho_conv3x3_winograd_stats_nhwc_f32: full maps, one group, no stored gate");
            VPHO_REQUIRE(tbs <= bn->cap, "vpho_conv3x3_winograd_stats_nhwc_f32: %d partial rows exceed stats_cap %d", tbs, bn->cap);
            VPHO_REQUIRE(!bn->x || (bn->mean && bn->invstd && bn->gamma && bn->beta), "vpho_conv3x3_winograd_stats_nhwc_f32: bn_x needs the four BatchNorm vectors");
            VPHO_REQUIRE(!bn->x || (((uintptr_t)bn->x & 15) == 0 && y_ld % 4 == 0 && 4.0 * N * H * W * y_ld < 3.9e9),
                         "vpho_conv3x3_winograd_stats_nhwc_f32: bn_x must be 16-byte aligned, y_ld %% 4 == 0, below 3.9 GB (it is fetched by 16-byte LDS-DMA)");
            a.stats = bn->stats; a.bn_x = bn->x; a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_gamma = bn->gamma; a.bn_beta = bn->beta;
            *bn->rows = tbs;
            const bool st1 = staged && wino_staged_ok(a.TH, a.TW, W);
            VPHO_DYN_LDS(conv_winograd_bn_kernel<1>, lds_staged);
            VPHO_DYN_LDS(conv_winograd_bn_kernel<0>, lds);
            VPHO_DYN_LDS(conv_winograd_bnb_kernel<1>, lds_staged);
            VPHO_DYN_LDS(conv_winograd_bnb_kernel<0>, lds);
            if (bn->x) {
                if (st1) hipLaunchKernelGGL(conv_winograd_bnb_kernel<1>, dim3(blocks, 1), dim3(256), lds_staged, (hipStream_t)stream, a);
                else     hipLaunchKernelGGL(conv_winograd_bnb_kernel<0>, dim3(blocks, 1), dim3(256), lds, (hipStream_t)stream, a);
            } else {
                if (st1) hipLaunchKernelGGL(conv_winograd_bn_kernel<1>, dim3(blocks, 1), dim3(256), lds_staged, (hipStream_t)stream, a);
                else     hipLaunchKernelGGL(conv_winograd_bn_kernel<0>, dim3(blocks, 1), dim3(256), lds, (hipStream_t)stream, a);
            }
            return vpho::check_launch("conv_winograd_bn_kernel");
        }
        // (Round 6, built and NOT kept: a persistent tile walk like conv_igemm_pers_kernel's -- the last super-stage of a tile prepares pixels, V and U of
        // the next tile's first stage, the epilogue runs with them in LDS -- bit-identical, and 1-6 % SLOWER on seven of nine layers (64 x 64 x 64 images
        // 256 -> 256: 1 233 against 1 220 us; 64 x 32 x 32 128 -> 128: 95.8 against 90.0), +1 % on two; docs/LOG.md round 6.  The one-tile kernel's
        // hand-over costs less than the walk's per-tile bookkeeping and its stage-end wait on the previous tile's stores.)
        if (staged && wins)
            hipLaunchKernelGGL(conv_winograd_kernel<2>, dim3(blocks, groups), dim3(256), lds_staged, (hipStream_t)stream, a);
        else if (staged && wino_staged_ok(a.TH, a.TW, W))
            hipLaunchKernelGGL(conv_winograd_kernel<1>, dim3(blocks, groups), dim3(256), lds_staged, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(conv_winograd_kernel<0>, dim3(blocks, groups), dim3(256), lds, (hipStream_t)stream, a);
        return vpho::check_launch("conv_winograd_kernel");
    }
    VPHO_DYN_LDS(conv_winograd8_kernel, lds);
    hipLaunchKernelGGL(conv_winograd8_kernel, dim3(blocks, groups), dim3(512), lds, (hipStream_t)stream, a);
    return vpho::check_launch("conv_winograd8_kernel");
}
