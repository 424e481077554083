// Depthwise 2-D convolution (groups == channels) on NCHW activations: forward, backward-data, backward-weight.
// The MBConv depthwise layers of the EfficientNet backbones inside the train step (reference:
// models/efficientnet_pytorch/model.py:59-64,100 `_depthwise_conv`, TF "SAME" padding from utils.py).  MIOpen has
// no tuned gfx950 solver for them (naive_conv_*), ATen's kernels issue k*k global loads per output and are
// address-unit bound (>50 % of the train step after the BN fusion).  The op is HBM streaming with a small stencil:
//   * every (image, channel) plane is independent; a work item = one plane x one band of output rows, handled by ONE
//     wave with a private LDS tile (fp32, input rows with halo, zero-filled outside the image -> asymmetric SAME
//     padding needs no padded copy of the input), so there are no workgroup barriers and 16 waves/CU hide latency;
//   * each lane owns one output column and a strip of R rows: a (R-1)*S+K tall input column per tap column is read
//     once from LDS into registers and reused by the R outputs (10 LDS reads per output at K=5, R=4 instead of 25);
//     consecutive lanes touch consecutive LDS words and store consecutive outputs;
//   * the filter taps of the plane's channel are wave-uniform (scalar registers).
// backward-data, stride 1 = the same kernel on dy with the flipped filter; stride 2 = a gather in "phase" form: a
// lane produces a 2x2 block of dx from a ceil(K/2)^2 window of dy (no divergence, no zero-insertion);
// backward-weight = per-lane K*K fp32 accumulators over the strips of all planes a wave visits for one channel,
// wave-reduced once at the end into partial[c][split][K*K] and summed in fixed order by a finalize kernel.
#include "common.hpp"

namespace moma {
namespace {

constexpr int DW_WAVES = 4;                      // waves per workgroup, each with its own LDS tile

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_raw>(bf16_raw* p, float v) { *p = f32_to_bf16(v); }

struct DwShape {
    int NC, C, H, W, OH, OW, pt, pl;             // planes, channels, input / output plane size, top / left padding
    int TH, nbands;                              // output rows per band, bands per plane
    int IR, IWS;                                 // LDS tile rows / row pitch (floats)
};

// rows [y0, y0+rows) x cols [x0, x0+cols) of one plane -> lds[r*pitch + col] (fp32), zero outside the plane.
// Loads are unconditional from clamped addresses, pinned, then masked (a guarded load is sunk under its guard and
// waited on one by one).
template <typename T>
__device__ __forceinline__ void load_tile(float* lds, const T* __restrict__ plane, int H, int W, int y0, int x0, int rows,
                                          int cols, int pitch, int lane) {
    for (int c0 = 0; c0 < cols; c0 += 64) {
        const int col = c0 + lane, ix = x0 + col;
        const bool cok = col < cols && ix >= 0 && ix < W;
        const int ixc = min(max(ix, 0), W - 1);
        int r = 0;
        // 16 loads in flight per lane (the pin below is what forces the wait; 4 per group left HBM latency exposed)
        for (; r + 16 <= rows; r += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ld<T>(plane + (size_t)min(max(y0 + r + u, 0), H - 1) * W + ixc);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("" : "+v"(v[u]));
                const int iy = y0 + r + u;
                if (col < cols) lds[(r + u) * pitch + col] = (cok && iy >= 0 && iy < H) ? v[u] : 0.f;
            }
        }
        for (; r + 4 <= rows; r += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld<T>(plane + (size_t)min(max(y0 + r + u, 0), H - 1) * W + ixc);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("" : "+v"(v[u]));
                const int iy = y0 + r + u;
                if (col < cols) lds[(r + u) * pitch + col] = (cok && iy >= 0 && iy < H) ? v[u] : 0.f;
            }
        }
        for (; r < rows; ++r) {
            float v = ld<T>(plane + (size_t)min(max(y0 + r, 0), H - 1) * W + ixc);
            asm volatile("" : "+v"(v));
            const int iy = y0 + r;
            if (col < cols) lds[r * pitch + col] = (cok && iy >= 0 && iy < H) ? v : 0.f;
        }
    }
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- forward (and stride-1 backward-data with FLIP) -------------------------------------------------
// y[p, oy, ox] = sum_{ky,kx} w[c, ky, kx] * x[p, oy*S + ky - pt, ox*S + kx - pl]
template <typename T, int K, int S, int R, bool FLIP>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             T* __restrict__ y, DwShape sh) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = smem + wave * sh.IR * sh.IWS;
    const int nitems = sh.NC * sh.nbands;
    const int nstrips = (sh.TH + R - 1) / R;
    constexpr int CR = (R - 1) * S + K;          // input rows a strip needs
    for (int item = uniform(blockIdx.x * DW_WAVES + wave); item < nitems; item += gridDim.x * DW_WAVES) {
        const int plane = item / sh.nbands, band = item - plane * sh.nbands;
        const int c = plane % sh.C;
        const int oy0 = band * sh.TH;
        const int th = min(sh.TH, sh.OH - oy0);
        load_tile<T>(tile, x + (size_t)plane * sh.H * sh.W, sh.H, sh.W, oy0 * S - sh.pt, -sh.pl, sh.IR, sh.IWS, sh.IWS, lane);
        float wk[K * K];                         // wave-uniform taps
#pragma unroll
        for (int t = 0; t < K * K; ++t) wk[t] = w[c * K * K + (FLIP ? K * K - 1 - t : t)];
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's tile stores are done (no other wave reads it)
        const int nwork = nstrips * sh.OW;
        for (int i = lane; i < nwork; i += 64) {
            const int strip = i / sh.OW, ox = i - strip * sh.OW;
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.f;
            const float* base = tile + (strip * R * S) * sh.IWS + ox * S;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float col[CR];
#pragma unroll
                for (int rr = 0; rr < CR; ++rr) col[rr] = base[rr * sh.IWS + kx];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) acc[r] = fmaf(wk[ky * K + kx], col[r * S + ky], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oy = oy0 + strip * R + r;
                if (strip * R + r < th) st<T>(y + ((size_t)plane * sh.OH + oy) * sh.OW + ox, acc[r]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // tile reads done before the next item overwrites it
    }
}

// ---- backward-data, stride 2 ------------------------------------------------------------------------
// padded coordinates u = iy + pt, v = ix + pl;  dx[u = 2a+e, v = 2b+f] = sum_{t,s} w[e+2t, f+2s] * dy[a-t, b-s]
template <typename T, int K>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_bwd_data_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                     T* __restrict__ dx, DwShape sh) {
    // here sh.H/W = dx plane, sh.OH/OW = dy plane; a band = sh.TH block rows `a`; tile rows = a0-HT .. a0+TH-1
    extern __shared__ float smem[];
    constexpr int HT = (K - 1) / 2;              // halo in dy rows / cols (1 for K=3, 2 for K=5)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = smem + wave * sh.IR * sh.IWS;
    const int nitems = sh.NC * sh.nbands;
    const int nb = (sh.W + sh.pl + 2) / 2;       // block columns b = 0 .. nb-1 cover v = 0 .. W+pl
    for (int item = uniform(blockIdx.x * DW_WAVES + wave); item < nitems; item += gridDim.x * DW_WAVES) {
        const int plane = item / sh.nbands, band = item - plane * sh.nbands;
        const int c = plane % sh.C;
        const int a0 = band * sh.TH;
        load_tile<T>(tile, dy + (size_t)plane * sh.OH * sh.OW, sh.OH, sh.OW, a0 - HT, -HT, sh.TH + HT, sh.IWS, sh.IWS, lane);
        float wk[K * K];
#pragma unroll
        for (int t = 0; t < K * K; ++t) wk[t] = w[c * K * K + t];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int nwork = sh.TH * nb;
        for (int i = lane; i < nwork; i += 64) {
            const int ar = i / nb, b = i - ar * nb;          // block row (local), block column
            float g[HT + 1][HT + 1];                          // g[t][s] = dy[a-t, b-s]
#pragma unroll
            for (int t = 0; t <= HT; ++t)
#pragma unroll
                for (int s = 0; s <= HT; ++s) g[t][s] = tile[(ar + HT - t) * sh.IWS + (b + HT - s)];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    float acc = 0.f;
#pragma unroll
                    for (int t = 0; e + 2 * t < K; ++t)
#pragma unroll
                        for (int s = 0; f + 2 * s < K; ++s) acc = fmaf(wk[(e + 2 * t) * K + f + 2 * s], g[t][s], acc);
                    const int iy = 2 * (a0 + ar) + e - sh.pt, ix = 2 * b + f - sh.pl;
                    if (iy >= 0 && iy < sh.H && ix >= 0 && ix < sh.W) st<T>(dx + ((size_t)plane * sh.H + iy) * sh.W + ix, acc);
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// ---- backward-weight --------------------------------------------------------------------------------
// dw[c, ky, kx] = sum_{n, oy, ox} dy[n,c,oy,ox] * x[n,c, oy*S+ky-pt, ox*S+kx-pl];  grid (nsplit, C): one wave visits
// planes n = first, first + stride, ... of channel c, all bands, and keeps K*K accumulators per lane.
template <typename T, int K, int S, int R>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    float* __restrict__ partial, DwShape sh, int N) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gtile = sh.TH * sh.OW;             // dy tile (floats) after the x tile
    float* tile = smem + wave * (sh.IR * sh.IWS + gtile);
    float* gt = tile + sh.IR * sh.IWS;
    const int c = blockIdx.y;
    const int nsplit = gridDim.x * DW_WAVES, split = blockIdx.x * DW_WAVES + wave;
    const int nstrips = (sh.TH + R - 1) / R;
    constexpr int CR = (R - 1) * S + K;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    const int nitems = N * sh.nbands;
    for (int item = uniform(split); item < nitems; item += nsplit) {
        const int n = item / sh.nbands, band = item - n * sh.nbands;
        const size_t plane = (size_t)n * sh.C + c;
        const int oy0 = band * sh.TH;
        const int th = min(sh.TH, sh.OH - oy0);
        // always the full tile: rows past the image are written as zeros, so a short last band multiplies zeros (never
        // stale LDS contents) by the zero-filled dy rows below
        load_tile<T>(tile, x + plane * sh.H * sh.W, sh.H, sh.W, oy0 * S - sh.pt, -sh.pl, sh.IR, sh.IWS, sh.IWS, lane);
        // dy rows of the band, zero rows past the plane's end (so that strips need no row guard)
        {
            const T* gp = dy + (plane * sh.OH + oy0) * sh.OW;
            for (int i = lane; i < sh.TH * sh.OW; i += 64) {
                const int r = i / sh.OW;
                float v = ld<T>(gp + min(i, th * sh.OW - 1));
                asm volatile("" : "+v"(v));
                gt[i] = r < th ? v : 0.f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int nwork = nstrips * sh.OW;
        for (int i = lane; i < nwork; i += 64) {
            const int strip = i / sh.OW, ox = i - strip * sh.OW;
            float g[R];
#pragma unroll
            for (int r = 0; r < R; ++r) g[r] = (strip * R + r < sh.TH) ? gt[(strip * R + r) * sh.OW + ox] : 0.f;
            const float* base = tile + (strip * R * S) * sh.IWS + ox * S;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float col[CR];
#pragma unroll
                for (int rr = 0; rr < CR; ++rr) col[rr] = base[rr * sh.IWS + kx];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) acc[ky * K + kx] = fmaf(g[r], col[r * S + ky], acc[ky * K + kx]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = wave_sum(acc[t]);
    if (lane == 0) {
        float* p = partial + ((size_t)c * nsplit + split) * K * K;
#pragma unroll
        for (int t = 0; t < K * K; ++t) p[t] = acc[t];
    }
}

__global__ void dw_bwd_weight_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dw, int total, int nsplit,
                                              int kk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // i = c*kk + t
    if (i >= total) return;
    const int c = i / kk, t = i - c * kk;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[((size_t)c * nsplit + k) * kk + t];
    dw[i] = s;
}

constexpr int LDS_BUDGET = 64 * 1024;            // dynamic LDS per workgroup (4 waves)

int pick_th(int OH, int rows_per_out_num, int K, int pitch, int extra_per_row, int R) {
    // largest multiple-of-R band (<= 32 rows, <= OH rounded up to R) whose per-wave tile fits the budget
    int th = ((min(OH, 32) + R - 1) / R) * R;
    while (th > R) {
        const long floats = (long)((th - 1) * rows_per_out_num + K) * pitch + (long)th * extra_per_row;
        if (floats * 4 * DW_WAVES <= LDS_BUDGET) break;
        th -= R;
    }
    return th;
}

unsigned grid_for(long nitems) {
    long g = (nitems + DW_WAVES - 1) / DW_WAVES;
    if (g > 256 * 16) g = 256 * 16;                // grid-stride beyond 16 workgroups per CU
    return (unsigned)max(1L, g);
}

template <typename T, int K, int S>
hipError_t fwd_t(const T* x, const float* w, T* y, int NC, int C, int H, int W, int OH, int OW, int pt, int pl, bool flip,
                 hipStream_t st) {
    DwShape sh{NC, C, H, W, OH, OW, pt, pl, 0, 0, 0, 0};
    sh.IWS = (OW - 1) * S + K;
    const int R = OH >= 28 ? 4 : (OH >= 14 ? 2 : 1);
    sh.TH = pick_th(OH, S, K, sh.IWS, 0, R);
    sh.nbands = (OH + sh.TH - 1) / sh.TH;
    sh.IR = (sh.TH - 1) * S + K;
    const size_t lds = (size_t)DW_WAVES * sh.IR * sh.IWS * sizeof(float);
    const dim3 grid(grid_for((long)NC * sh.nbands)), block(DW_WAVES * 64);
#define MOMA_DW_FWD(RR, FL) hipLaunchKernelGGL((dw_fwd_kernel<T, K, S, RR, FL>), grid, block, lds, st, x, w, y, sh)
    if (flip) { if (R == 4) MOMA_DW_FWD(4, true); else if (R == 2) MOMA_DW_FWD(2, true); else MOMA_DW_FWD(1, true); }
    else { if (R == 4) MOMA_DW_FWD(4, false); else if (R == 2) MOMA_DW_FWD(2, false); else MOMA_DW_FWD(1, false); }
#undef MOMA_DW_FWD
    return hipGetLastError();
}

template <typename T, int K>
hipError_t bwd_data_s2_t(const T* dy, const float* w, T* dx, int NC, int C, int H, int W, int OH, int OW, int pt, int pl,
                         hipStream_t st) {
    constexpr int HT = (K - 1) / 2;
    DwShape sh{NC, C, H, W, OH, OW, pt, pl, 0, 0, 0, 0};
    const int nb = (W + pl + 2) / 2, na = (H + pt + 2) / 2;     // block columns / rows covering the padded plane
    sh.IWS = nb + HT;
    int th = min(na, 16);
    while (th > 1 && (long)(th + HT) * sh.IWS * 4 * DW_WAVES > LDS_BUDGET) --th;
    sh.TH = th;
    sh.nbands = (na + th - 1) / th;
    sh.IR = th + HT;
    const size_t lds = (size_t)DW_WAVES * sh.IR * sh.IWS * sizeof(float);
    hipLaunchKernelGGL((dw_bwd_data_s2_kernel<T, K>), dim3(grid_for((long)NC * sh.nbands)), dim3(DW_WAVES * 64), lds, st, dy, w,
                       dx, sh);
    return hipGetLastError();
}

template <typename T, int K, int S>
hipError_t bwd_weight_t(const T* x, const T* dy, float* dw, float* ws, size_t ws_floats, int N, int C, int H, int W, int OH,
                        int OW, int pt, int pl, hipStream_t st) {
    DwShape sh{N * C, C, H, W, OH, OW, pt, pl, 0, 0, 0, 0};
    sh.IWS = (OW - 1) * S + K;
    const int R = OH >= 28 ? 4 : (OH >= 14 ? 2 : 1);
    sh.TH = pick_th(OH, S, K, sh.IWS, OW, R);
    sh.nbands = (OH + sh.TH - 1) / sh.TH;
    sh.IR = (sh.TH - 1) * S + K;
    const size_t lds = (size_t)DW_WAVES * (sh.IR * sh.IWS + sh.TH * sh.OW) * sizeof(float);
    // ~8192 waves in all, at least one item per wave, partials must fit the workspace
    long nsplit_wg = max(1L, min((long)(2048 + C - 1) / C, ((long)N * sh.nbands + DW_WAVES - 1) / DW_WAVES));
    while (nsplit_wg > 1 && (size_t)C * nsplit_wg * DW_WAVES * K * K > ws_floats) --nsplit_wg;
    if ((size_t)C * nsplit_wg * DW_WAVES * K * K > ws_floats) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nsplit_wg, C), block(DW_WAVES * 64);
#define MOMA_DW_BW(RR) hipLaunchKernelGGL((dw_bwd_weight_kernel<T, K, S, RR>), grid, block, lds, st, x, dy, ws, sh, N)
    if (R == 4) MOMA_DW_BW(4); else if (R == 2) MOMA_DW_BW(2); else MOMA_DW_BW(1);
#undef MOMA_DW_BW
    const int total = C * K * K;
    hipLaunchKernelGGL(dw_bwd_weight_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws, dw, total,
                       (int)nsplit_wg * DW_WAVES, K * K);
    return hipGetLastError();
}
}  // namespace

bool dwconv_supported(int K, int S) { return (K == 3 || K == 5) && (S == 1 || S == 2); }
size_t dwconv_workspace_floats(int C, int K) { return (size_t)C * 64 * DW_WAVES * K * K; }

#define MOMA_DW_DISPATCH(CALL)                                                   \
    if (K == 3 && S == 1) return CALL(3, 1);                                      \
    if (K == 3 && S == 2) return CALL(3, 2);                                      \
    if (K == 5 && S == 1) return CALL(5, 1);                                      \
    if (K == 5 && S == 2) return CALL(5, 2);                                      \
    return hipErrorInvalidValue;

hipError_t launch_dw_fwd(const void* x, const float* w, void* y, int N, int C, int H, int W, int OH, int OW, int K, int S,
                         int pt, int pl, int dtype, hipStream_t st) {
    if (dtype == MOMA_DT_BF16) {
#define CALL(KK, SS) fwd_t<bf16_raw, KK, SS>((const bf16_raw*)x, w, (bf16_raw*)y, N * C, C, H, W, OH, OW, pt, pl, false, st)
        MOMA_DW_DISPATCH(CALL)
#undef CALL
    }
#define CALL(KK, SS) fwd_t<float, KK, SS>((const float*)x, w, (float*)y, N * C, C, H, W, OH, OW, pt, pl, false, st)
    MOMA_DW_DISPATCH(CALL)
#undef CALL
}

hipError_t launch_dw_bwd_data(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int OH, int OW, int K,
                              int S, int pt, int pl, int dtype, hipStream_t st) {
    if (S == 1) {
        // correlation of dy with the flipped filter, padding K-1-pt / K-1-pl; output plane = the input plane
        if (dtype == MOMA_DT_BF16) {
            if (K == 3) return fwd_t<bf16_raw, 3, 1>((const bf16_raw*)dy, w, (bf16_raw*)dx, N * C, C, OH, OW, H, W, 2 - pt, 2 - pl, true, st);
            if (K == 5) return fwd_t<bf16_raw, 5, 1>((const bf16_raw*)dy, w, (bf16_raw*)dx, N * C, C, OH, OW, H, W, 4 - pt, 4 - pl, true, st);
        } else {
            if (K == 3) return fwd_t<float, 3, 1>((const float*)dy, w, (float*)dx, N * C, C, OH, OW, H, W, 2 - pt, 2 - pl, true, st);
            if (K == 5) return fwd_t<float, 5, 1>((const float*)dy, w, (float*)dx, N * C, C, OH, OW, H, W, 4 - pt, 4 - pl, true, st);
        }
        return hipErrorInvalidValue;
    }
    if (dtype == MOMA_DT_BF16) {
        if (K == 3) return bwd_data_s2_t<bf16_raw, 3>((const bf16_raw*)dy, w, (bf16_raw*)dx, N * C, C, H, W, OH, OW, pt, pl, st);
        if (K == 5) return bwd_data_s2_t<bf16_raw, 5>((const bf16_raw*)dy, w, (bf16_raw*)dx, N * C, C, H, W, OH, OW, pt, pl, st);
    } else {
        if (K == 3) return bwd_data_s2_t<float, 3>((const float*)dy, w, (float*)dx, N * C, C, H, W, OH, OW, pt, pl, st);
        if (K == 5) return bwd_data_s2_t<float, 5>((const float*)dy, w, (float*)dx, N * C, C, H, W, OH, OW, pt, pl, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_dw_bwd_weight(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats, int N, int C, int H,
                                int W, int OH, int OW, int K, int S, int pt, int pl, int dtype, hipStream_t st) {
    if (dtype == MOMA_DT_BF16) {
#define CALL(KK, SS) bwd_weight_t<bf16_raw, KK, SS>((const bf16_raw*)x, (const bf16_raw*)dy, dw, ws, ws_floats, N, C, H, W, OH, OW, pt, pl, st)
        MOMA_DW_DISPATCH(CALL)
#undef CALL
    }
#define CALL(KK, SS) bwd_weight_t<float, KK, SS>((const float*)x, (const float*)dy, dw, ws, ws_floats, N, C, H, W, OH, OW, pt, pl, st)
    MOMA_DW_DISPATCH(CALL)
#undef CALL
}

}  // namespace moma
