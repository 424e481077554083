// Depthwise 2-D convolution (groups == channels) on NCHW activations: forward, backward-data, backward-weight.
// The MBConv depthwise layers of the EfficientNet backbones inside the train step (reference:
// models/efficientnet_pytorch/model.py:59-64,100 `_depthwise_conv`, TF "SAME" padding from utils.py).  MIOpen has
// no tuned gfx950 solver for them (naive_conv_*), ATen's kernels issue k*k global loads per output and are
// address-unit bound (>50 % of the train step after the BN fusion).  The op is HBM streaming with a small stencil:
//   * every (image, channel) plane is independent; a work item = PB planes x one band of output rows (large planes:
//     PB = 1 and several bands; small planes: the whole plane and PB = 2 / 4 / 8 planes), handled by ONE wave with a
//     private LDS tile (fp32, rows with halo) -- no workgroup barriers, 8-16 waves per CU hide the latency;
//   * the tile is filled with 16-B / 8-B / 4-B vector loads over the planes' contiguous memory (a vector never
//     straddles a row: VEC | W), several vectors per lane in flight, and written with one ds_write of the same
//     width (image column 0 sits at the 16-B aligned tile column XO = 4).  Halo columns are zeroed once per wave and
//     never written again; rows outside the image are zeroed per item -> asymmetric SAME padding needs no padded copy;
//   * each lane owns one output column and a strip of R rows: a (R-1)*S+K tall input column per tap column is read
//     once from LDS into registers and reused by the R outputs (10 LDS reads per output at K=5, R=4 instead of 25);
//     consecutive lanes touch consecutive LDS words and store consecutive outputs; the taps of the plane's channel
//     are wave-uniform (staged per item in LDS, read back as broadcasts).
// backward-data, stride 1 = the same kernel on dy with the flipped filter; stride 2 = a gather in "phase" form: a
// lane produces a 2x2 block of dx from a ceil(K/2)^2 window of dy (no divergence, no zero-insertion);
// backward-weight = per-lane K*K fp32 accumulators over the strips of all planes a wave visits for one channel
// (its PB planes are the same channel of consecutive images), wave-reduced once at the end into
// partial[c][split][K*K] and summed in fixed order by a finalize kernel (deterministic).
#include "common.hpp"

namespace moma {
namespace {

constexpr int DW_WAVES = 4;                      // waves per workgroup, each with its own LDS tile
constexpr int XO = 4;                            // tile column of image column 0 (16-B aligned, >= max left halo)
constexpr int LDS_BUDGET = 64 * 1024;            // dynamic LDS per workgroup

template <typename T, int VEC> struct Vec;
template <> struct Vec<float, 4> { static __device__ void ld(const float* p, float* v) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; } };
template <> struct Vec<float, 2> { static __device__ void ld(const float* p, float* v) { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[1] = a.y; } };
template <> struct Vec<float, 1> { static __device__ void ld(const float* p, float* v) { v[0] = *p; } };
template <> struct Vec<bf16_raw, 8> {
    static __device__ void ld(const bf16_raw* p, float* v) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
};
template <> struct Vec<bf16_raw, 4> {
    static __device__ void ld(const bf16_raw* p, float* v) {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    }
};
template <> struct Vec<bf16_raw, 2> {
    static __device__ void ld(const bf16_raw* p, float* v) {
        const unsigned a = *reinterpret_cast<const unsigned*>(p);
        v[0] = __uint_as_float(a << 16); v[1] = __uint_as_float(a & 0xffff0000u);
    }
};
template <> struct Vec<bf16_raw, 1> { static __device__ void ld(const bf16_raw* p, float* v) { v[0] = bf16_to_f32(*p); } };

template <int VEC> __device__ __forceinline__ void lds_store(float* p, const float* v) {
    if constexpr (VEC == 8) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
        *p = v[0];
    }
}
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_raw>(bf16_raw* p, float v) { *p = f32_to_bf16(v); }

struct DwShape {
    int NC, C, H, W, OH, OW, pt, pl;             // planes, channels, source / result plane sizes, top / left padding
    int TH, nbands, PB, ngroups;                 // band rows, bands per plane, planes per item, plane groups
    int IR, pitch;                               // tile rows per plane, row pitch (floats, multiple of 4)
    int GR, gpitch;                              // second tile (backward-weight: dy rows, pitch)
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// i / d for 0 <= i < 2^20 (a work index inside one tile), d >= 1: the float quotient is within one of the answer
// (3 instructions + the fix-up instead of the ~25 of an integer division by a run-time value)
__device__ __forceinline__ int fast_div(int i, int d, float inv_d) {
    int q = (int)((float)i * inv_d);
    q += (q + 1) * d <= i;
    q -= q * d > i;
    return q;
}

// Fill tile rows r in [0, rows) of `np` planes: tile[(p*rows + r)*pitch + xo + col] = src_p[(y0 + r)*W + col] for image
// rows, 0 for rows outside [0, H).  src_p = src + p*plane_stride.  Vector loads over each plane's contiguous row range.
template <typename T, int VEC>
__device__ __forceinline__ void fill_tile(float* tile, const T* __restrict__ src, size_t plane_stride, int np, int H, int W,
                                          int y0, int rows, int pitch, int xo, int lane, bool zero_rows) {
    const int ya = max(y0, 0), yb = min(y0 + rows, H);           // image rows present in the tile
    const int wv = W / VEC, nvpp = max(yb - ya, 0) * wv, total = nvpp * np;
    const float inv_nvpp = 1.0f / (float)max(nvpp, 1), inv_wv = 1.0f / (float)wv;
    constexpr int U = 4;                                          // vectors in flight per lane
    for (int v0 = 0; v0 < total; v0 += 64 * U) {
        float val[U][VEC];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = min(v0 + u * 64 + lane, total - 1);
            const int p = fast_div(v, nvpp, inv_nvpp), e = v - p * nvpp, r = fast_div(e, wv, inv_wv), cv = e - r * wv;
            Vec<T, VEC>::ld(src + (size_t)p * plane_stride + (size_t)(ya + r) * W + cv * VEC, val[u]);
            dst[u] = (p * rows + (ya - y0) + r) * pitch + xo + cv * VEC;
        }
        // pin the loads as unconditional (else hipcc sinks each under the store's guard: load, wait, store, one by one)
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(val[u][0]));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (v0 + u * 64 + lane < total) lds_store<VEC>(tile + dst[u], val[u]);
    }
    // rows outside the image (wave-uniform, only at a plane's first / last band).  With one band per plane the row
    // mapping never changes and those rows keep the zeros of the initial clear: zero_rows = false.
    const int ztop = ya - y0, zbot = y0 + rows - max(yb, ya);
    if (zero_rows && (ztop > 0 || zbot > 0)) {
        for (int p = 0; p < np; ++p)
            for (int r = 0; r < rows; ++r)
                if (r < ztop || r >= rows - zbot)
                    for (int c = lane; c < W; c += 64) tile[(p * rows + r) * pitch + xo + c] = 0.f;
    }
}

template <typename T, int K, int S, int R, bool FLIP, int VEC>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             T* __restrict__ y, DwShape sh) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tfl = sh.PB * sh.IR * sh.pitch;                    // (multiple of 4 floats: tiles stay 16-B aligned)
    float* tile = smem + wave * (tfl + ((sh.PB * K * K + 3) & ~3));
    float* wl = tile + tfl;
    for (int i = lane; i < tfl; i += 64) tile[i] = 0.f;          // halo columns stay zero for good
    const int nitems = sh.ngroups * sh.nbands;
    const int nstrips = (sh.TH + R - 1) / R;
    constexpr int CR = (R - 1) * S + K;
    const int cbase = XO - sh.pl;
    for (int item = uniform(blockIdx.x * DW_WAVES + wave); item < nitems; item += gridDim.x * DW_WAVES) {
        const int grp = item / sh.nbands, band = item - grp * sh.nbands;
        const int plane0 = grp * sh.PB, np = min(sh.PB, sh.NC - plane0);
        const int oy0 = band * sh.TH, th = min(sh.TH, sh.OH - oy0);
        for (int i = lane; i < np * K * K; i += 64) {
            const int p = i / (K * K), t = i - p * (K * K);
            wl[i] = w[((plane0 + p) % sh.C) * K * K + (FLIP ? K * K - 1 - t : t)];
        }
        fill_tile<T, VEC>(tile, x + (size_t)plane0 * sh.H * sh.W, (size_t)sh.H * sh.W, np, sh.H, sh.W, oy0 * S - sh.pt, sh.IR,
                          sh.pitch, XO, lane, sh.nbands > 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's tile stores are done (no other wave reads it)
        const int nwork = nstrips * sh.OW;
        const float inv_ow = 1.0f / (float)sh.OW;
        for (int p = 0; p < np; ++p) {
            float wk[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) wk[t] = wl[p * K * K + t];
            const float* pt_ = tile + p * sh.IR * sh.pitch + cbase;
            T* yp = y + (size_t)(plane0 + p) * sh.OH * sh.OW;
            for (int i = lane; i < nwork; i += 64) {
                const int strip = fast_div(i, sh.OW, inv_ow), ox = i - strip * sh.OW;
                float acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = 0.f;
                const float* base = pt_ + (strip * R * S) * sh.pitch + ox * S;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    float col[CR];
#pragma unroll
                    for (int rr = 0; rr < CR; ++rr) col[rr] = base[rr * sh.pitch + kx];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int ky = 0; ky < K; ++ky) acc[r] = fmaf(wk[ky * K + kx], col[r * S + ky], acc[r]);
                }
                T* o = yp + (size_t)(oy0 + strip * R) * sh.OW + ox;
                if (strip * R + R <= th) {       // whole strip inside the band: no per-row guards
#pragma unroll
                    for (int r = 0; r < R; ++r) st<T>(o + r * sh.OW, acc[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (strip * R + r < th) st<T>(o + r * sh.OW, acc[r]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // tile reads done before the next item overwrites it
    }
}

// ---- small planes (OH = OW = 7 or 14): one output ROW per lane ------------------------------------------
// With 49 / 196 outputs per plane the strip mapping above leaves lanes idle and spends most of its instructions on
// LDS reads and index arithmetic (0.5 - 1.2 TB/s).  Here a work item is PB whole planes (PB * OWT <= 64), lane =
// (plane, output row): it reads its K input rows ((OWT-1)*S + K values each) from the tile once, keeps them in
// registers together with the K*K taps of its plane, and produces the OWT outputs of the row fully unrolled:
// (K*((OWT-1)*S+K) + K*K) LDS reads and OWT*K*K FMAs per OWT outputs (K=5, OWT=14: 8.2 reads per output instead of 15).
template <typename T, int K, int S, int OWT, bool FLIP, int VEC>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_small_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               T* __restrict__ y, DwShape sh) {
    extern __shared__ float smem[];
    constexpr int IW = (OWT - 1) * S + K;        // input columns one output row needs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tfl = sh.PB * sh.IR * sh.pitch;
    float* tile = smem + wave * (tfl + ((sh.PB * K * K + 3) & ~3));
    float* wl = tile + tfl;
    for (int i = lane; i < tfl; i += 64) tile[i] = 0.f;          // padding rows / columns stay zero for good
    const int p = lane / OWT, oy = lane - p * OWT;               // this lane's plane (of the item) and output row
    const int cbase = XO - sh.pl;
    for (int item = uniform(blockIdx.x * DW_WAVES + wave); item < sh.ngroups; item += gridDim.x * DW_WAVES) {
        const int plane0 = item * sh.PB, np = min(sh.PB, sh.NC - plane0);
        for (int i = lane; i < np * K * K; i += 64) {
            const int pp = i / (K * K), t = i - pp * (K * K);
            wl[i] = w[((plane0 + pp) % sh.C) * K * K + (FLIP ? K * K - 1 - t : t)];
        }
        fill_tile<T, VEC>(tile, x + (size_t)plane0 * sh.H * sh.W, (size_t)sh.H * sh.W, np, sh.H, sh.W, -sh.pt, sh.IR, sh.pitch,
                          XO, lane, false);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (p < np) {
            float wk[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) wk[t] = wl[p * K * K + t];
            float acc[OWT];
#pragma unroll
            for (int ox = 0; ox < OWT; ++ox) acc[ox] = 0.f;
            const float* rowp = tile + (p * sh.IR + oy * S) * sh.pitch + cbase;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float xr[IW];
#pragma unroll
                for (int c = 0; c < IW; ++c) xr[c] = rowp[ky * sh.pitch + c];
#pragma unroll
                for (int ox = 0; ox < OWT; ++ox)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc[ox] = fmaf(wk[ky * K + kx], xr[ox * S + kx], acc[ox]);
            }
            T* o = y + ((size_t)(plane0 + p) * OWT + oy) * OWT;
#pragma unroll
            for (int ox = 0; ox < OWT; ++ox) st<T>(o + ox, acc[ox]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// ---- backward-data, stride 2 ------------------------------------------------------------------------
// padded coordinates u = iy + pt, v = ix + pl;  dx[u = 2a+e, v = 2b+f] = sum_{t,s} w[e+2t, f+2s] * dy[a-t, b-s]
// here sh.H/W = the dy plane (source), sh.OH/OW = the dx plane (result); a band = sh.TH block rows `a`
template <typename T, int K, int VEC>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_bwd_data_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                     T* __restrict__ dx, DwShape sh) {
    extern __shared__ float smem[];
    constexpr int HT = (K - 1) / 2;              // halo in dy rows / cols (1 for K=3, 2 for K=5)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tfl = sh.PB * sh.IR * sh.pitch;
    float* tile = smem + wave * (tfl + ((sh.PB * K * K + 3) & ~3));
    float* wl = tile + tfl;
    for (int i = lane; i < tfl; i += 64) tile[i] = 0.f;
    const int nitems = sh.ngroups * sh.nbands;
    const int nb = (sh.OW + sh.pl + 2) / 2;      // block columns b = 0 .. nb-1 cover v = 0 .. OW+pl
    for (int item = uniform(blockIdx.x * DW_WAVES + wave); item < nitems; item += gridDim.x * DW_WAVES) {
        const int grp = item / sh.nbands, band = item - grp * sh.nbands;
        const int plane0 = grp * sh.PB, np = min(sh.PB, sh.NC - plane0);
        const int a0 = band * sh.TH;
        for (int i = lane; i < np * K * K; i += 64) {
            const int p = i / (K * K), t = i - p * (K * K);
            wl[i] = w[((plane0 + p) % sh.C) * K * K + t];
        }
        fill_tile<T, VEC>(tile, dy + (size_t)plane0 * sh.H * sh.W, (size_t)sh.H * sh.W, np, sh.H, sh.W, a0 - HT, sh.IR, sh.pitch,
                          XO, lane, sh.nbands > 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int nwork = sh.TH * nb;
        const float inv_nb = 1.0f / (float)nb;
        for (int p = 0; p < np; ++p) {
            float wk[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) wk[t] = wl[p * K * K + t];
            const float* pt_ = tile + p * sh.IR * sh.pitch + XO;
            T* xp = dx + (size_t)(plane0 + p) * sh.OH * sh.OW;
            for (int i = lane; i < nwork; i += 64) {
                const int ar = fast_div(i, nb, inv_nb), b = i - ar * nb;       // block row (local), block column
                float g[HT + 1][HT + 1];                       // g[t][s] = dy[a-t, b-s]; tile row of dy row a-t is ar+HT-t
#pragma unroll
                for (int t = 0; t <= HT; ++t)
#pragma unroll
                    for (int s = 0; s <= HT; ++s) g[t][s] = pt_[(ar + HT - t) * sh.pitch + (b - s)];
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
                        if (iy >= 0 && iy < sh.OH && ix >= 0 && ix < sh.OW) st<T>(xp + (size_t)iy * sh.OW + ix, acc);
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// ---- backward-weight --------------------------------------------------------------------------------
// dw[c, ky, kx] = sum_{n, oy, ox} dy[n,c,oy,ox] * x[n,c, oy*S+ky-pt, ox*S+kx-pl];  grid (nsplit/4, C): one wave visits
// image groups (PB consecutive images of channel c) x bands and keeps K*K accumulators per lane.
// GDIRECT: dy goes global -> registers inside the strip loop instead of through a second LDS tile (each dy value is used
// by one lane only).  Without the dy tile the x tile can be twice as tall at the same occupancy, which wins where the
// halo is large (K = 5) or the planes are small; on the K = 3 layers with 56^2 / 112^2 planes the loads' latency in the
// inner loop costs more than the shorter bands (measured both ways, scripts/bench_dw.py).
template <typename T, int K, int S, int R, int VEC, int GVEC, bool GDIRECT>
__global__ __launch_bounds__(DW_WAVES * 64) void dw_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    float* __restrict__ partial, DwShape sh, int N) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xt = sh.PB * sh.IR * sh.pitch, gtn = GDIRECT ? 0 : sh.PB * sh.GR * sh.gpitch;
    float* tile = smem + wave * (xt + gtn);
    float* gt = tile + xt;
    for (int i = lane; i < xt + gtn; i += 64) tile[i] = 0.f;
    const int c = blockIdx.y;
    const int nsplit = gridDim.x * DW_WAVES, split = blockIdx.x * DW_WAVES + wave;
    const int nstrips = (sh.TH + R - 1) / R;
    constexpr int CR = (R - 1) * S + K;
    const int cbase = XO - sh.pl;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    const size_t xs = (size_t)sh.C * sh.H * sh.W, gs = (size_t)sh.C * sh.OH * sh.OW;      // image strides
    const int nitems = sh.ngroups * sh.nbands;
    for (int item = uniform(split); item < nitems; item += nsplit) {
        const int grp = item / sh.nbands, band = item - grp * sh.nbands;
        const int n0 = grp * sh.PB, np = min(sh.PB, N - n0);
        const int oy0 = band * sh.TH;
        fill_tile<T, VEC>(tile, x + ((size_t)n0 * sh.C + c) * sh.H * sh.W, xs, np, sh.H, sh.W, oy0 * S - sh.pt, sh.IR, sh.pitch, XO,
                          lane, sh.nbands > 1);
        // dy rows of the band; rows past the plane's end are zero, so that a short last band contributes nothing
        if constexpr (!GDIRECT)
            fill_tile<T, GVEC>(gt, dy + ((size_t)n0 * sh.C + c) * sh.OH * sh.OW, gs, np, sh.OH, sh.OW, oy0, sh.GR, sh.gpitch, 0,
                               lane, sh.nbands > 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int nwork = nstrips * sh.OW;
        const float inv_ow = 1.0f / (float)sh.OW;
        for (int p = 0; p < np; ++p) {
            const float* pt_ = tile + p * sh.IR * sh.pitch + cbase;
            const float* gp = gt + p * sh.GR * sh.gpitch;
            const T* gd = dy + ((size_t)(n0 + p) * sh.C + c) * sh.OH * sh.OW;
            for (int i = lane; i < nwork; i += 64) {
                const int strip = fast_div(i, sh.OW, inv_ow), ox = i - strip * sh.OW;
                float g[R];
                if constexpr (GDIRECT) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float t1[1];
                        Vec<T, 1>::ld(gd + (size_t)min(oy0 + strip * R + r, sh.OH - 1) * sh.OW + ox, t1);
                        g[r] = t1[0];
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        asm volatile("" : "+v"(g[r]));         // keep the loads unconditional and together
                        g[r] = (oy0 + strip * R + r < sh.OH) ? g[r] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) g[r] = gp[(strip * R + r) * sh.gpitch + ox];
                }
                const float* base = pt_ + (strip * R * S) * sh.pitch + ox * S;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    float col[CR];
#pragma unroll
                    for (int rr = 0; rr < CR; ++rr) col[rr] = base[rr * sh.pitch + kx];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int ky = 0; ky < K; ++ky) acc[ky * K + kx] = fmaf(g[r], col[r * S + ky], acc[ky * K + kx]);
                }
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

// ---- host-side shape planning -------------------------------------------------------------------------
int round4(int v) { return (v + 3) & ~3; }
int pick_vec(int W, int elem_bytes, const void* a) {
    const int maxv = 16 / elem_bytes;                           // 8 for bf16, 4 for fp32
    for (int v = maxv; v > 1; v >>= 1)
        if (W % v == 0 && (uintptr_t)a % (v * elem_bytes) == 0) return v;
    return 1;
}
int strip_rows(int OH) { return OH >= 28 ? 4 : (OH >= 14 ? 2 : 1); }
int planes_per_item(int OH, int OW, int nbands) {
    if (nbands > 1) return 1;
    const int px = OH * OW;
    return px <= 64 ? 8 : (px <= 256 ? 4 : (px <= 1024 ? 2 : 1));
}
// band height: the largest multiple of R (<= OH rounded up) whose per-wave tile stays within WAVE_TILE_FLOATS --
// occupancy matters more than halo re-reads: ~10 KiB per wave keeps 16 waves per CU resident (measured on the
// 112 x 112 layers: 28-row bands / 8 waves per CU 1.86 TB/s, 16-row bands / 16 waves 3.1 TB/s)
constexpr int WAVE_TILE_FLOATS = 2560;
int pick_th(int OH, int S, int K, int pitch, int extra_floats_per_row, int R) {
    int th = ((OH + R - 1) / R) * R;
    while (th > R && ((th - 1) * S + K) * (long)pitch + (long)th * extra_floats_per_row > WAVE_TILE_FLOATS) th -= R;
    const int nbands = (OH + th - 1) / th;       // equal bands instead of full ones plus a remainder
    return (((OH + nbands - 1) / nbands + R - 1) / R) * R;
}
int fit_planes(int pb, long floats_per_plane) {
    while (pb > 1 && pb * floats_per_plane > WAVE_TILE_FLOATS) pb >>= 1;
    return pb;
}
unsigned grid_for(long nitems) {
    long g = (nitems + DW_WAVES - 1) / DW_WAVES;
    if (g > 256 * 16) g = 256 * 16;                // grid-stride beyond 16 workgroups per CU
    return (unsigned)max(1L, g);
}

// run BODY with `V` = the compile-time vector width for the run-time `vec` (8 only exists for 2-byte elements)
#define MOMA_DW_VEC_SWITCH(vec, MAXV, BODY)                                    \
    if (MAXV >= 8 && (vec) == 8) { constexpr int V = (MAXV >= 8 ? 8 : 4); BODY; } \
    else if ((vec) >= 4) { constexpr int V = 4; BODY; }                        \
    else if ((vec) == 2) { constexpr int V = 2; BODY; }                        \
    else { constexpr int V = 1; BODY; }

// y (OH x OW) from x (H x W); also backward-data stride 1 with flip = true (then x = dy and y = dx)
template <typename T, int K, int S>
hipError_t fwd_t(const T* x, const float* w, T* y, int NC, int C, int H, int W, int OH, int OW, int pt, int pl, bool flip,
                 hipStream_t st) {
    constexpr int MAXV = 16 / sizeof(T);
    DwShape sh{};
    sh.NC = NC; sh.C = C; sh.H = H; sh.W = W; sh.OH = OH; sh.OW = OW; sh.pt = pt; sh.pl = pl;
    sh.pitch = round4(XO + max(W, (OW - 1) * S + K - pl) + 1);
    if (S == 1 && OH == OW && (OW == 7 || OW == 14)) {
        // small planes: one output row per lane, whole planes per item (stride 2 measured faster on the strip kernel)
        sh.TH = OH; sh.nbands = 1;
        sh.IR = (OH - 1) * S + K;
        sh.PB = OW == 7 ? 8 : 4;
        sh.pitch |= 1;           // lanes read rows one pitch apart: an odd pitch spreads them over all LDS banks
        sh.ngroups = (NC + sh.PB - 1) / sh.PB;
        const size_t lds = (size_t)DW_WAVES * (sh.PB * sh.IR * sh.pitch + round4(sh.PB * K * K)) * sizeof(float);
        if (lds <= (size_t)LDS_BUDGET) {
            const dim3 grid(grid_for(sh.ngroups)), block(DW_WAVES * 64);
            const int vec = 1;       // (odd pitch: element stores into the tile)
#define MOMA_DW_SMALL(OWT, FL) hipLaunchKernelGGL((dw_small_kernel<T, K, S, OWT, FL, V>), grid, block, lds, st, x, w, y, sh)
            MOMA_DW_VEC_SWITCH(vec, MAXV, {
                if (OW == 7) { if (flip) MOMA_DW_SMALL(7, true); else MOMA_DW_SMALL(7, false); }
                else { if (flip) MOMA_DW_SMALL(14, true); else MOMA_DW_SMALL(14, false); }
            })
#undef MOMA_DW_SMALL
            return hipGetLastError();
        }
    }
    const int R = strip_rows(OH);
    sh.TH = pick_th(OH, S, K, sh.pitch, 0, R);
    sh.nbands = (OH + sh.TH - 1) / sh.TH;
    sh.IR = (sh.TH - 1) * S + K;
    sh.PB = fit_planes(planes_per_item(OH, OW, sh.nbands), (long)sh.IR * sh.pitch);
    sh.ngroups = (NC + sh.PB - 1) / sh.PB;
    const size_t lds = (size_t)DW_WAVES * (sh.PB * sh.IR * sh.pitch + round4(sh.PB * K * K)) * sizeof(float);
    if (lds > (size_t)LDS_BUDGET) return hipErrorInvalidValue;
    const dim3 grid(grid_for((long)sh.ngroups * sh.nbands)), block(DW_WAVES * 64);
    const int vec = pick_vec(W, sizeof(T), x);
#define MOMA_DW_GO(RR, FL) hipLaunchKernelGGL((dw_fwd_kernel<T, K, S, RR, FL, V>), grid, block, lds, st, x, w, y, sh)
    MOMA_DW_VEC_SWITCH(vec, MAXV, {
        if (flip) { if (R == 4) MOMA_DW_GO(4, true); else if (R == 2) MOMA_DW_GO(2, true); else MOMA_DW_GO(1, true); }
        else { if (R == 4) MOMA_DW_GO(4, false); else if (R == 2) MOMA_DW_GO(2, false); else MOMA_DW_GO(1, false); }
    })
#undef MOMA_DW_GO
    return hipGetLastError();
}

// dx (H x W) from dy (OH x OW), stride 2
template <typename T, int K>
hipError_t bwd_data_s2_t(const T* dy, const float* w, T* dx, int NC, int C, int H, int W, int OH, int OW, int pt, int pl,
                         hipStream_t st) {
    constexpr int MAXV = 16 / sizeof(T);
    constexpr int HT = (K - 1) / 2;
    DwShape sh{};
    sh.NC = NC; sh.C = C; sh.H = OH; sh.W = OW; sh.OH = H; sh.OW = W; sh.pt = pt; sh.pl = pl;       // source = dy, result = dx
    const int nb = (W + pl + 2) / 2, na = (H + pt + 2) / 2;     // block columns / rows covering the padded plane
    sh.pitch = round4(XO + max(OW, nb) + 1);
    int th = na;
    while (th > 1 && (long)(th + HT) * sh.pitch > WAVE_TILE_FLOATS) --th;
    th = (na + (na + th - 1) / th - 1) / ((na + th - 1) / th);       // equal bands
    sh.TH = th;
    sh.nbands = (na + th - 1) / th;
    sh.IR = th + HT;
    sh.PB = fit_planes(planes_per_item(H, W, sh.nbands), (long)sh.IR * sh.pitch);
    sh.ngroups = (NC + sh.PB - 1) / sh.PB;
    const size_t lds = (size_t)DW_WAVES * (sh.PB * sh.IR * sh.pitch + round4(sh.PB * K * K)) * sizeof(float);
    if (lds > (size_t)LDS_BUDGET) return hipErrorInvalidValue;
    const dim3 grid(grid_for((long)sh.ngroups * sh.nbands)), block(DW_WAVES * 64);
    const int vec = pick_vec(OW, sizeof(T), dy);
    MOMA_DW_VEC_SWITCH(vec, MAXV, { hipLaunchKernelGGL((dw_bwd_data_s2_kernel<T, K, V>), grid, block, lds, st, dy, w, dx, sh); })
    return hipGetLastError();
}

template <typename T, int K, int S>
hipError_t bwd_weight_t(const T* x, const T* dy, float* dw, float* ws, size_t ws_floats, int N, int C, int H, int W, int OH,
                        int OW, int pt, int pl, hipStream_t st) {
    constexpr int MAXV = 16 / sizeof(T);
    DwShape sh{};
    sh.NC = N * C; sh.C = C; sh.H = H; sh.W = W; sh.OH = OH; sh.OW = OW; sh.pt = pt; sh.pl = pl;
    sh.pitch = round4(XO + max(W, (OW - 1) * S + K - pl) + 1);
    const bool gdirect = !(K == 3 && OH >= 56);
    sh.gpitch = gdirect ? 0 : round4(OW + 1);
    const int R = strip_rows(OH);
    sh.TH = pick_th(OH, S, K, sh.pitch, sh.gpitch, R);
    sh.nbands = (OH + sh.TH - 1) / sh.TH;
    sh.IR = (sh.TH - 1) * S + K;
    sh.GR = gdirect ? 0 : sh.TH;
    sh.PB = fit_planes(min(planes_per_item(OH, OW, sh.nbands), N), (long)sh.IR * sh.pitch + (long)sh.GR * sh.gpitch);
    sh.ngroups = (N + sh.PB - 1) / sh.PB;          // groups of images (per channel)
    const size_t lds = (size_t)DW_WAVES * sh.PB * (sh.IR * sh.pitch + sh.GR * sh.gpitch) * sizeof(float);
    if (lds > (size_t)LDS_BUDGET) return hipErrorInvalidValue;
    // ~8192 waves in all, at least one item per wave, partials must fit the workspace
    long nsplit_wg = max(1L, min((long)(2048 + C - 1) / C, ((long)sh.ngroups * sh.nbands + DW_WAVES - 1) / DW_WAVES));
    while (nsplit_wg > 1 && (size_t)C * nsplit_wg * DW_WAVES * K * K > ws_floats) --nsplit_wg;
    if ((size_t)C * nsplit_wg * DW_WAVES * K * K > ws_floats) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nsplit_wg, C), block(DW_WAVES * 64);
    const int vec = pick_vec(W, sizeof(T), x), gvec = pick_vec(OW, sizeof(T), dy);
    // (the dy tile takes the x tile's vector width when that divides OW too, else scalar: keeps the instantiations down)
#define MOMA_DW_GO(RR, GV, GD) hipLaunchKernelGGL((dw_bwd_weight_kernel<T, K, S, RR, V, GV, GD>), grid, block, lds, st, x, dy, ws, sh, N)
    MOMA_DW_VEC_SWITCH(vec, MAXV, {
        if (gdirect) { if (R == 4) MOMA_DW_GO(4, 1, true); else if (R == 2) MOMA_DW_GO(2, 1, true); else MOMA_DW_GO(1, 1, true); }
        else if (gvec >= V) { if (R == 4) MOMA_DW_GO(4, V, false); else if (R == 2) MOMA_DW_GO(2, V, false); else MOMA_DW_GO(1, V, false); }
        else { if (R == 4) MOMA_DW_GO(4, 1, false); else if (R == 2) MOMA_DW_GO(2, 1, false); else MOMA_DW_GO(1, 1, false); }
    })
#undef MOMA_DW_GO
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
        // correlation of dy with the flipped filter, padding K-1-pt / K-1-pl; result plane = the input plane
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
