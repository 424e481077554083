// BatchNorm2d (+ fused SiLU / ReLU) forward and backward on NCHW activations -- the normalisation + activation
// pairs of the backbones inside the train step (reference: models/efficientnet_pytorch/model.py:96-102,114
// `_swish(_bn0(...))`, `_swish(_bn1(...))`, `_bn2(...)`; torch semantics of nn.BatchNorm2d in training and eval
// mode incl. the running-statistics update).  Pure HBM streaming work:
//   forward  train: read x (statistics), read x again + write out          = 3 passes over the activation
//   backward      : read x, dout (reductions), read x, dout + write dx     = 5 passes
// (unfused torch: BN forward + activation forward = 5 passes, backward 8, and the pre-activation tensor is kept).
// The backward recomputes the pre-activation y = xhat*gamma + beta in fp32 from x and the saved statistics, so
// nothing but x (which autograd keeps for the convolution anyway) has to be stored.
//
// Layout: channel c of image n is a run of HW contiguous elements at ((n*C + c)*HW); a channel's N runs are
// C*HW apart.  Vector width VEC = 8 / 4 / 1 elements by divisibility of HW (and pointer alignment), so every
// vector lies inside one run.  Reductions: grid (S, C), split s of channel c walks the channel's vectors with stride
// S*256; statistics are merged as (count, mean, M2) triples with Chan's update (no E[x^2]-mean^2 cancellation),
// backward sums as plain fp32 sums; a finalize kernel (one thread per channel) merges the S partials in fixed order
// (deterministic), forms scale/shift and updates the running statistics.
#include "common.hpp"

namespace moma {
namespace {

constexpr int BN_THREADS = 256;

template <typename T, int VEC>
struct VecIO;
template <int VEC>
struct VecIO<float, VEC> {
    __device__ static void load(const float* p, float (&v)[VEC]) {
        if constexpr (VEC == 8) {
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else if constexpr (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(p);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        } else {
            v[0] = *p;
        }
    }
    __device__ static void store(float* p, const float (&v)[VEC]) {
        if constexpr (VEC == 8) {
            *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            *p = v[0];
        }
    }
};
template <int VEC>
struct VecIO<bf16_raw, VEC> {
    __device__ static void load(const bf16_raw* p, float (&v)[VEC]) {
        if constexpr (VEC == 8) {
            const uint4 a = *reinterpret_cast<const uint4*>(p);
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        } else if constexpr (VEC == 4) {
            const uint2 a = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
            v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
        } else {
            v[0] = bf16_to_f32(*p);
        }
    }
    __device__ static void store(bf16_raw* p, const float (&v)[VEC]) {
        if constexpr (VEC == 8) {
            uint4 a;
            a.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            a.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            a.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
            a.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<uint4*>(p) = a;
        } else if constexpr (VEC == 4) {
            uint2 a;
            a.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            a.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(p) = a;
        } else {
            *p = f32_to_bf16(v[0]);
        }
    }
};

__device__ __forceinline__ float act_fwd(float y, int act) {
    if (act == MOMA_ACT_SILU) return y / (1.f + __expf(-y));
    if (act == MOMA_ACT_RELU) return fmaxf(y, 0.f);
    return y;
}
// d act(y) / dy
__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == MOMA_ACT_SILU) {
        const float s = 1.f / (1.f + __expf(-y));
        return s * (1.f + y * (1.f - s));
    }
    if (act == MOMA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}

// Chan's parallel update of (count, mean, M2)
__device__ __forceinline__ void merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    const float nn = n + nb;
    if (nn > 0.f) {
        const float delta = meanb - mean, f = nb / nn;
        mean += delta * f;
        m2 += m2b + delta * delta * n * f;
    }
    n = nn;
}

// ---- forward statistics: partial[(c*S + s)*3 ..] = (count, mean, M2) ------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                              int N, int C, int HW) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const unsigned hwv = HW / VEC, total = (unsigned)N * hwv;
    float sum = 0.f, sq = 0.f, cnt = 0.f;
    // per-thread plain sums over its (few hundred) elements only; everything wider is merged as (n, mean, M2)
    for (unsigned i = s * BN_THREADS + threadIdx.x; i < total; i += S * BN_THREADS) {
        const unsigned n = i / hwv, r = i - n * hwv;
        float v[VEC];
        VecIO<T, VEC>::load(x + ((size_t)n * C + c) * HW + (size_t)r * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            sum += v[j];
            sq = fmaf(v[j], v[j], sq);
        }
        cnt += (float)VEC;
    }
    float mean = cnt > 0.f ? sum / cnt : 0.f;
    float m2 = fmaxf(sq - sum * mean, 0.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float nb = __shfl_xor(cnt, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
        merge(cnt, mean, m2, nb, mb, qb);
    }
    __shared__ float red[BN_THREADS / 64][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[w][0] = cnt; red[w][1] = mean; red[w][2] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < BN_THREADS / 64; ++k) merge(cnt, mean, m2, red[k][0], red[k][1], red[k][2]);
        float* p = partial + ((size_t)c * S + s) * 3;
        p[0] = cnt; p[1] = mean; p[2] = m2;
    }
}

// one wave per channel (4 channels per workgroup): lane s holds partial s, the S <= 64 partials are merged with a
// butterfly of Chan updates (fixed order -> deterministic); writes save_mean / save_invstd / scale / shift and
// updates the running statistics
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int S,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          float* __restrict__ scale_shift, int C, int training,
                                                          float momentum, float eps) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float mean, invstd;
    if (training) {
        float n = 0.f, m2 = 0.f;
        mean = 0.f;
        if (lane < S) {
            const float* p = partial + ((size_t)c * S + lane) * 3;
            n = p[0]; mean = p[1]; m2 = p[2];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
            // both partners must end with the same triple: merge in a fixed (lower lane first) order
            if (lane & o) {
                float n2 = nb, mean2 = mb, m22 = qb;
                merge(n2, mean2, m22, n, mean, m2);
                n = n2; mean = mean2; m2 = m22;
            } else {
                merge(n, mean, m2, nb, mb, qb);
            }
        }
        const float var = m2 / n;                               // biased, used for normalisation
        invstd = rsqrtf(var + eps);
        if (lane == 0) {
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
        }
    } else {
        mean = running_mean[c];
        invstd = rsqrtf(running_var[c] + eps);
    }
    if (lane == 0) {
        if (save_mean) save_mean[c] = mean;
        if (save_invstd) save_invstd[c] = invstd;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale_shift[2 * c] = g * invstd;
        scale_shift[2 * c + 1] = b - mean * g * invstd;
    }
}

// out = act(x * scale[c] + shift[c])
template <typename T, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                              const float* __restrict__ scale_shift, int C, int HW,
                                                              size_t nvec, int act) {
    const unsigned hwv = HW / VEC;
    for (size_t i = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (size_t)gridDim.x * BN_THREADS) {
        const unsigned plane = (unsigned)(i / hwv), c = plane % (unsigned)C;
        const float sc = scale_shift[2 * c], sh = scale_shift[2 * c + 1];
        float v[VEC];
        VecIO<T, VEC>::load(x + i * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = act_fwd(fmaf(v[j], sc, sh), act);
        VecIO<T, VEC>::store(out + i * VEC, v);
    }
}

// same, one wave per (image, channel) plane, also emitting the plane's mean of the OUTPUT (the squeeze of the
// squeeze-excite block that follows BN1 + SiLU in an MBConv block: saves a separate read pass over the activation)
template <typename T, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_mean_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                                   T* __restrict__ pmean,
                                                                   const float* __restrict__ scale_shift, int NC, int C,
                                                                   int HW, int act) {
    const int lane = threadIdx.x & 63;
    const int nv = HW / VEC;
    for (int plane = blockIdx.x * (BN_THREADS / 64) + (threadIdx.x >> 6); plane < NC; plane += gridDim.x * (BN_THREADS / 64)) {
        const int c = plane % C;
        const float sc = scale_shift[2 * c], sh = scale_shift[2 * c + 1];
        const T* p = x + (size_t)plane * HW;
        T* o = out + (size_t)plane * HW;
        float acc = 0.f;
        for (int i = lane; i < nv; i += 64) {
            float v[VEC];
            VecIO<T, VEC>::load(p + (size_t)i * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                v[j] = act_fwd(fmaf(v[j], sc, sh), act);
                acc += v[j];
            }
            VecIO<T, VEC>::store(o + (size_t)i * VEC, v);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float m[1] = {acc / (float)HW};
            VecIO<T, 1>::store(pmean + plane, m);
        }
    }
}

// ---- backward reductions: partial[(c*S + s)*2 ..] = (sum dy, sum dy*xhat), dy = dout * act'(y) --------
// `dpl` (nullable): gradient of the plane means emitted by bn_apply_mean_kernel; every element of plane p then carries
// dout + dpl[p] / HW  (what autograd would otherwise materialise and add in two more passes)
template <typename T, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ save_mean,
                                                                   const float* __restrict__ save_invstd,
                                                                   float* __restrict__ partial, int N, int C, int HW,
                                                                   int act, const T* __restrict__ dpl) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const unsigned hwv = HW / VEC, total = (unsigned)N * hwv;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float inv_hw = 1.f / (float)HW;
    float s1 = 0.f, s2 = 0.f;
    for (unsigned i = s * BN_THREADS + threadIdx.x; i < total; i += S * BN_THREADS) {
        const unsigned n = i / hwv, r = i - n * hwv;
        const size_t off = ((size_t)n * C + c) * HW + (size_t)r * VEC;
        float xv[VEC], dv[VEC];
        VecIO<T, VEC>::load(x + off, xv);
        VecIO<T, VEC>::load(dout + off, dv);
        float add = 0.f;
        if (dpl) { float t1[1]; VecIO<T, 1>::load(dpl + (size_t)n * C + c, t1); add = t1[0] * inv_hw; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mean) * invstd;
            const float dy = (dv[j] + add) * act_grad(fmaf(xh, g, b), act);
            s1 += dy;
            s2 = fmaf(dy, xh, s2);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    __shared__ float red[BN_THREADS / 64][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[w][0] = s1; red[w][1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < BN_THREADS / 64; ++k) { s1 += red[k][0]; s2 += red[k][1]; }
        partial[((size_t)c * S + s) * 2] = s1;
        partial[((size_t)c * S + s) * 2 + 1] = s2;
    }
}

// per channel (one wave each): dgamma, dbeta and the coefficients of dx = a*dy - a*b1 - a*b2*xhat (training); eval: dx = a*dy
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int S,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef, int C,
                                                              float count, int training) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    if (lane < S) {
        s1 = partial[((size_t)c * S + lane) * 2];
        s2 = partial[((size_t)c * S + lane) * 2 + 1];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane != 0) return;
    if (dgamma) dgamma[c] = s2;
    if (dbeta) dbeta[c] = s1;
    const float a = (gamma ? gamma[c] : 1.f) * save_invstd[c];
    coef[3 * c] = a;
    coef[3 * c + 1] = training ? s1 / count : 0.f;
    coef[3 * c + 2] = training ? s2 / count : 0.f;
}

template <typename T, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                                  T* __restrict__ dx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd,
                                                                  const float* __restrict__ coef, int C, int HW,
                                                                  size_t nvec, int act, const T* __restrict__ dpl) {
    const unsigned hwv = HW / VEC;
    const float inv_hw = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (size_t)gridDim.x * BN_THREADS) {
        const unsigned plane = (unsigned)(i / hwv), c = plane % (unsigned)C;
        const float mean = save_mean[c], invstd = save_invstd[c];
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float a = coef[3 * c], b1 = coef[3 * c + 1], b2 = coef[3 * c + 2];
        float xv[VEC], dv[VEC];
        VecIO<T, VEC>::load(x + i * VEC, xv);
        VecIO<T, VEC>::load(dout + i * VEC, dv);
        float add = 0.f;
        if (dpl) { float t1[1]; VecIO<T, 1>::load(dpl + plane, t1); add = t1[0] * inv_hw; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - mean) * invstd;
            const float dy = (dv[j] + add) * act_grad(fmaf(xh, g, b), act);
            xv[j] = a * (dy - b1 - xh * b2);
        }
        VecIO<T, VEC>::store(dx + i * VEC, xv);
    }
}

int pick_vec(int HW, int elem_bytes, const void* a, const void* b, const void* c) {
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    if (HW % 8 == 0 && bits % (8 * elem_bytes) == 0) return 8;
    if (HW % 4 == 0 && bits % (4 * elem_bytes) == 0) return 4;
    return 1;
}
int pick_splits(int N, int C, int HW, int vec) {
    const long per_channel = (long)N * (HW / vec);
    long S = (4096 + C - 1) / C;                                  // >= ~4096 workgroups in all
    const long max_s = (per_channel + BN_THREADS * 2 - 1) / (BN_THREADS * 2);   // >= 2 vectors per thread
    if (S > max_s) S = max_s;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return (int)S;
}
unsigned apply_grid(size_t nvec) {
    size_t g = (nvec + BN_THREADS * 4 - 1) / (BN_THREADS * 4);   // ~4 vectors per thread
    if (g > 256u * 64u) g = 256u * 64u;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <typename T>
hipError_t bn_fwd_t(const T* x, T* out, const float* gamma, const float* beta, float* rm, float* rv, float* save_mean,
                    float* save_invstd, float* ws, int N, int C, int HW, int act, int training, float momentum,
                    float eps, T* pmean, hipStream_t st) {
    const int vec = pick_vec(HW, sizeof(T), x, out, nullptr);
    const int S = training ? pick_splits(N, C, HW, vec) : 1;
    float* scale_shift = ws;                   // [C][2]
    float* partial = ws + 2 * (size_t)C;       // [C][S][3]
    const size_t nvec = (size_t)N * C * HW / vec;
    if (training) {
        dim3 grid(S, C);
        if (vec == 8) hipLaunchKernelGGL((bn_stats_kernel<T, 8>), grid, dim3(BN_THREADS), 0, st, x, partial, N, C, HW);
        else if (vec == 4) hipLaunchKernelGGL((bn_stats_kernel<T, 4>), grid, dim3(BN_THREADS), 0, st, x, partial, N, C, HW);
        else hipLaunchKernelGGL((bn_stats_kernel<T, 1>), grid, dim3(BN_THREADS), 0, st, x, partial, N, C, HW);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, S, gamma, beta, rm, rv,
                       save_mean, save_invstd, scale_shift, C, training, momentum, eps);
    if (pmean) {
        long gp = ((long)N * C + BN_THREADS / 64 - 1) / (BN_THREADS / 64);
        if (gp > 256 * 32) gp = 256 * 32;
        const dim3 pg((unsigned)gp);
        if (vec == 8) hipLaunchKernelGGL((bn_apply_mean_kernel<T, 8>), pg, dim3(BN_THREADS), 0, st, x, out, pmean, scale_shift, N * C, C, HW, act);
        else if (vec == 4) hipLaunchKernelGGL((bn_apply_mean_kernel<T, 4>), pg, dim3(BN_THREADS), 0, st, x, out, pmean, scale_shift, N * C, C, HW, act);
        else hipLaunchKernelGGL((bn_apply_mean_kernel<T, 1>), pg, dim3(BN_THREADS), 0, st, x, out, pmean, scale_shift, N * C, C, HW, act);
        return hipGetLastError();
    }
    const unsigned g = apply_grid(nvec);
    if (vec == 8) hipLaunchKernelGGL((bn_apply_kernel<T, 8>), dim3(g), dim3(BN_THREADS), 0, st, x, out, scale_shift, C, HW, nvec, act);
    else if (vec == 4) hipLaunchKernelGGL((bn_apply_kernel<T, 4>), dim3(g), dim3(BN_THREADS), 0, st, x, out, scale_shift, C, HW, nvec, act);
    else hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(g), dim3(BN_THREADS), 0, st, x, out, scale_shift, C, HW, nvec, act);
    return hipGetLastError();
}

template <typename T>
hipError_t bn_bwd_t(const T* x, const T* dout, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, T* dx, float* dgamma, float* dbeta, float* ws, int N, int C, int HW,
                    int act, int training, const T* dpl, hipStream_t st) {
    const int vec = pick_vec(HW, sizeof(T), x, dout, dx);
    const int S = pick_splits(N, C, HW, vec);
    float* coef = ws;                          // [C][3]
    float* partial = ws + 3 * (size_t)C;       // [C][S][2]
    const size_t nvec = (size_t)N * C * HW / vec;
    dim3 grid(S, C);
    if (vec == 8) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 8>), grid, dim3(BN_THREADS), 0, st, x, dout, gamma, beta, save_mean, save_invstd, partial, N, C, HW, act, dpl);
    else if (vec == 4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 4>), grid, dim3(BN_THREADS), 0, st, x, dout, gamma, beta, save_mean, save_invstd, partial, N, C, HW, act, dpl);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), grid, dim3(BN_THREADS), 0, st, x, dout, gamma, beta, save_mean, save_invstd, partial, N, C, HW, act, dpl);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, S, gamma, save_invstd,
                       dgamma, dbeta, coef, C, (float)N * (float)HW, training);
    if (dx) {
        const unsigned g = apply_grid(nvec);
        if (vec == 8) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 8>), dim3(g), dim3(BN_THREADS), 0, st, x, dout, dx, gamma, beta, save_mean, save_invstd, coef, C, HW, nvec, act, dpl);
        else if (vec == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(g), dim3(BN_THREADS), 0, st, x, dout, dx, gamma, beta, save_mean, save_invstd, coef, C, HW, nvec, act, dpl);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), dim3(g), dim3(BN_THREADS), 0, st, x, dout, dx, gamma, beta, save_mean, save_invstd, coef, C, HW, nvec, act, dpl);
    }
    return hipGetLastError();
}
}  // namespace

size_t bn_workspace_floats(int C) { return (size_t)C * (3 + 64 * 3); }

hipError_t launch_bn_fwd(const void* x, void* out, const float* gamma, const float* beta, float* rm, float* rv,
                         float* save_mean, float* save_invstd, float* ws, int N, int C, int HW, int dtype, int act,
                         int training, float momentum, float eps, void* plane_mean, hipStream_t st) {
    if (dtype == MOMA_DT_BF16)
        return bn_fwd_t<bf16_raw>((const bf16_raw*)x, (bf16_raw*)out, gamma, beta, rm, rv, save_mean, save_invstd, ws, N,
                                  C, HW, act, training, momentum, eps, (bf16_raw*)plane_mean, st);
    return bn_fwd_t<float>((const float*)x, (float*)out, gamma, beta, rm, rv, save_mean, save_invstd, ws, N, C, HW, act,
                           training, momentum, eps, (float*)plane_mean, st);
}
hipError_t launch_bn_bwd(const void* x, const void* dout, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, void* dx, float* dgamma, float* dbeta, float* ws, int N, int C, int HW,
                         int dtype, int act, int training, const void* dplane_mean, hipStream_t st) {
    if (dtype == MOMA_DT_BF16)
        return bn_bwd_t<bf16_raw>((const bf16_raw*)x, (const bf16_raw*)dout, gamma, beta, save_mean, save_invstd,
                                  (bf16_raw*)dx, dgamma, dbeta, ws, N, C, HW, act, training, (const bf16_raw*)dplane_mean, st);
    return bn_bwd_t<float>((const float*)x, (const float*)dout, gamma, beta, save_mean, save_invstd, (float*)dx, dgamma,
                           dbeta, ws, N, C, HW, act, training, (const float*)dplane_mean, st);
}

}  // namespace moma
