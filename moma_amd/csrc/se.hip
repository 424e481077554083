// Squeeze-excite helpers on NCHW activations: per-plane mean (the "squeeze") and the sigmoid gate
// out = x * sigmoid(s[n,c]) with its backward (reference: models/efficientnet_pytorch/model.py:104-110 --
// `x_squeezed = F.adaptive_avg_pool2d(x, 1)` ... `x = torch.sigmoid(x_squeezed) * x`).  HBM streaming: one wave per
// (image, channel) plane, vector loads along the plane's contiguous memory.  The gate's backward makes
// dx = dout * sigmoid(s) and ds = sigmoid'(s) * sum_hw(dout * x) in ONE pass over (dout, x) -- stock autograd takes a
// multiply, another multiply plus a full-size temporary, and a reduction for the same two results.
#include "common.hpp"

namespace moma {
namespace {

template <typename T, int VEC> struct PV;
template <> struct PV<float, 4> {
    static __device__ void ld(const float* p, float* v) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
    static __device__ void st(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct PV<float, 1> {
    static __device__ void ld(const float* p, float* v) { v[0] = *p; }
    static __device__ void st(float* p, const float* v) { *p = v[0]; }
};
template <> struct PV<bf16_raw, 8> {
    static __device__ void ld(const bf16_raw* p, float* v) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ void st(bf16_raw* p, const float* v) {
        uint4 a;
        a.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        a.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        a.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        a.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        *reinterpret_cast<uint4*>(p) = a;
    }
};
template <> struct PV<bf16_raw, 4> {
    static __device__ void ld(const bf16_raw* p, float* v) {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    }
    static __device__ void st(bf16_raw* p, const float* v) {
        uint2 a;
        a.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        a.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = a;
    }
};
template <> struct PV<bf16_raw, 1> {
    static __device__ void ld(const bf16_raw* p, float* v) { v[0] = bf16_to_f32(*p); }
    static __device__ void st(bf16_raw* p, const float* v) { *p = f32_to_bf16(v[0]); }
};
template <typename T> __device__ __forceinline__ float ld1(const T* p) { float v; PV<T, 1>::ld(p, &v); return v; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { PV<T, 1>::st(p, &v); }

constexpr int SE_WAVES = 4;

// out[plane] = mean over the plane (fp32 accumulation)
template <typename T, int VEC>
__global__ __launch_bounds__(SE_WAVES * 64) void plane_mean_kernel(const T* __restrict__ x, T* __restrict__ out, int NC, int HW) {
    const int lane = threadIdx.x & 63;
    const int nv = HW / VEC;
    for (int plane = blockIdx.x * SE_WAVES + (threadIdx.x >> 6); plane < NC; plane += gridDim.x * SE_WAVES) {
        const T* p = x + (size_t)plane * HW;
        float s = 0.f;
        for (int i = lane; i < nv; i += 64) {
            float v[VEC];
            PV<T, VEC>::ld(p + (size_t)i * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += v[j];
        }
        s = wave_sum(s);
        if (lane == 0) st1<T>(out + plane, s / (float)HW);
    }
}

// out = x * sigmoid(s[plane])
template <typename T, int VEC>
__global__ __launch_bounds__(SE_WAVES * 64) void se_gate_fwd_kernel(const T* __restrict__ x, const T* __restrict__ s,
                                                                  T* __restrict__ out, int NC, int HW) {
    const int lane = threadIdx.x & 63;
    const int nv = HW / VEC;
    for (int plane = blockIdx.x * SE_WAVES + (threadIdx.x >> 6); plane < NC; plane += gridDim.x * SE_WAVES) {
        const float g = 1.f / (1.f + __expf(-ld1<T>(s + plane)));
        const T* p = x + (size_t)plane * HW;
        T* o = out + (size_t)plane * HW;
        for (int i = lane; i < nv; i += 64) {
            float v[VEC];
            PV<T, VEC>::ld(p + (size_t)i * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] *= g;
            PV<T, VEC>::st(o + (size_t)i * VEC, v);
        }
    }
}

// dx = dout * sigmoid(s);  ds[plane] = sigmoid(s) * (1 - sigmoid(s)) * sum(dout * x)
template <typename T, int VEC>
__global__ __launch_bounds__(SE_WAVES * 64) void se_gate_bwd_kernel(const T* __restrict__ x, const T* __restrict__ s,
                                                                  const T* __restrict__ dout, T* __restrict__ dx,
                                                                  T* __restrict__ ds, int NC, int HW) {
    const int lane = threadIdx.x & 63;
    const int nv = HW / VEC;
    for (int plane = blockIdx.x * SE_WAVES + (threadIdx.x >> 6); plane < NC; plane += gridDim.x * SE_WAVES) {
        const float g = 1.f / (1.f + __expf(-ld1<T>(s + plane)));
        const T* p = x + (size_t)plane * HW;
        const T* d = dout + (size_t)plane * HW;
        T* o = dx + (size_t)plane * HW;
        float acc = 0.f;
        for (int i = lane; i < nv; i += 64) {
            float v[VEC], dv[VEC];
            PV<T, VEC>::ld(p + (size_t)i * VEC, v);
            PV<T, VEC>::ld(d + (size_t)i * VEC, dv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc = fmaf(dv[j], v[j], acc);
                dv[j] *= g;
            }
            PV<T, VEC>::st(o + (size_t)i * VEC, dv);
        }
        acc = wave_sum(acc);
        if (lane == 0) st1<T>(ds + plane, acc * g * (1.f - g));
    }
}

int se_vec(int HW, int elem_bytes, uintptr_t bits) {
    const int maxv = 16 / elem_bytes;
    for (int v = maxv; v > 1; v >>= 1)
        if (v != 2 && HW % v == 0 && bits % (v * elem_bytes) == 0) return v;
    return 1;
}
unsigned se_grid(int NC) {
    long g = ((long)NC + SE_WAVES - 1) / SE_WAVES;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)(g < 1 ? 1 : g);
}
}  // namespace

#define MOMA_SE_LAUNCH(KERNEL, T, MAXV, ...)                                                                       \
    do {                                                                                                           \
        if (MAXV == 8 && vec == 8) hipLaunchKernelGGL((KERNEL<T, MAXV>), dim3(se_grid(NC)), dim3(SE_WAVES * 64), 0, st, __VA_ARGS__); \
        else if (vec >= 4) hipLaunchKernelGGL((KERNEL<T, 4>), dim3(se_grid(NC)), dim3(SE_WAVES * 64), 0, st, __VA_ARGS__);            \
        else hipLaunchKernelGGL((KERNEL<T, 1>), dim3(se_grid(NC)), dim3(SE_WAVES * 64), 0, st, __VA_ARGS__);                          \
    } while (0)

hipError_t launch_plane_mean(const void* x, void* out, int NC, int HW, int dtype, hipStream_t st) {
    if (dtype == MOMA_DT_BF16) {
        const int vec = se_vec(HW, 2, (uintptr_t)x);
        MOMA_SE_LAUNCH(plane_mean_kernel, bf16_raw, 8, (const bf16_raw*)x, (bf16_raw*)out, NC, HW);
    } else {
        const int vec = se_vec(HW, 4, (uintptr_t)x);
        MOMA_SE_LAUNCH(plane_mean_kernel, float, 4, (const float*)x, (float*)out, NC, HW);
    }
    return hipGetLastError();
}
hipError_t launch_se_gate_fwd(const void* x, const void* s, void* out, int NC, int HW, int dtype, hipStream_t st) {
    if (dtype == MOMA_DT_BF16) {
        const int vec = se_vec(HW, 2, (uintptr_t)x | (uintptr_t)out);
        MOMA_SE_LAUNCH(se_gate_fwd_kernel, bf16_raw, 8, (const bf16_raw*)x, (const bf16_raw*)s, (bf16_raw*)out, NC, HW);
    } else {
        const int vec = se_vec(HW, 4, (uintptr_t)x | (uintptr_t)out);
        MOMA_SE_LAUNCH(se_gate_fwd_kernel, float, 4, (const float*)x, (const float*)s, (float*)out, NC, HW);
    }
    return hipGetLastError();
}
hipError_t launch_se_gate_bwd(const void* x, const void* s, const void* dout, void* dx, void* ds, int NC, int HW, int dtype,
                              hipStream_t st) {
    if (dtype == MOMA_DT_BF16) {
        const int vec = se_vec(HW, 2, (uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx);
        MOMA_SE_LAUNCH(se_gate_bwd_kernel, bf16_raw, 8, (const bf16_raw*)x, (const bf16_raw*)s, (const bf16_raw*)dout,
                       (bf16_raw*)dx, (bf16_raw*)ds, NC, HW);
    } else {
        const int vec = se_vec(HW, 4, (uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx);
        MOMA_SE_LAUNCH(se_gate_bwd_kernel, float, 4, (const float*)x, (const float*)s, (const float*)dout, (float*)dx,
                       (float*)ds, NC, HW);
    }
    return hipGetLastError();
}

}  // namespace moma
