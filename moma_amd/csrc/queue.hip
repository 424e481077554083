// K3 ring-buffer enqueue and K4 multi-tensor EMA: pure HBM streaming kernels (16 B per lane where the
// operands are 16-B aligned).
#include "common.hpp"

namespace moma {
namespace {

// queue[(index + i) mod K, :] = rows[i, :]      (MoMA/mem_moco.py:17-27)
// one workgroup per source row; rows whose slot is rewritten by a later row (n > K) are skipped so the
// result equals a serial index_copy_ (last writer wins) without a write race.
template <typename TQ, bool VEC4>
__global__ __launch_bounds__(256) void enqueue_kernel(TQ* __restrict__ queue, const float* __restrict__ rows, int n,
                                                       int64_t index, int K, int d) {
    const int i = blockIdx.x;
    if ((int64_t)i + K < n) return;
    const int64_t slot = (index + i) % K;
    const float* src = rows + (int64_t)i * d;
    TQ* dst = queue + slot * d;
    if constexpr (VEC4) {
        for (int c = threadIdx.x * 4; c < d; c += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            if constexpr (sizeof(TQ) == 4) {
                *reinterpret_cast<float4*>(dst + c) = v;
            } else {
                ushort4 o;
                o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
                *reinterpret_cast<ushort4*>(dst + c) = o;
            }
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) {
            if constexpr (sizeof(TQ) == 4) dst[c] = src[c];
            else dst[c] = f32_to_bf16(src[c]);
        }
    }
}

// the same for an fp32 queue AND its bf16 mirror in one launch: both rows come from one read of the source row
__global__ __launch_bounds__(256) void enqueue_mirror_kernel(float* __restrict__ queue, bf16_raw* __restrict__ mirror,
                                                             const float* __restrict__ rows, int n, int64_t index, int K, int d) {
    const int i = blockIdx.x;
    if ((int64_t)i + K < n) return;
    const int64_t slot = (index + i) % K;
    const float* src = rows + (int64_t)i * d;
    float* dst = queue + slot * d;
    bf16_raw* dm = mirror + slot * d;
    const bool vec = ((((uintptr_t)queue | (uintptr_t)rows) & 15) == 0) && (((uintptr_t)mirror & 7) == 0) && d % 4 == 0;
    if (vec) {
        for (int c = threadIdx.x * 4; c < d; c += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            *reinterpret_cast<float4*>(dst + c) = v;
            ushort4 o;
            o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
            *reinterpret_cast<ushort4*>(dm + c) = o;
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) {
            dst[c] = src[c];
            dm[c] = f32_to_bf16(src[c]);
        }
    }
}

// Touch `bytes` of memory with streaming 16-B loads and drop the data: pulls the lines into the memory-side Infinity Cache
// (256 MiB) so that the kernel that streams them next reads them at cache latency instead of HBM latency.  A hint only.
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, size_t n16) {
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];                 // (default cache policy: the point is that the lines STAY in the caches)
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    asm volatile("" ::"v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));          // keep the loads
}

// bf16 -> fp32 (exact) of a whole queue: the exact-fp32 K2 policy over a bf16-stored queue streams the widened copy (api.hip)
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, size_t n8,
                                                         const bf16_raw* __restrict__ tail_src, float* __restrict__ tail_dst, int ntail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        dst[2 * i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                                 __uint_as_float(v.y & 0xffff0000u));
        dst[2 * i + 1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16),
                                     __uint_as_float(v.w & 0xffff0000u));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail_dst[threadIdx.x] = bf16_to_f32(tail_src[threadIdx.x]);
}

// ema = fma(1-m, p, ema*m) over a table of tensors   (learning/contrast_trainer.py:207-211)
__global__ __launch_bounds__(256) void ema_kernel(const int64_t* __restrict__ table, int n_tensors, float m, float om) {
    const int64_t blk = blockIdx.x;
    // binary search: last t with first_block[t] <= blk
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 4 + 3] <= blk) lo = mid;
        else hi = mid - 1;
    }
    float* __restrict__ e = reinterpret_cast<float*>(table[lo * 4 + 0]);
    const float* __restrict__ p = reinterpret_cast<const float*>(table[lo * 4 + 1]);
    const int64_t numel = table[lo * 4 + 2];
    const int64_t off = (blk - table[lo * 4 + 3]) * MOMA_EMA_BLOCK_ELEMS;
    const int64_t rem = numel - off;
    const bool vec = (((uintptr_t)e | (uintptr_t)p) & 15) == 0;
    if (vec && rem >= MOMA_EMA_BLOCK_ELEMS) {
#pragma unroll
        for (int i = 0; i < MOMA_EMA_BLOCK_ELEMS / (256 * 4); ++i) {
            const int64_t idx = off + (int64_t)(i * 256 + threadIdx.x) * 4;
            float4 ev = *reinterpret_cast<const float4*>(e + idx);
            const float4 pv = *reinterpret_cast<const float4*>(p + idx);
            ev.x = fmaf(om, pv.x, ev.x * m);
            ev.y = fmaf(om, pv.y, ev.y * m);
            ev.z = fmaf(om, pv.z, ev.z * m);
            ev.w = fmaf(om, pv.w, ev.w * m);
            *reinterpret_cast<float4*>(e + idx) = ev;
        }
    } else {
        const int64_t end = rem < MOMA_EMA_BLOCK_ELEMS ? rem : MOMA_EMA_BLOCK_ELEMS;
        for (int64_t j = threadIdx.x; j < end; j += 256) e[off + j] = fmaf(om, p[off + j], e[off + j] * m);
    }
}
}  // namespace

hipError_t launch_enqueue(void* queue, const float* rows, int n, int64_t index, int K, int d, int qdtype, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const bool a16 = (((uintptr_t)queue | (uintptr_t)rows) & 15) == 0;
    dim3 grid(n), block(256);
    if (qdtype == MOMA_DT_BF16) {
        if (a16 && d % 8 == 0) hipLaunchKernelGGL((enqueue_kernel<bf16_raw, true>), grid, block, 0, st, (bf16_raw*)queue, rows, n, index, K, d);
        else hipLaunchKernelGGL((enqueue_kernel<bf16_raw, false>), grid, block, 0, st, (bf16_raw*)queue, rows, n, index, K, d);
    } else {
        if (a16 && d % 4 == 0) hipLaunchKernelGGL((enqueue_kernel<float, true>), grid, block, 0, st, (float*)queue, rows, n, index, K, d);
        else hipLaunchKernelGGL((enqueue_kernel<float, false>), grid, block, 0, st, (float*)queue, rows, n, index, K, d);
    }
    return hipGetLastError();
}

hipError_t launch_enqueue_mirror(float* queue, void* mirror, const float* rows, int n, int64_t index, int K, int d, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(enqueue_mirror_kernel, dim3(n), dim3(256), 0, st, queue, (bf16_raw*)mirror, rows, n, index, K, d);
    return hipGetLastError();
}

hipError_t launch_prefetch(const void* p, size_t bytes, hipStream_t st) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(prefetch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)p, n16);
    return hipGetLastError();
}

hipError_t launch_widen_bf16(const void* src, float* dst, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const size_t n8 = n / 8;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const bf16_raw* s16 = (const bf16_raw*)src;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)src, (float4*)dst, n8, s16 + n8 * 8,
                       dst + n8 * 8, (int)(n - n8 * 8));
    return hipGetLastError();
}

hipError_t launch_ema(const int64_t* table, int n_tensors, int64_t total_blocks, float m, float om, hipStream_t st) {
    if (n_tensors <= 0 || total_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, table, n_tensors, m, om);
    return hipGetLastError();
}

}  // namespace moma
