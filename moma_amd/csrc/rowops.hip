// Row-wise kernels around the generic GEMM stages: softmax and its backward over score rows, bias
// gradients, the positive logit, and the cross-entropy-with-label-0 reduction over materialised logits.
// One 64-lane wave per row where rows are short (attention: N columns), one 256-thread workgroup per
// row where they are long (InfoNCE: K+1 columns); reductions by wave shuffles.
#include "common.hpp"

namespace moma {
namespace {

// in place: s[r,:] = softmax(s[r,:])      (MoMA/criterion_moco_att.py:160)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long rows, int cols) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = s + row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, p[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(p[c] - m);
        p[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < cols; c += 64) p[c] *= inv;
}

// in place on dp: ds = p * (dp - sum_j dp_j p_j) * scale   (backward of softmax then of the *scale)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                                long rows, int cols, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * cols;
    float* dr = dp + row * cols;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - dot) * scale;
}

// out[c] = sum_r x[r*ld + c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                      int cols, long ld) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(long)r * ld + c];
    out[c] = s;
}

// same, 64 columns per block as 16 float4 lanes x 16 row groups: every thread has rows/16 independent loads in
// flight (the scalar kernel above is one dependent chain of `rows` loads per thread), fixed summation order
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                          int cols, long ld) {
    __shared__ float4 red[16][16];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cx * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
#pragma unroll 8
        for (int r = ry; r < rows; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) {
#pragma unroll
        for (int g = 1; g < 16; ++g) {
            const float4 v = red[g][cx];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + c) = acc;
    }
}

// out[b*ld_out] = <q_b, k_b> * inv_T        (MoMA/mem_moco.py:37-38,45)
__global__ __launch_bounds__(256) void pos_logit_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         float* __restrict__ out, long ld_out, int B, int d,
                                                         float inv_T) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s = fmaf(q[(long)row * d + c], k[(long)row * d + c], s);
    s = wave_sum(s);
    if (lane == 0) out[(long)row * ld_out] = s * inv_T;
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// CrossEntropy(logits, label 0) per row + top-1 flag; optionally overwrite the row with d(sum loss)/dlogits
// = softmax(row) - onehot(0)   (helper/loops_moma.py:322,332-335; learning/contrast_trainer.py:197-204)
__global__ __launch_bounds__(256) void infonce_rows_kernel(float* __restrict__ logits, int ncols,
                                                            float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                            int32_t* __restrict__ top1, int write_probs) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float* row = logits + (long)b * ncols;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < ncols; c += 256) m = fmaxf(m, row[c]);
    m = block_reduce(m, red, true);
    float s = 0.f;
    for (int c = threadIdx.x; c < ncols; c += 256) s += expf(row[c] - m);
    s = block_reduce(s, red, false);
    const float lse = m + logf(s);
    const float s0 = row[0];
    __syncthreads();
    if (write_probs) {
        for (int c = threadIdx.x; c < ncols; c += 256) {
            float p = expf(row[c] - lse);
            if (c == 0) p -= 1.f;
            row[c] = p;
        }
    }
    if (threadIdx.x == 0) {
        lse_out[b] = lse;
        loss_rows[b] = lse - s0;
        top1[b] = (s0 >= m) ? 1 : 0;
    }
}

// dq[b,:] = dlogits[b,0] * k[b,:] * inv_T   (initialises dq before the split-K queue product is added)
__global__ __launch_bounds__(256) void pos_grad_init_kernel(const float* __restrict__ dlogits, long ld,
                                                             const float* __restrict__ k, float* __restrict__ dq,
                                                             int B, int d, float inv_T) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * d) return;
    const int b = (int)(i / d);
    dq[i] = dlogits[(long)b * ld] * k[i] * inv_T;
}

// dst += the partial products of a split-K GEMM, in split order (bitwise reproducible where atomics are not)
__global__ __launch_bounds__(256) void add_partials_kernel(float* __restrict__ dst, const float* __restrict__ parts, int nparts, long n,
                                                            long stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = dst[i];
    for (int s = 0; s < nparts; ++s) v += parts[(long)s * stride + i];
    dst[i] = v;
}
}  // namespace

hipError_t launch_softmax_rows(float* s, long rows, int cols, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, s, rows, cols);
    return hipGetLastError();
}
hipError_t launch_softmax_bwd_rows(const float* p, float* dp, long rows, int cols, float scale, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, dp, rows, cols, scale);
    return hipGetLastError();
}
hipError_t launch_colsum(const float* x, float* out, int rows, int cols, long ld, hipStream_t st) {
    if (cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, x, out, rows, cols, ld);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, x, out, rows, cols, ld);
    return hipGetLastError();
}
hipError_t launch_pos_logit(const float* q, const float* k, float* out, long ld_out, int B, int d, float inv_T, hipStream_t st) {
    hipLaunchKernelGGL(pos_logit_kernel, dim3((B + 3) / 4), dim3(256), 0, st, q, k, out, ld_out, B, d, inv_T);
    return hipGetLastError();
}
hipError_t launch_infonce_rows(float* logits, int B, int ncols, float* loss_rows, float* lse, int32_t* top1,
                               int write_probs, hipStream_t st) {
    hipLaunchKernelGGL(infonce_rows_kernel, dim3(B), dim3(256), 0, st, logits, ncols, loss_rows, lse, top1, write_probs);
    return hipGetLastError();
}
hipError_t launch_add_partials(float* dst, const float* parts, int nparts, long n, long stride, hipStream_t st) {
    if (n <= 0 || nparts <= 0) return hipSuccess;
    hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, parts, nparts, n, stride);
    return hipGetLastError();
}
hipError_t launch_pos_grad_init(const float* dlogits, long ld, const float* k, float* dq, int B, int d, float inv_T,
                                hipStream_t st) {
    const long n = (long)B * d;
    hipLaunchKernelGGL(pos_grad_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dlogits, ld, k, dq, B, d, inv_T);
    return hipGetLastError();
}

}  // namespace moma
