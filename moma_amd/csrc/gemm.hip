// Generic batched MFMA GEMM used by the non-fused stages (projection linears, their gradients, the
// per-head score / context products and the materialised-logits compatibility path).
//   C[m,n] (+)= alpha * sum_k A(m,k) * B(n,k) + bias[n]
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave 2x2 MFMA 16x16 tiles), K in
// steps of 32 through fp32 LDS tiles.  Two arithmetic policies on the same tiles:
//   MOMA_PREC_F32  : v_mfma_f32_16x16x4_f32  (exact fp32 fma chain = the reference's arithmetic)
//   MOMA_PREC_BF16 : v_mfma_f32_16x16x32_bf16 (operands rounded to bf16 at fragment load, fp32 accumulate)
#include "common.hpp"

namespace moma {

namespace {
constexpr int BM = 64, BN = 64, BK = 32, LDK = BK + 4;  // +4 floats: 16-B aligned rows, spreads banks

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_raw>(const bf16_raw* p) { return bf16_to_f32(*p); }

// Stage a [64 rows] x [BK k] tile of X into S[row][k] (fp32), zero-filling out-of-range elements.
//   X(row,k) = trans ? X[k*ld + row] : X[row*ld + k]
template <typename T>
__device__ __forceinline__ void stage_tile(float (*S)[LDK], const T* __restrict__ X, long ld, int trans, int r0,
                                           int rows, int k0, int kend, int tid) {
    if (!trans) {
#pragma unroll
        for (int i = 0; i < (64 * BK) / 256; ++i) {
            const int idx = tid + i * 256;
            const int r = idx >> 5, c = idx & 31;
            const int gr = r0 + r, gk = k0 + c;
            float v = 0.f;
            if (gr < rows && gk < kend) v = ldf<T>(X + (long)gr * ld + gk);
            S[r][c] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < (64 * BK) / 256; ++i) {
            const int idx = tid + i * 256;
            const int kr = idx >> 6, c = idx & 63;
            const int gr = r0 + c, gk = k0 + kr;
            float v = 0.f;
            if (gr < rows && gk < kend) v = ldf<T>(X + (long)gk * ld + gr);
            S[c][kr] = v;
        }
    }
}

template <int PREC, typename TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int batch = blockIdx.z / g.splitk, split = blockIdx.z % g.splitk;
    const float* A = g.A + (long)batch * g.strideA;
    const TB* B = reinterpret_cast<const TB*>(g.B) + (long)batch * g.strideB;
    float* C = g.C + (long)batch * g.strideC;

    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kbeg = split * per * BK;
    const int kend = min(g.K, kbeg + per * BK);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage_tile<float>(As, A, g.lda, g.transA, m0, g.M, k0, kend, tid);
        stage_tile<TB>(Bs, B, g.ldb, g.transB, n0, g.N, k0, kend, tid);
        __syncthreads();
        if constexpr (PREC == MOMA_PREC_BF16) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* pa = &As[wm * 32 + i * 16 + fr][fg * 8];
                const float* pb = &Bs[wn * 32 + i * 16 + fr][fg * 8];
                const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
                af[i] = bf16x8{(__bf16)a0.x, (__bf16)a0.y, (__bf16)a0.z, (__bf16)a0.w,
                               (__bf16)a1.x, (__bf16)a1.y, (__bf16)a1.z, (__bf16)a1.w};
                bf[i] = bf16x8{(__bf16)b0.x, (__bf16)b0.y, (__bf16)b0.z, (__bf16)b0.w,
                               (__bf16)b1.x, (__bf16)b1.y, (__bf16)b1.z, (__bf16)b1.w};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = As[wm * 32 + i * 16 + fr][kk * 4 + fg];
                    bf[i] = Bs[wn * 32 + i * 16 + fr][kk * 4 + fg];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // C/D layout of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + fg * 4 + r;
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][r] * g.alpha;
                    if (g.bias != nullptr && split == 0) v += g.bias[col];
                    float* dst = C + (long)row * g.ldc + col;
                    if (g.atomic) atomicAdd(dst, v);
                    else *dst = v;
                }
            }
        }
}
}  // namespace

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return hipSuccess;
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.batch * a.splitk);
    dim3 block(256);
    if (a.prec == MOMA_PREC_BF16) {
        if (a.b_dtype == MOMA_DT_BF16) hipLaunchKernelGGL((gemm_kernel<MOMA_PREC_BF16, bf16_raw>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<MOMA_PREC_BF16, float>), grid, block, 0, s, a);
    } else {
        if (a.b_dtype == MOMA_DT_BF16) hipLaunchKernelGGL((gemm_kernel<MOMA_PREC_F32, bf16_raw>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<MOMA_PREC_F32, float>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace moma
