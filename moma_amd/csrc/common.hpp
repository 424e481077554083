// Shared device helpers and internal launch declarations for libmoma_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/moma_hip.h"

namespace moma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_raw;  // storage type of a bf16 queue element

constexpr int WAVE = 64;

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) {
    return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    // plain cast -> v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_raw*>(&b);
}

// two floats -> one word of two bf16 (lo in bits 0..15), round to nearest even: ONE v_cvt_pk_bf16_f32.  Written as
// `f32_to_bf16(lo) | f32_to_bf16(hi) << 16` hipcc pairs the conversions of a 16-byte store the wrong way round (values 0,2 / 1,3)
// and puts the words together again with v_and / v_lshl / 2 x v_or_sdwa: 20 VALU instructions per store instead of 12 -- seen
// in round 4 in the partial stores of the one-pass K2 kernel, whose last tile took 2.9 us instead of ~1
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// value of lane l ^ 32 (the other half of the wave) by one v_permlane32_swap -- a VALU instruction; __shfl_xor(v, 32) is a
// ds_bpermute, i.e. an LDS round trip with a wait
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = {lo, lo}, r[1] = {hi, hi}
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// Reductions over the 32 lanes of a wave HALF (lanes 0-31 / 32-63), result in every lane, without LDS: four DPP rotations inside
// the 16-lane rows (x op= ror 8, 4, 2, 1: a butterfly, so every lane of a row ends with the same bits) and one v_permlane16_swap
// across the two rows of the half.  __shfl_xor is a ds_bpermute each -- five dependent LDS round trips per reduction.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float half32_sum(float v) {
    v += dpp_row<0x128>(v);            // row_ror:8
    v += dpp_row<0x124>(v);            // row_ror:4
    v += dpp_row<0x122>(v);            // row_ror:2
    v += dpp_row<0x121>(v);            // row_ror:1
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // r[0] = rows {0, 0, 2, 2}, r[1] = rows {1, 1, 3, 3}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half32_max(float v) {
    v = fmaxf(v, dpp_row<0x128>(v));
    v = fmaxf(v, dpp_row<0x124>(v));
    v = fmaxf(v, dpp_row<0x122>(v));
    v = fmaxf(v, dpp_row<0x121>(v));
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- generic batched GEMM:  C[m,n] (+)= alpha * sum_k A(m,k) * B(n,k) + bias[n] ----------------
//   A(m,k) = transA ? A[k*lda + m] : A[m*lda + k]      (fp32)
//   B(n,k) = transB ? B[k*ldb + n] : B[n*ldb + k]      (fp32 or bf16 storage)
//   batch b adds b*strideX elements to each base pointer; split-K partial sums are combined with
//   fp32 atomics into a C the caller has initialised (atomic != 0).
struct GemmArgs {
    const float* A;
    const void* B;
    float* C;
    const float* bias;   // nullable, length N
    int M, N, K;
    long lda, ldb, ldc;
    long strideA, strideB, strideC;
    int batch;
    int transA, transB;
    float alpha;
    int splitk;          // >= 1
    int atomic;          // 1: atomicAdd into C
    long splitC;         // splitk > 1 without atomics: split s writes its partial product to C + s * splitC (the caller sums them)
    int b_dtype;         // MOMA_DT_*
    int prec;            // MOMA_PREC_*
    float* colsum_a;     // nullable: also write sum_k A(m,k) for every m (the bias gradient next to dW = dY^T X); only honoured
                         // where gemm_fuses_colsum(args) holds, zero-initialised by value-initialisation otherwise ignored
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
bool gemm_fuses_colsum(const GemmArgs& a);     // true: launch_gemm fills a.colsum_a itself (no separate column-sum launch)

// ---- row-wise helpers (rowops.hip) ---------------------------------------------------------------
hipError_t launch_softmax_rows(float* s, long rows, int cols, hipStream_t st);
hipError_t launch_softmax_bwd_rows(const float* p, float* dp, long rows, int cols, float scale, hipStream_t st);
hipError_t launch_colsum(const float* x, float* out, int rows, int cols, long ld, hipStream_t st);
hipError_t launch_pos_logit(const float* q, const float* k, float* out, long ld_out, int B, int d, float inv_T, hipStream_t st);
hipError_t launch_infonce_rows(float* logits, int B, int ncols, float* loss_rows, float* lse, int32_t* top1,
                               int write_probs, hipStream_t st);
// dst[i] = ((dst[i] + parts[i]) + parts[stride + i]) + ... over nparts partial arrays: the fixed-order end of a split-K product
hipError_t launch_add_partials(float* dst, const float* parts, int nparts, long n, long stride, hipStream_t st);
hipError_t launch_pos_grad_init(const float* dlogits, long ld, const float* k, float* dq, int B, int d, float inv_T,
                                hipStream_t st);

// ---- queue.hip -----------------------------------------------------------------------------------
hipError_t launch_enqueue(void* queue, const float* rows, int n, int64_t index, int K, int d, int qdtype, hipStream_t st);
hipError_t launch_enqueue_mirror(float* queue, void* mirror, const float* rows, int n, int64_t index, int K, int d, hipStream_t st);
hipError_t launch_prefetch(const void* p, size_t bytes, hipStream_t st);
hipError_t launch_widen_bf16(const void* src, float* dst, size_t n, hipStream_t st);      // bf16 -> fp32, n elements (src 16-B aligned)
hipError_t launch_ema(const int64_t* table, int n_tensors, int64_t total_blocks, float m, float om, hipStream_t st);

// ---- infonce_fused.hip (one-pass flash-style kernel) ----------------------------------------------
bool infonce_flash_supported(int B, int d, int K, int qdtype, int prec);
size_t infonce_flash_workspace_bytes(int B, int d, int K);
int set_k2_target_wg(int n);                            // debug knob of the K2 plan (moma_debug_set_k2_target_wg)
// the enqueue that follows a K2 call (MoMA/mem_moco.py:97-99), carried by the call's LAST launch where there is one to carry it
struct EnqueueJob {
    void* queue16;        // bf16 queue rows are rounded into (the queue K2 has just read), or nullptr
    float* queue32;       // fp32 queue (alone, or next to its bf16 mirror `queue16`), or nullptr
    const float* rows;    // [n, d] fp32
    int n;                // 0: nothing to enqueue
    int64_t index;        // ring pointer: rows[i] -> slot (index + i) mod K
    int K, d;
};
hipError_t launch_infonce_flash(const float* q, const float* k, const void* queue, int B, int d, int K, float inv_T,
                                float* loss_rows, float* lse, int32_t* top1, float* dq, void* ws, int qdtype,
                                hipStream_t st, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr,
                                const void* q_packed = nullptr, hipEvent_t ev_call_end = nullptr,
                                const EnqueueJob* enq = nullptr);
size_t infonce_qpack_bytes(int B, int d);
bool infonce_multi_supported(int n_terms, int B, int d, int K, int qdtype, int prec);
size_t infonce_multi_workspace_bytes(int n_terms, int B, int d, int K);
hipError_t launch_infonce_multi(const moma_infonce_term_t* terms, int n_terms, int B, int d, int K, float inv_T, void* ws,
                                hipStream_t st);

// ---- infonce_f32.hip (one pass over an fp32 queue in exact fp32 arithmetic) ------------------------
bool infonce_f32_flash_supported(int B, int d, int K, int qdtype, int prec);
size_t infonce_f32_flash_workspace_bytes(int B, int d, int K);
hipError_t launch_infonce_f32_flash(const float* q, const float* k, const float* queue, int B, int d, int K, float inv_T,
                                    float* loss_rows, float* lse, int32_t* top1, float* dq, void* ws, hipStream_t st,
                                    hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);

// ---- k1_fast.hip (batch-token attention, bf16 fast path) -------------------------------------------
bool mha_fast_supported(int N, int d, int H, int prec);
hipError_t launch_mha_pack(const float* w_qkv, const float* w_proj, void* pack, int d, int with_t, hipStream_t st);
hipError_t launch_mha_fwd_fast(const moma_mha_module_t* mods, int n_modules, int N, int d, int H, hipStream_t st);
size_t mha_bwd_fast_workspace_bytes(int N, int d);
hipError_t launch_mha_bwd_fast(const void* pack, const void* x, int x_dtype, const void* qkv16, const void* attn16, const float* lse,
                               const float* dy, float* dx, float* dw_qkv, float* db_qkv, float* dw_proj, float* db_proj,
                               void* workspace, int N, int d, int H, hipStream_t st);

// ---- bn.hip (BatchNorm2d + activation, NCHW) ------------------------------------------------------
size_t bn_workspace_floats(int C);
hipError_t launch_bn_fwd(const void* x, void* out, const float* gamma, const float* beta, float* rm, float* rv,
                         float* save_mean, float* save_invstd, float* ws, int N, int C, int HW, int dtype, int act,
                         int training, float momentum, float eps, void* plane_mean, hipStream_t st);
hipError_t launch_bn_bwd(const void* x, const void* dout, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, void* dx, float* dgamma, float* dbeta, float* ws, int N, int C, int HW,
                         int dtype, int act, int training, const void* dplane_mean, hipStream_t st);

// ---- dwconv.hip (depthwise convolution, NCHW) -----------------------------------------------------
bool dwconv_supported(int K, int S);
size_t dwconv_workspace_floats(int C, int K);
hipError_t launch_dw_fwd(const void* x, const float* w, void* y, int N, int C, int H, int W, int OH, int OW, int K, int S,
                         int pt, int pl, int dtype, hipStream_t st);
hipError_t launch_dw_bwd_data(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int OH, int OW, int K,
                              int S, int pt, int pl, int dtype, hipStream_t st);
hipError_t launch_dw_bwd_weight(const void* x, const void* dy, float* dw, float* ws, size_t ws_floats, int N, int C, int H,
                                int W, int OH, int OW, int K, int S, int pt, int pl, int dtype, hipStream_t st);

// ---- se.hip (squeeze-excite: per-plane mean, sigmoid gate) ----------------------------------------------
hipError_t launch_plane_mean(const void* x, void* out, int NC, int HW, int dtype, hipStream_t st);
hipError_t launch_se_gate_fwd(const void* x, const void* s, void* out, int NC, int HW, int dtype, hipStream_t st);
hipError_t launch_se_gate_bwd(const void* x, const void* s, const void* dout, void* dx, void* ds, int NC, int HW, int dtype,
                              hipStream_t st);

}  // namespace moma
