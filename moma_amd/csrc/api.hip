// extern "C" entry points of libmoma_hip.so (see include/moma_hip.h for the contract of each).
#include "common.hpp"

using namespace moma;

namespace {
inline int hip_rc(hipError_t e) { return e == hipSuccess ? MOMA_OK : (int)e; }
inline bool bad_prec(int p) { return p != MOMA_PREC_F32 && p != MOMA_PREC_BF16; }
inline bool bad_dt(int t) { return t != MOMA_DT_F32 && t != MOMA_DT_BF16; }
inline bool misaligned(const void* p, size_t a) { return ((uintptr_t)p % a) != 0; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define MOMA_TRY(expr)                         \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) return (int)_e;  \
    } while (0)

GemmArgs gemm(const float* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int tA,
              int tB, float alpha, int prec) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.strideA = g.strideB = g.strideC = 0; g.batch = 1;
    g.transA = tA; g.transB = tB; g.alpha = alpha; g.splitk = 1; g.atomic = 0; g.splitC = 0;
    g.b_dtype = MOMA_DT_F32; g.prec = prec;
    return g;
}
}  // namespace

extern "C" {

int moma_version(void) { return MOMA_ABI_VERSION; }

int moma_debug_set_k2_target_wg(int n) { return set_k2_target_wg(n); }

const char* moma_error_string(int code) {
    switch (code) {
        case MOMA_OK: return "ok";
        case MOMA_E_NULL: return "moma: required pointer is NULL";
        case MOMA_E_SHAPE: return "moma: bad or inconsistent dimension";
        case MOMA_E_DTYPE: return "moma: unknown precision / queue dtype";
        case MOMA_E_ALIGN: return "moma: pointer not aligned to its element size";
        case MOMA_E_WORKSPACE: return "moma: workspace too small";
        case MOMA_E_UNSUPPORTED: return "moma: unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "moma: unknown error";
    }
}

int moma_ema_multi(const int64_t* table, int n_tensors, int64_t total_blocks, float m, float one_minus_m,
                   moma_stream_t stream) {
    if (n_tensors == 0) return MOMA_OK;
    if (!table) return MOMA_E_NULL;
    if (n_tensors < 0 || total_blocks < 0 || total_blocks > 0x7fffffffLL) return MOMA_E_SHAPE;
    return hip_rc(launch_ema(table, n_tensors, total_blocks, m, one_minus_m, (hipStream_t)stream));
}

int moma_enqueue(void* queue, const float* rows, int n, int64_t index, int K, int d, int qdtype,
                 moma_stream_t stream) {
    if (n == 0) return MOMA_OK;
    if (!queue || !rows) return MOMA_E_NULL;
    if (n < 0 || K <= 0 || d <= 0 || index < 0 || index >= K) return MOMA_E_SHAPE;
    if (bad_dt(qdtype)) return MOMA_E_DTYPE;
    if (misaligned(rows, 4) || misaligned(queue, qdtype == MOMA_DT_BF16 ? 2 : 4)) return MOMA_E_ALIGN;
    return hip_rc(launch_enqueue(queue, rows, n, index, K, d, qdtype, (hipStream_t)stream));
}

int moma_enqueue_mirror(float* queue, void* mirror_bf16, const float* rows, int n, int64_t index, int K, int d,
                        moma_stream_t stream) {
    if (n == 0) return MOMA_OK;
    if (!queue || !mirror_bf16 || !rows) return MOMA_E_NULL;
    if (n < 0 || K <= 0 || d <= 0 || index < 0 || index >= K) return MOMA_E_SHAPE;
    if (misaligned(rows, 4) || misaligned(queue, 4) || misaligned(mirror_bf16, 2)) return MOMA_E_ALIGN;
    return hip_rc(launch_enqueue_mirror(queue, mirror_bf16, rows, n, index, K, d, (hipStream_t)stream));
}

int moma_queue_prefetch(const void* queue, size_t bytes, moma_stream_t stream) {
    if (bytes == 0) return MOMA_OK;
    if (!queue) return MOMA_E_NULL;
    if (misaligned(queue, 16)) return MOMA_E_ALIGN;
    return hip_rc(launch_prefetch(queue, bytes, (hipStream_t)stream));
}

int moma_infonce_logits(const float* q, const float* k, const void* queue, float* out, int B, int d, int K,
                        float inv_T, int qdtype, int prec, moma_stream_t stream) {
    if (!q || !k || !queue || !out) return MOMA_E_NULL;
    if (B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_dt(qdtype) || bad_prec(prec)) return MOMA_E_DTYPE;
    hipStream_t st = (hipStream_t)stream;
    const long ld = (long)K + 1;
    MOMA_TRY(launch_pos_logit(q, k, out, ld, B, d, inv_T, st));
    GemmArgs g = gemm(q, queue, out + 1, B, K, d, d, d, ld, 0, 0, inv_T, prec);   // neg = q . queue^T
    g.b_dtype = qdtype;
    return hip_rc(launch_gemm(g, st));
}

// dq = dlogits[:,0] k inv_T + dlogits[:,1:] . queue inv_T: the contraction over K is split over workgroups (B x d is a handful of
// output tiles).  With `parts` (splitk x B x d floats) every split writes its own partial product and a last launch adds them in
// split order -- bitwise reproducible; without it the splits meet in fp32 atomics (equal up to the order of the additions).
static int logits_bwd_splitk(int B, int d, int K) {
    const int tiles = ((B + 63) / 64) * ((d + 63) / 64);
    int splitk = (1024 + tiles - 1) / tiles;
    const int ktiles = (K + 31) / 32;
    if (splitk > ktiles) splitk = ktiles;
    return splitk < 1 ? 1 : splitk;
}
static size_t logits_bwd_parts_bytes(int B, int d, int K) {
    const int sk = logits_bwd_splitk(B, d, K);
    return sk > 1 ? align_up((size_t)sk * B * d * sizeof(float), 256) : 0;
}
static int logits_bwd_impl(const float* dlogits, const float* k, const void* queue, float* dq, int B, int d, int K, float inv_T,
                           int qdtype, int prec, hipStream_t st, float* parts) {
    const long ld = (long)K + 1;
    MOMA_TRY(launch_pos_grad_init(dlogits, ld, k, dq, B, d, inv_T, st));
    GemmArgs g = gemm(dlogits + 1, queue, dq, B, d, K, ld, d, d, 0, 1, inv_T, prec);
    g.b_dtype = qdtype;
    g.splitk = logits_bwd_splitk(B, d, K);
    if (parts != nullptr && g.splitk > 1) {
        g.C = parts;
        g.splitC = (long)B * d;
        MOMA_TRY(launch_gemm(g, st));
        return hip_rc(launch_add_partials(dq, parts, g.splitk, (long)B * d, (long)B * d, st));
    }
    g.atomic = 1;                     // (also splitk == 1: one split adds its product to the initialised dq)
    return hip_rc(launch_gemm(g, st));
}

int moma_infonce_logits_bwd(const float* dlogits, const float* k, const void* queue, float* dq, int B, int d,
                            int K, float inv_T, int qdtype, int prec, moma_stream_t stream) {
    if (!dlogits || !k || !queue || !dq) return MOMA_E_NULL;
    if (B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_dt(qdtype) || bad_prec(prec)) return MOMA_E_DTYPE;
    return logits_bwd_impl(dlogits, k, queue, dq, B, d, K, inv_T, qdtype, prec, (hipStream_t)stream, nullptr);
}

size_t moma_infonce_logits_bwd_workspace_bytes(int B, int d, int K) {
    if (B <= 0 || d <= 0 || K <= 0) return 0;
    const size_t b = logits_bwd_parts_bytes(B, d, K);
    return b ? b : 256;               // (never 0 for a valid shape: 0 is the error value of the *_bytes functions)
}

int moma_infonce_logits_bwd_ws(const float* dlogits, const float* k, const void* queue, float* dq, int B, int d, int K,
                               float inv_T, int qdtype, int prec, void* workspace, size_t workspace_bytes,
                               moma_stream_t stream) {
    if (!dlogits || !k || !queue || !dq || !workspace) return MOMA_E_NULL;
    if (B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_dt(qdtype) || bad_prec(prec)) return MOMA_E_DTYPE;
    if (workspace_bytes < moma_infonce_logits_bwd_workspace_bytes(B, d, K)) return MOMA_E_WORKSPACE;
    if (misaligned(workspace, 16)) return MOMA_E_ALIGN;
    return logits_bwd_impl(dlogits, k, queue, dq, B, d, K, inv_T, qdtype, prec, (hipStream_t)stream, (float*)workspace);
}

int moma_infonce_logits_bwd_kq(const float* dlogits, const float* q, float* dk, float* dqueue, int B, int d, int K,
                               float inv_T, int prec, moma_stream_t stream) {
    if (!dlogits || !q) return MOMA_E_NULL;
    if (B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_prec(prec)) return MOMA_E_DTYPE;
    hipStream_t st = (hipStream_t)stream;
    const long ld = (long)K + 1;
    if (dk) MOMA_TRY(launch_pos_grad_init(dlogits, ld, q, dk, B, d, inv_T, st));
    if (dqueue) {
        // dqueue[j,c] = sum_b dlogits[b,1+j] * q[b,c]
        GemmArgs g = gemm(dlogits + 1, q, dqueue, K, d, B, ld, d, d, 1, 1, inv_T, prec);
        MOMA_TRY(launch_gemm(g, st));
    }
    return MOMA_OK;
}

// Exact-fp32 policy over a bf16-STORED queue at a width the one-pass fp32 kernel takes: the queue is widened (bf16 -> fp32 is
// exact) into the workspace and that kernel streams the copy -- 64 MB read + 128 MB written at K x d = 65536 x 512 (~35 us) in front
// of a 378 us pass, against 807 us for the staged path with its [B,K+1] logits (round 5).  The copy lives for the call only.
static bool f32_policy_widens_queue(int B, int d, int K, int qdtype, int prec) {
    return prec == MOMA_PREC_F32 && qdtype == MOMA_DT_BF16 && infonce_f32_flash_supported(B, d, K, MOMA_DT_F32, prec);
}

size_t moma_infonce_fused_workspace_bytes(int B, int d, int K, int qdtype, int prec) {
    if (B <= 0 || d <= 0 || K <= 0) return 0;
    if (infonce_flash_supported(B, d, K, qdtype, prec)) return infonce_flash_workspace_bytes(B, d, K);
    if (infonce_f32_flash_supported(B, d, K, qdtype, prec)) return infonce_f32_flash_workspace_bytes(B, d, K);
    if (f32_policy_widens_queue(B, d, K, qdtype, prec))
        return align_up((size_t)K * d * sizeof(float), 256) + infonce_f32_flash_workspace_bytes(B, d, K);
    return align_up((size_t)B * ((size_t)K + 1) * sizeof(float), 256) + logits_bwd_parts_bytes(B, d, K);     // logits | split-K partials of dq
}

int moma_infonce_fused(const float* q, const float* k, const void* queue, int B, int d, int K, float inv_T,
                       float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                       size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream) {
    return moma_infonce_fused_ex(q, k, queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, workspace_bytes,
                                 qdtype, prec, stream, nullptr, nullptr);
}

int moma_infonce_fused_ex(const float* q, const float* k, const void* queue, int B, int d, int K, float inv_T,
                          float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                          size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream, void* ev_begin,
                          void* ev_end) {
    return moma_infonce_fused_q(q, nullptr, k, queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, workspace_bytes,
                                qdtype, prec, stream, ev_begin, ev_end, nullptr);
}

size_t moma_infonce_qpack_bytes(int B, int d) { return infonce_qpack_bytes(B, d); }

// K2 (+ K3 when `job` names rows): the enqueue rides on the call's last launch where the path has a combine launch to carry it
// (*carried = 1), else the caller launches it behind the call
static int fused_impl(const float* q, const void* q_packed, const float* k, const void* queue, int B, int d, int K,
                      float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                      size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream, void* ev_begin, void* ev_end,
                      void* ev_call_end, const EnqueueJob* job, int* carried) {
    if (carried) *carried = 0;
    if (!q || !k || !queue || !loss_rows || !lse || !top1 || !workspace) return MOMA_E_NULL;
    if (B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_dt(qdtype) || bad_prec(prec)) return MOMA_E_DTYPE;
    if (workspace_bytes < moma_infonce_fused_workspace_bytes(B, d, K, qdtype, prec)) return MOMA_E_WORKSPACE;
    if (misaligned(workspace, 16)) return MOMA_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (q_packed && (misaligned(q_packed, 16) || infonce_qpack_bytes(B, d) == 0 || !infonce_flash_supported(B, d, K, qdtype, prec)))
        return q_packed && misaligned(q_packed, 16) ? MOMA_E_ALIGN : MOMA_E_UNSUPPORTED;
    // the one-pass kernels move q / k / dq in 16-byte pieces and the queue by 16-byte LDS-DMA: operands that start inside a vector
    // (an element-granular view of a larger buffer) are refused here instead of faulting there
    const bool one_pass = infonce_flash_supported(B, d, K, qdtype, prec) || infonce_f32_flash_supported(B, d, K, qdtype, prec) ||
                          f32_policy_widens_queue(B, d, K, qdtype, prec);
    if (one_pass && (misaligned(q, 16) || misaligned(k, 16) || misaligned(queue, 16) || (dq && misaligned(dq, 16)))) return MOMA_E_ALIGN;
    if (!one_pass && (misaligned(q, 4) || misaligned(k, 4) || misaligned(queue, qdtype == MOMA_DT_BF16 ? 2 : 4) || (dq && misaligned(dq, 4))))
        return MOMA_E_ALIGN;
    if (infonce_flash_supported(B, d, K, qdtype, prec)) {
        if (carried) *carried = 1;
        return hip_rc(launch_infonce_flash(q, k, queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, qdtype, st,
                                           (hipEvent_t)ev_begin, (hipEvent_t)ev_end, q_packed, (hipEvent_t)ev_call_end, job));
    }
    if (infonce_f32_flash_supported(B, d, K, qdtype, prec))      // exact fp32, fp32 queue: one pass, no [B,K+1] logits
    {
        const int rc = hip_rc(launch_infonce_f32_flash(q, k, (const float*)queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, st,
                                                       (hipEvent_t)ev_begin, (hipEvent_t)ev_end));
        if (ev_call_end) (void)hipEventRecord((hipEvent_t)ev_call_end, st);
        return rc;
    }
    if (f32_policy_widens_queue(B, d, K, qdtype, prec)) {
        if (misaligned(queue, 16)) return MOMA_E_ALIGN;
        float* wide = (float*)workspace;
        void* ws2 = (char*)workspace + align_up((size_t)K * d * sizeof(float), 256);
        // (the measurement span opens IN FRONT of the widening pass: 64 MB read + 128 MB written per call at the bench shape belong
        //  to what this policy costs -- ADVICE r5)
        if (ev_begin) (void)hipEventRecord((hipEvent_t)ev_begin, st);
        MOMA_TRY(launch_widen_bf16(queue, wide, (size_t)K * d, st));
        const int rc = hip_rc(launch_infonce_f32_flash(q, k, wide, B, d, K, inv_T, loss_rows, lse, top1, dq, ws2, st,
                                                       (hipEvent_t) nullptr, (hipEvent_t)ev_end));
        if (ev_call_end) (void)hipEventRecord((hipEvent_t)ev_call_end, st);
        return rc;
    }
    // staged path (any shape, exact fp32 available): logits -> row reduction -> gradient product
    float* logits = (float*)workspace;
    if (ev_begin) (void)hipEventRecord((hipEvent_t)ev_begin, st);
    int rc = moma_infonce_logits(q, k, queue, logits, B, d, K, inv_T, qdtype, prec, stream);
    if (ev_end) (void)hipEventRecord((hipEvent_t)ev_end, st);
    if (rc != MOMA_OK) return rc;
    MOMA_TRY(launch_infonce_rows(logits, B, K + 1, loss_rows, lse, top1, dq != nullptr, st));
    if (dq) {
        float* parts = (float*)((char*)workspace + align_up((size_t)B * ((size_t)K + 1) * sizeof(float), 256));
        rc = logits_bwd_impl(logits, k, queue, dq, B, d, K, inv_T, qdtype, prec, st, logits_bwd_parts_bytes(B, d, K) ? parts : nullptr);
    }
    if (ev_call_end) (void)hipEventRecord((hipEvent_t)ev_call_end, st);
    return rc;
}

int moma_infonce_fused_q(const float* q, const void* q_packed, const float* k, const void* queue, int B, int d, int K,
                         float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                         size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream, void* ev_begin, void* ev_end,
                         void* ev_call_end) {
    return fused_impl(q, q_packed, k, queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, workspace_bytes, qdtype, prec, stream,
                      ev_begin, ev_end, ev_call_end, nullptr, nullptr);
}

int moma_infonce_fused_enqueue(const float* q, const void* q_packed, const float* k, void* queue, int B, int d, int K,
                               float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                               size_t workspace_bytes, int qdtype, int prec, const float* rows, int n, int64_t index,
                               float* queue_f32, moma_stream_t stream, void* ev_begin, void* ev_end, void* ev_call_end) {
    if (n < 0 || K <= 0 || index < 0 || index >= K) return MOMA_E_SHAPE;
    if (n > 0 && !rows) return MOMA_E_NULL;
    if (bad_dt(qdtype)) return MOMA_E_DTYPE;
    if (queue_f32 && qdtype != MOMA_DT_BF16) return MOMA_E_DTYPE;          // (a mirror is the bf16 image of an fp32 queue)
    if (n > 0 && (misaligned(rows, 4) || misaligned(queue, qdtype == MOMA_DT_BF16 ? 2 : 4) || (queue_f32 && misaligned(queue_f32, 4))))
        return MOMA_E_ALIGN;
    EnqueueJob job{qdtype == MOMA_DT_BF16 ? queue : nullptr, qdtype == MOMA_DT_BF16 ? queue_f32 : (float*)queue, rows, n, index, K, d};
    int carried = 0;
    const int rc = fused_impl(q, q_packed, k, queue, B, d, K, inv_T, loss_rows, lse, top1, dq, workspace, workspace_bytes, qdtype, prec,
                              stream, ev_begin, ev_end, ev_call_end, n > 0 ? &job : nullptr, &carried);
    if (rc != MOMA_OK || carried || n == 0) return rc;
    // a path without a combine launch (exact fp32, staged): the enqueue as its own launch, behind the call on the stream
    if (queue_f32) return hip_rc(launch_enqueue_mirror(queue_f32, queue, rows, n, index, K, d, (hipStream_t)stream));
    return hip_rc(launch_enqueue(queue, rows, n, index, K, d, qdtype, (hipStream_t)stream));
}

size_t moma_infonce_fused_multi_workspace_bytes(int n_terms, int B, int d, int K, int qdtype, int prec) {
    if (n_terms < 1 || n_terms > 4 || B <= 0 || d <= 0 || K <= 0 || bad_dt(qdtype) || bad_prec(prec)) return 0;
    if (infonce_multi_supported(n_terms, B, d, K, qdtype, prec)) return infonce_multi_workspace_bytes(n_terms, B, d, K);
    // no one-sweep kernel for this configuration (wide rows, exact fp32, fp32 queue under the bf16 policy): the terms go through
    // moma_infonce_fused's own path one after the other on the stream and share ONE workspace of that call's size
    return moma_infonce_fused_workspace_bytes(B, d, K, qdtype, prec);
}

int moma_infonce_fused_multi(const moma_infonce_term_t* terms, int n_terms, int B, int d, int K, float inv_T, void* workspace,
                             size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream) {
    if (!terms || !workspace) return MOMA_E_NULL;
    if (n_terms < 1 || n_terms > 4 || B <= 0 || d <= 0 || K <= 0) return MOMA_E_SHAPE;
    if (bad_dt(qdtype) || bad_prec(prec)) return MOMA_E_DTYPE;
    for (int i = 0; i < n_terms; ++i) {
        const moma_infonce_term_t& t = terms[i];
        if (!t.q || !t.k || !t.queue || !t.loss_rows || !t.lse || !t.top1) return MOMA_E_NULL;
        if ((t.dq != nullptr) != (terms[0].dq != nullptr)) return MOMA_E_UNSUPPORTED;     // every term with dq, or none
        if (misaligned(t.q, 16) || misaligned(t.k, 4) || misaligned(t.queue, 16)) return MOMA_E_ALIGN;
    }
    if (workspace_bytes < moma_infonce_fused_multi_workspace_bytes(n_terms, B, d, K, qdtype, prec)) return MOMA_E_WORKSPACE;
    if (misaligned(workspace, 16)) return MOMA_E_ALIGN;
    if (infonce_multi_supported(n_terms, B, d, K, qdtype, prec))
        return hip_rc(launch_infonce_multi(terms, n_terms, B, d, K, inv_T, workspace, (hipStream_t)stream));
    // term by term (stream order makes the shared workspace safe).  At these shapes every term's passes already fill the chip and
    // are bound by the matrix pipe / L2 delivery, not by HBM (d = 1280 bf16: 1.2 TB/s; exact fp32: 0.35 TB/s): merging the terms'
    // launches would not read the queues any less
    for (int i = 0; i < n_terms; ++i) {
        const moma_infonce_term_t& t = terms[i];
        const int rc = moma_infonce_fused_q(t.q, nullptr, t.k, t.queue, B, d, K, inv_T, t.loss_rows, t.lse, t.top1, t.dq, workspace,
                                            workspace_bytes, qdtype, prec, stream, nullptr, nullptr, nullptr);
        if (rc != MOMA_OK) return rc;
    }
    return MOMA_OK;
}

int moma_mha_saved_state(int N, int d, int H, int prec) {
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0 || bad_prec(prec)) return MOMA_E_SHAPE;
    return mha_fast_supported(N, d, H, prec) ? MOMA_MHA_SAVE_LSE : MOMA_MHA_SAVE_PROBS;
}

int moma_mha_fwd(const float* x, const float* w_qkv, const float* b_qkv, const float* w_proj, const float* b_proj,
                 float* y, float* qkv, float* probs, float* attn_out, int N, int d, int H, int prec,
                 moma_stream_t stream) {
    if (!x || !w_qkv || !w_proj || !b_proj || !y || !qkv || !attn_out || !probs) return MOMA_E_NULL;
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0) return MOMA_E_SHAPE;
    if (bad_prec(prec)) return MOMA_E_DTYPE;
    hipStream_t st = (hipStream_t)stream;
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    // qkv = x Wqkv^T + b                                   (MoMA/criterion_moco_att.py:156)
    GemmArgs g = gemm(x, w_qkv, qkv, N, 3 * d, d, d, d, 3L * d, 0, 0, 1.f, prec);
    g.bias = b_qkv;
    MOMA_TRY(launch_gemm(g, st));
    // per head: S = (q k^T) * scale                          (:159)
    g = gemm(qkv, qkv + d, probs, N, N, hd, 3L * d, 3L * d, N, 0, 0, scale, prec);
    g.batch = H; g.strideA = hd; g.strideB = hd; g.strideC = (long)N * N;
    MOMA_TRY(launch_gemm(g, st));
    MOMA_TRY(launch_softmax_rows(probs, (long)H * N, N, st));                      // :160
    // per head: a[:, h*hd:(h+1)*hd] = P v                   (:163)
    g = gemm(probs, qkv + 2 * d, attn_out, N, hd, N, N, 3L * d, d, 0, 1, 1.f, prec);
    g.batch = H; g.strideA = (long)N * N; g.strideB = hd; g.strideC = hd;
    MOMA_TRY(launch_gemm(g, st));
    // y = a Wproj^T + b                                      (:164)
    g = gemm(attn_out, w_proj, y, N, d, d, d, d, d, 0, 0, 1.f, prec);
    g.bias = b_proj;
    return hip_rc(launch_gemm(g, st));
}

size_t moma_mha_bwd_workspace_bytes(int N, int d, int H, int prec) {
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0 || bad_prec(prec)) return 0;
    // dA [N,d] + dP [H,N,N] + dqkv [N,3d]
    return align_up(((size_t)N * d + (size_t)H * N * N + (size_t)N * 3 * d) * sizeof(float), 256);
}

int moma_mha_bwd(const float* x, const float* w_qkv, const float* w_proj, const float* qkv, const float* probs,
                 const float* attn_out, const float* dy, float* dx, float* dw_qkv, float* db_qkv,
                 float* dw_proj, float* db_proj, void* workspace, size_t workspace_bytes, int N, int d, int H, int prec,
                 moma_stream_t stream) {
    if (!x || !w_qkv || !w_proj || !qkv || !attn_out || !dy || !workspace || !probs) return MOMA_E_NULL;
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0) return MOMA_E_SHAPE;
    if (bad_prec(prec)) return MOMA_E_DTYPE;
    if (workspace_bytes < moma_mha_bwd_workspace_bytes(N, d, H, prec)) return MOMA_E_WORKSPACE;
    if (misaligned(workspace, 16)) return MOMA_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int hd = d / H;
    const float scale = 1.0f / sqrtf((float)hd);
    float* dA = (float*)workspace;
    float* dP = dA + (size_t)N * d;
    float* dqkv = dP + (size_t)H * N * N;
    GemmArgs g;
    // proj: dWproj = dy^T a ; dbproj = colsum(dy) ; dA = dy Wproj
    bool db_done = false;
    if (dw_proj) {
        g = gemm(dy, attn_out, dw_proj, d, d, N, d, d, d, 1, 1, 1.f, prec);
        if (db_proj && gemm_fuses_colsum(g)) { g.colsum_a = db_proj; db_done = true; }       // the bias gradient rides along
        MOMA_TRY(launch_gemm(g, st));
    }
    if (db_proj && !db_done) MOMA_TRY(launch_colsum(dy, db_proj, N, d, d, st));
    g = gemm(dy, w_proj, dA, N, d, d, d, d, d, 0, 1, 1.f, prec);
    MOMA_TRY(launch_gemm(g, st));
    // per head: dV = P^T dA_h  -> dqkv[:, 2d + h*hd ...]
    g = gemm(probs, dA, dqkv + 2 * d, N, hd, N, N, d, 3L * d, 1, 1, 1.f, prec);
    g.batch = H; g.strideA = (long)N * N; g.strideB = hd; g.strideC = hd;
    MOMA_TRY(launch_gemm(g, st));
    // per head: dP = dA_h V^T
    g = gemm(dA, qkv + 2 * d, dP, N, N, hd, d, 3L * d, N, 0, 0, 1.f, prec);
    g.batch = H; g.strideA = hd; g.strideB = hd; g.strideC = (long)N * N;
    MOMA_TRY(launch_gemm(g, st));
    // dS = P * (dP - rowsum(dP*P)) * scale   (in place on dP)
    MOMA_TRY(launch_softmax_bwd_rows(probs, dP, (long)H * N, N, scale, st));
    // per head: dQ = dS K ; dK = dS^T Q
    g = gemm(dP, qkv + d, dqkv, N, hd, N, N, 3L * d, 3L * d, 0, 1, 1.f, prec);
    g.batch = H; g.strideA = (long)N * N; g.strideB = hd; g.strideC = hd;
    MOMA_TRY(launch_gemm(g, st));
    g = gemm(dP, qkv, dqkv + d, N, hd, N, N, 3L * d, 3L * d, 1, 1, 1.f, prec);
    g.batch = H; g.strideA = (long)N * N; g.strideB = hd; g.strideC = hd;
    MOMA_TRY(launch_gemm(g, st));
    // qkv linear: dWqkv = dqkv^T x ; dbqkv = colsum(dqkv) ; dx = dqkv Wqkv
    db_done = false;
    if (dw_qkv) {
        g = gemm(dqkv, x, dw_qkv, 3 * d, d, N, 3L * d, d, d, 1, 1, 1.f, prec);
        if (db_qkv && gemm_fuses_colsum(g)) { g.colsum_a = db_qkv; db_done = true; }
        MOMA_TRY(launch_gemm(g, st));
    }
    if (db_qkv && !db_done) MOMA_TRY(launch_colsum(dqkv, db_qkv, N, 3 * d, 3L * d, st));
    if (dx) {
        g = gemm(dqkv, w_qkv, dx, N, d, 3 * d, 3L * d, d, d, 0, 1, 1.f, prec);
        MOMA_TRY(launch_gemm(g, st));
    }
    return MOMA_OK;
}

// ---- K1 fast path (k1_fast.hip) ----------------------------------------------------------------------------------
size_t moma_mha_pack_bytes(int d) { return d > 0 ? (size_t)16 * d * d : 0; }

int moma_mha_pack_weights(const float* w_qkv, const float* w_proj, void* pack, int d, int with_transposed,
                          moma_stream_t stream) {
    if (!w_qkv || !w_proj || !pack) return MOMA_E_NULL;
    if (d <= 0 || d % 16 != 0) return MOMA_E_SHAPE;
    if (misaligned(pack, 16) || misaligned(w_qkv, 4) || misaligned(w_proj, 4)) return MOMA_E_ALIGN;
    return hip_rc(launch_mha_pack(w_qkv, w_proj, pack, d, with_transposed != 0, (hipStream_t)stream));
}

int moma_mha_fwd_fast(const moma_mha_module_t* mods, int n_modules, int N, int d, int H, moma_stream_t stream) {
    if (!mods) return MOMA_E_NULL;
    if (n_modules < 1 || n_modules > 4 || N <= 0 || d <= 0 || H <= 0 || d % H != 0) return MOMA_E_SHAPE;
    if (!mha_fast_supported(N, d, H, MOMA_PREC_BF16)) return MOMA_E_UNSUPPORTED;
    for (int i = 0; i < n_modules; ++i) {
        const moma_mha_module_t& m = mods[i];
        if (!m.x || !m.pack || !m.b_proj || !m.y || !m.qkv16 || !m.attn16) return MOMA_E_NULL;
        if (bad_dt(m.x_dtype)) return MOMA_E_DTYPE;
        if (misaligned(m.x, 16) || misaligned(m.pack, 16) || misaligned(m.y, 16) || misaligned(m.qkv16, 16) ||
            misaligned(m.attn16, 16) || misaligned(m.b_proj, 16) || (m.b_qkv && misaligned(m.b_qkv, 16)) ||
            (m.qpack && misaligned(m.qpack, 16)))
            return MOMA_E_ALIGN;
    }
    return hip_rc(launch_mha_fwd_fast(mods, n_modules, N, d, H, (hipStream_t)stream));
}

size_t moma_mha_bwd_fast_workspace_bytes(int N, int d, int H) {
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0) return 0;
    return mha_bwd_fast_workspace_bytes(N, d);
}

int moma_mha_bwd_fast(const void* pack, const void* x, int x_dtype, const void* qkv16, const void* attn16, const float* lse,
                      const float* dy, float* dx, float* dw_qkv, float* db_qkv, float* dw_proj, float* db_proj,
                      void* workspace, size_t workspace_bytes, int N, int d, int H, moma_stream_t stream) {
    if (!pack || !x || !qkv16 || !attn16 || !lse || !dy || !workspace) return MOMA_E_NULL;
    if (N <= 0 || d <= 0 || H <= 0 || d % H != 0) return MOMA_E_SHAPE;
    if (!mha_fast_supported(N, d, H, MOMA_PREC_BF16)) return MOMA_E_UNSUPPORTED;
    if (bad_dt(x_dtype)) return MOMA_E_DTYPE;
    if ((db_qkv && !dw_qkv) || (db_proj && !dw_proj)) return MOMA_E_UNSUPPORTED;      // a bias gradient rides on its weight gradient
    if (workspace_bytes < mha_bwd_fast_workspace_bytes(N, d)) return MOMA_E_WORKSPACE;
    if (misaligned(workspace, 16) || misaligned(pack, 16) || misaligned(x, 16) || misaligned(qkv16, 16) ||
        misaligned(attn16, 16) || misaligned(dy, 16) || (dx && misaligned(dx, 16)) || (dw_qkv && misaligned(dw_qkv, 8)) ||
        (dw_proj && misaligned(dw_proj, 8)))
        return MOMA_E_ALIGN;
    return hip_rc(launch_mha_bwd_fast(pack, x, x_dtype, qkv16, attn16, lse, dy, dx, dw_qkv, db_qkv, dw_proj, db_proj, workspace, N, d,
                                      H, (hipStream_t)stream));
}

size_t moma_bn_workspace_bytes(int C) { return C > 0 ? align_up(bn_workspace_floats(C) * sizeof(float), 256) : 0; }

static int bn_check(int N, int C, int HW, int dtype, int act, const void* ws, size_t ws_bytes) {
    if (N <= 0 || C <= 0 || HW <= 0 || (long)N * C > 0x7fffffffL || (long)N * HW > 0x7fffffffL) return MOMA_E_SHAPE;
    if (dtype != MOMA_DT_F32 && dtype != MOMA_DT_BF16) return MOMA_E_DTYPE;
    if (act != MOMA_ACT_NONE && act != MOMA_ACT_SILU && act != MOMA_ACT_RELU) return MOMA_E_DTYPE;
    if (!ws) return MOMA_E_NULL;
    if (ws_bytes < moma_bn_workspace_bytes(C)) return MOMA_E_WORKSPACE;
    return MOMA_OK;
}

int moma_bn_fwd(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                int N, int C, int HW, int dtype, int act, int training, float momentum, float eps, void* plane_mean,
                moma_stream_t stream) {
    if (!x || !out) return MOMA_E_NULL;
    if (!training && (!running_mean || !running_var)) return MOMA_E_NULL;
    const int rc = bn_check(N, C, HW, dtype, act, workspace, workspace_bytes);
    if (rc != MOMA_OK) return rc;
    return hip_rc(launch_bn_fwd(x, out, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                                (float*)workspace, N, C, HW, dtype, act, training, momentum, eps, plane_mean, (hipStream_t)stream));
}

int moma_bn_bwd(const void* x, const void* dout, const float* gamma, const float* beta, const float* save_mean,
                const float* save_invstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                size_t workspace_bytes, int N, int C, int HW, int dtype, int act, int training,
                const void* dplane_mean, moma_stream_t stream) {
    if (!x || !dout || !save_mean || !save_invstd) return MOMA_E_NULL;
    const int rc = bn_check(N, C, HW, dtype, act, workspace, workspace_bytes);
    if (rc != MOMA_OK) return rc;
    return hip_rc(launch_bn_bwd(x, dout, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, (float*)workspace, N,
                                C, HW, dtype, act, training, dplane_mean, (hipStream_t)stream));
}

size_t moma_dwconv_workspace_bytes(int C, int K) {
    return (C > 0 && K > 0) ? align_up(dwconv_workspace_floats(C, K) * sizeof(float), 256) : 0;
}

static int dw_check(int N, int C, int H, int W, int OH, int OW, int K, int S, int pt, int pl, int dtype) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || (long)N * C > 0x7fffffffL) return MOMA_E_SHAPE;
    if (!dwconv_supported(K, S)) return MOMA_E_SHAPE;
    if (pt < 0 || pl < 0 || pt >= K || pl >= K) return MOMA_E_SHAPE;
    // every output must read at least its first tap row/column inside the padded image
    if ((long)(OH - 1) * S - pt >= H || (long)(OW - 1) * S - pl >= W) return MOMA_E_SHAPE;
    if (dtype != MOMA_DT_F32 && dtype != MOMA_DT_BF16) return MOMA_E_DTYPE;
    return MOMA_OK;
}

int moma_dwconv_fwd(const void* x, const float* w, void* y, int N, int C, int H, int W, int OH, int OW, int K, int stride,
                    int pad_top, int pad_left, int dtype, moma_stream_t stream) {
    if (!x || !w || !y) return MOMA_E_NULL;
    const int rc = dw_check(N, C, H, W, OH, OW, K, stride, pad_top, pad_left, dtype);
    if (rc != MOMA_OK) return rc;
    return hip_rc(launch_dw_fwd(x, w, y, N, C, H, W, OH, OW, K, stride, pad_top, pad_left, dtype, (hipStream_t)stream));
}

int moma_dwconv_bwd_data(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int OH, int OW, int K,
                         int stride, int pad_top, int pad_left, int dtype, moma_stream_t stream) {
    if (!dy || !w || !dx) return MOMA_E_NULL;
    const int rc = dw_check(N, C, H, W, OH, OW, K, stride, pad_top, pad_left, dtype);
    if (rc != MOMA_OK) return rc;
    return hip_rc(launch_dw_bwd_data(dy, w, dx, N, C, H, W, OH, OW, K, stride, pad_top, pad_left, dtype, (hipStream_t)stream));
}

int moma_dwconv_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int C,
                           int H, int W, int OH, int OW, int K, int stride, int pad_top, int pad_left, int dtype,
                           moma_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return MOMA_E_NULL;
    const int rc = dw_check(N, C, H, W, OH, OW, K, stride, pad_top, pad_left, dtype);
    if (rc != MOMA_OK) return rc;
    if (workspace_bytes < moma_dwconv_workspace_bytes(C, K)) return MOMA_E_WORKSPACE;
    return hip_rc(launch_dw_bwd_weight(x, dy, dw, (float*)workspace, workspace_bytes / sizeof(float), N, C, H, W, OH, OW, K,
                                       stride, pad_top, pad_left, dtype, (hipStream_t)stream));
}

static int se_check(int NC, int HW, int dtype) {
    if (NC <= 0 || HW <= 0) return MOMA_E_SHAPE;
    if (dtype != MOMA_DT_F32 && dtype != MOMA_DT_BF16) return MOMA_E_DTYPE;
    return MOMA_OK;
}
int moma_plane_mean(const void* x, void* mean, int NC, int HW, int dtype, moma_stream_t stream) {
    if (!x || !mean) return MOMA_E_NULL;
    const int rc = se_check(NC, HW, dtype);
    return rc != MOMA_OK ? rc : hip_rc(launch_plane_mean(x, mean, NC, HW, dtype, (hipStream_t)stream));
}
int moma_se_gate_fwd(const void* x, const void* s, void* out, int NC, int HW, int dtype, moma_stream_t stream) {
    if (!x || !s || !out) return MOMA_E_NULL;
    const int rc = se_check(NC, HW, dtype);
    return rc != MOMA_OK ? rc : hip_rc(launch_se_gate_fwd(x, s, out, NC, HW, dtype, (hipStream_t)stream));
}
int moma_se_gate_bwd(const void* x, const void* s, const void* dout, void* dx, void* ds, int NC, int HW, int dtype,
                     moma_stream_t stream) {
    if (!x || !s || !dout || !dx || !ds) return MOMA_E_NULL;
    const int rc = se_check(NC, HW, dtype);
    return rc != MOMA_OK ? rc : hip_rc(launch_se_gate_bwd(x, s, dout, dx, ds, NC, HW, dtype, (hipStream_t)stream));
}

}  // extern "C"
