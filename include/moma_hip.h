/*
 * moma_hip.h -- C ABI of libmoma_hip.so: the MI355X (gfx950) kernels behind MoMA's
 * contrastive-distillation training step.
 *
 * The reference (trinhvg/MoMA) is pure Python on PyTorch and has no native interface of its own, so
 * every entry point below replaces a *chain of ATen ops* at a reference call site; the call site is
 * cited per function (paths relative to the reference root).  A maintainer binds these with ctypes
 * (see INTEGRATION.md); moma_amd/_lib.py is that binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (torch tensors' data_ptr()); nothing is
 *     retained past the call; the library allocates no device memory: workspaces are caller-provided
 *     and sized by the *_workspace_bytes() queries;
 *   - every launch function takes the hipStream_t to enqueue on (as void*), is asynchronous on it and
 *     performs no host synchronisation (safe under hipGraph capture);
 *   - return value: 0 = ok, <0 = argument check failed (MOMA_E_*), >0 = hipError_t of a failed launch;
 *     nothing throws; no environment variable is read; no mutable global state apart from once-per-process kernel-attribute
 *     setups (dynamic-LDS opt-ins) behind std::call_once and the debug knob moma_debug_set_k2_target_wg() -- they are made for the device that is current at the first call:
 *     ONE DEVICE PER PROCESS (the DDP model: one host thread per process / GPU); re-entrant within that;
 *   - matrices are row-major and dense unless a leading dimension is given;
 *   - `prec`   : arithmetic of the contractions. MOMA_PREC_F32 = f32-input MFMA (exact fp32 fma chain,
 *                the reference's arithmetic), MOMA_PREC_BF16 = bf16-input MFMA with fp32 accumulate;
 *   - `qdtype` : storage type of the K x d feature queue (MOMA_DT_F32 = reference storage,
 *                MOMA_DT_BF16 = 2-byte rows).
 */
#ifndef MOMA_HIP_H
#define MOMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOMA_ABI_VERSION 4

enum { MOMA_PREC_F32 = 0, MOMA_PREC_BF16 = 1 };
enum { MOMA_DT_F32 = 0, MOMA_DT_BF16 = 1 };

enum {
    MOMA_OK = 0,
    MOMA_E_NULL = -1,      /* a required pointer is NULL                        */
    MOMA_E_SHAPE = -2,     /* a dimension is <= 0 or inconsistent               */
    MOMA_E_DTYPE = -3,     /* unknown prec / qdtype                             */
    MOMA_E_ALIGN = -4,     /* pointer not aligned to its element size -- or, for the operands the fast paths move in 16-byte pieces (K1 fast
                              path, one-pass K2: q, k, queue, dq), not to 16 bytes */
    MOMA_E_WORKSPACE = -5, /* workspace too small                               */
    MOMA_E_UNSUPPORTED = -6
};

typedef void* moma_stream_t; /* hipStream_t */

/* ABI version (== MOMA_ABI_VERSION of the header the library was built from). */
int moma_version(void);
/* Human-readable text for a return code of this library (static storage). */
const char* moma_error_string(int code);
/* DEBUG knob, for plan sweeps only (scripts/sweep_k2_plan.sh): cut the K2 passes over the queue into about `n` workgroups instead
 * of the product's own plan (one per compute unit, fewer for short passes).  n = 0 restores the plan; 8 <= n <= 1024 otherwise.
 * Returns the previous value, or -1 when n is refused.  Process-wide and not synchronised with calls in flight: the launch plan AND
 * moma_infonce_fused*_workspace_bytes() follow it, so set it once, before the first workspace query, and size every workspace
 * after that.  This is the one piece of caller-set state in the library and the reason it reads no environment variable. */
int moma_debug_set_k2_target_wg(int n);

/* ---------------------------------------------------------------------------------------------
 * K4  multi-tensor EMA -- replaces ContrastTrainer.momentum_update
 *     (learning/contrast_trainer.py:207-211: per-tensor p2.mul_(m).add_(p1, alpha=1-m)).
 *     One launch for the whole parameter list:  ema = fma(fl32(1-m), p, fl32(ema*fl32(m))).
 *     `table` is a device array of n_tensors records {int64 ema_ptr, int64 p_ptr, int64 numel,
 *     int64 first_block}; first_block = exclusive prefix sum of ceil(numel / MOMA_EMA_BLOCK_ELEMS);
 *     total_blocks = that sum.  The table is built once per model pair by the host.
 * ------------------------------------------------------------------------------------------- */
#define MOMA_EMA_BLOCK_ELEMS 4096
int moma_ema_multi(const int64_t* table, int n_tensors, int64_t total_blocks,
                   float m, float one_minus_m, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  ring-buffer enqueue -- replaces BaseMoCo._update_memory (MoMA/mem_moco.py:17-27:
 *     arange+index -> fmod(K) -> index_copy_).   queue[(index+i) mod K, :] = rows[i, :], i in [0,n).
 *     n > K (duplicate slots) is last-writer-wins like a serial index_copy_.  The pointer update
 *     index = (index+n) mod K (MoMA/mem_moco.py:14-15) stays a host integer in the caller.
 *     rows are fp32 [n,d]; queue is [K,d] in `qdtype` (bf16 rows are rounded to nearest even).
 * ------------------------------------------------------------------------------------------- */
int moma_enqueue(void* queue, const float* rows, int n, int64_t index, int K, int d,
                 int qdtype, moma_stream_t stream);
/* The same enqueue into an fp32 queue (the reference's storage) AND its bf16 mirror (what the bf16-policy one-pass K2 streams:
 * half the HBM bytes per step, no conversion pass over the K x d queue) in ONE launch from one read of the rows. */
int moma_enqueue_mirror(float* queue, void* mirror_bf16, const float* rows, int n, int64_t index, int K, int d,
                        moma_stream_t stream);

/* Cache hint for K2: streams `bytes` of the queue once (16-B loads, data dropped) so that it sits in the 256 MiB memory-side
 * Infinity Cache when moma_infonce_fused reads it next.  In the training step the K x d queue was last touched a whole step
 * (gigabytes of activations) ago; issued on a second stream under the latency-bound attention launches that precede K2, the
 * sweep costs no step time and K2 then runs at its cache-resident rate.  Purely a performance hint: no result depends on it.
 * (Round 5: the training loop no longer issues it by default -- a second read of the queue per step for 0.6 us of K2 at B = 256.) */
int moma_queue_prefetch(const void* queue, size_t bytes, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  InfoNCE over the queue.
 *
 * moma_infonce_logits -- replaces BaseMoCo._compute_logit (MoMA/mem_moco.py:29-49):
 *     out[b,0] = <q_b,k_b>*inv_T ; out[b,1+j] = <queue_j,q_b>*inv_T ; out is [B,K+1] fp32 contiguous.
 * moma_infonce_logits_bwd -- the autograd backward of the above w.r.t. q:
 *     dq[b,:] = (dlogits[b,0]*k_b + sum_j dlogits[b,1+j]*queue_j) * inv_T          (dq is [B,d] fp32)
 *     The contraction over K is split over workgroups.  moma_infonce_logits_bwd_ws (round 6) keeps one partial product per split
 *     in a caller-owned workspace (moma_infonce_logits_bwd_workspace_bytes()) and adds them in split order: bitwise reproducible,
 *     like every other result of this library.  moma_infonce_logits_bwd, the form without a workspace, lets the splits meet in
 *     fp32 atomics: equal up to the order of the additions (last bits differ from run to run).
 * moma_infonce_fused -- replaces the whole chain MoCo.forward (MoMA/mem_moco.py:77-100, minus the
 *     enqueue) + nn.CrossEntropyLoss(logits, 0) + top-1 accuracy (helper/loops_moma.py:322,331-335,
 *     learning/contrast_trainer.py:189-205) and its backward, in one pass over the queue:
 *       lse[b]       = logsumexp_j out[b,j]
 *       loss_rows[b] = lse[b] - out[b,0]                  (loss_kd = mean_b loss_rows[b])
 *       top1[b]      = 1 if out[b,0] >= max_j out[b,j] else 0
 *       dq[b,:]      = d(sum_b loss_rows)/dq_b = ((p_b0-1)*k_b + sum_j p_bj*queue_j)*inv_T
 *     dq may be NULL (forward only).  workspace: moma_infonce_fused_workspace_bytes() -- it depends on (qdtype, prec): under
 *     prec = fp32 with a bf16-stored queue it includes room for a widened (fp32, exact) copy of the queue, made and dropped inside
 *     the call, at the widths the one-pass fp32 kernel takes.
 * ------------------------------------------------------------------------------------------- */
int moma_infonce_logits(const float* q, const float* k, const void* queue, float* out,
                        int B, int d, int K, float inv_T, int qdtype, int prec, moma_stream_t stream);
int moma_infonce_logits_bwd(const float* dlogits, const float* k, const void* queue, float* dq,
                            int B, int d, int K, float inv_T, int qdtype, int prec,
                            moma_stream_t stream);
size_t moma_infonce_logits_bwd_workspace_bytes(int B, int d, int K);
int moma_infonce_logits_bwd_ws(const float* dlogits, const float* k, const void* queue, float* dq,
                               int B, int d, int K, float inv_T, int qdtype, int prec,
                               void* workspace, size_t workspace_bytes, moma_stream_t stream);
/* gradients of the logits w.r.t. the key / queue operands (needed by the MoCoAtt cross-attention variants,
 * MoMA/mem_moco.py:103-161, where k and the queue are attention outputs that carry gradient):
 *     dk[b,:]     = dlogits[b,0] * q_b * inv_T                          (may be NULL)
 *     dqueue[j,:] = sum_b dlogits[b,1+j] * q_b * inv_T   ([K,d] fp32)    (may be NULL) */
int moma_infonce_logits_bwd_kq(const float* dlogits, const float* q, float* dk, float* dqueue,
                               int B, int d, int K, float inv_T, int prec, moma_stream_t stream);
size_t moma_infonce_fused_workspace_bytes(int B, int d, int K, int qdtype, int prec);
int moma_infonce_fused(const float* q, const float* k, const void* queue, int B, int d, int K,
                       float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq,
                       void* workspace, size_t workspace_bytes, int qdtype, int prec,
                       moma_stream_t stream);
/* Same, with an optional measurement hook: ev_begin / ev_end are hipEvent_t handles (or NULL) recorded on
 * `stream` immediately before / after the dominant kernel of the call (the one pass over the queue), so a
 * benchmark can time exactly the kernel its roofline is quoted for.  No synchronisation is performed. */
int moma_infonce_fused_ex(const float* q, const float* k, const void* queue, int B, int d, int K,
                          float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq,
                          void* workspace, size_t workspace_bytes, int qdtype, int prec,
                          moma_stream_t stream, void* ev_begin, void* ev_end);
/* Same, with the query ALSO handed over pre-packed: q_packed (nullable) holds q * inv_T * log2(e) rounded to bf16 in the
 * MFMA-operand order the one pass over the queue loads it in,
 *     unit[((row / 32) * (d / 16) + col / 16) * 64 + ((col / 8) & 1) * 32 + row % 32] = 8 bf16 = columns 8 * (col / 8) .. + 7 of row,
 * 16 bytes per unit, rows padded to a multiple of 128 with zeros: moma_infonce_qpack_bytes(B, d) bytes (0 = this width takes
 * no pre-packed query: d must be 128 / 256 / 384 / 512).  The producer of q writes it -- moma_mha_fwd_fast does when asked
 * (moma_mha_module_t.qpack) -- and the call then runs no pre-pack launch.  q itself (fp32) is still read for the exact positive
 * logit.  The results are bit-identical to moma_infonce_fused_ex on the same q.
 * ev_call_end (hipEvent_t or NULL): recorded when the LAST kernel of the call has finished (on its dispatch), so that
 * ev_begin .. ev_call_end spans the kernels of the call without the latency of event packets around it. */
size_t moma_infonce_qpack_bytes(int B, int d);
int moma_infonce_fused_q(const float* q, const void* q_packed, const float* k, const void* queue, int B, int d, int K,
                         float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq,
                         void* workspace, size_t workspace_bytes, int qdtype, int prec,
                         moma_stream_t stream, void* ev_begin, void* ev_end, void* ev_call_end);
/* K2 + K3 in one call: moma_infonce_fused_q(...) followed by the enqueue of `rows` [n,d] at ring pointer `index`
 * (MoCo.forward, MoMA/mem_moco.py:77-100: logits from the PRE-enqueue queue, then _update_memory).  Where the call ends in the
 * combine launch (the bf16 policy's one-pass and two-pass paths) the enqueue rides on THAT launch -- extra workgroups behind every
 * read of the queue -- instead of a launch of its own; elsewhere it is issued behind the call.  Same results as the two calls.
 *   queue     : what K2 reads, written in place: fp32 rows (qdtype F32) or rows rounded to bf16 (qdtype BF16);
 *   queue_f32 : NULL, or -- qdtype BF16 only -- the fp32 queue whose bf16 mirror `queue` is: both take the rows (moma_enqueue_mirror).
 * The ring pointer stays with the caller (a host integer: bit-exact contract); n = 0 is K2 alone. */
int moma_infonce_fused_enqueue(const float* q, const void* q_packed, const float* k, void* queue, int B, int d, int K,
                               float inv_T, float* loss_rows, float* lse, int32_t* top1, float* dq, void* workspace,
                               size_t workspace_bytes, int qdtype, int prec, const float* rows, int n, int64_t index,
                               float* queue_f32, moma_stream_t stream, void* ev_begin, void* ev_end, void* ev_call_end);

/* Several InfoNCE terms over queues of the same shape in ONE sweep -- replaces the 2 / 4 `_compute_logit` calls + CrossEntropy of
 * the dual-queue memories MoCoST.forward / MoCoSSTT.forward (MoMA/mem_moco.py:165-253):
 *     (q, k, memory_s), (q, k_t, memory_t) [, (q_t, k, memory_s), (q_t, k_t, memory_t)]
 * One pre-pack launch over the distinct q pointers, ONE launch of the one-pass kernel over every (term, query tile, key chunk) --
 * sized to one workgroup per compute unit over all terms, the queues streamed back to back --, one combine launch: 3 launches
 * for any number of terms (<= 4).  Per term the outputs of moma_infonce_fused; dq either for every term or for none (a query
 * shared by two terms gets two dq buffers: the caller adds them).  The one-sweep kernel takes the bf16 policy with bf16 queues and
 * d in {128, 256, 384, 512}; every other configuration moma_infonce_fused accepts (wide rows, exact fp32, fp32 queues) is served
 * by the same entry point term by term on the stream, through ONE workspace of moma_infonce_fused's size -- at those shapes a
 * term's own passes fill the chip and are not HBM-bound, so there is no sweep to share.  Queues are bf16 or fp32 per `qdtype`. */
typedef struct moma_infonce_term {
    const float* q;        /* [B,d] fp32  */
    const float* k;        /* [B,d] fp32  */
    const void* queue;     /* [K,d] bf16 / fp32 (qdtype) */
    float* loss_rows;      /* [B] out     */
    float* lse;            /* [B] out     */
    int32_t* top1;         /* [B] out     */
    float* dq;             /* [B,d] out or NULL */
} moma_infonce_term_t;
size_t moma_infonce_fused_multi_workspace_bytes(int n_terms, int B, int d, int K, int qdtype, int prec);
int moma_infonce_fused_multi(const moma_infonce_term_t* terms, int n_terms, int B, int d, int K, float inv_T,
                             void* workspace, size_t workspace_bytes, int qdtype, int prec, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1  batch-token multi-head attention -- replaces Attention.forward
 *     (MoMA/criterion_moco_att.py:153-167) and its autograd backward.
 *     x [N,d] -> qkv = x Wqkv^T + bqkv -> per head softmax(q k^T * hd^-1/2) v -> y = a Wproj^T + bproj.
 *
 *   Two families of entry points; moma_mha_saved_state() says which one a configuration takes:
 *
 *   MOMA_MHA_SAVE_LSE  ->  the FAST path (moma_mha_pack_weights / moma_mha_fwd_fast / moma_mha_bwd_fast): MOMA_PREC_BF16 with a
 *     head dim that is a multiple of 16: up to 128 (every `--head mlp` configuration) at any N, wider heads up to 1024
 *     (`--head None`: EfficientNet-B0 1280 / 4 = 320, ResNet-50 2048 / 4 = 512) at N <= 256.  Flash-style: the forward keeps the
 *     row log-sum-exp lse [H,N] (log2 units, scale included) and the backward recomputes P per tile, so no [H,N,N] array
 *     exists at any N (attn = 'all' runs over N = 2B + K tokens).  Everything a launch reads more than once is bf16:
 *       pack    caller-owned, moma_mha_pack_bytes(d) bytes: [Wqkv | Wproj | Wqkv^T | Wproj^T] as bf16, written by
 *               moma_mha_pack_weights (one launch; with_transposed = 0 fills only the first half -- enough for a forward);
 *               the caller refreshes it when the fp32 weights change (after optimizer.step());
 *       qkv16   [N,3d] bf16, the Q third pre-scaled by hd^-1/2 * log2(e);  attn16 [N,d] bf16  (both saved for the backward,
 *               together with x, lse and the pack);
 *     moma_mha_fwd_fast runs n_modules <= 4 modules of equal (N, d, H) in the SAME three launches (qkv linear, per-head core,
 *     proj linear: blockIdx selects the module) -- the loop's atts_k and atts_queue (helper/loops_moma.py:327-329) run as one
 *     group.  A module with qpack != NULL also gets y * qpack_scale in the packed bf16 MFMA-operand layout that
 *     moma_infonce_fused_q consumes (so K2 needs no pre-pack launch; the buffer has moma_infonce_qpack_bytes(N, d) bytes and
 *     its pad rows must be zero -- zero it once).
 *     moma_mha_bwd_fast: 3 launches -- {dA = dy Wproj with the row dots D, dWproj, dbproj} -> core (dQ | dK | dV) ->
 *     {dWqkv, dbqkv, dx}.  dx / (dw_qkv, db_qkv) / (dw_proj, db_proj) may be NULL to skip them; a bias gradient is only
 *     produced together with its weight gradient.  workspace: moma_mha_bwd_fast_workspace_bytes().
 *
 *   MOMA_MHA_SAVE_PROBS  ->  the STAGED path (moma_mha_fwd / moma_mha_bwd): the reference's op chain on one generic MFMA GEMM
 *     plus row softmax kernels -- exact fp32 under MOMA_PREC_F32 (f32-input MFMA), any N, d, H; keeps qkv [N,3d],
 *     attn_out [N,d] and probs [H,N,N] in fp32 for the backward.
 *     bwd writes dx [N,d], dw_qkv [3d,d], db_qkv [3d], dw_proj [d,d], db_proj [d]; any of the five may be NULL to skip it.
 *     bwd workspace: moma_mha_bwd_workspace_bytes().
 *   No atomics anywhere: results are bitwise reproducible run to run.  One device per process (kernel attributes are set once).
 * ------------------------------------------------------------------------------------------- */
enum { MOMA_MHA_SAVE_PROBS = 0, MOMA_MHA_SAVE_LSE = 1 };
int moma_mha_saved_state(int N, int d, int H, int prec);
int moma_mha_fwd(const float* x, const float* w_qkv, const float* b_qkv, const float* w_proj,
                 const float* b_proj, float* y, float* qkv, float* probs, float* attn_out,
                 int N, int d, int H, int prec, moma_stream_t stream);
size_t moma_mha_bwd_workspace_bytes(int N, int d, int H, int prec);
int moma_mha_bwd(const float* x, const float* w_qkv, const float* w_proj, const float* qkv,
                 const float* probs, const float* attn_out, const float* dy, float* dx,
                 float* dw_qkv, float* db_qkv, float* dw_proj, float* db_proj, void* workspace,
                 size_t workspace_bytes, int N, int d, int H, int prec, moma_stream_t stream);

typedef struct moma_mha_module {
    const void* x;        /* [N,d] input, fp32 or bf16 (x_dtype)                           */
    const void* pack;     /* bf16 weight pack (moma_mha_pack_weights)                      */
    const float* b_qkv;   /* [3d] or NULL                                                  */
    const float* b_proj;  /* [d]                                                           */
    float* y;             /* [N,d] fp32 output                                             */
    void* qkv16;          /* [N,3d] bf16 out (saved for the backward / scratch)            */
    void* attn16;         /* [N,d]  bf16 out (saved for the backward / scratch)            */
    float* lse;           /* [H,N] out, or NULL for a forward without backward             */
    void* qpack;          /* NULL, or moma_infonce_qpack_bytes(N,d) bytes: packed y * qpack_scale for moma_infonce_fused_q */
    float qpack_scale;
    int x_dtype;          /* MOMA_DT_F32 / MOMA_DT_BF16: storage type of x (a bf16 x -- the output of a bf16-autocast head -- is
                             consumed as it stands: the products round x to bf16 either way, the values are identical)        */
} moma_mha_module_t;
size_t moma_mha_pack_bytes(int d);
int moma_mha_pack_weights(const float* w_qkv, const float* w_proj, void* pack, int d, int with_transposed,
                          moma_stream_t stream);
int moma_mha_fwd_fast(const moma_mha_module_t* modules, int n_modules, int N, int d, int H, moma_stream_t stream);
size_t moma_mha_bwd_fast_workspace_bytes(int N, int d, int H);
int moma_mha_bwd_fast(const void* pack, const void* x, int x_dtype, const void* qkv16, const void* attn16, const float* lse,
                      const float* dy, float* dx, float* dw_qkv, float* db_qkv, float* dw_proj, float* db_proj,
                      void* workspace, size_t workspace_bytes, int N, int d, int H, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BN  BatchNorm2d + fused activation on NCHW activations -- the `_swish(_bn(conv(x)))` pairs of the
 *     backbones inside the step (models/efficientnet_pytorch/model.py:96-102,114; nn.BatchNorm2d
 *     semantics: batch statistics + running-stat update (momentum, unbiased variance) when
 *     training != 0, running statistics otherwise).  Not one of the KD-term kernels: an HBM-streaming
 *     helper for the step's throughput (3 passes forward, 5 backward instead of 5 + 8 unfused).
 *     x, out, dout, dx: [N, C, HW] contiguous, MOMA_DT_F32 or MOMA_DT_BF16; gamma/beta/running/save: fp32 [C]
 *     (gamma, beta, running_* may be NULL when training).  act: MOMA_ACT_*.  The backward recomputes the
 *     pre-activation from x and save_mean/save_invstd (written by the forward, eval mode included).
 *     workspace >= moma_bn_workspace_bytes(C) for either call.
 *     plane_mean (nullable, [N*C], activation dtype): the forward also emits mean_hw(out[n,c]) -- the squeeze of
 *     a squeeze-excite block that follows (replaces a separate F.adaptive_avg_pool2d pass); dplane_mean (nullable)
 *     is its gradient, folded into the backward as dout + dplane_mean / HW.
 * ------------------------------------------------------------------------------------------- */
enum { MOMA_ACT_NONE = 0, MOMA_ACT_SILU = 1, MOMA_ACT_RELU = 2 };
size_t moma_bn_workspace_bytes(int C);
int moma_bn_fwd(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float* save_mean, float* save_invstd, void* workspace,
                size_t workspace_bytes, int N, int C, int HW, int dtype, int act, int training,
                float momentum, float eps, void* plane_mean, moma_stream_t stream);
int moma_bn_bwd(const void* x, const void* dout, const float* gamma, const float* beta,
                const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta,
                void* workspace, size_t workspace_bytes, int N, int C, int HW, int dtype, int act,
                int training, const void* dplane_mean, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DW  depthwise convolution (groups == channels) on NCHW activations -- the MBConv `_depthwise_conv`
 *     layers of the backbones inside the step (models/efficientnet_pytorch/model.py:59-64,100; TF "SAME"
 *     padding, possibly asymmetric: pad_top / pad_left are given, bottom / right follow from OH / OW).
 *     Like BN a throughput helper for the step, not a KD-term kernel.  K in {3,5}, stride in {1,2}.
 *       y[n,c,oy,ox] = sum_{ky,kx} w[c,ky,kx] * x[n,c, oy*S+ky-pad_top, ox*S+kx-pad_left]   (zero outside)
 *     x, dx [N,C,H,W]; y, dy [N,C,OH,OW]: MOMA_DT_F32 or MOMA_DT_BF16;  w, dw [C,K,K] fp32.
 *     bwd_weight needs workspace >= moma_dwconv_workspace_bytes(C, K).
 * ------------------------------------------------------------------------------------------- */
size_t moma_dwconv_workspace_bytes(int C, int K);
int moma_dwconv_fwd(const void* x, const float* w, void* y, int N, int C, int H, int W, int OH, int OW,
                    int K, int stride, int pad_top, int pad_left, int dtype, moma_stream_t stream);
int moma_dwconv_bwd_data(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int OH,
                         int OW, int K, int stride, int pad_top, int pad_left, int dtype,
                         moma_stream_t stream);
int moma_dwconv_bwd_weight(const void* x, const void* dy, float* dw, void* workspace,
                           size_t workspace_bytes, int N, int C, int H, int W, int OH, int OW, int K,
                           int stride, int pad_top, int pad_left, int dtype, moma_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SE  squeeze-excite helpers on NCHW activations (models/efficientnet_pytorch/model.py:104-110):
 *     moma_plane_mean     mean[n,c] = mean_hw x[n,c,:,:]                       (F.adaptive_avg_pool2d(x, 1))
 *     moma_se_gate_fwd    out = x * sigmoid(s[n,c])                            (torch.sigmoid(x_squeezed) * x)
 *     moma_se_gate_bwd    dx = dout * sigmoid(s);  ds = sigmoid'(s) * sum_hw(dout * x)   -- one pass
 *     x, out, dout, dx: [NC, HW] contiguous; mean, s, ds: [NC]; all of dtype MOMA_DT_F32 or MOMA_DT_BF16.
 *     Throughput helpers of the step like BN / DW, not KD-term kernels.
 * ------------------------------------------------------------------------------------------- */
int moma_plane_mean(const void* x, void* mean, int NC, int HW, int dtype, moma_stream_t stream);
int moma_se_gate_fwd(const void* x, const void* s, void* out, int NC, int HW, int dtype, moma_stream_t stream);
int moma_se_gate_bwd(const void* x, const void* s, const void* dout, void* dx, void* ds, int NC, int HW,
                     int dtype, moma_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOMA_HIP_H */
