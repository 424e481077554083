// K2: one-pass ("flash") InfoNCE over the K x d feature queue -- the HBM/MFMA roofline kernel of the path.
//
//   InfoNCE with label 0 over [k_b ; queue] is flash-attention forward with keys == values == the queue and
//   one query per sample: one tiled pass with an online softmax yields lse (-> loss, top-1) AND the
//   un-normalised sum_j p_bj * queue_j (-> dq) from a single read of the queue.
//
// Decomposition (bf16 MFMA, fp32 accumulate; queue stored bf16):
//   grid  = nbt x nchunk workgroups (nbt = ceil(B/128) query tiles, nchunk key chunks, ~1 WG per CU);
//           the nbt workgroups that share a key chunk get block ids 8 apart -> same XCD, same L2.
//   WG    = 4 waves, ONE wave per SIMD with the whole 512-register file; wave w owns 32 query rows:
//             Q fragments [32 x D] bf16 as the B operand of the score MFMA   (D/4  VGPRs, resident)
//             O accumulator [32 x D] fp32                                    (D/2  AGPR/VGPRs, resident)
//   tile  = 32 keys x D bf16, double-buffered in LDS, filled by LDS-DMA (global_load_lds_dwordx4; the
//           bank swizzle is applied on the per-lane SOURCE address, the LDS image is lane-linear per piece).
//   score : X[key, q] = K_tile . Q^T      v_mfma_f32_32x32x16_bf16, A = keys (ds_read_b128), B = Q (regs)
//           -> the query sits on the lane, the 32 keys of the tile in the 16 registers x 2 lane halves,
//              so row max / sum are in-register (+1 cross-half shuffle) -- no LDS, no other wave involved;
//   P     : exp2(X - m) in registers, packed to bf16: registers 8s..8s+7 ARE the A fragment of k-step s of
//   P.K   : O[q, :] += P[q, keys] . K_tile  with B = keys read column-wise by ds_read_b64_tr_b16 in the
//           permuted k order key(s,h,j) = 16s + 8(j>>2) + 4h + (j&3).
//   The running max m is only raised when a tile would exceed it by more than 2^12 (rare); O is never
//   rescaled in registers: on that event the accumulator is MERGED into the workgroup's partial slot in
//   memory (slot = slot * 2^(m_slot - m) + O) and restarted from zero, which is also how the final partial is
//   written.  Each WG leaves (m, l, max, O) per query row; a small combine kernel merges the chunks, adds the
//   positive logit (exact fp32) and writes loss / lse / top-1 / dq.
//
// LDS image of a key tile: D/128 segments of [32 keys][128 cols] with 256-B rows,
//   off(seg,row,ch) = seg*8192 + row*256 + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))      ch = 16-B chunk 0..15
// which is conflict-free for both the row reads (ds_read_b128) and the transposed reads.
#include "common.hpp"

namespace moma {
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int QROWS_WG = 128;        // query rows per workgroup (4 waves x 32)
constexpr int KT = 32;               // keys per tile
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 12.0f; // log2 units: P <= 2^12 before the reference max is raised

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int D>
__device__ __forceinline__ void dma_tile(const bf16_raw* __restrict__ queue, long key0, int K, char* buf, int wave,
                                         int lane) {
    // tile = D/16 pieces of 1 KiB (4 rows x 256 B of one segment); wave w issues pieces w, w+4, ...
    constexpr int NPIECE = D / 16;
    const int rl = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int i = 0; i < NPIECE / 4; ++i) {
        const int pc = i * 4 + wave;            // wave-uniform
        const int seg = pc >> 3, rg = pc & 7;
        const int row = rg * 4 + rl;
        long key = key0 + row;
        if (key >= K) key = K - 1;              // clamp (masked in the softmax)
        const int ch = slot ^ swz(row);
        const char* src = reinterpret_cast<const char*>(queue) + key * (long)(D * 2) + seg * 256 + ch * 16;
        char* dst = buf + seg * 8192 + rg * 1024;   // wave-uniform LDS base; lane L lands at +16*L
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int D, bool WITH_DQ>
__global__ __launch_bounds__(256, 1) void infonce_flash_kernel(const float* __restrict__ q,
                                                               const bf16_raw* __restrict__ queue, int B, int K,
                                                               float scale_log2, int nbt, int nchunk,
                                                               int tiles_per_chunk, int Bpad,
                                                               float* __restrict__ o_part, float* __restrict__ m_part,
                                                               float* __restrict__ l_part, float* __restrict__ x_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16;       // k-steps of the score product
    constexpr int NCT = D / 32;      // 32-column tiles of O
    constexpr int TILE_BYTES = KT * D * 2;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;

    // block -> (query tile, key chunk): the nbt tiles of one chunk are 8 block ids apart (same XCD)
    int bt, chunk;
    {
        const int id = blockIdx.x;
        if ((nchunk & 7) == 0) {
            const int g = id / (8 * nbt), r = id % (8 * nbt);
            bt = r >> 3;
            chunk = g * 8 + (r & 7);
        } else {
            bt = id % nbt;
            chunk = id / nbt;
        }
    }
    const int ntiles = (K + KT - 1) / KT;
    const int t0 = chunk * tiles_per_chunk;
    const int t1 = min(t0 + tiles_per_chunk, ntiles);
    const int qrow = bt * QROWS_WG + wave * 32 + n;      // the query this lane carries in the score layout

    // ---- Q fragments: B operand of X = K.Q^T ; lane (q=n, h) holds Q[q][16ks + 8h + j], pre-scaled by log2e/T
    bf16x8 qf[KS];
    {
        const bool ok = qrow < B;
        const float* qp = q + (long)(ok ? qrow : 0) * D + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (ok) {
                a = *reinterpret_cast<const float4*>(qp + 16 * ks);
                b = *reinterpret_cast<const float4*>(qp + 16 * ks + 4);
            }
            qf[ks] = bf16x8{(__bf16)(a.x * scale_log2), (__bf16)(a.y * scale_log2), (__bf16)(a.z * scale_log2),
                            (__bf16)(a.w * scale_log2), (__bf16)(b.x * scale_log2), (__bf16)(b.y * scale_log2),
                            (__bf16)(b.z * scale_log2), (__bf16)(b.w * scale_log2)};
        }
    }

    f32x16 O[WITH_DQ ? NCT : 1];
    if constexpr (WITH_DQ) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    }
    float m_run = NEG_BIG, l_run = 0.f, mx = NEG_BIG;

    // per-lane LDS offsets
    //  row read (A operand of the score MFMA): key row n, chunk 2*(ks&7)+h of segment ks>>3
    int a_off[8];
    {
        const int g = swz(n) ^ h;
#pragma unroll
        for (int c = 0; c < 8; ++c) a_off[c] = n * 256 + (((2 * c) ^ g) << 4);
    }
    //  transposed read (B operand of P.K): 16-lane group -> 4 keys x 16 columns
    int b_off[4][2];
    {
        const int i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
        const int e = (2 * g2 + (p >> 1)) ^ h;
        const int base = (4 * h + q4) * 256 + 8 * (p & 1);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u) b_off[c][u] = base + (((((c ^ q4) << 2) | (e ^ (2 * u)))) << 4);
    }

    // partial slot of this wave's 32 query rows in chunk `chunk`
    const long prow = (long)chunk * Bpad + bt * QROWS_WG + wave * 32;
    bool o_dirty = false;        // O holds un-flushed contributions
    bool slot_used = false;      // the slot already holds a flushed accumulator (relative to m_slot)
    float m_slot = NEG_BIG;
    auto merge_store = [&](float m_ref) {
        // slot = slot * 2^(m_slot - m_ref) + O ; O = 0.  O carries the query on registers / lane half.
        if constexpr (WITH_DQ) {
            if (slot_used) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const float a_q = __builtin_amdgcn_exp2f(m_slot - m_ref);
                float av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = __shfl(a_q, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* dst = o_part + (prow + (r & 3) + 8 * (r >> 2) + 4 * h) * D + c * 32 + n;
                        *dst = fmaf(*dst, av[r], O[c][r]);
                        O[c][r] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o_part[(prow + (r & 3) + 8 * (r >> 2) + 4 * h) * D + c * 32 + n] = O[c][r];
                        O[c][r] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            slot_used = true;
            m_slot = m_ref;
        }
    };

    if (t0 < t1) dma_tile<D>(queue, (long)t0 * KT, K, smem, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        char* buf = smem + ((t - t0) & 1) * TILE_BYTES;
        if (t + 1 < t1) dma_tile<D>(queue, (long)(t + 1) * KT, K, smem + (((t - t0) & 1) ^ 1) * TILE_BYTES, wave, lane);

        // ---- scores: X[key, q] over D ; fragments fetched one group (4 k-steps) ahead of the MFMAs that use them
        f32x16 x;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
        {
            constexpr int G = 4, NG = KS / G;
            bf16x8 kf[2][G];
#pragma unroll
            for (int i = 0; i < G; ++i)
                kf[0][i] = *reinterpret_cast<const bf16x8*>(buf + (i >> 3) * 8192 + a_off[i & 7]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int ks = (g + 1) * G + i;
                        kf[(g + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(buf + (ks >> 3) * 8192 + a_off[ks & 7]);
                    }
                }
#pragma unroll
                for (int i = 0; i < G; ++i)
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g & 1][i], qf[g * G + i], x, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // key of register r on this lane: (r&3) + 8*(r>>2) + 4*h
        if ((t + 1) * KT > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KT + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= K) x[r] = NEG_BIG;
            }
        }
        float tmax = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        mx = fmaxf(mx, tmax);
        if (!__all(tmax - m_run <= RESCALE_THR)) {
            // raise the reference max: flush what O holds (relative to the old m_run) into the partial slot
            if constexpr (WITH_DQ) {
                if (o_dirty) {
                    merge_store(m_run);
                    o_dirty = false;
                }
            }
            const float m_new = fmaxf(m_run, tmax);
            l_run *= __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = __builtin_amdgcn_exp2f(x[r] - m_run);
            psum += x[r];
        }
        l_run += psum;

        if constexpr (WITH_DQ) {
            bf16x8 pa[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                pa[s] = bf16x8{(__bf16)x[8 * s + 0], (__bf16)x[8 * s + 1], (__bf16)x[8 * s + 2], (__bf16)x[8 * s + 3],
                               (__bf16)x[8 * s + 4], (__bf16)x[8 * s + 5], (__bf16)x[8 * s + 6], (__bf16)x[8 * s + 7]};
            // ---- O[q, cols] += P[q, keys] . K_tile[keys, cols] ; column tile c+1's key fragments are fetched
            //      while tile c's MFMAs run
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            auto ld_kb = [&](int c, int s) -> bf16x8 {
                const char* pb = buf + (c >> 2) * 8192 + (16 * s) * 256;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(pb + b_off[c & 3][0]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(pb + 8 * 256 + b_off[c & 3][1]));
                const s16x8 kb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                return __builtin_bit_cast(bf16x8, kb);
            };
            bf16x8 kb[2][2];
            kb[0][0] = ld_kb(0, 0);
            kb[0][1] = ld_kb(0, 1);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                if (c + 1 < NCT) {
                    kb[(c + 1) & 1][0] = ld_kb(c + 1, 0);
                    kb[(c + 1) & 1][1] = ld_kb(c + 1, 1);
                }
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], kb[c & 1][0], O[c], 0, 0, 0);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], kb[c & 1][1], O[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            o_dirty = true;
        }
        // next tile's DMA must have landed, and every wave must be done reading this buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- partials: per (chunk, query row): m, l, true max ; O[chunk][row][D] (relative to m)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (h == 0) {
        m_part[prow + n] = m_run;
        l_part[prow + n] = l_tot;
        x_part[prow + n] = mx;
    }
    merge_store(m_run);
}

// merge the key chunks of one query row, add the positive logit (exact fp32), emit loss / lse / top-1 / dq
__global__ __launch_bounds__(256) void infonce_combine_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               int B, int D, float inv_T, int nchunk, int Bpad,
                                                               const float* __restrict__ o_part,
                                                               const float* __restrict__ m_part,
                                                               const float* __restrict__ l_part,
                                                               const float* __restrict__ x_part,
                                                               float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                               int32_t* __restrict__ top1, float* __restrict__ dq) {
    __shared__ float red[4];
    __shared__ float wts[1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    // positive logit
    float s = 0.f;
    for (int c = tid; c < D; c += 256) s = fmaf(q[(long)b * D + c], k[(long)b * D + c], s);
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    const float s0 = (red[0] + red[1] + red[2] + red[3]) * inv_T;
    const float s0l = s0 * LOG2E;
    __syncthreads();
    float M = NEG_BIG, X = NEG_BIG;
    for (int c = tid; c < nchunk; c += 256) {
        M = fmaxf(M, m_part[(long)c * Bpad + b]);
        X = fmaxf(X, x_part[(long)c * Bpad + b]);
    }
    M = wave_max(M);
    X = wave_max(X);
    if (lane == 0) red[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), s0l);
    __syncthreads();
    if (lane == 0) red[wid] = X;
    __syncthreads();
    X = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float L = 0.f;
    for (int c = tid; c < nchunk; c += 256) {
        const float w = exp2f(m_part[(long)c * Bpad + b] - M);
        wts[c] = w;
        L += w * l_part[(long)c * Bpad + b];
    }
    L = wave_sum(L);
    if (lane == 0) red[wid] = L;
    __syncthreads();
    const float p0u = exp2f(s0l - M);
    L = red[0] + red[1] + red[2] + red[3] + p0u;
    const float lse = (M + log2f(L)) * LN2;
    if (tid == 0) {
        lse_out[b] = lse;
        loss_rows[b] = lse - s0;
        top1[b] = (s0l >= X) ? 1 : 0;
    }
    if (dq != nullptr) {
        const float invL = 1.f / L;
        const float cpos = p0u * invL - 1.f;
        for (int c = tid; c < D; c += 256) {
            float acc = 0.f;
            for (int j = 0; j < nchunk; ++j) acc = fmaf(wts[j], o_part[((long)j * Bpad + b) * D + c], acc);
            dq[(long)b * D + c] = (cpos * k[(long)b * D + c] + acc * invL) * inv_T;
        }
    }
}

struct FlashPlan {
    int nbt, nchunk, tiles_per_chunk, Bpad;
};

FlashPlan plan(int B, int K) {
    FlashPlan p;
    p.nbt = (B + QROWS_WG - 1) / QROWS_WG;
    p.Bpad = p.nbt * QROWS_WG;
    const int ntiles = (K + KT - 1) / KT;
    int want = 256 / p.nbt;                    // ~1 workgroup per CU
    if (want < 8) want = 8;
    if (want > 1024) want = 1024;
    want = (want / 8) * 8;
    int tpc = (ntiles + want - 1) / want;
    if (tpc < 1) tpc = 1;
    p.tiles_per_chunk = tpc;
    p.nchunk = (ntiles + tpc - 1) / tpc;       // no empty chunk by construction
    return p;
}

}  // namespace

bool infonce_flash_supported(int B, int d, int K, int qdtype, int prec) {
    if (prec != MOMA_PREC_BF16 || qdtype != MOMA_DT_BF16) return false;
    if (d != 256 && d != 384 && d != 512) return false;
    return B >= 1 && K >= 1;
}

size_t infonce_flash_workspace_bytes(int B, int d, int K) {
    const FlashPlan p = plan(B, K);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    return (rows * d + 3 * rows) * sizeof(float) + 256;
}

hipError_t launch_infonce_flash(const float* q, const float* k, const void* queue, int B, int d, int K, float inv_T,
                                float* loss_rows, float* lse, int32_t* top1, float* dq, void* ws, int qdtype,
                                hipStream_t st) {
    const FlashPlan p = plan(B, K);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    float* m_part = (float*)ws;
    float* l_part = m_part + rows;
    float* x_part = l_part + rows;
    float* o_part = x_part + rows;
    const float scale_log2 = inv_T * 1.4426950408889634f;
    const dim3 grid(p.nbt * p.nchunk), block(256);
    const size_t lds = 2 * (size_t)KT * d * 2;
    const bf16_raw* qu = (const bf16_raw*)queue;
#define MOMA_FLASH_LAUNCH(DD)                                                                                      \
    do {                                                                                                           \
        if (dq) {                                                                                                  \
            hipFuncSetAttribute((const void*)infonce_flash_kernel<DD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((infonce_flash_kernel<DD, true>), grid, block, lds, st, q, qu, B, K, scale_log2, p.nbt,  \
                               p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, m_part, l_part, x_part);               \
        } else {                                                                                                   \
            hipFuncSetAttribute((const void*)infonce_flash_kernel<DD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((infonce_flash_kernel<DD, false>), grid, block, lds, st, q, qu, B, K, scale_log2, p.nbt, \
                               p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, m_part, l_part, x_part);               \
        }                                                                                                          \
    } while (0)
    if (d == 512) MOMA_FLASH_LAUNCH(512);
    else if (d == 384) MOMA_FLASH_LAUNCH(384);
    else MOMA_FLASH_LAUNCH(256);
#undef MOMA_FLASH_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(infonce_combine_kernel, dim3(B), dim3(256), 0, st, q, k, B, d, inv_T, p.nchunk, p.Bpad, o_part,
                       m_part, l_part, x_part, loss_rows, lse, top1, dq);
    return hipGetLastError();
}

}  // namespace moma
