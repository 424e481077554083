// K1 core: fused per-head batch-token attention forward (bf16 MFMA policy).
//
//   a[:, h*hd:(h+1)*hd] = softmax(q_h k_h^T * hd^-1/2) v_h          (MoMA/criterion_moco_att.py:159-163)
//
// replaces the three launches "scores GEMM -> row softmax -> context GEMM" of the staged path.  The problem is tiny
// (N = per-rank batch = 256 tokens, 4 heads) and therefore latency bound, so the shape of the kernel is chosen
// for parallelism and few dependent memory round trips, not for operand reuse:
//   * one workgroup = one head x 32 query rows; its 8 waves split the KEYS (tile t of 32 keys goes to wave t%8),
//     so N=256, H=4 already gives 32 workgroups x 8 waves with ONE key tile per wave and each K/V element is used by exactly one wave of the
//     group -- operands therefore go global -> registers directly (fp32 -> bf16 on the way), no LDS staging:
//       K as the A operand: lane (key n, half h2) reads 8 consecutive floats of its key row per k-step (32 B),
//       V as the B operand: lane (col n, half h2) reads 8 keys of column 32c + n per k-step -- every load
//       instruction covers two full 128-B row segments.
//   * scores X[key, q] = K_tile . Q^T with v_mfma_f32_32x32x16_bf16: the query sits on the lane, the tile's 32
//     keys in 16 registers x 2 lane halves -> row max / sum are in-register plus one cross-half wave shuffle.
//   * each wave gets an (m, l) pair per query over its keys, merged across the waves through LDS into the row
//     log-sum-exp; then (N <= 256: from the score registers it still holds; larger N: a second pass that
//     recomputes the scores) it forms NORMALISED probabilities and accumulates O += P . V per wave; the partial O are
//     summed through LDS and stored as coalesced rows.  The probabilities are never written: what the backward keeps is
//     the row log-sum-exp lse[h][q] (log2 units, scale included) -- it recomputes P tile by tile (flash-style), so no
//     [H, N, N] array exists at any N (attn = 'all' runs over N = 2B + K tokens).
#include "common.hpp"

namespace moma {
namespace {

constexpr int KT = 32;
constexpr int DP = 128;                       // padded head dim
constexpr int KS = DP / 16;                   // k-steps of the score product
constexpr int NCT = DP / 32;                  // 32-column tiles of O
constexpr int NW = 8;                         // waves per workgroup (split over keys)
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b, float s) {
    return bf16x8{(__bf16)(a.x * s), (__bf16)(a.y * s), (__bf16)(a.z * s), (__bf16)(a.w * s),
                  (__bf16)(b.x * s), (__bf16)(b.y * s), (__bf16)(b.z * s), (__bf16)(b.w * s)};
}

// 8 x 32 B of one matrix row as MFMA A/B fragments: lane (row, h2) -> columns 16ks + 8h2 .. +8
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[KS], const float* __restrict__ base, long ld, int row,
                                               int N, int hd, int h2, float s) {
    float4 a[KS], b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int col = 16 * ks + 8 * h2;
        // unconditional loads from a clamped address (a guarded load becomes a branch + a full wait per load)
        const float* p = base + (long)min(row, N - 1) * ld + min(col, hd - 8);
        a[ks] = *reinterpret_cast<const float4*>(p);
        b[ks] = *reinterpret_cast<const float4*>(p + 4);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float sk = (row < N && 16 * ks + 8 * h2 < hd) ? s : 0.f;
        f[ks] = cvt8(a[ks], b[ks], sk);
    }
}

template <bool ONE_TILE>
__global__ __launch_bounds__(NW * 64) void mha_core_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ attn_out,
                                                            float* __restrict__ lse_out, int N, int d, int H) {
    // [NW][32][DP] fp32 partial O (128 KiB) | [NW][32][2] (m, l)
    __shared__ __attribute__((aligned(16))) float s_o[NW * 32 * DP];
    __shared__ float s_ml[NW * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, hd = d / H;
    const int q0 = blockIdx.x * 32;
    const long ld = 3L * d;
    const float* qbase = qkv + head * hd;
    const float* kbase = qkv + d + head * hd;
    const float* vbase = qkv + 2 * d + head * hd;
    const float scale_log2 = 1.4426950408889634f / sqrtf((float)hd);
    const int ntiles = (N + KT - 1) / KT;

    bf16x8 qf[KS];                              // B operand: lane (q = n, h2), pre-scaled by hd^-1/2 * log2(e)
    bf16x8 kf[KS];                              // A operand: lane (key n, h2)
    float vv[NCT][2][8];                        // B operand of the context product, fp32 as loaded
    load_row_frags(qf, qbase, ld, q0 + n, N, hd, h2, scale_log2);
    auto load_v = [&](int t) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // the k index of k-step s enumerates the keys in the order the score registers hold them:
                    // register 8s + j is key 16s + 8*(j>>2) + 4*h2 + (j&3)
                    const int key = t * KT + 16 * s + 8 * (j >> 2) + 4 * h2 + (j & 3), col = 32 * c + n;
                    vv[c][s][j] = vbase[(long)min(key, N - 1) * ld + min(col, hd - 1)];
                }
    };
    // loads stay unconditional from clamped addresses (a guarded load is sunk under its guard: branch + full wait
    // per load); the pin keeps them so, the select zeroes what lies past N / hd
    auto mask_v = [&](int t) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = t * KT + 16 * s + 8 * (j >> 2) + 4 * h2 + (j & 3), col = 32 * c + n;
                    asm volatile("" : "+v"(vv[c][s][j]));
                    vv[c][s][j] = (key < N && col < hd) ? vv[c][s][j] : 0.f;
                }
    };
    auto scores = [&](int t, f32x16& x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], x, 0, 0, 0);
        if ((t + 1) * KT > N) {                 // keys past N: register r holds key (r&3) + 8*(r>>2) + 4*h2
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) x[r] = NEG_BIG;
        }
    };
    auto tile_ml = [&](const f32x16& x, float& m, float& l) {
        float tmax = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(x[r] - mn);
        ps += __shfl_xor(ps, 32, 64);
        l = l * __builtin_amdgcn_exp2f(m - mn) + ps;
        m = mn;
    };
    // merge the per-wave (m, l) into the row log-sum-exp (log2 units); contains the workgroup barrier
    auto merge_lse = [&](float m, float l) -> float {
        if (h2 == 0) {
            s_ml[(wave * 32 + n) * 2 + 0] = m;
            s_ml[(wave * 32 + n) * 2 + 1] = l;
        }
        __syncthreads();
        float mm = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, s_ml[(w * 32 + n) * 2]);
        float ll = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ll += s_ml[(w * 32 + n) * 2 + 1] * __builtin_amdgcn_exp2f(s_ml[(w * 32 + n) * 2] - mm);
        return mm + __builtin_amdgcn_logf(ll);                  // v_log_f32 = log2
    };
    f32x16 O[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    auto context = [&](const f32x16& p) {      // O += P . V for one tile
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 pa = bf16x8{(__bf16)p[8 * s + 0], (__bf16)p[8 * s + 1], (__bf16)p[8 * s + 2],
                                     (__bf16)p[8 * s + 3], (__bf16)p[8 * s + 4], (__bf16)p[8 * s + 5],
                                     (__bf16)p[8 * s + 6], (__bf16)p[8 * s + 7]};
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const bf16x8 vb = bf16x8{(__bf16)vv[c][s][0], (__bf16)vv[c][s][1], (__bf16)vv[c][s][2],
                                         (__bf16)vv[c][s][3], (__bf16)vv[c][s][4], (__bf16)vv[c][s][5],
                                         (__bf16)vv[c][s][6], (__bf16)vv[c][s][7]};
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, O[c], 0, 0, 0);
            }
        }
    };
    if constexpr (ONE_TILE) {
        // N <= 32*NW: one key tile per wave (a wave past the last tile runs fully masked).  Straight line: one round
        // of loads (Q, K, V all requested up front), scores once, two barriers, one round of stores.
        load_row_frags(kf, kbase, ld, wave * KT + n, N, hd, h2, 1.f);
        load_v(wave);
        f32x16 x;
        scores(wave, x);
        float m = NEG_BIG, l = 0.f;
        tile_ml(x, m, l);
        const float lse2 = merge_lse(m, l);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - lse2);
        mask_v(wave);
        context(x);
        if (lse_out != nullptr && wave == 0 && h2 == 0 && q0 + n < N) lse_out[(long)head * N + q0 + n] = lse2;
    } else {
        // pass 1: per-wave (m, l) over its key tiles; pass 2 recomputes the scores
        float m = NEG_BIG, l = 0.f;
        for (int t = wave; t < ntiles; t += NW) {
            load_row_frags(kf, kbase, ld, t * KT + n, N, hd, h2, 1.f);
            f32x16 x;
            scores(t, x);
            tile_ml(x, m, l);
        }
        const float lse2 = merge_lse(m, l);
        for (int t = wave; t < ntiles; t += NW) {
            load_v(t);
            load_row_frags(kf, kbase, ld, t * KT + n, N, hd, h2, 1.f);
            f32x16 x;
            scores(t, x);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - lse2);
            mask_v(t);
            context(x);
        }
        if (lse_out != nullptr && wave == 0 && h2 == 0 && q0 + n < N) lse_out[(long)head * N + q0 + n] = lse2;
    }
    // ---- sum the partial O through LDS; O[c][r] is query row (r&3) + 8*(r>>2) + 4*h2, column 32c + n ----
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            s_o[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * DP + 32 * c + n] = O[c][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 1024 / (NW * 64); ++i) {
        const int idx = tid + i * NW * 64;      // 1024 float4 = 32 rows x 32 chunks
        const int row = idx >> 5, col = (idx & 31) * 4;
        float4 acc = *reinterpret_cast<const float4*>(&s_o[row * DP + col]);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * DP + col]);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (q0 + row < N && col < hd) *reinterpret_cast<float4*>(&attn_out[(long)(q0 + row) * d + head * hd + col]) = acc;
    }
}

// =====================================================================================================
// Backward of the per-head core (bf16 policy), flash-style: P is RECOMPUTED per tile from Q, K and the forward's row
// log-sum-exp -- nothing of size [H, N, N] is read or written.
//   D_i  = sum_c dA[i,c] * O[i,c]                  (mha_rowdot_kernel, per head)
//   S    = Q K^T * scale ;  P = 2^(S*log2e - lse2) ;  dP = dA_h V_h^T ;  dS = P o (dP - D) * scale
//   dQ   = dS K ;  dK = dS^T Q ;  dV = P^T dA_h
// Same shape as the forward: operands go global -> registers -> MFMA directly, a workgroup's waves split the
// reduction tiles and their partial results are summed through LDS in a fixed order (no atomics: bitwise reproducible).
// One launch, two roles (blockIdx.z):
//   ROLE_Q  : workgroup = (head, 32 queries); loop over key tiles; the tiles X[key, q] = V_tile . dA_blk^T and
//             S[key, q] = K_tile . Qs_blk^T have the query on the lane (as the forward's score tile); dQ += dS . K_tile.
//   ROLE_KV : workgroup = (head, 32 keys);    loop over query tiles; the tiles are built transposed,
//             X'[q, key] = dA_tile . V_blk^T and S'[q, key] = Q_tile . Ks_blk^T, so that the key sits on the lane:
//             dV += P'-as-A . dA_tile and dK += dS'-as-A . Q_tile use them as MFMA A operands [key, q] as they stand.
// The k index of the second product enumerates the tile's rows in the order the tile registers hold them
// (register 8s + j <-> row 16s + 8*(j>>2) + 4*h2 + (j&3)), as in the forward.
constexpr int BW = 4;                         // waves per workgroup in the backward (512 registers per lane each)

__global__ __launch_bounds__(256) void mha_rowdot_kernel(const float* __restrict__ dA, const float* __restrict__ O,
                                                         float* __restrict__ D, int N, int d, int H) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // item = head * N + row
    if (item >= N * H) return;
    const int head = item / N, row = item - head * N, hd = d / H;
    const float* a = dA + (long)row * d + head * hd;
    const float* o = O + (long)row * d + head * hd;
    float sacc = 0.f;
    for (int c = lane; c < hd; c += 64) sacc = fmaf(a[c], o[c], sacc);
    sacc = wave_sum(sacc);
    if (lane == 0) D[item] = sacc;
}

// B operand of the second product from a row-major [rows, ld] matrix: lane (col n, h2) takes, for k-step s, the 8 rows
// row0 + 16s + 8*(j>>2) + 4*h2 + (j&3) of column col0 + 32c + n (every load instruction covers two 128-B row segments)
__device__ __forceinline__ void load_strided(float (&vv)[NCT][2][8], const float* __restrict__ base, long ld, int row0,
                                             int N, int hd, int n, int h2) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = row0 + 16 * s2 + 8 * (j >> 2) + 4 * h2 + (j & 3), col = 32 * c + n;
                vv[c][s2][j] = base[(long)min(row, N - 1) * ld + min(col, hd - 1)];
            }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = row0 + 16 * s2 + 8 * (j >> 2) + 4 * h2 + (j & 3), col = 32 * c + n;
                asm volatile("" : "+v"(vv[c][s2][j]));     // keep the loads unconditional (see the forward)
                vv[c][s2][j] = (row < N && col < hd) ? vv[c][s2][j] : 0.f;
            }
}

// acc[c] += A(tile registers t, as [lane-row, k]) . B(vv)
__device__ __forceinline__ void tile_product(f32x16 (&acc)[NCT], const f32x16& t, const float (&vv)[NCT][2][8]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pa = bf16x8{(__bf16)t[8 * s2 + 0], (__bf16)t[8 * s2 + 1], (__bf16)t[8 * s2 + 2], (__bf16)t[8 * s2 + 3],
                                 (__bf16)t[8 * s2 + 4], (__bf16)t[8 * s2 + 5], (__bf16)t[8 * s2 + 6], (__bf16)t[8 * s2 + 7]};
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const bf16x8 vb = bf16x8{(__bf16)vv[c][s2][0], (__bf16)vv[c][s2][1], (__bf16)vv[c][s2][2], (__bf16)vv[c][s2][3],
                                     (__bf16)vv[c][s2][4], (__bf16)vv[c][s2][5], (__bf16)vv[c][s2][6], (__bf16)vv[c][s2][7]};
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, acc[c], 0, 0, 0);
        }
    }
}

// sum the BW waves' accumulators through LDS and store rows [r0, r0+32) x hd of `out` (row stride ld, column offset in `out`)
__device__ __forceinline__ void reduce_store(float* s_o, const f32x16 (&acc)[NCT], float* __restrict__ out, long ld, int r0,
                                             int N, int hd, int tid, int wave, int n, int h2) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_o[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * DP + 32 * c + n] = acc[c][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 1024 / (BW * 64); ++i) {
        const int idx = tid + i * BW * 64;
        const int row = idx >> 5, col = (idx & 31) * 4;
        float4 a = *reinterpret_cast<const float4*>(&s_o[row * DP + col]);
#pragma unroll
        for (int w = 1; w < BW; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * DP + col]);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (r0 + row < N && col < hd) *reinterpret_cast<float4*>(&out[(long)(r0 + row) * ld + col]) = a;
    }
    __syncthreads();
}

template <bool ROLE_KV>
__device__ __forceinline__ void mha_core_bwd_role(float* s_o, const float* __restrict__ qkv, const float* __restrict__ lse,
                                                  const float* __restrict__ dA, const float* __restrict__ Dv,
                                                  float* __restrict__ dqkv, int N, int d, int H) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, hd = d / H;
    const int b0 = blockIdx.x * 32;                         // this workgroup's 32 queries (ROLE_Q) or keys (ROLE_KV)
    const long ld = 3L * d;
    const float* qb = qkv + head * hd;
    const float* kb = qkv + d + head * hd;
    const float* vb = qkv + 2 * d + head * hd;
    const float* dab = dA + head * hd;
    const float* Lh = lse + (long)head * N;
    const float* Dh = Dv + (long)head * N;
    const float scale = 1.0f / sqrtf((float)hd);
    const float scale_log2 = scale * 1.4426950408889634f;
    const int ntiles = (N + KT - 1) / KT;

    // resident B fragments of the block: ROLE_Q: dA rows and (pre-scaled) Q rows; ROLE_KV: V rows and K rows
    bf16x8 res[KS], sres[KS];
    if constexpr (!ROLE_KV) {
        load_row_frags(res, dab, d, b0 + n, N, hd, h2, 1.f);
        load_row_frags(sres, qb, ld, b0 + n, N, hd, h2, scale_log2);
    } else {
        load_row_frags(res, vb, ld, b0 + n, N, hd, h2, 1.f);
        load_row_frags(sres, kb, ld, b0 + n, N, hd, h2, 1.f);          // (the scale rides on Q in both roles: same bf16 roundings
    }                                                                   //  as the forward, so the recomputed P is the forward's P)
    f32x16 acc0[NCT], acc1[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[c][r] = 0.f; acc1[c][r] = 0.f; }
    const bool lane_ok = b0 + n < N;                                        // the block element on this lane exists
    const float Dn = (!ROLE_KV && lane_ok) ? Dh[b0 + n] : 0.f;              // ROLE_Q: D and lse of the lane's query
    const float Ln = (!ROLE_KV && lane_ok) ? Lh[b0 + n] : 0.f;

    for (int t = wave; t < ntiles; t += BW) {
        const int t0 = t * KT;
        // X = (rows of the tile as A) . (block as B): ROLE_Q: V_tile . dA_blk^T = dP[key, q] ; ROLE_KV: dA_tile . V_blk^T
        // S = likewise:                               ROLE_Q: K_tile . Qs_blk^T          ; ROLE_KV: Q_tile . Ks_blk^T
        bf16x8 af[KS];
        f32x16 x, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r] = 0.f; sc[r] = 0.f; }
        if constexpr (!ROLE_KV) load_row_frags(af, vb, ld, t0 + n, N, hd, h2, 1.f);
        else load_row_frags(af, dab, d, t0 + n, N, hd, h2, 1.f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], res[ks], x, 0, 0, 0);
        if constexpr (!ROLE_KV) load_row_frags(af, kb, ld, t0 + n, N, hd, h2, 1.f);
        else load_row_frags(af, qb, ld, t0 + n, N, hd, h2, scale_log2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], sres[ks], sc, 0, 0, 0);
        // P of the tile (tile row on the registers, block element on the lane) from the row log-sum-exp; dS = P (dP - D) scale
        f32x16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = t0 + (r & 3) + 8 * (r >> 2) + 4 * h2;        // tile row of register r
            float l_, d_;
            if constexpr (ROLE_KV) {                                    // the query is the tile row
                l_ = Lh[min(tr, N - 1)];
                d_ = Dh[min(tr, N - 1)];
            } else {
                l_ = Ln;
                d_ = Dn;
            }
            const float pv = __builtin_amdgcn_exp2f(sc[r] - l_);
            p[r] = (tr < N && lane_ok) ? pv : 0.f;
            ds[r] = p[r] * (x[r] - d_) * scale;
        }
        float vv[NCT][2][8];
        if constexpr (!ROLE_KV) {
            load_strided(vv, kb, ld, t0, N, hd, n, h2);                  // dQ += dS . K_tile
            tile_product(acc0, ds, vv);
        } else {
            load_strided(vv, dab, d, t0, N, hd, n, h2);                  // dV += P' . dA_tile
            tile_product(acc1, p, vv);
            load_strided(vv, qb, ld, t0, N, hd, n, h2);                  // dK += dS' . Q_tile
            tile_product(acc0, ds, vv);
        }
    }
    if constexpr (!ROLE_KV) {
        reduce_store(s_o, acc0, dqkv + head * hd, ld, b0, N, hd, tid, wave, n, h2);
    } else {
        reduce_store(s_o, acc0, dqkv + d + head * hd, ld, b0, N, hd, tid, wave, n, h2);
        reduce_store(s_o, acc1, dqkv + 2 * d + head * hd, ld, b0, N, hd, tid, wave, n, h2);
    }
}

__global__ __launch_bounds__(BW * 64) void mha_core_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ lse,
                                                             const float* __restrict__ dA, const float* __restrict__ Dv,
                                                             float* __restrict__ dqkv, int N, int d, int H) {
    __shared__ __attribute__((aligned(16))) float s_o[BW * 32 * DP];
    if (blockIdx.z == 0) mha_core_bwd_role<false>(s_o, qkv, lse, dA, Dv, dqkv, N, d, H);
    else mha_core_bwd_role<true>(s_o, qkv, lse, dA, Dv, dqkv, N, d, H);
}
}  // namespace

bool mha_core_fused_supported(int N, int d, int H, int prec) {
    if (prec != MOMA_PREC_BF16 || H <= 0 || d % H) return false;
    const int hd = d / H;
    return hd % 16 == 0 && hd <= DP && d % 4 == 0 && N >= 1;
}

// D scratch: H*N floats.  dqkv [N,3d] receives dQ | dK | dV of every head.  lse: the forward's [H,N] row log-sum-exp.
hipError_t launch_mha_core_bwd(const float* qkv, const float* lse, const float* attn_out, const float* dA, float* D,
                               float* dqkv, int N, int d, int H, hipStream_t st) {
    hipLaunchKernelGGL(mha_rowdot_kernel, dim3((N * H + 3) / 4), dim3(256), 0, st, dA, attn_out, D, N, d, H);
    const dim3 grid((N + 31) / 32, H, 2), block(BW * 64);              // z = role: 0 dQ, 1 dK | dV
    hipLaunchKernelGGL(mha_core_bwd_kernel, grid, block, 0, st, qkv, lse, dA, D, dqkv, N, d, H);
    return hipGetLastError();
}

hipError_t launch_mha_core_fwd(const float* qkv, float* attn_out, float* lse, int N, int d, int H, hipStream_t st) {
    dim3 grid((N + 31) / 32, H), block(NW * 64);
    if (N <= KT * NW) hipLaunchKernelGGL((mha_core_fwd_kernel<true>), grid, block, 0, st, qkv, attn_out, lse, N, d, H);
    else hipLaunchKernelGGL((mha_core_fwd_kernel<false>), grid, block, 0, st, qkv, attn_out, lse, N, d, H);
    return hipGetLastError();
}

}  // namespace moma
