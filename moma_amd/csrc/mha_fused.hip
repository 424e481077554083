// K1 core: fused per-head batch-token attention forward (bf16 MFMA policy).
//
//   a[:, h*hd:(h+1)*hd] = softmax(q_h k_h^T * hd^-1/2) v_h          (MoMA/criterion_moco_att.py:159-163)
//
// replaces the three launches "scores GEMM -> row softmax -> context GEMM" of the staged path.  The problem is tiny
// (N = per-rank batch = 256 tokens, 4 heads) and therefore latency bound, so the shape of the kernel is chosen
// for parallelism and few dependent memory round trips, not for operand reuse:
//   * one workgroup = one head x 32 query rows; its 8 waves split the KEYS (tile t of 32 keys goes to wave t%8),
//     so N=256, H=4 already gives 32 workgroups x 8 waves with ONE key tile per wave and each K/V element is used by exactly one wave of the
//     group:
//       Q, K as B / A operands of the score product: lane (row n, half h2) reads 8 consecutive floats of its row per
//       k-step (32 B), global -> registers directly (fp32 -> bf16 on the way);
//       V as the B operand of the context product needs the keys along k, i.e. a TRANSPOSED fragment: the wave's V tile
//       (32 keys x hd) is loaded in full rows (16 x 16 B per lane, coalesced), converted to bf16 and staged in the wave's own
//       slice of LDS as the swizzled [32 keys][128 cols] image of K2 (infonce_fused.hip), then read column-wise with
//       ds_read_b64_tr_b16 (round 1: 64 strided 4-byte loads per lane).  Rows past N and columns past hd are staged as zeros.
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
    load_row_frags(qf, qbase, ld, q0 + n, N, hd, h2, scale_log2);
    // ---- V tile of this wave: global (full rows) -> registers -> bf16 -> LDS image -> transposed fragments
    //   slot i of lane L: key row 4i + (L>>4), 16-B bf16 chunk L&15 (columns 8*(L&15) .. +8) = two float4 loads
    float4 va[8][2];
    auto load_v = [&](int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = t * KT + 4 * i + (lane >> 4), col = 8 * (lane & 15);
            // unconditional loads from clamped addresses (a guarded load is sunk under its guard: branch + full wait per load)
            const float* p = vbase + (long)min(key, N - 1) * ld + min(col, hd - 8);
            va[i][0] = *reinterpret_cast<const float4*>(p);
            va[i][1] = *reinterpret_cast<const float4*>(p + 4);
        }
    };
    char* s_v = reinterpret_cast<char*>(s_o) + wave * (32 * DP * 4);          // the wave's slice (its partial O goes there later)
    const unsigned lds_v = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)s_v;
    auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
    auto stage_v = [&](int t) {                  // rows past N / columns past hd become zeros
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (lane >> 4), ch = lane & 15;
            const unsigned keep = (t * KT + row < N && 8 * ch < hd) ? 0xffffffffu : 0u;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = __builtin_bit_cast(u32x4, cvt8(va[i][0], va[i][1], 1.f));
            w &= keep;                                                        // (loads unconditional from clamped addresses)
            const unsigned addr = lds_v + row * 256 + ((ch ^ swz(row)) << 4);
            asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(w) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the wave's own writes, in order before its reads
    };
    // transposed-read lane offsets of the image (as the P.K product of K2): 16-lane group -> 4 keys x 16 columns
    unsigned b_off;
    {
        const int i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
        const int e = (2 * g2 + (p >> 1)) ^ h2;
        b_off = lds_v + (4 * h2 + q4) * 256 + 8 * (p & 1) + (((q4 << 2) | e) << 4);
    }
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
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            // the 8 transposed reads of a k-step and their wait are ONE statement with early-clobber outputs: an asm load's
            // destination counts as written at the end of its statement, so with the wait in a later statement hipcc would be
            // free to copy a destination register before the data has landed
            static_assert(NCT == 4, "four column tiles per k-step");
            s16x4 kb[NCT][2];
            const unsigned a00 = b_off, a01 = a00 ^ 32u, a10 = b_off ^ 64u, a11 = a10 ^ 32u;
            const unsigned a20 = b_off ^ 128u, a21 = a20 ^ 32u, a30 = b_off ^ 192u, a31 = a30 ^ 32u;
            if (s == 0)
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:2048\n\t"
                             "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11 offset:2048\n\t"
                             "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13 offset:2048\n\t"
                             "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(kb[0][0]), "=&v"(kb[0][1]), "=&v"(kb[1][0]), "=&v"(kb[1][1]), "=&v"(kb[2][0]), "=&v"(kb[2][1]),
                               "=&v"(kb[3][0]), "=&v"(kb[3][1])
                             : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31) : "memory");
            else
                asm volatile("ds_read_b64_tr_b16 %0, %8 offset:4096\n\tds_read_b64_tr_b16 %1, %9 offset:6144\n\t"
                             "ds_read_b64_tr_b16 %2, %10 offset:4096\n\tds_read_b64_tr_b16 %3, %11 offset:6144\n\t"
                             "ds_read_b64_tr_b16 %4, %12 offset:4096\n\tds_read_b64_tr_b16 %5, %13 offset:6144\n\t"
                             "ds_read_b64_tr_b16 %6, %14 offset:4096\n\tds_read_b64_tr_b16 %7, %15 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(kb[0][0]), "=&v"(kb[0][1]), "=&v"(kb[1][0]), "=&v"(kb[1][1]), "=&v"(kb[2][0]), "=&v"(kb[2][1]),
                               "=&v"(kb[3][0]), "=&v"(kb[3][1])
                             : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31) : "memory");
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const s16x8 vb = __builtin_shufflevector(kb[c][0], kb[c][1], 0, 1, 2, 3, 4, 5, 6, 7);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8, vb), O[c], 0, 0, 0);
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
        stage_v(wave);
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
            stage_v(t);
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
