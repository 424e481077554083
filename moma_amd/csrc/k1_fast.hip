// K1 fast path: batch-token multi-head attention (MoMA/criterion_moco_att.py:153-167) under the bf16 policy for head dims that
// are multiples of 16: up to 128 -- every `--head mlp` configuration -- at any N, wider heads (`--head None`: 1280 / 4 = 320) at
// N <= 256.  The problem is tiny (N = 256 tokens, d = 512: 0.67 GFLOP
// per module forward) and therefore bound by launches and by the bytes each compute unit can pull from L2 (60-70 GB/s per CU),
// not by the matrix pipe.  The path is built around that:
//   * every operand a kernel reads more than once per launch is stored as bf16 (weights: a pack refreshed when the optimizer
//     changed them -- k1_pack_kernel; activations between the launches of a module: qkv, attn_out, dA, dqkv), half the bytes of
//     round 2's fp32 intermediates; the arithmetic is unchanged (those kernels rounded the same values to bf16 at fragment load);
//   * operands reach the MFMA through LDS in FULL LINES: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, the bank
//     swizzle applied on the per-lane source address) into the [32 rows][256 B] image of K2 (infonce_fused.hip), which serves row
//     reads (ds_read_b128: A / B fragments with k contiguous) and transposed reads (ds_read_b64_tr_b16: B fragments with k along
//     the tile rows) alike -- round 2 loaded fragment-shaped 32-B pieces of 32 rows per instruction, and the backward core
//     gathered its transposed operands with 4-byte loads;
//   * launches are GROUPED: one launch runs several independent products (k1_gemm_kernel takes a job list: dA | dWproj, then
//     dWqkv | dx) or several modules (the forward of atts_k and atts_queue together: blockIdx.z = module).
// Forward of a module: qkv linear -> core -> proj linear (3 launches; n modules at once cost the same 3).  Backward: {dA (+ the
// row dots D = rowsum(dA o a) as partial sums per 16 columns), dWproj, dbproj} -> core (dQ | dK | dV from the forward's row
// log-sum-exp, flash-style) -> {dWqkv, dbqkv, dx}: 3 launches.  No atomics: every output is one fixed-order sum.
#include "common.hpp"
#include <mutex>
#include <type_traits>

namespace moma {
namespace {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// ---- the LDS image of a [32 rows][<= 128 bf16] tile: off(row, ch) = row*256 + 16*(ch ^ swz(row)), ch = 16-B chunk 0..15 -------
// LDS-DMA of rows [row0, row0 + 32) x columns [col0, col0 + 8*nch) of a row-major bf16 matrix (row pitch ld elements) into `img`:
// piece p = rows 4p .. 4p+3 (1 KiB, lane L lands at +16 L).  Rows past nrows and chunks past nch are CLAMPED to the last valid
// one (finite data that no product uses: k-steps past the valid chunks are skipped, rows past the end are masked by the caller).
__device__ __forceinline__ void dma_piece(const bf16_raw* __restrict__ base, long ld, int row0, int nrows, int col0, int nch,
                                          char* img, int p, int lane) {
    const int row = 4 * p + (lane >> 4);
    const int ch = min((lane & 15) ^ swz(row), nch - 1);
    const int gr = min(row0 + row, nrows - 1);
    const char* src = reinterpret_cast<const char*>(base + (long)gr * ld + col0) + ch * 16;
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(img + p * 1024), 16, 0, 0);
}
__device__ __forceinline__ void dma_tile32(const bf16_raw* __restrict__ base, long ld, int row0, int nrows, int col0, int nch,
                                           char* img, int lane) {
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_piece(base, ld, row0, nrows, col0, nch, img, p, lane);
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// row read: the 32x32x16 A / B fragment of k-step ks for lane (row n, half h): 8 consecutive k of the lane's row
__device__ __forceinline__ bf16x8 row_frag(const char* img, int n, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(img + n * 256 + 16 * ((2 * ks + h) ^ swz(n)));
}
// transposed read: B fragment of k-step s (s = 0, 1) for column tile c with k along the tile ROWS in the order the 32x32 score
// registers enumerate them: element j of lane half h <-> tile row 16 s + 8 (j >> 2) + 4 h + (j & 3).
//   boff = tr_lane_offset(lane); the two reads of a fragment sit 8 rows (2048 B) apart with the swizzle's bit 1 flipped.
__device__ __forceinline__ unsigned tr_lane_offset(int lane) {
    const int h2 = lane >> 5, i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
    const int e = (2 * g2 + (p >> 1)) ^ h2;
    return (unsigned)((4 * h2 + q4) * 256 + 8 * (p & 1) + (((q4 << 2) | e) << 4));
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
// all column tiles of one k-step in ONE statement with its wait (an asm load's destination counts as written at the end of its
// statement: with the wait in a later statement hipcc may copy a destination before the data has landed)
__device__ __forceinline__ void tr_frags4(unsigned img_lds, unsigned boff, int s, bf16x8 (&f)[4]) {
    s16x4 kb[4][2];
    const unsigned a0 = img_lds + boff + (unsigned)s * 4096u;
    const unsigned a00 = a0, a01 = a0 ^ 32u, a10 = a0 ^ 64u, a11 = a10 ^ 32u, a20 = a0 ^ 128u, a21 = a20 ^ 32u, a30 = a0 ^ 192u,
                   a31 = a30 ^ 32u;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:2048\n\t"
                 "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11 offset:2048\n\t"
                 "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13 offset:2048\n\t"
                 "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(kb[0][0]), "=&v"(kb[0][1]), "=&v"(kb[1][0]), "=&v"(kb[1][1]), "=&v"(kb[2][0]), "=&v"(kb[2][1]),
                   "=&v"(kb[3][0]), "=&v"(kb[3][1])
                 : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(a20), "v"(a21), "v"(a30), "v"(a31)
                 : "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c)
        f[c] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(kb[c][0], kb[c][1], 0, 1, 2, 3, 4, 5, 6, 7));
}
// the 32x32 score-tile registers 8 s .. 8 s + 7 as the A fragment of k-step s
__device__ __forceinline__ bf16x8 tile_as_a(const f32x16& t, int s) {
    return bf16x8{(__bf16)t[8 * s + 0], (__bf16)t[8 * s + 1], (__bf16)t[8 * s + 2], (__bf16)t[8 * s + 3],
                  (__bf16)t[8 * s + 4], (__bf16)t[8 * s + 5], (__bf16)t[8 * s + 6], (__bf16)t[8 * s + 7]};
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    return __builtin_bit_cast(unsigned, bf16x2{(__bf16)lo, (__bf16)hi});
}

// =====================================================================================================================
// Grouped small products of the module (the "linears" and their gradients), one 256-thread workgroup per output tile, the four
// waves split K, partial tiles summed through LDS in a fixed order.
//   KC job: C[m, n] = sum_k A[m, k] * B[n, k]  -- both operands K-contiguous (x . W^T, dy . Wproj, dqkv . Wqkv with transposed
//           weight copies): 32 x 32 tile; wave w takes the 128-wide K segments w, w + 4, ...; B (bf16) by LDS-DMA, A by LDS-DMA
//           (bf16) or by full-row 16-B loads converted on the way into the same image (fp32: x, dy).
//   KS job: C[m, n] = sum_k A[k, m] * B[k, n]  -- both operands token-major (dW = dY^T X): 64 x 64 tile; a lane loads TWO adjacent
//           columns per token (one dword of bf16, or 8 B of fp32) so that every load instruction covers whole 128-B / 256-B
//           segments, and feeds two MFMA row / column sets (even and odd columns) from them.
// =====================================================================================================================
struct K1Job {
    int kind;                 // 0 = KC, 1 = KS
    int M, N, K;
    const void* A;            // KC: [M, K] fp32 (a_f32) or bf16, pitch lda;  KS: [K, M] fp32 or bf16
    const void* B;            // KC: [N, K] bf16, pitch ldb;                  KS: [K, N] fp32 (b_f32) or bf16
    long lda, ldb;
    int a_f32, b_f32;
    float* C32; long ldc32;   // nullable fp32 output
    bf16_raw* C16; long ldc16;// nullable bf16 output (KC)
    const float* bias;        // nullable [N] (KC)
    float scale; int scale_cols;   // KC: C16 columns < scale_cols are multiplied by `scale` before rounding (the Q pre-scale)
    float* dpart; const bf16_raw* R; long ldr;   // KC, nullable: dpart[(col / 16) * M + row] = sum over the 16 columns of C * R
    bf16_raw* qpack; float qpack_scale;          // KC, nullable: K2's packed-Q image of C (infonce_fused.hip: infonce_qpack_kernel)
    float* colsum;            // KS, nullable [M]: sum_k A[k, m] (the bias gradient beside dW)
    int tiles_n, tile0, tile_end;
};
struct K1Jobs {
    K1Job j[4];
    int n;
};

// NH = 1: 32 x 32 tile.  (NH = 2, a 32 x 64 tile that stages the A panel once for two B images, was measured for the qkv linear
// -- 384 tiles on 256 compute units -- and lost: 8.5 vs 8.0 us for one module, 13.1 vs 10.4 us for two grouped: 96 KiB of LDS
// leave one workgroup per compute unit, and these products live on overlapping the load round trips of co-resident workgroups.)
template <int NH>
__device__ __forceinline__ void kc_body(const K1Job& J, int tile, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int tm = tile / J.tiles_n, tn = tile - tm * J.tiles_n;
    const int m0 = tm * 32, n0 = tn * 32 * NH;
    char* imgA = smem + wave * (8192 * (1 + NH));
    char* imgB = imgA + 8192;
    f32x16 acc[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const int nseg = (J.K + 127) >> 7;
    for (int seg = wave; seg < nseg; seg += 4) {
        const int k0 = seg * 128;
        const int nks = min(8, (J.K - k0) >> 4);
#pragma unroll
        for (int u = 0; u < NH; ++u)
            dma_tile32(reinterpret_cast<const bf16_raw*>(J.B), J.ldb, n0 + 32 * u, J.N, k0, 2 * nks, imgB + 8192 * u, lane);
        if (J.a_f32) {
            // 32 rows x 128 fp32 in full rows: instruction i = rows 2i, 2i+1; lane -> floats 4 (lane & 31) .. + 3 of its row
            const float* A = reinterpret_cast<const float*>(J.A);
            float4 v[16];
            const int c4 = min(lane & 31, 4 * nks - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gr = min(m0 + 2 * i + h, J.M - 1);
                v[i] = *reinterpret_cast<const float4*>(A + (long)gr * J.lda + k0 + 4 * c4);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 2 * i + h, ch = (lane & 31) >> 1;
                uint2 w;
                w.x = pack_bf16(v[i].x, v[i].y);
                w.y = pack_bf16(v[i].z, v[i].w);
                *reinterpret_cast<uint2*>(imgA + row * 256 + 16 * (ch ^ swz(row)) + 8 * (lane & 1)) = w;
            }
        } else {
            dma_tile32(reinterpret_cast<const bf16_raw*>(J.A), J.lda, m0, J.M, k0, 2 * nks, imgA, lane);
        }
        wait_dma();
        if (nks == 8) {                                     // the whole segment: straight line, every fragment read up front
            bf16x8 fa[8], fb[NH][8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                fa[ks] = row_frag(imgA, n, h, ks);
#pragma unroll
                for (int u = 0; u < NH; ++u) fb[u][ks] = row_frag(imgB + 8192 * u, n, h, ks);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int u = 0; u < NH; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[u][ks], acc[u], 0, 0, 0);
        } else {
            for (int ks = 0; ks < nks; ++ks) {
                const bf16x8 fa = row_frag(imgA, n, h, ks);
#pragma unroll
                for (int u = 0; u < NH; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, row_frag(imgB + 8192 * u, n, h, ks), acc[u], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();                                        // every wave is done with its images: the space is reused
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][16][64]
    float* tileS = red + 4 * 16 * 64;                       // [32][36]
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        if (u > 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[u][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 4 + i;
            const float v = ((red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + red[(32 + r) * 64 + lane]) + red[(48 + r) * 64 + lane];
            tileS[(8 * (r >> 2) + 4 * h + (r & 3)) * 36 + n] = v;
        }
        __syncthreads();
        // ---- epilogue: thread -> (row, 4 columns) of the 32 x 32 half ----
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const int grow = m0 + row, gcol = n0 + 32 * u + c4;
        const bool ok = grow < J.M && gcol < J.N;
        float4 v = *reinterpret_cast<const float4*>(&tileS[row * 36 + c4]);
        if (J.bias != nullptr && ok) {
            const float4 b = *reinterpret_cast<const float4*>(J.bias + gcol);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (J.C32 != nullptr && ok) *reinterpret_cast<float4*>(J.C32 + (long)grow * J.ldc32 + gcol) = v;
        if (J.C16 != nullptr && ok) {
            const float sc = gcol < J.scale_cols ? J.scale : 1.f;
            uint2 w;
            w.x = pack_bf16(v.x * sc, v.y * sc);
            w.y = pack_bf16(v.z * sc, v.w * sc);
            *reinterpret_cast<uint2*>(J.C16 + (long)grow * J.ldc16 + gcol) = w;
        }
        if (J.dpart != nullptr) {                           // (workgroup-uniform: the shuffles below run with every lane active)
            float p = 0.f;
            if (ok) {
                const uint2 rw = *reinterpret_cast<const uint2*>(J.R + (long)grow * J.ldr + gcol);
                p = v.x * __uint_as_float(rw.x << 16) + v.y * __uint_as_float(rw.x & 0xffff0000u) +
                    v.z * __uint_as_float(rw.y << 16) + v.w * __uint_as_float(rw.y & 0xffff0000u);
            }
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            if (ok && (tid & 3) == 0) J.dpart[(long)(gcol >> 4) * J.M + grow] = p;
        }
        if (J.qpack != nullptr && ok) {
            // qpack[((row_tile * KS + ks) * 64 + lane')] = 8 bf16 = C[32 row_tile + (lane' & 31)][16 ks + 8 (lane' >> 5) + 0..7] * scale
            const int KS = J.N >> 4;
            const long unit = ((long)(grow >> 5) * KS + (gcol >> 4)) * 64 + ((gcol >> 3) & 1) * 32 + (grow & 31);
            uint2 w;
            w.x = pack_bf16(v.x * J.qpack_scale, v.y * J.qpack_scale);
            w.y = pack_bf16(v.z * J.qpack_scale, v.w * J.qpack_scale);
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(J.qpack) + unit * 16 + (gcol & 4) * 2) = w;
        }
    }
}

// two adjacent columns of token `tok` as a packed bf16 pair (lo = even column).  (The bias gradients below sum these rounded
// values; summing an fp32 operand before the rounding was measured: +1.4 us on the dA | dWproj launch and +2.3 us on the
// dWqkv | dx launch -- the extra live registers break up the batch of loads -- for a difference far below the policy's tolerance.)
template <bool F32>
__device__ __forceinline__ unsigned ks_load_pair(const void* base, long ld, int tok, int col) {
    if constexpr (F32) {
        const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(base) + (long)tok * ld + col);
        return pack_bf16(v.x, v.y);
    }
    return *reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16_raw*>(base) + (long)tok * ld + col);
}

template <bool AF32, bool BF32>
__device__ __forceinline__ void ks_body(const K1Job& J, int tile, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int tm = tile / J.tiles_n, tn = tile - tm * J.tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int steps = (J.K + 15) >> 4;
    const int base = steps >> 2, rem = steps & 3;
    const int sb = wave * base + min(wave, rem), ns = base + (wave < rem ? 1 : 0);
    const int ma = min(m0 + 2 * c, J.M - 2), nb = min(n0 + 2 * c, J.N - 2);      // clamped: masked at the store
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool want_cs = J.colsum != nullptr && tn == 0;
    float cs0 = 0.f, cs1 = 0.f;
    constexpr int CH = 4;                                   // k-steps in flight
    auto step = [&](const unsigned (&da)[8], const unsigned (&db)[8]) {
        if (want_cs) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cs0 += __uint_as_float(da[j] << 16);
                cs1 += __uint_as_float(da[j] & 0xffff0000u);
            }
        }
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 ae, ao, be, bo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ae[w] = __builtin_amdgcn_perm(da[2 * w + 1], da[2 * w], 0x05040100u);
            ao[w] = __builtin_amdgcn_perm(da[2 * w + 1], da[2 * w], 0x07060302u);
            be[w] = __builtin_amdgcn_perm(db[2 * w + 1], db[2 * w], 0x05040100u);
            bo[w] = __builtin_amdgcn_perm(db[2 * w + 1], db[2 * w], 0x07060302u);
        }
        const bf16x8 fae = __builtin_bit_cast(bf16x8, ae), fao = __builtin_bit_cast(bf16x8, ao);
        const bf16x8 fbe = __builtin_bit_cast(bf16x8, be), fbo = __builtin_bit_cast(bf16x8, bo);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fae, fbe, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fae, fbo, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fao, fbe, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fao, fbo, acc[1][1], 0, 0, 0);
    };
    int c0 = 0;
    for (; c0 + CH <= ns && (sb + c0 + CH) * 16 <= J.K; c0 += CH) {     // whole chunks of whole k-steps: straight line
        unsigned da[CH][8], db[CH][8];
#pragma unroll
        for (int s = 0; s < CH; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tok = (sb + c0 + s) * 16 + 8 * h + j;
                da[s][j] = ks_load_pair<AF32>(J.A, J.lda, tok, ma);
                db[s][j] = ks_load_pair<BF32>(J.B, J.ldb, tok, nb);
            }
#pragma unroll
        for (int s = 0; s < CH; ++s) step(da[s], db[s]);
    }
    for (; c0 < ns; ++c0) {                                  // the rest, one k-step at a time; tokens past K contribute zero
        unsigned da[8], db[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tok = min((sb + c0) * 16 + 8 * h + j, J.K - 1);
            da[j] = ks_load_pair<AF32>(J.A, J.lda, tok, ma);
            db[j] = ks_load_pair<BF32>(J.B, J.ldb, tok, nb);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = (sb + c0) * 16 + 8 * h + j < J.K;
            da[j] = live ? da[j] : 0u;
            db[j] = live ? db[j] : 0u;
        }
        step(da, db);
    }
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][4 acc][16][64] = 64 KiB
    if (want_cs) {                                           // column sums first (the space is reused below)
        cs0 += other_half(cs0);
        cs1 += other_half(cs1);
        if (h == 0) { red[(wave * 32 + c) * 2] = cs0; red[(wave * 32 + c) * 2 + 1] = cs1; }
        __syncthreads();
        if (tid < 64) {                                      // thread -> column m0 + tid: pair tid >> 1, member tid & 1
            const int pc = tid >> 1, mem = tid & 1;
            const float v = ((red[(pc) * 2 + mem] + red[(32 + pc) * 2 + mem]) + red[(64 + pc) * 2 + mem]) + red[(96 + pc) * 2 + mem];
            if (m0 + tid < J.M) J.colsum[m0 + tid] = v;
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * 4 + a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 4 + i;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int q = (a * 2 + b) * 16 + r;
                v[b] = ((red[q * 64 + lane] + red[(64 + q) * 64 + lane]) + red[(128 + q) * 64 + lane]) + red[(192 + q) * 64 + lane];
            }
            const int row = m0 + 2 * (8 * (r >> 2) + 4 * h + (r & 3)) + a, col = n0 + 2 * c;
            if (row < J.M && col < J.N) *reinterpret_cast<float2*>(J.C32 + (long)row * J.ldc32 + col) = make_float2(v[0], v[1]);
        }
    }
}

__global__ __launch_bounds__(256, 2) void k1_gemm_kernel(K1Jobs jobs) {     // two workgroups per compute unit (registers, LDS)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB
    const int id = blockIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < jobs.n && id >= jobs.j[i].tile0 && id < jobs.j[i].tile_end) {
            const K1Job& J = jobs.j[i];
            const int t = id - J.tile0;
            if (J.kind == 0) {
                kc_body<1>(J, t, smem);
            } else if (J.a_f32 && !J.b_f32) ks_body<true, false>(J, t, smem);      // dWproj = dy^T a
            else if (!J.a_f32 && J.b_f32) ks_body<false, true>(J, t, smem);        // dWqkv = dqkv^T x
            else if (J.a_f32) ks_body<true, true>(J, t, smem);
            else ks_body<false, false>(J, t, smem);
            return;
        }
    }
}

// ---- bf16 weight pack: [Wqkv (3d x d) | Wproj (d x d) | Wqkv^T (d x 3d) | Wproj^T (d x d)], the transposed halves only when
// the module runs a backward.  One 32 x 32 tile per workgroup (the transposition goes through LDS).
__global__ __launch_bounds__(256) void k1_pack_kernel(const float* __restrict__ w_qkv, const float* __restrict__ w_proj,
                                                      bf16_raw* __restrict__ pack, int d, int with_t) {
    __shared__ float t[32][33];
    const int tiles_c = (d + 31) / 32, tiles_qkv = ((3 * d + 31) / 32) * tiles_c;
    int id = blockIdx.x;
    const bool is_proj = id >= tiles_qkv;
    if (is_proj) id -= tiles_qkv;
    const float* W = is_proj ? w_proj : w_qkv;
    const int rows = is_proj ? d : 3 * d;
    bf16_raw* out = pack + (is_proj ? 3L * d * d : 0);
    bf16_raw* out_t = pack + 4L * d * d + (is_proj ? 3L * d * d : 0);
    const int r0 = (id / tiles_c) * 32, c0 = (id % tiles_c) * 32;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 4;
    // d % 16 == 0 (checked by the entry point): a thread's 4 columns are all inside or all outside, rows are guarded
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + tr < rows && c0 + tc < d) {
        v = *reinterpret_cast<const float4*>(W + (long)(r0 + tr) * d + c0 + tc);
        uint2 w;
        w.x = pack_bf16(v.x, v.y);
        w.y = pack_bf16(v.z, v.w);
        *reinterpret_cast<uint2*>(out + (long)(r0 + tr) * d + c0 + tc) = w;
    }
    if (!with_t) return;
    t[tr][tc] = v.x; t[tr][tc + 1] = v.y; t[tr][tc + 2] = v.z; t[tr][tc + 3] = v.w;
    __syncthreads();
    // out_t[c][r] = W[r][c]: thread -> column c0 + tr, rows r0 + tc .. + 3 (rows % 16 == 0: all four inside or outside)
    if (c0 + tr < d && r0 + tc < rows) {
        uint2 w;
        w.x = pack_bf16(t[tc][tr], t[tc + 1][tr]);
        w.y = pack_bf16(t[tc + 2][tr], t[tc + 3][tr]);
        *reinterpret_cast<uint2*>(out_t + (long)(c0 + tr) * rows + r0 + tc) = w;
    }
}

// =====================================================================================================================
// Per-head core, forward:  a[:, head] = softmax(q_h k_h^T * hd^-1/2) v_h  from the bf16 qkv of the qkv linear (Q pre-scaled by
// hd^-1/2 log2 e), one workgroup per (32 queries, head, module), its 8 waves split the keys; keeps the row log-sum-exp.
// =====================================================================================================================
constexpr int KT = 32;
constexpr int NW = 8;
constexpr float NEG_BIG = -1.0e30f;
constexpr int CORE_LDS = 8192 + NW * 16384 + 2048;        // Q image | per wave: K image, V image (later its partial O) | (m, l)

struct CoreMod {
    const bf16_raw* qkv;      // [N, 3d] bf16, Q pre-scaled
    bf16_raw* out;            // [N, d] bf16
    float* lse;               // nullable [H, N]
};
struct CoreFwdArgs {
    CoreMod m[4];
    int N, d, H;
};

// Workgroup -> (32-row block, head, z = module / role) on a 1-D grid.  Workgroups are dealt round-robin over the 8 XCDs (ids b and
// b + 8 share one), each with its own L2, and what the workgroups of a core launch re-read is per (head, z): the head's K / V (or
// dA / Q) columns of qkv16.  With the plain order (32-row block fastest) the 8 row blocks of a head land on 8 different XCDs and every
// XCD pulls every head's operands through its own cold L2 (round 4 counters: 6.0 MB fetched for 0.77 MB of distinct bytes, 10 % L2
// hits).  Here XCD x takes a CONTIGUOUS range of the (z, head, block) order -- at N = 256, H = 4: one or two (head, z) pairs per
// XCD.  Speed only: any placement is correct.  (total % 8 != 0: the plain order.)
struct CoreBlock { int blk, head, z; };
__device__ __forceinline__ CoreBlock core_block(int N, int H) {
    const int nb = (N + 31) >> 5, total = gridDim.x;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
    CoreBlock c;
    c.blk = L % nb;
    const int g = L / nb;
    c.head = g % H;
    c.z = g / H;
    return c;
}

// FULL: head dim 128 (every k-step and column tile of the images is live: no guards in the MFMA chains)
template <bool ONE_TILE, bool FULL>
__global__ __launch_bounds__(NW * 64) void k1_core_fwd_kernel(CoreFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, d = a.d, hd = d / a.H;
    const CoreBlock cb = core_block(N, a.H);
    const CoreMod M = a.m[cb.z];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h2 = lane >> 5;
    const int head = cb.head, q0 = cb.blk * 32;
    const long ld = 3L * d;
    const int nch = FULL ? 16 : hd >> 3, kse = FULL ? 8 : hd >> 4, nct = FULL ? 4 : (hd + 31) >> 5;
    const bf16_raw* qb = M.qkv + head * hd;
    const bf16_raw* kb = M.qkv + d + head * hd;
    const bf16_raw* vb = M.qkv + 2 * d + head * hd;
    char* imgQ = smem;
    char* imgK = smem + 8192 + wave * 16384;
    char* imgV = imgK + 8192;
    float* s_ml = reinterpret_cast<float*>(smem + 8192 + NW * 16384);
    const int ntiles = (N + KT - 1) / KT;
    const unsigned boff = tr_lane_offset(lane);

    auto scores = [&](int t, f32x16& x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < kse) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(imgK, n, h2, ks), row_frag(imgQ, n, h2, ks), x, 0, 0, 0);
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
        tmax = fmaxf(tmax, other_half(tmax));
        const float mn = fmaxf(m, tmax);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(x[r] - mn);
        ps += other_half(ps);
        l = l * __builtin_amdgcn_exp2f(m - mn) + ps;
        m = mn;
    };
    auto merge_lse = [&](float m, float l) -> float {      // contains the workgroup barrier
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
    f32x16 O[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    auto context = [&](const f32x16& p) {      // O += P . V for one tile
        const unsigned vl = lds_addr(imgV);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 vf[4];
            tr_frags4(vl, boff, s, vf);
            const bf16x8 pa = tile_as_a(p, s);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nct) O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vf[c], O[c], 0, 0, 0);
        }
    };

    dma_piece(qb, ld, q0, N, 0, nch, imgQ, wave, lane);        // the query block: one piece per wave
    float lse2;
    if constexpr (ONE_TILE) {
        // N <= 32 * NW: one key tile per wave (a wave past the last tile runs fully masked).  One round of loads (Q, K, V all
        // requested up front), scores once, two barriers, one round of stores.
        dma_tile32(kb, ld, wave * KT, N, 0, nch, imgK, lane);
        dma_tile32(vb, ld, wave * KT, N, 0, nch, imgV, lane);
        wait_dma();
        __syncthreads();
        f32x16 x;
        scores(wave, x);
        float m = NEG_BIG, l = 0.f;
        tile_ml(x, m, l);
        lse2 = merge_lse(m, l);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - lse2);
        context(x);
    } else {
        wait_dma();
        __syncthreads();
        float m = NEG_BIG, l = 0.f;
        for (int t = wave; t < ntiles; t += NW) {           // pass 1: the row statistics
            dma_tile32(kb, ld, t * KT, N, 0, nch, imgK, lane);
            wait_dma();
            f32x16 x;
            scores(t, x);
            tile_ml(x, m, l);
            asm volatile("" ::: "memory");
        }
        lse2 = merge_lse(m, l);
        for (int t = wave; t < ntiles; t += NW) {           // pass 2 recomputes the scores
            dma_tile32(kb, ld, t * KT, N, 0, nch, imgK, lane);
            dma_tile32(vb, ld, t * KT, N, 0, nch, imgV, lane);
            wait_dma();
            f32x16 x;
            scores(t, x);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - lse2);
            context(x);
            asm volatile("" ::: "memory");
        }
    }
    if (M.lse != nullptr && wave == 0 && h2 == 0 && q0 + n < N) M.lse[(long)head * N + q0 + n] = lse2;
    // ---- sum the partial O through LDS (the wave's own slice: its tiles are consumed); O[c][r] = query (r&3)+8(r>>2)+4h2, col 32c+n
    float* s_o = reinterpret_cast<float*>(smem + 8192);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_o[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * 128 + 32 * c + n] = O[c][r];
    __syncthreads();
    {
        const int row = tid >> 4, col = (tid & 15) * 8;     // 512 threads = 32 rows x 16 chunks of 8 columns
        float4 s0 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col]);
        float4 s1 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col + 4]);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 u0 = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * 128 + col]);
            const float4 u1 = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * 128 + col + 4]);
            s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
            s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
        }
        if (q0 + row < N && col < hd) {
            uint4 w;
            w.x = pack_bf16(s0.x, s0.y); w.y = pack_bf16(s0.z, s0.w); w.z = pack_bf16(s1.x, s1.y); w.w = pack_bf16(s1.z, s1.w);
            *reinterpret_cast<uint4*>(M.out + (long)(q0 + row) * d + head * hd + col) = w;
        }
    }
}


// ---- wide heads (128 < head dim, e.g. `--head None` on EfficientNet-B0: d = 1280, 4 heads of 320), N <= 32 * NW * TW ------------
// The products are separable over 128-column SEGMENTS of the head: S = sum_seg Q_seg K_seg^T, O_seg = P V_seg.  Every wave owns up
// to TW key tiles (tiles wave, wave + NW, ...: TW = 1 up to N = 256, 2 up to 512, 4 up to 1024 -- the concatenated [q ; k] token
// sets of the MoCoAtt variants) and keeps their score tiles in registers; the segment images stream through the same two 8 KiB
// images per wave (and the one Q image of the workgroup), so the LDS budget grows neither with the head dim nor with N: phase 1
// accumulates the scores of the wave's tiles over the segments, the row statistics follow, phase 2 forms one segment of O at a
// time (4 accumulator tiles, summed over the wave's tiles) and sums it over the waves through LDS.
template <int TW>
__global__ __launch_bounds__(NW * 64) void k1_core_fwd_wide_kernel(CoreFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, d = a.d, hd = d / a.H;
    const CoreBlock cb = core_block(N, a.H);
    const CoreMod M = a.m[cb.z];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h2 = lane >> 5;
    const int head = cb.head, q0 = cb.blk * 32;
    const long ld = 3L * d;
    const int nseg = (hd + 127) >> 7;
    const bf16_raw* qb = M.qkv + head * hd;
    const bf16_raw* kb = M.qkv + d + head * hd;
    const bf16_raw* vb = M.qkv + 2 * d + head * hd;
    char* imgQ = smem;
    char* imgK = smem + 8192 + wave * 16384;
    char* imgV = imgK + 8192;
    float* s_ml = reinterpret_cast<float*>(smem + 8192 + NW * 16384);
    const unsigned boff = tr_lane_offset(lane);
    // the wave's tile j covers keys [(wave + NW j) KT, +KT); its images alternate between the wave's two buffers so that the next
    // tile's DMA runs under this tile's MFMAs (8 pieces per tile image: "all but the newest tile landed" = vmcnt(8))
    auto tile_row0 = [&](int j) { return (wave + NW * j) * KT; };
    auto wait_newest_in_flight = [&](bool more) __attribute__((always_inline)) {
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- phase 1: the wave's score tiles, accumulated over the segments
    f32x16 x[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[j][r] = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {
        const int c0 = sg * 128, w = min(128, hd - c0), nch = w >> 3, kse = w >> 4;
        if (sg) __syncthreads();                                 // every wave is done with the previous Q image
        dma_piece(qb, ld, q0, N, c0, nch, imgQ, wave, lane);
        dma_tile32(kb, ld, tile_row0(0), N, c0, nch, imgK, lane);
        if (TW == 1 && sg == 0) dma_tile32(vb, ld, tile_row0(0), N, 0, nch, imgV, lane);      // (one tile: V's first segment rides along)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            char* img = (j & 1) ? imgV : imgK;
            if (j + 1 < TW) dma_tile32(kb, ld, tile_row0(j + 1), N, c0, nch, (j & 1) ? imgK : imgV, lane);
            wait_newest_in_flight(j + 1 < TW);
            if (j == 0) __syncthreads();                         // the Q image is complete
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (ks < kse) x[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(img, n, h2, ks), row_frag(imgQ, n, h2, ks), x[j], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
    }
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        if (tile_row0(j) + KT > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (tile_row0(j) + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) x[j][r] = NEG_BIG;
        }
    }
    // ---- the row log-sum-exp over all tiles of all waves
    float lse2;
    {
        float tmax = NEG_BIG;
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, x[j][r]);
        tmax = fmaxf(tmax, other_half(tmax));
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(x[j][r] - tmax);
        ps += other_half(ps);
        if (h2 == 0) {
            s_ml[(wave * 32 + n) * 2 + 0] = tmax;
            s_ml[(wave * 32 + n) * 2 + 1] = ps;
        }
        __syncthreads();
        float mm = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, s_ml[(w * 32 + n) * 2]);
        float ll = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ll += s_ml[(w * 32 + n) * 2 + 1] * __builtin_amdgcn_exp2f(s_ml[(w * 32 + n) * 2] - mm);
        lse2 = mm + __builtin_amdgcn_logf(ll);                   // v_log_f32 = log2
    }
    if (M.lse != nullptr && wave == 0 && h2 == 0 && q0 + n < N) M.lse[(long)head * N + q0 + n] = lse2;
    bf16x8 pa[TW][2];
#pragma unroll
    for (int j = 0; j < TW; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[j][r] = __builtin_amdgcn_exp2f(x[j][r] - lse2);
        pa[j][0] = tile_as_a(x[j], 0);
        pa[j][1] = tile_as_a(x[j], 1);
    }

    // ---- phase 2: O_seg = sum over the waves (and their tiles) of P . V_seg
    float* s_o = reinterpret_cast<float*>(smem + 8192);
    for (int sg = 0; sg < nseg; ++sg) {
        const int c0 = sg * 128, w = min(128, hd - c0), nch = w >> 3, nct = (w + 31) >> 5;
        f32x16 O[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
        if (TW > 1 || sg) dma_tile32(vb, ld, tile_row0(0), N, c0, nch, imgV, lane);
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            char* img = (j & 1) ? imgK : imgV;
            if (j + 1 < TW) dma_tile32(vb, ld, tile_row0(j + 1), N, c0, nch, (j & 1) ? imgV : imgK, lane);
            wait_newest_in_flight(j + 1 < TW);
            const unsigned vl = lds_addr(img);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 vf[4];
                tr_frags4(vl, boff, s, vf);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nct) O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[j][s], vf[c], O[c], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        // the wave's slice (its own two images: consumed) takes its partial; O[c][r] = query (r&3)+8(r>>2)+4h2, column 32c+n
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_o[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * 128 + 32 * c + n] = O[c][r];
        __syncthreads();
        {
            const int row = tid >> 4, col = (tid & 15) * 8;      // 512 threads = 32 rows x 16 chunks of 8 columns
            float4 s0 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col]);
            float4 s1 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col + 4]);
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) {
                const float4 u0 = *reinterpret_cast<const float4*>(&s_o[(ww * 32 + row) * 128 + col]);
                const float4 u1 = *reinterpret_cast<const float4*>(&s_o[(ww * 32 + row) * 128 + col + 4]);
                s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
                s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
            }
            if (q0 + row < N && col < w) {
                uint4 o;
                o.x = pack_bf16(s0.x, s0.y); o.y = pack_bf16(s0.z, s0.w); o.z = pack_bf16(s1.x, s1.y); o.w = pack_bf16(s1.z, s1.w);
                *reinterpret_cast<uint4*>(M.out + (long)(q0 + row) * d + head * hd + c0 + col) = o;
            }
        }
        __syncthreads();                                         // the slices are free again
    }
}

// =====================================================================================================================
// Per-head core, backward (flash-style: P recomputed per tile from Q, K and the forward's row log-sum-exp).
//   S = Qs K^T (log2 units) ; P = 2^(S - lse2) ; dP = dA_h V_h^T ; D = rowsum(dA o a) ; dS = P o (dP - D) * scale
//   dQ = dS K ; dK = dS^T Q = (P o (dP - D))^T Qs * ln 2 ; dV = P^T dA_h
// One launch, two roles (blockIdx.z), 8 waves split the reduction tiles, partial results summed through LDS in a fixed order:
//   ROLE_Q  : workgroup = (head, 32 queries); per key tile the images V_t, K_t: X[key, q] = V_t . dA_blk^T, S[key, q] = K_t . Qs_blk^T
//             (row reads), dQ += dS . K_t (K_t read transposed).
//   ROLE_KV : workgroup = (head, 32 keys); per query tile the images dA_t, Qs_t: X'[q, key] = dA_t . V_blk^T, S'[q, key] =
//             Qs_t . K_blk^T, dV += P' . dA_t, dK += dS' . Qs_t (the tiles read transposed).
// D arrives as partial sums per 16 columns from the dA product (dpart[(col / 16) * N + row]).
// =====================================================================================================================
constexpr int BWD_LDS = 2 * 8192 + NW * 16384 + NW * 256;  // block images | per wave: two tile images (later its partials) | L, D

struct CoreBwdArgs {
    const bf16_raw* qkv;      // [N, 3d] bf16 (Q pre-scaled)
    const bf16_raw* dA;       // [N, d] bf16
    const float* lse;         // [H, N]
    const float* dpart;       // [d / 16, N]
    bf16_raw* dqkv;           // [N, 3d] bf16
    int N, d, H;
};

// sum the NW partial [32 x 128] tiles (fp32, the waves' slices) and store rows [r0, r0 + 32) x hd as bf16
__device__ __forceinline__ void reduce_store16(float* s_o, const f32x16 (&acc)[4], bf16_raw* __restrict__ out, long ld, int r0, int N,
                                               int hd, int tid, int wave, int n, int h2) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_o[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2) * 128 + 32 * c + n] = acc[c][r];
    __syncthreads();
    const int row = tid >> 4, col = (tid & 15) * 8;
    float4 s0 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col]);
    float4 s1 = *reinterpret_cast<const float4*>(&s_o[row * 128 + col + 4]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const float4 u0 = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * 128 + col]);
        const float4 u1 = *reinterpret_cast<const float4*>(&s_o[(w * 32 + row) * 128 + col + 4]);
        s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
        s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
    }
    if (r0 + row < N && col < hd) {
        uint4 w;
        w.x = pack_bf16(s0.x, s0.y); w.y = pack_bf16(s0.z, s0.w); w.z = pack_bf16(s1.x, s1.y); w.w = pack_bf16(s1.z, s1.w);
        *reinterpret_cast<uint4*>(out + (long)(r0 + row) * ld + col) = w;
    }
    __syncthreads();
}

template <bool ROLE_KV, bool FULL>
__device__ __forceinline__ void k1_core_bwd_role(char* smem, const CoreBwdArgs& a) {
    const int N = a.N, d = a.d, hd = d / a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h2 = lane >> 5;
    const CoreBlock cb = core_block(N, a.H);
    const int head = cb.head, b0 = cb.blk * 32;
    const long ld = 3L * d;
    const int nch = FULL ? 16 : hd >> 3, kse = FULL ? 8 : hd >> 4, nct = FULL ? 4 : (hd + 31) >> 5, npart = FULL ? 8 : hd >> 4;
    const bf16_raw* qb = a.qkv + head * hd;
    const bf16_raw* kb = a.qkv + d + head * hd;
    const bf16_raw* vb = a.qkv + 2 * d + head * hd;
    const bf16_raw* dab = a.dA + head * hd;
    const float* Lh = a.lse + (long)head * N;
    const float* Dp = a.dpart + (long)head * npart * N;
    const float scale = 1.0f / sqrtf((float)hd);
    const int ntiles = (N + KT - 1) / KT;
    char* imgR = smem;                      // ROLE_Q: dA_blk ; ROLE_KV: V_blk   (B operand of X)
    char* imgS = smem + 8192;               // ROLE_Q: Qs_blk ; ROLE_KV: K_blk   (B operand of S)
    char* img1 = smem + 16384 + wave * 16384;   // ROLE_Q: V_t ; ROLE_KV: dA_t
    char* img2 = img1 + 8192;                   // ROLE_Q: K_t ; ROLE_KV: Qs_t
    float* sLD = reinterpret_cast<float*>(smem + 16384 + NW * 16384) + wave * 64;    // [32] lse | [32] D of the tile rows
    const unsigned boff = tr_lane_offset(lane);

    // the block's two images: wave w loads piece w of each
    if constexpr (!ROLE_KV) {
        dma_piece(dab, d, b0, N, 0, nch, imgR, wave, lane);
        dma_piece(qb, ld, b0, N, 0, nch, imgS, wave, lane);
    } else {
        dma_piece(vb, ld, b0, N, 0, nch, imgR, wave, lane);
        dma_piece(kb, ld, b0, N, 0, nch, imgS, wave, lane);
    }
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[c][r] = 0.f; acc1[c][r] = 0.f; }
    const bool lane_ok = b0 + n < N;                            // the block element on this lane exists
    float Ln = 0.f, Dn = 0.f;                                   // ROLE_Q: lse and D of the lane's query
    if (!ROLE_KV && lane_ok) {
        Ln = Lh[b0 + n];
        for (int i = 0; i < npart; ++i) Dn += Dp[(long)i * N + b0 + n];
    }
    wait_dma();
    __syncthreads();

    for (int t = wave; t < ntiles; t += NW) {
        const int t0 = t * KT;
        if constexpr (!ROLE_KV) {
            dma_tile32(vb, ld, t0, N, 0, nch, img1, lane);
            dma_tile32(kb, ld, t0, N, 0, nch, img2, lane);
        } else {
            dma_tile32(dab, d, t0, N, 0, nch, img1, lane);
            dma_tile32(qb, ld, t0, N, 0, nch, img2, lane);
            // lse and D of the tile's queries: lane-wise (coalesced), then redistributed to the register index through LDS
            if (h2 == 0) {
                const int q = min(t0 + n, N - 1);
                float dsum = 0.f;
                for (int i = 0; i < npart; ++i) dsum += Dp[(long)i * N + q];
                sLD[n] = Lh[q];
                sLD[32 + n] = dsum;
            }
        }
        wait_dma();
        f32x16 x, sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r] = 0.f; sc[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < kse) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(img1, n, h2, ks), row_frag(imgR, n, h2, ks), x, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < kse) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(img2, n, h2, ks), row_frag(imgS, n, h2, ks), sc, 0, 0, 0);
        f32x16 p, ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 l4, d4;
            if constexpr (ROLE_KV) {                            // the query is the tile row: rows 8g + 4h2 + 0..3
                l4 = *reinterpret_cast<const float4*>(&sLD[8 * g + 4 * h2]);
                d4 = *reinterpret_cast<const float4*>(&sLD[32 + 8 * g + 4 * h2]);
            } else {
                l4 = make_float4(Ln, Ln, Ln, Ln);
                d4 = make_float4(Dn, Dn, Dn, Dn);
            }
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const int tr = t0 + 8 * g + 4 * h2 + i;         // tile row of register r
                const float pv = __builtin_amdgcn_exp2f(sc[r] - lv[i]);
                p[r] = (tr < N && lane_ok) ? pv : 0.f;
                ds[r] = p[r] * (x[r] - dv[i]) * (ROLE_KV ? 0.6931471805599453f : scale);
            }
        }
        if constexpr (!ROLE_KV) {
            const unsigned kl = lds_addr(img2);                 // dQ += dS . K_t
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 f[4];
                tr_frags4(kl, boff, s, f);
                const bf16x8 pa = tile_as_a(ds, s);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nct) acc0[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, f[c], acc0[c], 0, 0, 0);
            }
        } else {
            const unsigned al = lds_addr(img1), ql = lds_addr(img2);
#pragma unroll
            for (int s = 0; s < 2; ++s) {                        // dV += P' . dA_t
                bf16x8 f[4];
                tr_frags4(al, boff, s, f);
                const bf16x8 pa = tile_as_a(p, s);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nct) acc1[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, f[c], acc1[c], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {                        // dK += dS' . Qs_t  (ln 2 folded into dS')
                bf16x8 f[4];
                tr_frags4(ql, boff, s, f);
                const bf16x8 pa = tile_as_a(ds, s);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nct) acc0[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, f[c], acc0[c], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();                                            // all tiles consumed: the slices take the partial sums
    float* s_o = reinterpret_cast<float*>(smem + 16384);
    if constexpr (!ROLE_KV) {
        reduce_store16(s_o, acc0, a.dqkv + head * hd, ld, b0, N, hd, tid, wave, n, h2);
    } else {
        reduce_store16(s_o, acc0, a.dqkv + d + head * hd, ld, b0, N, hd, tid, wave, n, h2);
        reduce_store16(s_o, acc1, a.dqkv + 2 * d + head * hd, ld, b0, N, hd, tid, wave, n, h2);
    }
}

template <bool FULL>
__global__ __launch_bounds__(NW * 64) void k1_core_bwd_kernel(CoreBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (core_block(a.N, a.H).z == 0) k1_core_bwd_role<false, FULL>(smem, a);
    else k1_core_bwd_role<true, FULL>(smem, a);
}


// ---- wide heads, backward: the same two roles with the head's columns in 128-column segments and up to TW reduction tiles per
// wave (N <= 32 * NW * TW).  Phase 1: X and S of every tile of the wave accumulate over the segments (the block images and the
// wave's tile images are re-filled per segment and tile); P and dS follow once per tile; phase 2 forms one segment of dQ (or dK
// and dV) at a time, summed over the wave's tiles in registers and over the waves through LDS.
template <int TW> constexpr int bwd_wide_lds() { return 2 * 8192 + NW * 16384 + NW * 256 * TW; }

template <bool ROLE_KV, int TW>
__device__ __forceinline__ void k1_core_bwd_wide_role(char* smem, const CoreBwdArgs& a) {
    const int N = a.N, d = a.d, hd = d / a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h2 = lane >> 5;
    const CoreBlock cb = core_block(N, a.H);
    const int head = cb.head, b0 = cb.blk * 32;
    const long ld = 3L * d;
    const int nseg = (hd + 127) >> 7, npart = hd >> 4;
    const bf16_raw* qb = a.qkv + head * hd;
    const bf16_raw* kb = a.qkv + d + head * hd;
    const bf16_raw* vb = a.qkv + 2 * d + head * hd;
    const bf16_raw* dab = a.dA + head * hd;
    const float* Lh = a.lse + (long)head * N;
    const float* Dp = a.dpart + (long)head * npart * N;
    const float scale = 1.0f / sqrtf((float)hd);
    char* imgR = smem;                      // ROLE_Q: dA_blk ; ROLE_KV: V_blk   (B operand of X)
    char* imgS = smem + 8192;               // ROLE_Q: Qs_blk ; ROLE_KV: K_blk   (B operand of S)
    char* img1 = smem + 16384 + wave * 16384;   // ROLE_Q: V_t ; ROLE_KV: dA_t
    char* img2 = img1 + 8192;                   // ROLE_Q: K_t ; ROLE_KV: Qs_t
    float* sLD = reinterpret_cast<float*>(smem + 16384 + NW * 16384) + wave * 64 * TW;   // per tile: [32] lse | [32] D of the tile rows
    const unsigned boff = tr_lane_offset(lane);
    auto tile_row0 = [&](int j) { return (wave + NW * j) * KT; };
    const bool lane_ok = b0 + n < N;
    float Ln = 0.f, Dn = 0.f;
    if (!ROLE_KV && lane_ok) {
        Ln = Lh[b0 + n];
        for (int i = 0; i < npart; ++i) Dn += Dp[(long)i * N + b0 + n];
    }
    if (ROLE_KV && h2 == 0) {
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int q = min(tile_row0(j) + n, N - 1);
            float dsum = 0.f;
            for (int i = 0; i < npart; ++i) dsum += Dp[(long)i * N + q];
            sLD[j * 64 + n] = Lh[q];
            sLD[j * 64 + 32 + n] = dsum;
        }
    }
    // Phase 1 ONE TILE AT A TIME: the X and S tiles of the tile in work live in registers across the segments (2 x 16 registers),
    // its P / dS fragments (16 registers per tile) are what persists -- with the X and S of two or more tiles alive at once (plus
    // the row fragments hipcc requests ahead) three and four tiles per wave spill.  The block images are re-filled per tile and
    // segment (L2 hits).
    constexpr int GP = 1;
    bf16x8 dsa[TW][2], pa[TW][2];
#pragma unroll
    for (int g0 = 0; g0 < TW; g0 += GP) {
        f32x16 x[GP], sc[GP];
#pragma unroll
        for (int jj = 0; jj < GP; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[jj][r] = 0.f; sc[jj][r] = 0.f; }
        for (int sg = 0; sg < nseg; ++sg) {
            const int c0 = sg * 128, w = min(128, hd - c0), nch = w >> 3, kse = w >> 4;
            if (sg || g0) __syncthreads();                       // every wave is done with the previous block images
            if constexpr (!ROLE_KV) {
                dma_piece(dab, d, b0, N, c0, nch, imgR, wave, lane);
                dma_piece(qb, ld, b0, N, c0, nch, imgS, wave, lane);
            } else {
                dma_piece(vb, ld, b0, N, c0, nch, imgR, wave, lane);
                dma_piece(kb, ld, b0, N, c0, nch, imgS, wave, lane);
            }
#pragma unroll
            for (int jj = 0; jj < GP; ++jj) {
                if (g0 + jj < TW) {
                    int t0 = tile_row0(g0 + jj);
                    asm volatile("" : "+v"(t0));                 // (opaque per use: hipcc otherwise hoists the row offsets of every
                    if constexpr (!ROLE_KV) {                    //  tile, segment and sweep out of the loops -- 64-bit pairs, spilled)
                        dma_tile32(vb, ld, t0, N, c0, nch, img1, lane);
                        dma_tile32(kb, ld, t0, N, c0, nch, img2, lane);
                    } else {
                        dma_tile32(dab, d, t0, N, c0, nch, img1, lane);
                        dma_tile32(qb, ld, t0, N, c0, nch, img2, lane);
                    }
                    wait_dma();
                    if (jj == 0) __syncthreads();                // the block images are complete
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        if (ks < kse) x[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(img1, n, h2, ks), row_frag(imgR, n, h2, ks), x[jj], 0, 0, 0);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        if (ks < kse) sc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(img2, n, h2, ks), row_frag(imgS, n, h2, ks), sc[jj], 0, 0, 0);
                    asm volatile("" ::: "memory");
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < GP; ++jj) {
            if (g0 + jj < TW) {
                const int j = g0 + jj;
                const int t0 = tile_row0(j);
                f32x16 p, ds;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 l4, d4;
                    if constexpr (ROLE_KV) {
                        l4 = *reinterpret_cast<const float4*>(&sLD[j * 64 + 8 * g + 4 * h2]);
                        d4 = *reinterpret_cast<const float4*>(&sLD[j * 64 + 32 + 8 * g + 4 * h2]);
                    } else {
                        l4 = make_float4(Ln, Ln, Ln, Ln);
                        d4 = make_float4(Dn, Dn, Dn, Dn);
                    }
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const int tr = t0 + 8 * g + 4 * h2 + i;
                        const float pv = __builtin_amdgcn_exp2f(sc[jj][r] - lv[i]);
                        p[r] = (tr < N && lane_ok) ? pv : 0.f;
                        ds[r] = p[r] * (x[jj][r] - dv[i]) * (ROLE_KV ? 0.6931471805599453f : scale);
                    }
                }
                dsa[j][0] = tile_as_a(ds, 0); dsa[j][1] = tile_as_a(ds, 1);
                asm volatile("" : "+v"(dsa[j][0]), "+v"(dsa[j][1]));         // (formed HERE: left to hipcc the P / dS of every tile sink
                if constexpr (ROLE_KV) {                                     //  behind the last tile and the X / S tiles of all of them
                    pa[j][0] = tile_as_a(p, 0); pa[j][1] = tile_as_a(p, 1);  //  stay alive: 43 / 158 spilled registers at 3 / 4 tiles)
                    asm volatile("" : "+v"(pa[j][0]), "+v"(pa[j][1]));
                }
            }
        }
    }
    float* s_o = reinterpret_cast<float*>(smem + 16384);
    __syncthreads();
    // One sweep over the wave's tiles per output (dQ; dV, then dK): ONE set of four accumulator tiles is live at a time (with dV
    // and dK side by side two or more tiles per wave spill).  `which`: 0 = dQ (ROLE_Q), 1 = dV, 2 = dK (ROLE_KV).
    auto sweep = [&](int sg, auto which_tag) __attribute__((always_inline)) {
        constexpr int WHICH = decltype(which_tag)::value;
        const int c0 = sg * 128, w = min(128, hd - c0), nch = w >> 3, nct = (w + 31) >> 5;
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            int t0 = tile_row0(j);
            asm volatile("" : "+v"(t0));
            char* img = WHICH == 1 ? img1 : img2;                 // dV reads dA_t; dQ reads K_t; dK reads Qs_t
            // (one tile per wave: the last segment's images of phase 1 are still in place -- until the first reduction of this
            //  role has used the waves' image slices for its partial sums: the dK sweep always fetches)
            if (TW > 1 || sg != nseg - 1 || WHICH == 2) {
                dma_tile32(WHICH == 0 ? kb : (WHICH == 1 ? dab : qb), WHICH == 1 ? (long)d : ld, t0, N, c0, nch, img, lane);
                wait_dma();
            }
            const unsigned il = lds_addr(img);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 f[4];
                tr_frags4(il, boff, s, f);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nct) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WHICH == 1 ? pa[j][s] : dsa[j][s], f[c], acc[c], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        bf16_raw* out = a.dqkv + (WHICH == 0 ? 0 : (WHICH == 2 ? d : 2 * d)) + head * hd + c0;
        reduce_store16(s_o, acc, out, ld, b0, N, w, tid, wave, n, h2);
    };
    for (int sg = nseg - 1; sg >= 0; --sg) {
        if constexpr (!ROLE_KV) {
            sweep(sg, std::integral_constant<int, 0>{});          // dQ_seg = sum_t dS . K_t,seg
        } else {
            sweep(sg, std::integral_constant<int, 1>{});          // dV_seg = sum_t P' . dA_t,seg
            sweep(sg, std::integral_constant<int, 2>{});          // dK_seg = sum_t dS' . Qs_t,seg  (ln 2 folded into dS')
        }
    }
}

template <int TW>
__global__ __launch_bounds__(NW * 64) void k1_core_bwd_wide_kernel(CoreBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (core_block(a.N, a.H).z == 0) k1_core_bwd_wide_role<false, TW>(smem, a);
    else k1_core_bwd_wide_role<true, TW>(smem, a);
}

void core_attrs_once() {
    static std::once_flag once;                           // (one device per process: include/moma_hip.h)
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_wide_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_fwd_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS);
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_wide_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_wide_lds<1>());
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_wide_lds<2>());
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_wide_lds<3>());
        (void)hipFuncSetAttribute((const void*)k1_core_bwd_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_wide_lds<4>());
        (void)hipFuncSetAttribute((const void*)k1_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    });
}

constexpr int GEMM_LDS = 65536;
// tile geometry and block ranges of a job list; returns the dynamic LDS size of the launch
int finish_jobs(K1Jobs& js) {
    int t = 0, lds = GEMM_LDS;
    for (int i = 0; i < js.n; ++i) {
        K1Job& j = js.j[i];
        int TM = 64, TN = 64;
        if (j.kind == 0) {
            TM = 32;
            TN = 32;
        }
        j.tiles_n = (j.N + TN - 1) / TN;
        j.tile0 = t;
        t += j.tiles_n * ((j.M + TM - 1) / TM);
        j.tile_end = t;
    }
    return lds;
}
hipError_t launch_jobs(K1Jobs& js, hipStream_t st) {
    const int lds = finish_jobs(js);
    const int total = js.j[js.n - 1].tile_end;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k1_gemm_kernel, dim3(total), dim3(256), lds, st, js);
    return hipGetLastError();
}
K1Job kc_job(const void* A, int a_f32, long lda, const bf16_raw* B, long ldb, int M, int N, int K) {
    K1Job j{};
    j.kind = 0; j.A = A; j.a_f32 = a_f32; j.lda = lda; j.B = B; j.ldb = ldb; j.M = M; j.N = N; j.K = K;
    j.scale = 1.f; j.scale_cols = 0;
    return j;
}
K1Job ks_job(const void* A, int a_f32, long lda, const void* B, int b_f32, long ldb, float* C, long ldc, int M, int N, int K,
             float* colsum) {
    K1Job j{};
    j.kind = 1; j.A = A; j.a_f32 = a_f32; j.lda = lda; j.B = B; j.b_f32 = b_f32; j.ldb = ldb; j.M = M; j.N = N; j.K = K;
    j.C32 = C; j.ldc32 = ldc; j.colsum = colsum;
    return j;
}
}  // namespace

bool mha_fast_supported(int N, int d, int H, int prec) {
    if (prec != MOMA_PREC_BF16 || H <= 0 || d % H || N < 1) return false;
    const int hd = d / H;
    // head dims up to 128: any N.  Wider heads (segments of 128 columns, k1_core_*_wide_kernel): up to 4 key tiles per wave
    return hd % 16 == 0 && (hd <= 128 || (hd <= 1024 && N <= KT * NW * 4));
}

hipError_t launch_mha_pack(const float* w_qkv, const float* w_proj, void* pack, int d, int with_t, hipStream_t st) {
    const int tiles_c = (d + 31) / 32;
    const int tiles = ((3 * d + 31) / 32 + (d + 31) / 32) * tiles_c;
    hipLaunchKernelGGL(k1_pack_kernel, dim3(tiles), dim3(256), 0, st, w_qkv, w_proj, (bf16_raw*)pack, d, with_t);
    return hipGetLastError();
}

hipError_t launch_mha_fwd_fast(const moma_mha_module_t* mods, int n_modules, int N, int d, int H, hipStream_t st) {
    core_attrs_once();
    const int hd = d / H;
    const float scale_log2 = 1.4426950408889634f / sqrtf((float)hd);
    K1Jobs js{};
    js.n = n_modules;
    for (int i = 0; i < n_modules; ++i) {                   // qkv = x Wqkv^T + b  -> bf16, Q pre-scaled   (:156)
        const moma_mha_module_t& m = mods[i];
        K1Job j = kc_job(m.x, m.x_dtype == MOMA_DT_F32, d, (const bf16_raw*)m.pack, d, N, 3 * d, d);
        j.bias = m.b_qkv;
        j.C16 = (bf16_raw*)m.qkv16; j.ldc16 = 3L * d;
        j.scale = scale_log2; j.scale_cols = d;
        js.j[i] = j;
    }
    hipError_t e = launch_jobs(js, st);
    if (e != hipSuccess) return e;
    CoreFwdArgs ca{};
    ca.N = N; ca.d = d; ca.H = H;
    for (int i = 0; i < n_modules; ++i) ca.m[i] = CoreMod{(const bf16_raw*)mods[i].qkv16, (bf16_raw*)mods[i].attn16, mods[i].lse};
    const dim3 grid(((N + 31) / 32) * H * n_modules), block(NW * 64);      // 1-D: core_block() places the workgroups
    const bool one = N <= KT * NW, full = hd == 128;                                                       // (:159-163)
    const int tw = (N + KT * NW - 1) / (KT * NW);                  // key tiles per wave of the wide-head cores: 1 .. 4
    if (hd > 128 && tw <= 1) hipLaunchKernelGGL(k1_core_fwd_wide_kernel<1>, grid, block, CORE_LDS, st, ca);
    else if (hd > 128 && tw == 2) hipLaunchKernelGGL(k1_core_fwd_wide_kernel<2>, grid, block, CORE_LDS, st, ca);
    else if (hd > 128 && tw == 3) hipLaunchKernelGGL(k1_core_fwd_wide_kernel<3>, grid, block, CORE_LDS, st, ca);
    else if (hd > 128) hipLaunchKernelGGL(k1_core_fwd_wide_kernel<4>, grid, block, CORE_LDS, st, ca);
    else if (one && full) hipLaunchKernelGGL((k1_core_fwd_kernel<true, true>), grid, block, CORE_LDS, st, ca);
    else if (one) hipLaunchKernelGGL((k1_core_fwd_kernel<true, false>), grid, block, CORE_LDS, st, ca);
    else if (full) hipLaunchKernelGGL((k1_core_fwd_kernel<false, true>), grid, block, CORE_LDS, st, ca);
    else hipLaunchKernelGGL((k1_core_fwd_kernel<false, false>), grid, block, CORE_LDS, st, ca);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int i = 0; i < n_modules; ++i) {                   // y = a Wproj^T + b   (:164)
        const moma_mha_module_t& m = mods[i];
        K1Job j = kc_job(m.attn16, 0, d, (const bf16_raw*)m.pack + 3L * d * d, d, N, d, d);
        j.bias = m.b_proj;
        j.C32 = m.y; j.ldc32 = d;
        j.qpack = (bf16_raw*)m.qpack; j.qpack_scale = m.qpack_scale;
        js.j[i] = j;
    }
    return launch_jobs(js, st);
}

size_t mha_bwd_fast_workspace_bytes(int N, int d) {
    // dA16 [N, d] | dqkv16 [N, 3d] | dpart [d / 16, N]
    return (((size_t)N * d + (size_t)N * 3 * d) * 2 + (size_t)(d / 16) * N * 4 + 255) / 256 * 256;
}

hipError_t launch_mha_bwd_fast(const void* pack, const void* x, int x_dtype, const void* qkv16, const void* attn16, const float* lse,
                               const float* dy, float* dx, float* dw_qkv, float* db_qkv, float* dw_proj, float* db_proj,
                               void* workspace, int N, int d, int H, hipStream_t st) {
    core_attrs_once();
    const bf16_raw* P = (const bf16_raw*)pack;
    const bf16_raw* wqkv_t = P + 4L * d * d;                 // [d, 3d]
    const bf16_raw* wproj_t = P + 7L * d * d;                // [d, d]
    bf16_raw* dA16 = (bf16_raw*)workspace;
    bf16_raw* dqkv16 = dA16 + (size_t)N * d;
    float* dpart = (float*)(dqkv16 + (size_t)N * 3 * d);
    K1Jobs js{};
    // launch 1: dA = dy Wproj (+ D partials), dWproj = dy^T a, dbproj = colsum(dy)
    js.n = 0;
    {
        K1Job j = kc_job(dy, 1, d, wproj_t, d, N, d, d);
        j.C16 = dA16; j.ldc16 = d;
        j.dpart = dpart; j.R = (const bf16_raw*)attn16; j.ldr = d;
        js.j[js.n++] = j;
    }
    if (dw_proj) js.j[js.n++] = ks_job(dy, 1, d, attn16, 0, d, dw_proj, d, d, d, N, db_proj);
    hipError_t e = launch_jobs(js, st);
    if (e != hipSuccess) return e;
    // launch 2: the per-head core
    CoreBwdArgs ca{(const bf16_raw*)qkv16, dA16, lse, dpart, dqkv16, N, d, H};
    const int tw = (N + KT * NW - 1) / (KT * NW);
    if (d / H > 128 && tw <= 1) hipLaunchKernelGGL(k1_core_bwd_wide_kernel<1>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), bwd_wide_lds<1>(), st, ca);
    else if (d / H > 128 && tw == 2) hipLaunchKernelGGL(k1_core_bwd_wide_kernel<2>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), bwd_wide_lds<2>(), st, ca);
    else if (d / H > 128 && tw == 3) hipLaunchKernelGGL(k1_core_bwd_wide_kernel<3>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), bwd_wide_lds<3>(), st, ca);
    else if (d / H > 128) hipLaunchKernelGGL(k1_core_bwd_wide_kernel<4>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), bwd_wide_lds<4>(), st, ca);
    else if (d / H == 128) hipLaunchKernelGGL(k1_core_bwd_kernel<true>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), BWD_LDS, st, ca);
    else hipLaunchKernelGGL(k1_core_bwd_kernel<false>, dim3(((N + 31) / 32) * H * 2), dim3(NW * 64), BWD_LDS, st, ca);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // launch 3: dWqkv = dqkv^T x, dbqkv = colsum(dqkv), dx = dqkv Wqkv
    js.n = 0;
    if (dw_qkv) js.j[js.n++] = ks_job(dqkv16, 0, 3L * d, x, x_dtype == MOMA_DT_F32, d, dw_qkv, d, 3 * d, d, N, db_qkv);
    if (dx) {
        K1Job j = kc_job(dqkv16, 0, 3L * d, wqkv_t, 3L * d, N, d, 3 * d);
        j.C32 = dx; j.ldc32 = d;
        js.j[js.n++] = j;
    }
    if (js.n == 0) return hipSuccess;
    return launch_jobs(js, st);
}

}  // namespace moma
