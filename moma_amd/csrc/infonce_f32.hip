// K2 under the exact-fp32 policy -- the reference's own arithmetic (MoMA/mem_moco.py:29-49,77-100 + CrossEntropy,
// helper/loops_moma.py:322,331-335 are fp32) -- as ONE pass over an fp32 queue: no [B,K+1] logits in HBM, no queue clone, no second
// read of the queue in the backward.  Same algorithm as the bf16 one-pass kernel (infonce_fused.hip): per (32 query rows, key
// chunk) an online softmax over 32-key tiles leaves (reference, sum, max, sum_j p_j key_j); a combine kernel merges the chunks,
// adds the positive logit and writes loss / lse / top-1 / dq.  The arithmetic is the f32-input MFMA (v_mfma_f32_32x32x2_f32: an
// exact fp32 fma chain at the fp32 vector rate, 157 TFLOP/s chip-wide), so the roofline of this kernel is that rate:
// 4 B d (K+1) flop = 34.4 GFLOP at the benchmark shape -> 219 us (the HBM side, 135 MB, is 17 us).
//
// Decomposition: Q (32 x d fp32) + O (32 x d fp32) do not fit one wave's registers, so the four waves of a workgroup split the
// COLUMNS: wave w owns the column slab [w d/4, (w+1) d/4) of Q (resident B operands), of every key tile (its own LDS-DMA stream
// into its own double buffer -- no barrier guards the tiles) and of O.  Per tile:
//   partial scores  X_w[key, q] = sum over the slab of K[key, c] Q[q, c]     d/8 MFMAs   (A: keys by ds_read_b128, 4 MFMAs each)
//   exchange        X = X_0 + X_1 + X_2 + X_3 through LDS (fixed order), two barriers
//   softmax         p = 2^(X - ref_q), row sums / maxima (every wave, redundantly: 48 VALU against 128 MFMAs of 64 cycles)
//   P.K             O_w[q, slab] += sum_key p[q, key] K[key, slab]            d/8 MFMAs   (A: p as it stands, B: ds_read_b32)
// The softmax reference of a row is fixed to (first tile's max + 32) as in the bf16 kernel (fp32 has the same exponent range);
// a score 2^96 above it makes the workgroup repeat its chunk with the true row maxima as references (rare; tested).
#include "common.hpp"
#include <mutex>

namespace moma {
namespace {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int KT = 32;
constexpr float NEG_BIG = -1.0e30f;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
constexpr float REF_MARGIN = 32.0f, OVERFLOW_THR = 96.0f;

struct F32Plan {
    int nrb, nchunk, tiles_per_chunk, Bpad;
};
F32Plan f32_plan(int B, int K) {
    F32Plan p;
    p.nrb = (B + 31) / 32;
    p.Bpad = p.nrb * 32;
    const int ntiles = (K + KT - 1) / KT;
    int want = 256 / p.nrb;                       // ~1 workgroup per CU
    if (want < 8) want = 8;
    want = (want / 8) * 8;
    int tpc = (ntiles + want - 1) / want;
    if (tpc < 1) tpc = 1;
    p.tiles_per_chunk = tpc;
    p.nchunk = (ntiles + tpc - 1) / tpc;
    return p;
}

// CT = column tiles (of 32) per wave: d = 128 CT.  The wave's slab of a key tile goes through LDS in NSEG = CT / SC SEGMENTS of SC
// column tiles (SC in {1, 2, 4}: power-of-two row pitch for the xor-stepped addresses): d = 128 / 256 / 512 are one segment
// (SC = CT), wider rows -- the reference CLI's default --head None: 384, 768, 1280, also 1024 -- stream 3 / 3 / 5 / 2
// segments per tile through the same two buffers per wave.  The scores need every segment before the softmax and P.K needs them
// again after it, so per tile the wave's DMA stream is S_0 .. S_{n-1}, P_{n-2} .. P_0 (P_{n-1} finds its segment still in place):
// 2n - 1 segment fetches, the repeats from L2; the kernel is bound by the f32 matrix rate, 2.3x above its HBM side, so the extra
// L2 -> LDS traffic is affordable where a materialised [B,K+1] logits matrix is not.
// LDS: per wave two segment buffers of 32 keys x 128 SC bytes | X exchange 16 KiB | flag
template <int SC> constexpr int f32_lds_bytes() { return 4 * 2 * 32 * 128 * SC + 16384 + 16; }

template <int CT, int SC, bool WITH_DQ>
__global__ __launch_bounds__(256) void infonce_f32_flash_kernel(const float* __restrict__ q, const float* __restrict__ queue, int B,
                                                                int K, float scale_log2, int nrb, int nchunk, int tiles_per_chunk,
                                                                int Bpad, float* __restrict__ o_part, float* __restrict__ m_part,
                                                                float* __restrict__ l_part, float* __restrict__ x_part) {
    static_assert(CT % SC == 0 && (SC == 1 || SC == 2 || SC == 4), "segments of 1 / 2 / 4 column tiles");
    constexpr int NSEG = CT / SC;                 // segments of the wave's slab
    constexpr int SLAB = 32 * CT;                 // columns per wave
    constexpr int SEGC = 32 * SC;                 // columns per segment
    constexpr int PITCH = SEGC * 4;               // bytes per key row of a segment image
    constexpr int NCH = PITCH / 16;               // 16-B chunks per row: 8 / 16 / 32
    constexpr int G = SEGC / 8;                   // groups of 8 columns per segment = 4 MFMAs each
    constexpr int GT = SLAB / 8;                  // ... per slab
    constexpr int NDMA = 32 * PITCH / 1024;       // LDS-DMA instructions per segment and wave
    constexpr int D = 4 * SLAB;
    constexpr int JOBS = WITH_DQ ? 2 * NSEG - 1 : NSEG;      // segment fetches per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    // block -> (row block, key chunk): the row blocks of one chunk are 8 block ids apart (same XCD: one L2 fetch of the keys)
    int rb, chunk;
    {
        const int id = blockIdx.x;
        if ((nchunk & 7) == 0) {
            const int g = id / (8 * nrb), r = id % (8 * nrb);
            rb = r >> 3;
            chunk = g * 8 + (r & 7);
        } else {
            rb = id % nrb;
            chunk = id / nrb;
        }
    }
    char* bufs = smem + wave * (2 * 32 * PITCH);
    float* xs = reinterpret_cast<float*>(smem + 4 * 2 * 32 * PITCH);        // [4 waves][4][64][4]
    unsigned* flag = reinterpret_cast<unsigned*>(smem + 4 * 2 * 32 * PITCH + 16384);
    const int ntiles = (K + KT - 1) / KT;
    const int t0 = chunk * tiles_per_chunk, t1 = min(t0 + tiles_per_chunk, ntiles);
    const int slab0 = wave * SLAB;

    // resident Q slab: lane (q = n, h) holds Q[q][slab0 + 8 g + 4 h + m], pre-scaled by log2(e) / T
    float qr[GT][4];
    {
        const int row = rb * 32 + n;
        const float sc = row < B ? scale_log2 : 0.f;          // rows past B: zero queries (unconditional loads from a clamped row)
        const float* qp = q + (long)min(row, B - 1) * D + slab0 + 4 * h;
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(qp + 8 * g);
            qr[g][0] = v.x * sc; qr[g][1] = v.y * sc; qr[g][2] = v.z * sc; qr[g][3] = v.w * sc;
        }
    }
    // LDS-DMA of this wave's slab of tile t: piece p covers image bytes [1024 p, 1024 p + 1024); lane L -> row, chunk position;
    // position cp of a row holds the row's chunk cp ^ (row & (NCH - 1) & 15)   (bank spread for the ds_read_b128 row reads)
    auto dma_seg = [&](int t, int seg, int b) __attribute__((always_inline)) {
        char* dst = bufs + b * (32 * PITCH);
#pragma unroll
        for (int p = 0; p < NDMA; ++p) {
            const int o = p * 1024 + lane * 16;
            const int row = o / PITCH, cp = (o % PITCH) >> 4;
            const int ch = cp ^ (row & (NCH - 1) & 15);
            const long key = min((long)t * KT + row, (long)K - 1);            // keys past K: clamped, masked below
            const char* src = reinterpret_cast<const char*>(queue + key * D + slab0 + seg * SEGC) + ch * 16;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + p * 1024), 16, 0, 0);
        }
    };
    // The wave's DMA stream: job j of tile t (j = 0 .. JOBS-1) fetches segment seg_of(j) into buffer (tile index * JOBS + j) & 1;
    // `advance` issues the job AFTER (t, j) -- possibly the next tile's first -- and waits until job (t, j) itself has landed.
    auto seg_of = [](int j) { return j < NSEG ? j : 2 * NSEG - 2 - j; };
    auto advance = [&](int t, int j) __attribute__((always_inline)) {
        const int jn = j + 1 < JOBS ? j + 1 : 0, tn = j + 1 < JOBS ? t : t + 1;
        if (tn < t1) {
            dma_seg(tn, seg_of(jn), (((tn - t0) * JOBS + jn) & 1));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // Lane terms of the two kinds of LDS reads, so that every read is ONE of a few base registers plus a compile-time constant
    // (left as one expression per read, hipcc keeps 64 + 16 precomputed addresses alive across the tile loop and runs out of
    // registers: every MFMA then waits for its own ds_read):
    //   row read (scores, A operand): chunk 2 g + h of key row n  ->  a_base ^ 32 g      (a_base = n PITCH + 16 ((h ^ n) & (NCH-1) & 15) ...)
    //   column read (P.K, B operand): K[key (s&3) + 8 (s>>2) + 4 h][32 c + n]  ->  pv_base[s & 3] + const(s, c)
    constexpr int SW = (NCH - 1) & 15;                       // swizzle mask: row & SW
    const unsigned a_base = (unsigned)(n * PITCH + 16 * ((n & SW) ^ h));
    unsigned pv_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pv_base[j] = (unsigned)(4 * h * PITCH + 16 * ((n >> 2) ^ ((j + 4 * h) & SW)) + 4 * (n & 3));

    float m_ref = NEG_BIG, l_run = 0.f, x_max = NEG_BIG;
    f32x16 O[CT];
    bool second = false;
    for (;;) {                                     // at most two passes over the chunk (the second only after an overflow)
        l_run = 0.f;
        float seen_max = NEG_BIG;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
        if (tid == 0) *flag = 0u;
        dma_seg(t0, 0, 0);
        for (int t = t0; t < t1; ++t) {
            const int jb = (t - t0) * JOBS;                                   // (parity of) this tile's first DMA job
            // ---- partial scores over the wave's slab, segment by segment
            // (measured at one segment and not taken: the next tile's 16 pieces issued one per group of four score MFMAs instead of
            //  up front -- 389 vs 377 us with dq, 233 vs 210 forward-only; the pieces by inline asm with one base + one xor each
            //  instead of the builtin's ~12 instructions -- 384 / 212 us, no change: the DMA issue is not what the tile waits for;
            //  round 6, ablation builds (scripts/build_k2_f32_variants.py; call = kernel + 12 us combine, d = 512, same box): 388 us as
            //  shipped, 358 without the exchange's two barriers (354 without the whole exchange, 396 without the second barrier
            //  alone: what they cost is the skew of the four waves, and one barrier pays all of it), 388 without the exponentials,
            //  354 without the refill, 390 without its vmcnt waits, 320 with none of the three = the two MFMA loops and their LDS
            //  reads alone; the pieces one per MFMA behind the first quarter of the score product: 387 vs 390, 227 vs 231 at
            //  d = 256, 140 vs 143 at d = 128 -- not taken for 1-2 %)
            f32x16 x;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = 0.f;
            // (the LDS reads of both products are inline asm with counted waits: left to hipcc every read is sunk to its MFMA and
            //  waited for with lgkmcnt(0) -- one exposed LDS round trip per MFMA pair, measured 496 us instead of the 219 us bound)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                advance(t, sg);                                               // next fetch on its way, this segment landed
                const char* img = bufs + ((jb + sg) & 1) * (32 * PITCH);
                const unsigned ab = lds_addr(img) + a_base;
                f32x4 kf[2];
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0]) : "v"(ab) : "memory");
#pragma unroll
                for (int g = 0; g < G; ++g) {                // the fragment of group g + 1 is requested ahead of group g's MFMAs
                    if (g + 1 < G) {
                        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[(g + 1) & 1]) : "v"(ab ^ (unsigned)(32 * (g + 1))) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kf[g & 1]) : : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[g & 1]) : : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 f = kf[g & 1];
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0], qr[sg * G + g][0], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(f[1], qr[sg * G + g][1], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(f[2], qr[sg * G + g][2], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(f[3], qr[sg * G + g][3], x, 0, 0, 0);
                }
            }
            // ---- exchange: every wave sums the four partial tiles in the same order
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(&xs[((wave * 4 + g4) * 64 + lane) * 4]) = make_float4(x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 s = *reinterpret_cast<const float4*>(&xs[((0 * 4 + g4) * 64 + lane) * 4]);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 u = *reinterpret_cast<const float4*>(&xs[((w * 4 + g4) * 64 + lane) * 4]);
                    s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
                }
                x[4 * g4] = s.x; x[4 * g4 + 1] = s.y; x[4 * g4 + 2] = s.z; x[4 * g4 + 3] = s.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // xs is free for the next tile
            // ---- softmax against the fixed reference (register r of lane half h is key (r&3) + 8 (r>>2) + 4 h)
            // (measured in round 6 and not taken: each numerator formed one k-step ahead of its use, in the shadow of P.K's 64-cycle
            //  MFMAs -- register s is the A operand of k-step s -- instead of all 16 up front: 400 vs 400 us per call at d = 512,
            //  230 vs 234 at d = 256, same box, three alternating runs: these 48 VALU instructions are not what the tile waits for)
            if ((t + 1) * KT > K) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + (r & 3) + 8 * (r >> 2) + 4 * h >= K) x[r] = NEG_BIG;
            }
            float tmax = x[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
            tmax = fmaxf(tmax, other_half(tmax));
            seen_max = fmaxf(seen_max, tmax);
            if (t == t0 && !second) m_ref = tmax + REF_MARGIN;
            if (tmax - m_ref > OVERFLOW_THR) *flag = 1u;                    // (any lane of any wave; same value from all)
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = __builtin_amdgcn_exp2f(x[r] - m_ref);
                ps += x[r];
            }
            ps += other_half(ps);
            l_run += ps;
            // ---- O_w += P . K_tile[:, slab]: A = p (lane (q, h), k-step s <-> register s: key (s&3) + 8 (s>>2) + 4 h),
            //      B = K[key(s, h)][slab column 32 c + n] by ds_read_b32; segments in REVERSE order (the last score segment is
            //      still in its buffer; the others are fetched again -- L2 hits)
            if constexpr (WITH_DQ) {
                constexpr int HB = NCH > 8 ? 1 : 0;            // (with 8 chunks per row the swizzle has no bit 3)
#pragma unroll
                for (int u = 0; u < NSEG; ++u) {
                    const int sg = NSEG - 1 - u, j = NSEG - 1 + u;               // segment, and the DMA job that brought / brings it
                    if (u > 0) advance(t, j);
                    const char* img = bufs + ((jb + j) & 1) * (32 * PITCH);
                    unsigned pb[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) pb[jj] = lds_addr(img) + pv_base[jj];
                    float kv[2][SC];
                    auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
                        for (int c = 0; c < SC; ++c)
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(kv[s & 1][c]) : "v"(pb[s & 3]),
                                         "n"(((s & 3) + 8 * (s >> 2)) * PITCH + 128 * (c ^ (((s >> 2) & 1) * HB))) : "memory");
                    };
                    auto wait_for = [&](int s, bool more) __attribute__((always_inline)) {
                        // the SC reads of step s are complete once at most the SC younger ones (step s + 1) are outstanding
                        if constexpr (SC == 4) {
                            if (more) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kv[s & 1][0]), "+v"(kv[s & 1][1]), "+v"(kv[s & 1][2]), "+v"(kv[s & 1][3]) : : "memory");
                            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[s & 1][0]), "+v"(kv[s & 1][1]), "+v"(kv[s & 1][2]), "+v"(kv[s & 1][3]) : : "memory");
                        } else if constexpr (SC == 2) {
                            if (more) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kv[s & 1][0]), "+v"(kv[s & 1][1]) : : "memory");
                            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[s & 1][0]), "+v"(kv[s & 1][1]) : : "memory");
                        } else {
                            if (more) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kv[s & 1][0]) : : "memory");
                            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[s & 1][0]) : : "memory");
                        }
                    };
                    issue(0);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {             // the B values of step s + 1 are requested ahead of step s's MFMAs
                        if (s + 1 < 16) issue(s + 1);
                        wait_for(s, s + 1 < 16);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c = 0; c < SC; ++c)
                            O[sg * SC + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s], kv[s & 1][c], O[sg * SC + c], 0, 0, 0);
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
        x_max = seen_max;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned over = *reinterpret_cast<volatile unsigned*>(flag);
        __builtin_amdgcn_s_barrier();
        if (over == 0u || second) break;
        second = true;                              // repeat the chunk with the true row maxima as references
        m_ref = seen_max;
    }
    // ---- partials of (chunk, row block): m = reference, l, true max (wave 0) and the wave's slab of O
    const long prow = (long)chunk * Bpad + rb * 32;
    if (wave == 0 && h == 0) {
        m_part[prow + n] = m_ref;
        l_part[prow + n] = l_run;
        x_part[prow + n] = x_max;
    }
    if constexpr (WITH_DQ) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                o_part[(prow + row) * D + slab0 + 32 * c + n] = O[c][r];
            }
    }
}

// one workgroup per query row: merges the chunk partials in a fixed order, adds the positive logit (exact fp32)
__global__ __launch_bounds__(256) void infonce_f32_combine_kernel(const float* __restrict__ q, const float* __restrict__ k, int B, int D,
                                                                  float inv_T, int nchunk, int Bpad, const float* __restrict__ o_part,
                                                                  const float* __restrict__ m_part, const float* __restrict__ l_part,
                                                                  const float* __restrict__ x_part, float* __restrict__ loss_rows,
                                                                  float* __restrict__ lse_out, int32_t* __restrict__ top1,
                                                                  float* __restrict__ dq) {
    __shared__ float red[4];
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto block_sum = [&](float v) {                // fixed order: bitwise reproducible
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    auto block_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    // positive logit in log2 units: x0 = <q_b, k_b> * inv_T * log2(e)
    float acc = 0.f;
    for (int c = tid; c < D; c += 256) acc = fmaf(q[(long)b * D + c], k[(long)b * D + c], acc);
    const float x0 = block_sum(acc) * inv_T * LOG2E;
    float mloc = NEG_BIG, xloc = NEG_BIG;
    for (int c = tid; c < nchunk; c += 256) {
        mloc = fmaxf(mloc, m_part[(long)c * Bpad + b]);
        xloc = fmaxf(xloc, x_part[(long)c * Bpad + b]);
    }
    const float M = fmaxf(x0, block_max(mloc));
    const float X = block_max(xloc);
    float lloc = 0.f;
    for (int c = tid; c < nchunk; c += 256) lloc += exp2f(m_part[(long)c * Bpad + b] - M) * l_part[(long)c * Bpad + b];
    const float L = exp2f(x0 - M) + block_sum(lloc);
    if (tid == 0) {
        const float lse2 = M + log2f(L);
        loss_rows[b] = (lse2 - x0) * LN2;
        lse_out[b] = lse2 * LN2;
        top1[b] = x0 >= X ? 1 : 0;
    }
    const float invL = 1.0f / L, p0 = exp2f(x0 - M) / L;
    if (dq == nullptr) return;
    for (int c0 = tid; c0 < D; c0 += 256) {
        float a = 0.f;
        for (int c = 0; c < nchunk; ++c)
            a = fmaf(exp2f(m_part[(long)c * Bpad + b] - M), o_part[((long)c * Bpad + b) * D + c0], a);
        dq[(long)b * D + c0] = ((p0 - 1.0f) * k[(long)b * D + c0] + a * invL) * inv_T;
    }
}

std::once_flag g_f32_attr_once;
void set_f32_attrs() {
#define MOMA_F32_ATTR(CT, SC)                                                                                                     \
    (void)hipFuncSetAttribute((const void*)infonce_f32_flash_kernel<CT, SC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, f32_lds_bytes<SC>()); \
    (void)hipFuncSetAttribute((const void*)infonce_f32_flash_kernel<CT, SC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, f32_lds_bytes<SC>())
    MOMA_F32_ATTR(1, 1); MOMA_F32_ATTR(2, 2); MOMA_F32_ATTR(4, 4);
    MOMA_F32_ATTR(3, 1); MOMA_F32_ATTR(6, 2); MOMA_F32_ATTR(8, 4); MOMA_F32_ATTR(10, 2);
#undef MOMA_F32_ATTR
}
}  // namespace

// widths of the one-pass fp32 kernel: (column tiles per wave, per segment) = (d / 128, SC)
static bool f32_dim(int d) { return d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024 || d == 1280; }      // (d = 1536: 12 column tiles per wave spill; d = 2048: Q + O alone are the register file)

bool infonce_f32_flash_supported(int B, int d, int K, int qdtype, int prec) {
    return prec == MOMA_PREC_F32 && qdtype == MOMA_DT_F32 && f32_dim(d) && B >= 1 && K >= 1 &&
           f32_plan(B, K).nchunk <= 4096;
}

size_t infonce_f32_flash_workspace_bytes(int B, int d, int K) {
    const F32Plan p = f32_plan(B, K);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    return rows * d * sizeof(float) + 3 * rows * sizeof(float) + 1024;
}

hipError_t launch_infonce_f32_flash(const float* q, const float* k, const float* queue, int B, int d, int K, float inv_T,
                                    float* loss_rows, float* lse, int32_t* top1, float* dq, void* ws, hipStream_t st,
                                    hipEvent_t ev_begin, hipEvent_t ev_end) {
    const F32Plan p = f32_plan(B, K);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    float* m_part = (float*)ws;
    float* l_part = m_part + rows;
    float* x_part = l_part + rows;
    float* o_part = (float*)(((uintptr_t)(x_part + rows) + 255) & ~(uintptr_t)255);
    std::call_once(g_f32_attr_once, set_f32_attrs);
    const float scale_log2 = inv_T * 1.4426950408889634f;
    const dim3 grid(p.nrb * p.nchunk), block(256);
    if (ev_begin) (void)hipEventRecord(ev_begin, st);
#define MOMA_F32_LAUNCH(CT, SC)                                                                                                   \
    do {                                                                                                                          \
        if (dq) hipLaunchKernelGGL((infonce_f32_flash_kernel<CT, SC, true>), grid, block, f32_lds_bytes<SC>(), st, q, queue, B, K, scale_log2, \
                                   p.nrb, p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, m_part, l_part, x_part);                  \
        else hipLaunchKernelGGL((infonce_f32_flash_kernel<CT, SC, false>), grid, block, f32_lds_bytes<SC>(), st, q, queue, B, K, scale_log2, \
                                p.nrb, p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, m_part, l_part, x_part);                     \
    } while (0)
    switch (d) {
        case 128: MOMA_F32_LAUNCH(1, 1); break;
        case 256: MOMA_F32_LAUNCH(2, 2); break;
        case 384: MOMA_F32_LAUNCH(3, 1); break;
        case 512: MOMA_F32_LAUNCH(4, 4); break;
        case 768: MOMA_F32_LAUNCH(6, 2); break;
        case 1024: MOMA_F32_LAUNCH(8, 4); break;
        case 1280: MOMA_F32_LAUNCH(10, 2); break;
        default: return hipErrorInvalidValue;
    }
#undef MOMA_F32_LAUNCH
    if (ev_end) (void)hipEventRecord(ev_end, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(infonce_f32_combine_kernel, dim3(B), dim3(256), 0, st, q, k, B, d, inv_T, p.nchunk, p.Bpad, o_part, m_part, l_part,
                       x_part, loss_rows, lse, top1, dq);
    return hipGetLastError();
}

}  // namespace moma
