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
//   tile  = 32 keys x D bf16 in a ring of 4 LDS slots, filled by LDS-DMA (global_load_lds_dwordx4; the
//           bank swizzle is applied on the per-lane SOURCE address, the LDS image is lane-linear per piece).
//   score : X[key, q] = K_tile . Q^T      v_mfma_f32_32x32x16_bf16, A = keys (ds_read_b128), B = Q (regs)
//           -> the query sits on the lane, the 32 keys of the tile in the 16 registers x 2 lane halves,
//              so row max / sum are in-register (+1 cross-half shuffle) -- no LDS, no other wave involved;
//   P     : exp2(X - m) in registers, packed to bf16: registers 8s..8s+7 ARE the A fragment of k-step s of
//   P.K   : O[q, :] += P[q, keys] . K_tile  with B = keys read column-wise by ds_read_b64_tr_b16 in the
//           permuted k order key(s,h,j) = 16s + 8(j>>2) + 4h + (j&3).
//   The softmax reference m of a query row is FIXED to the row max of its chunk's first tile + 32 (bf16 and fp32
//   share the exponent range, so P may exceed 1 without losing precision) and O is not rescaled per tile; should a
//   later tile exceed m by more than 2^96 the WORKGROUP repeats its chunk with the true row maxima as references
//   (rare, workgroup-uniform branch; tests force it).
//   Each WG leaves (m, l, max, O) per query row; a combine kernel merges the chunks, adds the positive
//   logit (exact fp32) and writes loss / lse / top-1 / dq.  Two launches per call after the Q pre-pack.
//   Wide rows (d > 512): two passes over the queue -- infonce_wide_scores_kernel / infonce_wide_pv2_kernel below --
//   or, for widths those do not take, column-slab passes of this body (MODE 1 / 2).
//
// LDS image of a key tile: D/128 segments of [32 keys][128 cols] with 256-B rows,
//   off(seg,row,ch) = seg*8192 + row*256 + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))      ch = 16-B chunk 0..15
// which is conflict-free for both the row reads (ds_read_b128) and the transposed reads.
#include "common.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <type_traits>

// Tuning constants of the kernels below (each fixed by measurement on MI355X; DESIGN.md section 4 records the sweeps):
#define MOMA_K2_WPV_SD 3        // wide P.K pass: key tiles (and their P) requested ahead; ring of SD + 1 slots of 16 KiB
#define MOMA_K2_SMALL_AUX 0     // small-batch kernel (every key tile is read by ONE workgroup): cache policy of its queue stream

namespace moma {
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int QROWS_WG = 128;        // query rows per workgroup (4 waves x 32)
constexpr int KT = 32;               // keys per tile
constexpr int NBUF = 4;              // LDS ring: tile t is consumed while t+1..t+3 are in flight (LDS-DMA)
constexpr float NEG_BIG = -1.0e30f;
// The fixed softmax reference of a wave is (row max of its chunk's first tile) + REF_MARGIN.  Only P's exponent moves
// with the reference (bf16 / fp32 share the 8-bit exponent, down to 2^-126), so starting 2^-32 low costs nothing
// and leaves REF_MARGIN + OVERFLOW_THR = 128 log2 units (61 nats of logit) of headroom above the first tile's max before
// the repair launch is needed; sums stay below 2^(96 + 9) per chunk.  (Un-normalised attention outputs as q / keys do
// reach logit ranges of tens of nats in training: with the margin at 0 and the threshold at 64 the repair pass --
// 4x the cost of the main pass -- fired every other step of the benchmark from step 20 on.)
constexpr float REF_MARGIN = 32.0f;
constexpr float OVERFLOW_THR = 96.0f;

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// LDS-DMA of one key tile (32 keys x D bf16) into `buf`: D/16 pieces of 1 KiB (4 rows x 256 B of one segment);
// wave w issues pieces w, w+4, ... (PPW = D/64 per wave).  `dma_piece` issues this wave's i-th piece so that the
// issue cost can be spread between MFMAs instead of stalling the wave at the tile top.
//   source offset of lane L in piece (seg, rg): row = 4*rg + (L>>4), chunk = (L&15) ^ swz(row)
//   = [rg*4*D*2 + seg*256]  (wave-uniform, scalar)  +  lane term  (1 VGPR, computed once per kernel)
//   Every piece of wave w has rg & 3 == w (pc = 4i + w), so a wave needs a single lane term.
struct DmaLane {
    unsigned term;        // (L>>4)*pitch + ((L&15) ^ swz(4*w + (L>>4)))*16
    unsigned pitch;       // bytes between queue rows: D*2, or the full row when the tile is a column slab of a wider queue
};
template <int D>
__device__ __forceinline__ DmaLane dma_lane_terms(int lane, int wave, unsigned pitch) {
    DmaLane t;
    const int rl = lane >> 4, sl = lane & 15;
    t.pitch = pitch;
    t.term = (unsigned)(rl * pitch + ((sl ^ swz(4 * wave + rl)) << 4));
    return t;
}
// AUX: cache policy bits of the load (0 = default, 2 = nt: the line is streamed, not kept)
template <int D, bool PARTIAL = true, int AUX = 0>
__device__ __forceinline__ void dma_piece(int i, const DmaLane& dl, const bf16_raw* __restrict__ queue, long key0, int K,
                                          char* buf, int wave, int lane) {
    const int pc = i * 4 + wave;                // wave-uniform
    const int seg = pc >> 3, rg = pc & 7;
    const char* tile = reinterpret_cast<const char*>(queue) + key0 * (long)dl.pitch + seg * 256;   // wave-uniform
    unsigned off = dl.term;
    if (PARTIAL && key0 + KT > K) {             // queue's last, partial tile: clamp rows past K (masked later)
        const int rl = lane >> 4;
        const int row = min(rg * 4 + rl, (int)(K - 1 - key0));
        off = off - (unsigned)rl * dl.pitch + (unsigned)row * dl.pitch;
    } else {
        tile += rg * 4 * (long)dl.pitch;
    }
    char* dst = buf + seg * 8192 + rg * 1024;   // wave-uniform LDS base; lane L lands at +16*L
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, AUX);
}

template <int D, int AUX = 0>
__device__ __forceinline__ void dma_tile(const DmaLane& dl, const bf16_raw* __restrict__ queue, long key0, int K,
                                         char* buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < D / 64; ++i) dma_piece<D, true, AUX>(i, dl, queue, key0, K, buf, wave, lane);
}

// Q operand pre-pack: scale by log2(e)/T, round to bf16 and store in MFMA-fragment order so that every
// workgroup of the main kernel fetches its Q tile with fully coalesced 16-B-per-lane loads:
//   qpack[((row_tile*KS + ks)*64 + lane)] = 8 bf16 = Q[32*row_tile + (lane&31)][col0 + 16*ks + 8*(lane>>5) + 0..7]
// (rows >= B are zero; col0 / dfull select a column slab of a wider q).  One 64-lane "virtual wave" per 32-row tile.
template <int D>
__global__ __launch_bounds__(256) void infonce_qpack_kernel(const float* __restrict__ q, int B, int dfull, int col0,
                                                            float scale_log2, uint4* __restrict__ qpack, int n_row_tiles) {
    constexpr int KS = D / 16;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);          // (row tile, k-step)
    if (item >= n_row_tiles * KS) return;
    const int rt = item / KS, ks = item % KS;
    const int row = rt * 32 + (lane & 31), h = lane >> 5;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row < B) {
        const float* qp = q + (long)row * dfull + col0 + 16 * ks + 8 * h;
        a = *reinterpret_cast<const float4*>(qp);
        b = *reinterpret_cast<const float4*>(qp + 4);
    }
    const bf16x8 f = bf16x8{(__bf16)(a.x * scale_log2), (__bf16)(a.y * scale_log2), (__bf16)(a.z * scale_log2),
                            (__bf16)(a.w * scale_log2), (__bf16)(b.x * scale_log2), (__bf16)(b.y * scale_log2),
                            (__bf16)(b.z * scale_log2), (__bf16)(b.w * scale_log2)};
    qpack[(long)item * 64 + lane] = __builtin_bit_cast(uint4, f);
}

// The softmax reference m of a query row is FIXED to (row max of its chunk's first tile) + REF_MARGIN: bf16 and fp32
//   share the 8-bit exponent, so P = 2^(x - m) may exceed 1 by many orders of magnitude without losing precision and O
//   is not rescaled per tile.  A wave that meets a score more than 2^OVERFLOW_THR above a row's m raises a word in LDS and
//   the workgroup repeats its chunk with the true row maxima (recorded by the first pass) as references -- `run_pass` below.
//   tests/test_gpu_kernels.py forces the branch (guide rule 26).
// MODE 0: the one-pass kernel described above (d = D).
// MODE 1 / 2: wide queues (d > 512) go through column SLABS of D <= 512 columns of the same key tiles, with the complete
//   score tiles kept in a scratch `xs` in register order (xs[((row-wave * ntiles + tile) * 64 + lane) * 16 + r]):
//   MODE 1 adds this slab's partial scores to the scratch (first slab: stores), nothing else;
//   MODE 2 reads the finished scores, forms P = 2^(x - chunk max) and accumulates this slab's columns of O (un-normalised:
//   the combine kernel weighs the chunks, as for the one-pass kernel).
//   `queue` then points at the slab's first column and `slab` carries the row pitch and the first-slab flag.
struct SlabArgs {
    float* xs;
    const float* mref;    // [nchunk][Bpad] log2 reference per (chunk, row) (MODE 2, P = 2^(x - mref): the combine kernel weighs)
    unsigned pitch;       // bytes between queue rows
    int first;            // MODE 1: this is the first slab (store instead of accumulate)
};

// O partial of one wave block (32 query rows x D) in REGISTER order, 16 B per lane and store:
//   o_part4[wave block][(c*2 + g)*64 + lane] = 4 words w = 0..3, word w = bf16 pair { O[c][8g+2w] , O[c][8g+2w+1] }
//   = rows 16g + 4h + {0,1 | 2,3 | 8,9 | 10,11}[w] of column 32c + (lane&31), h = lane>>5
// (O[c][r] is query row (r&3) + 8*(r>>2) + 4*h); the combine kernel gives one workgroup the 8 rows of a (block, g, h).
__device__ __forceinline__ int opart_row(int g, int h, int w, int u) { return 16 * g + 4 * h + 8 * (w >> 1) + 2 * (w & 1) + u; }
// Which partials exist.  The passes over the queue skip the O partial of a wave block (32 query rows) that lies past B entirely,
// and the combine kernel must then never read it: BOTH sides decide with these two predicates.  group_live relies on
// opart_row(g, h, 0, 0) being the SMALLEST of the 8 rows of a (g, h) group (w, u only add), so "its first row is live" <=> "the
// wave block that holds it was stored" -- a layout change to opart_row has to keep that, or change both predicates together.
__device__ __forceinline__ bool wave_block_live(int wb, int B) { return wb * 32 < B; }
__device__ __forceinline__ bool group_live(int wb, int g, int h, int B) { return wb * 32 + opart_row(g, h, 0, 0) < B; }

template <int D, bool WITH_DQ, int MODE = 0>
__device__ __forceinline__ void infonce_flash_body(const int id, char* smem, const uint4* __restrict__ qpack,
                                                   const bf16_raw* __restrict__ queue, int B, int K, int nbt,
                                                   int nchunk, int tiles_per_chunk, int Bpad,
                                                   uint4* __restrict__ o_part, float* __restrict__ m_part,
                                                   float* __restrict__ l_part, float* __restrict__ x_part,
                                                   SlabArgs slab = SlabArgs{}) {
    static_assert(MODE != 1 || !WITH_DQ, "score slabs carry no O");
    static_assert(MODE != 2 || WITH_DQ, "P.K slabs carry O");
    constexpr int KS = D / 16;       // k-steps of the score product
    constexpr int NCT = D / 32;      // 32-column tiles of O
    constexpr int TILE_BYTES = KT * D * 2;
    constexpr bool PIPELINED = WITH_DQ && MODE == 0;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;

    // block -> (query tile, key chunk): the nbt tiles of one chunk are 8 block ids apart (same XCD)
    int bt, chunk;
    {
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
    const DmaLane dl = dma_lane_terms<D>(lane, wave, MODE == 0 ? (unsigned)(D * 2) : slab.pitch);
    // partial slot of this wave's 32 query rows in chunk `chunk`
    const long prow = (long)chunk * Bpad + bt * QROWS_WG + wave * 32;

    float m_ref = NEG_BIG;
    if constexpr (MODE == 2)
        m_ref = slab.mref[prow + n];
    // MODE 2: the finished scores of tile t are loaded one tile ahead by inline asm (counted by hand like the Q loads: at the
    // first use of an ordinary load hipcc drains vmcnt to 0, the whole tile ring included, and the load's own latency -- a
    // round trip to the scratch in HBM / Infinity Cache -- was exposed once per tile)
    f32x16 xnext;
    auto load_scores = [&](int t) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
            const char* xa = reinterpret_cast<const char*>(slab.xs + (((long)(bt * 4 + wave) * ntiles + t) * 64 + lane) * 16);
            f32x4 v0, v1, v2, v3;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(xa) : "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) { xnext[r] = v0[r]; xnext[4 + r] = v1[r]; xnext[8 + r] = v2[r]; xnext[12 + r] = v3[r]; }
        }
    };
    if constexpr (MODE == 2) load_scores(t0);       // (older than every tile of the ring: landed with the prologue's first wait)

    // ---- Q fragments: B operand of X = K.Q^T ; lane (q=n, h) holds Q[q][16ks + 8h + j], pre-scaled by log2e/T,
    //      pre-packed in fragment order by infonce_qpack_kernel (coalesced 16 B per lane).  The loads are inline asm and
    //      counted by hand: left to hipcc, the first use of a Q register waits vmcnt(0), i.e. for every key tile of the
    //      prologue as well (LDS-DMA and ordinary loads share the counter).
    constexpr int PPW = D / 64;                       // DMA instructions per wave per tile
    constexpr int NQL = (MODE != 2) ? KS : 0;         // Q load instructions per wave
    bf16x8 qf[KS];
    auto load_q = [&]() __attribute__((always_inline)) {
        if constexpr (MODE != 2) {
            const char* qb = reinterpret_cast<const char*>(qpack + ((long)(bt * 4 + wave) * KS) * 64 + lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[ks]) : "v"(qb + (ks >> 2) * 4096), "n"((ks & 3) * 1024) : "memory");
            }
        }
    };
    load_q();

    // LDS-DMA ring.  Every wave issues PPW pieces per tile, in tile order, so "all but the newest j tiles of
    // this wave have landed" is s_waitcnt vmcnt(j*PPW); the workgroup barrier then makes the other waves'
    // pieces visible too.  __syncthreads() would drain vmcnt to 0, hence the raw s_barrier.
    auto wait_tiles_in_flight = [&](int j) __attribute__((always_inline)) {
        if (j >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (j == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (j == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert((NBUF & (NBUF - 1)) == 0 && NBUF == 4, "ring of four tiles");
    static_assert(3 * PPW + NQL <= 63, "vmcnt is a 6-bit counter");
    auto slot = [&](int t) __attribute__((always_inline)) -> char* { return smem + ((unsigned)(t - t0) & (NBUF - 1)) * TILE_BYTES; };

    // the whole ring is requested up front (Q first: the first score needs all of it, the tiles one after the other)
    const int npro = min(t1 - t0, NBUF);
#define MOMA_K2_PRO 4
    // Tiles requested BEFORE the first wait.  An LDS-DMA piece costs ~150 cycles of issue when nothing else runs (the CU's
    // address path moves 64 B/clk: the 4 x 32 KiB ring alone is 2 k cycles of it, measured 4.7 k) and ~nothing in the shadow of
    // MFMAs, so the pipelined kernel requests only two tiles up front and fills the rest of the ring from the first tile's
    // scores (tile t0+2) and the first loop iteration (tile t0+3), exactly like the steady-state refills.
    constexpr int PRO = (WITH_DQ && MODE == 0) ? 2 : MOMA_K2_PRO;
    static_assert(PRO == 2 || PRO == NBUF, "the non-pipelined loops find the whole ring requested");
    // The prologue's requests and its first wait branch on the SAME wave-uniform predicates, nested the same way: on every
    // path through the two the wait's count is the number of pieces that path has issued behind Q and tile t0 -- which is what
    // scripts/audit_isa.py checks on the compiled code (it walks every path and knows only that the same comparison of the
    // same registers gives the same answer).  Tile t0 always exists.
    auto issue_ring = [&](auto first_pass) __attribute__((always_inline)) {
        int tb = t0;
        asm volatile("" : "+s"(tb));                   // (opaque: the two passes do not share hoisted source addresses)
#pragma unroll
        for (int j = 0; j < PRO; ++j) {
            if (npro > j) dma_tile<D>(dl, queue, (long)(tb + j) * KT, K, smem + j * TILE_BYTES, wave, lane);
        }
    };
    // Q and tile t0 have landed; the other tiles of the prologue stay in flight
    auto wait_first_tile = [&]() __attribute__((always_inline)) {
        if (npro > 1) {
            if constexpr (PRO > 2) {
                if (npro > 2) {
                    if (npro > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
                } else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            } else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    issue_ring(std::true_type{});

    f32x16 O[WITH_DQ ? NCT : 1];
    float l_run = 0.f, mx = NEG_BIG;
    int ovf = 0;

    // per-lane LDS offsets
    //  row read (A operand of the score MFMA): key row n, chunk 2*(ks&7)+h of segment ks>>3
    //  chunk (2c ^ g) of the row = lane part ^ (c << 5): ONE register across tiles, the 8 addresses of a tile are xor-ed from it
    //  when the tile begins (tile slots are 8 KiB-aligned, so bits 4..7 of the lane part belong to the chunk alone).  An xor in
    //  front of every read instead (one address register in all) costs 4 us: the reads then wait on the VALU result.
    const int a_off = n * 256 + ((swz(n) ^ h) << 4);
    //  transposed read (B operand of P.K): 16-lane group -> 4 keys x 16 columns
    //  chunk ((c ^ q4) << 2 | (e ^ 2u)) = lane part ^ (c << 6) ^ (u << 5), same scheme
    int b_off;
    {
        const int i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
        const int e = (2 * g2 + (p >> 1)) ^ h;
        b_off = (4 * h + q4) * 256 + 8 * (p & 1) + (((q4 << 2) | e) << 4);
    }

    // ---- scores of one tile: X[key, q] over D.  A fragments (keys) are requested RD k-steps ahead
    // of the MFMAs that consume them; the sched_barriers pin "issue reads, then MFMAs" (left alone, hipcc sinks the
    // reads behind the MFMAs and exposes the LDS latency once per group).  One LDS-DMA piece of the refill tile is
    // issued after every 4th MFMA, so its issue cost hides behind the matrix pipe.
    auto wait_lgkm = [&](int n) __attribute__((always_inline)) {          // n is a constant after unrolling
        switch (n) {
            case 15: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;
            case 13: asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        }
    };
    // ---- transposed reads of P.K (B operand: keys column-wise).  They are inline asm: through the builtin hipcc orders every
    // ds_read_b64_tr_b16 behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would drain the tile ring.  Column tile c
    // takes 4 reads (lo: k-step 0, hi: k-step 1) into buffer c % (PF+1); they are counted by hand (LDS returns in order).
    // The state lives at function scope so that the pipelined loop can issue the FIRST PF column tiles of P.K(t) in the tail
    // of the score product of tile t+1 (no pipeline-fill bubble between the two MFMA phases).
#define MOMA_K2_ROT 1          // pipelined loop: next tile's first A fragments requested behind the end-of-iteration barrier
#define MOMA_K2_PF 2
    constexpr int PF = MOMA_K2_PF;                     // column tiles of transposed reads in flight
    s16x4 kb[PF + 1][4];
    unsigned ba[4][2];
    auto pv_setup = [&](const char* buf) __attribute__((always_inline)) {
        const unsigned b0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)buf + b_off;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            ba[c4][0] = b0 ^ (c4 << 6);
            ba[c4][1] = b0 ^ ((c4 << 6) | 32);
        }
    };
    auto issue_lo = [&](int c) __attribute__((always_inline)) {
        s16x4* k4 = kb[c % (PF + 1)];
        const int imm = (c >> 2) * 8192;
        (void)imm;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[0]) : "v"(ba[c & 3][0]), "i"(imm) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[1]) : "v"(ba[c & 3][1]), "i"(imm + 2048) : "memory");
    };
    auto issue_hi = [&](int c) __attribute__((always_inline)) {
        s16x4* k4 = kb[c % (PF + 1)];
        const int imm = (c >> 2) * 8192;
        (void)imm;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[2]) : "v"(ba[c & 3][0]), "i"(imm + 4096) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[3]) : "v"(ba[c & 3][1]), "i"(imm + 6144) : "memory");
    };

    // pvpre_tag (true_type): the last 2*PF gaps of the product also issue the first PF column tiles of the transposed reads of
    // `pvbuf` (the tile P.K runs on next), two reads per gap; the product's own waits then count those younger reads too, and
    // the s_nops behind the last MFMA are dropped (the caller keeps every VALU reader of x behind >= 4 further MFMAs).
#define MOMA_K2_RD 4      // (with the rotated loop the first fragments are early anyway; 6 and 8 spill at D = 512)
    constexpr int RD = MOMA_K2_RD < KS ? MOMA_K2_RD : KS;                             // LDS read distance in k-steps (RD x 32 cycles)
    static_assert(RD <= 12 && RD <= KS, "lgkmcnt is a 4-bit counter");
    // A fragments of the score product (keys, row-wise): a ring of RD k-steps, requested by inline asm and counted by hand
    // (hipcc's own waits were lgkmcnt(0) every 8 steps: the latency of the newest read exposed 4x per tile).  `score_begin`
    // requests the first RD k-steps of a tile; the pipelined loop calls it for the NEXT iteration's tile right behind the barrier
    // that makes the tile visible, so the LDS latency of the first fragments passes under the end-of-iteration bookkeeping.
    f32x4 kf[RD];
    unsigned aa[8];
    auto rd = [&](int ks) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ks % RD]) : "v"(aa[ks & 7]), "i"((ks >> 3) * 8192) : "memory");
    };
    auto score_begin = [&](const char* buf) __attribute__((always_inline)) {
        const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)buf + a_off;
#pragma unroll
        for (int c = 0; c < 8; ++c) aa[c] = a0 ^ (c << 5);
#pragma unroll
        for (int ks = 0; ks < RD; ++ks) rd(ks);
        __builtin_amdgcn_sched_barrier(0);
    };

    // pvpre_tag (true_type): the last 2*PF gaps of the product also issue the first PF column tiles of the transposed reads of
    // `pvbuf` (the tile P.K runs on next), two reads per gap; the product's own waits then count those younger reads too, and
    // the s_nops behind the last MFMA are dropped (the caller keeps every VALU reader of x behind >= 4 further MFMAs).
    // begun_tag (true_type): score_begin(buf) was already called for this tile.
    auto score = [&](auto refill_tag, auto pvpre_tag, auto begun_tag, const char* buf, f32x16& x, float init, long rkey0,
                     char* rbuf, const char* pvbuf) __attribute__((always_inline)) {
        constexpr bool PVPRE = decltype(pvpre_tag)::value;
        constexpr int NPRE = PVPRE ? 2 * PF : 0;                                  // gaps that carry a lo / hi pair
        // One wave per SIMD issues in order: what sits between two MFMAs runs in the shadow of the first (about 24
        // free issue cycles per 32-cycle MFMA; measured: one ds_read_b128 or one LDS-DMA piece per 4 hides fully),
        // while long runs of non-MFMA work idle the matrix pipe.  So each k-step is "one ds_read_b128 for k-step
        // ks+RD, [one DMA piece every 4th step], one MFMA", in exactly that order (sched_barriers).
        // x starts at -m_ref ("row constant as the initial accumulator"): no subtraction later.
        // refill_tag: 0 = no refill, 1 = refill with a full tile, 2 = refill with the queue's last (partial) tile.
        constexpr int REFILL = decltype(refill_tag)::value;
        static_assert(KS % 4 == 0 && KS / 4 == PPW, "one DMA piece per 4 k-steps");
        if constexpr (!decltype(begun_tag)::value) score_begin(buf);
        // the first MFMA takes the row constant as a separate C tuple (loop-invariant registers; copying it into x first was 8
        // v_mov_b64 per tile at the top of the product, where nothing overlaps them)
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = init;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // reads ks+1 .. min(ks+RD-1, KS-1) may stay in flight, and the transposed reads issued so far in the tail
            const int ahead = (KS - 1 - ks) < (RD - 1) ? (KS - 1 - ks) : (RD - 1);
            const int pre = ks > KS - NPRE ? 2 * (ks - (KS - NPRE)) : 0;
            wait_lgkm(ahead + pre);
            __builtin_amdgcn_sched_barrier(0);
            // inline-asm MFMA pins the score accumulator to VGPRs (through the builtin hipcc puts it in a[0:15] and
            // moves O's first column tile out and back every tile).  Hazards by hand: s_nop before the first MFMA
            // (VALU-written C), s_nops after the last one (VALU readers of D).
            if (ks == 0)
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]), "v"(c0));
            else if (ks == KS - 1 && !PVPRE)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));
            if (ks + RD < KS) rd(ks + RD);
            if constexpr (PVPRE) {
                static_assert(NPRE <= KS && PF <= NCT, "tail gaps and column tiles exist");
                if (ks == KS - NPRE) pv_setup(pvbuf);
                if (ks >= KS - NPRE) {
                    const int j = ks - (KS - NPRE);                            // 0 .. 2*PF-1 : (column tile j/2, lo / hi)
                    if ((j & 1) == 0) issue_lo(j >> 1);
                    else issue_hi(j >> 1);
                }
            }
            if constexpr (REFILL != 0) {
                if ((ks & 3) == 1) dma_piece<D, REFILL == 2>(ks >> 2, dl, queue, rkey0, K, rbuf, wave, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // run `score` with the refill mode chosen once per tile (one branch instead of three per DMA piece)
    auto score_dispatch = [&](const char* buf, f32x16& x, float init, bool refill, int rtile) __attribute__((always_inline)) {
        const long rkey0 = (long)rtile * KT;
        char* rbuf = slot(rtile);
        constexpr std::false_type F{};
        if (!refill) score(std::integral_constant<int, 0>{}, F, F, buf, x, init, rkey0, rbuf, nullptr);
        else if (rkey0 + KT > K) score(std::integral_constant<int, 2>{}, F, F, buf, x, init, rkey0, rbuf, nullptr);
        else score(std::integral_constant<int, 1>{}, F, F, buf, x, init, rkey0, rbuf, nullptr);
    };
    // pipelined loop: first A fragments already requested (score_begin), first transposed reads of P.K on `pvbuf` in the tail
    auto score_dispatch_pv = [&](const char* buf, f32x16& x, float init, bool refill, int rtile, const char* pvbuf)
                                 __attribute__((always_inline)) {
        const long rkey0 = (long)rtile * KT;
        char* rbuf = slot(rtile);
        constexpr std::true_type T{};
        constexpr std::integral_constant<bool, (MOMA_K2_ROT != 0)> R{};
        if (!refill) score(std::integral_constant<int, 0>{}, T, R, buf, x, init, rkey0, rbuf, pvbuf);
        else if (rkey0 + KT > K) score(std::integral_constant<int, 2>{}, T, R, buf, x, init, rkey0, rbuf, pvbuf);
        else score(std::integral_constant<int, 1>{}, T, R, buf, x, init, rkey0, rbuf, pvbuf);
    };
    // keys past K (only in the queue's last tile) get -inf scores; key of register r = (r&3) + 8*(r>>2) + 4*h
    auto mask_tail = [&](f32x16& x, int t) __attribute__((always_inline)) {
        if ((t + 1) * KT > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + (r & 3) + 8 * (r >> 2) + 4 * h >= K) x[r] = NEG_BIG;
        }
    };
    auto pack = [&](const f32x16& p, bf16x8 (&pa)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            pa[s] = bf16x8{(__bf16)p[8 * s + 0], (__bf16)p[8 * s + 1], (__bf16)p[8 * s + 2], (__bf16)p[8 * s + 3],
                           (__bf16)p[8 * s + 4], (__bf16)p[8 * s + 5], (__bf16)p[8 * s + 6], (__bf16)p[8 * s + 7]};
    };
    // softmax numerators of one tile of ABSOLUTE scores, all at once (prologue tile, forward-only loop, rescue):
    //   x <- 2^(x - m_ref), l_run += row sums.  first: fixes the reference.  Otherwise a row whose tile max exceeds its
    //   reference by more than OVERFLOW_THR moves its reference (textbook online softmax; there is no O to rescale on the
    //   paths that can get there: the pipelined loop handles that case in its rescue branch before calling this).
    auto softmax_plain = [&](f32x16& x, int t, bool first) __attribute__((always_inline)) {
        mask_tail(x, t);
        float tmax = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
        tmax = fmaxf(tmax, other_half(tmax));
        mx = fmaxf(mx, tmax);
        if (first) {
            m_ref = tmax + REF_MARGIN;
        } else if constexpr (!WITH_DQ) {
            const bool o = tmax - m_ref > OVERFLOW_THR;
            const float m_new = o ? tmax + REF_MARGIN : m_ref;
            l_run *= __builtin_amdgcn_exp2f(m_ref - m_new);
            m_ref = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = __builtin_amdgcn_exp2f(x[r] - m_ref);
            psum += x[r];
        }
        l_run += psum;
    };
    // ---- O[q, cols] += P[q, keys] . K_tile[keys, cols].  Fragments of column tile c+PF are requested before tile c's
    // MFMAs, so "tile c has arrived" is lgkmcnt(4*PF).  `between(c)` is executed in the shadow of column tile c's two MFMAs
    // (the pipelined loop puts the NEXT tile's softmax there).  prefetched_tag: the first PF column tiles were already requested
    // by the preceding score product (score_dispatch_pv).
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto pv = [&](auto prefetched_tag, const char* buf, const bf16x8 (&pa)[2], auto&& between_a, auto&& between) __attribute__((always_inline)) {
        if constexpr (WITH_DQ) {
            if constexpr (!decltype(prefetched_tag)::value) {
                pv_setup(buf);
#pragma unroll
                for (int c = 0; c < PF; ++c) {
                    issue_lo(c);
                    issue_hi(c);
                }
            }
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                // in flight when tile c's first MFMA issues: all of c+1 .. c+PF-1, nothing of c+PF yet
                const int ahead = (NCT - 1 - c) < (PF - 1) ? (NCT - 1 - c) : (PF - 1);
                if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                s16x4* k4 = kb[c % (PF + 1)];
                const s16x8 k0 = __builtin_shufflevector(k4[0], k4[1], 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8 k1 = __builtin_shufflevector(k4[2], k4[3], 0, 1, 2, 3, 4, 5, 6, 7);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], __builtin_bit_cast(bf16x8, k0), O[c], 0, 0, 0);
                if (c + PF < NCT) issue_lo(c + PF);               // LDS requests in the first MFMA's shadow
                between_a(c);                                      // (the gap behind the first MFMA carries no wait: room for 12 cycles)
                __builtin_amdgcn_sched_barrier(0);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], __builtin_bit_cast(bf16x8, k1), O[c], 0, 0, 0);
                if (c + PF < NCT) issue_hi(c + PF);
                between(c);                                        // one softmax step in the second MFMA's shadow
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // partial O of column tile c in register order, 16 B per lane and store (layout: opart_row above)
    auto opart_dst = [&]() __attribute__((always_inline)) -> uint4* {
        return o_part + ((long)chunk * (Bpad / 32) + bt * 4 + wave) * (long)(NCT * 2 * 64) + lane;
    };
    auto store_tile = [&](uint4* dst, int c) __attribute__((always_inline)) {
        if constexpr (WITH_DQ) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint4 v;
                v.x = pack_bf16x2(O[c][8 * g + 0], O[c][8 * g + 1]);
                v.y = pack_bf16x2(O[c][8 * g + 2], O[c][8 * g + 3]);
                v.z = pack_bf16x2(O[c][8 * g + 4], O[c][8 * g + 5]);
                v.w = pack_bf16x2(O[c][8 * g + 6], O[c][8 * g + 7]);
                dst[(c * 2 + g) * 64] = v;     // (non-temporal stores: same kernel time, +3 us on the combine that reads them back)
            }
        }
    };

    // A workgroup normally makes ONE pass over its chunk.  If one of its waves met a score more than 2^OVERFLOW_THR above a
    // row's fixed reference (its P may have overflowed), the whole workgroup -- the tile ring is shared -- repeats the chunk
    // with the references set to the true row maxima the first pass recorded, which cannot overflow.  Rare, workgroup-uniform.
    // (the second pass is a second copy of the code, not a loop: around a back edge hipcc's wait insertion puts a vmcnt(0)
    // -- a full drain of the tile ring -- in front of the first pass's first softmax)
    auto run_pass = [&](auto repass_tag) __attribute__((always_inline)) {
    constexpr bool repass = decltype(repass_tag)::value;
    if constexpr (repass) issue_ring(std::false_type{});
    if constexpr (WITH_DQ) {
        // O = 0 as the result of an MFMA on zero operands with the inline-constant accumulator 0: 16 matrix instructions
        // issued while the first tile is on its way, instead of 256 accumulator-register writes that hipcc rematerialises
        // between the first tile's scores and the loop.
        bf16x8 zq = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        asm volatile("" : "+v"(zq));                      // (opaque zeros: keeps the MFMAs from being folded away)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            asm volatile("" : "+v"(zq));                  // (a fresh opaque value per tile: no common-subexpression merge)
            O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zq, zq, z, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    l_run = 0.f;
    mx = NEG_BIG;
    ovf = 0;
    // Q and tile t0 have landed (the rest of the ring stays in flight); from here on the Q registers may be read
    wait_first_tile();
    if constexpr (MODE != 2) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    }
    __builtin_amdgcn_s_barrier();

    if constexpr (PIPELINED) {
        // ---- software-pipelined main loop (one wave per SIMD: nothing else hides the softmax's VALU time):
        //   iteration t:  X(t+1) = scores of tile t+1            [matrix pipe; DMA pieces of tile t+3 in its shadow]
        //                 O += P(t).K(t)  ||  P(t+1) = softmax numerators of X(t+1), one register per column tile
        // Live ring slots: t, t+1 ; in flight: t+2, t+3.
        f32x16 xa;
        bf16x8 pa[2];
        score_dispatch(slot(t0), xa, 0.f, t0 + 2 < t1, t0 + 2);          // (tile t0+2 requested in its shadow)
        softmax_plain(xa, t0, !repass);
        pack(xa, pa);
        // tile t0+1 must have landed before the loop's first score (t0+2 may stay in flight)
        wait_tiles_in_flight(max(min(t0 + 2, t1 - 1) - (t0 + 1), 0));
        __builtin_amdgcn_s_barrier();
        // every iteration of the loop has a next tile; the chunk's LAST tile is peeled off below
        const int tlast = t1 - 1;
        if (MOMA_K2_ROT && t0 < tlast) score_begin(slot(t0 + 1));            // first A fragments of the loop's first score product
#pragma unroll 1
        for (int t = t0; t < tlast; ++t) {
            f32x16 xb;
            {
                // the slot of tile t-1 (free since the barrier that ended iteration t-1; never used yet at t = t0) takes tile t+NBUF-1;
                // the product's tail already requests the first column tiles of P.K(t): the two MFMA phases run back to back
                const bool refill = t + NBUF - 1 < t1;
                score_dispatch_pv(slot(t + 1), xb, -m_ref, refill, t + NBUF - 1, slot(t));
            }
            float tmax = NEG_BIG, psum = 0.f;
            // One softmax step (register j of the score tile: running max, 2^x, running sum) per P.K column tile.  xb holds
            // score - m_ref.  The exponential and the sum are inline asm: as plain expressions hipcc sinks them to their use
            // BEHIND the 2*NCT MFMAs of P.K; volatile statements stay between the transposed reads of their column tile.  The
            // sum lags one step (no instruction reads a transcendental result right behind its v_exp_f32).  The steps start
            // SM_SHIFT column tiles into P.K: the score MFMAs were issued without trailing wait states, and no VALU may read
            // their accumulator before >= 4 further MFMAs have gone by.
#define MOMA_K2_SM_SHIFT 2
            constexpr int SM_SHIFT = MOMA_K2_SM_SHIFT;
            // The step is split over the two gaps of a column tile: maximum + exponential (12 issue cycles) behind the first MFMA,
            // whose gap carries only the two reads; the sum behind the second, whose gap also carries the wait (a whole step
            // there made that gap 36 cycles against the MFMA's 32, while the first one idled).
            auto sm_step_a = [&](int j) __attribute__((always_inline)) {
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(tmax) : "v"(xb[j]));       // (pinned here: left to hipcc the maxima sink
                asm volatile("v_exp_f32 %0, %0" : "+v"(xb[j]));                       //  behind the barrier and every x is copied first)
            };
            auto sm_step_b = [&](int j) __attribute__((always_inline)) {
                if (j >= 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(psum) : "v"(xb[j - 1]));
            };
            auto sm_step = [&](int j) __attribute__((always_inline)) { sm_step_a(j); sm_step_b(j); };
            pv(std::true_type{}, slot(t), pa,
               [&](int c) __attribute__((always_inline)) {
                   if (c == SM_SHIFT) mask_tail(xb, t + 1);
                   if (c >= SM_SHIFT && c - SM_SHIFT < 16) sm_step_a(c - SM_SHIFT);
               },
               [&](int c) __attribute__((always_inline)) {
                   if (c >= SM_SHIFT && c - SM_SHIFT < 16) sm_step_b(c - SM_SHIFT);
               });
#pragma unroll
            for (int j = (NCT > SM_SHIFT ? NCT - SM_SHIFT : 0); j < 16; ++j) {
                if (NCT <= SM_SHIFT && j == 0) mask_tail(xb, t + 1);
                sm_step(j);
            }
            psum += xb[15];
            // tile t+2 must have landed (t+3 may stay in flight); every wave is done with slot t
            if (t + NBUF - 1 < t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");     // (one two-way branch, not the general switch)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // the next iteration's first A fragments are requested right here, and the bookkeeping of THIS tile runs under
            // their LDS latency: per-lane-half statistics (tmax is relative to m_ref; the two halves of a row are merged once
            // after the loop -- a cross-half shuffle here would be an LDS round trip on the critical path of every tile) and
            // the bf16 packing of P(t+1)
            if constexpr (MOMA_K2_ROT) score_begin(slot(t + 2));       // (unconditional: behind the loop's last iteration the fragments are simply not used)
            ovf |= (tmax > OVERFLOW_THR) ? 1 : 0;
            mx = fmaxf(mx, tmax + m_ref);                       // back to absolute log2 units
            l_run += psum;
            pack(xb, pa);
        }
        // ---- last tile: O += P.K, with the partial of column tile c-1 converted and stored in the shadow of tile c's MFMAs
        // (nothing else is left to hide there; done after the loop the 256 accumulator reads, 128 conversions and 32 stores
        // of a wave are ~5 k cycles of pure issue).  A repeat pass simply stores again.
        {
            // (a wave whose 32 rows all lie past B -- small or ragged batches -- multiplied zeros: nothing of it is read back)
            const bool live = wave_block_live(bt * (QROWS_WG / 32) + wave, B);
            uint4* dst = opart_dst();
            pv(std::false_type{}, slot(t1 - 1), pa, [&](int) __attribute__((always_inline)) {}, [&](int c) __attribute__((always_inline)) {
                if (c >= 1 && live) store_tile(dst, c - 1);
            });
            if (live) store_tile(dst, NCT - 1);
        }
    } else {
        // ---- plain loop (forward-only and slab variants): score, softmax, (P.K), one tile at a time
#pragma unroll 1
        for (int t = t0; t < t1; ++t) {
            const bool refill = t >= t0 + 1 && t + NBUF - 1 < t1;          // slot of tile t-1, free since the last barrier
            if constexpr (MODE != 0) {
                // slab passes: the tile's complete scores live in the scratch, 16 floats per lane in register order
                float4* xa = reinterpret_cast<float4*>(slab.xs + (((long)(bt * 4 + wave) * ntiles + t) * 64 + lane) * 16);
                f32x16 x;
                if constexpr (MODE == 1) {
                    score_dispatch(slot(t), x, 0.f, refill, t + NBUF - 1);
                    if (!slab.first) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const float4 v = xa[g4];
                            x[4 * g4] += v.x; x[4 * g4 + 1] += v.y; x[4 * g4 + 2] += v.z; x[4 * g4 + 3] += v.w;
                        }
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) xa[g4] = make_float4(x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]);
                } else {
                    asm volatile("" : "+v"(xnext));                          // (landed: the wait at the end of the last iteration)
                    x = xnext;
                    if (t + 1 < t1) load_scores(t + 1);                      // older than this iteration's refill pieces
                    if (refill) dma_tile<D>(dl, queue, (long)(t + NBUF - 1) * KT, K, slot(t + NBUF - 1), wave, lane);
                    mask_tail(x, t);
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(x[r] - m_ref);      // m_ref = the row's lse (log2)
                    bf16x8 pa[2];
                    pack(x, pa);
                    pv(std::false_type{}, slot(t), pa, [&](int) __attribute__((always_inline)) {}, [&](int) __attribute__((always_inline)) {});
                }
            } else {
                f32x16 x;
                score_dispatch(slot(t), x, 0.f, refill, t + NBUF - 1);
                softmax_plain(x, t, t == t0);
            }
            // tile t+1 must have landed (tiles t+2.. may stay in flight); every wave must be done with this slot
            if constexpr (MODE == 2) {
                // ... and the scores of tile t+1: only this iteration's refill pieces are younger than they are
                if (refill) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else
            wait_tiles_in_flight(max(min(t + NBUF - 1, t1 - 1) - (t + 1), 0));
            __builtin_amdgcn_s_barrier();
        }
    }

    };
    run_pass(std::false_type{});
    mx = fmaxf(mx, other_half(mx));                                              // (the loop keeps per-half maxima)
    if constexpr (PIPELINED) {
        int* wflag = reinterpret_cast<int*>(smem + NBUF * TILE_BYTES);         // 4 words behind the ring
        const int any = __any(ovf) ? 1 : 0;
        if (lane == 0) wflag[wave] = any;
        __syncthreads();
        if ((wflag[0] | wflag[1] | wflag[2] | wflag[3]) != 0) {
            m_ref = mx;                                                          // true row maxima of the chunk
            __syncthreads();                                                     // flags read before the ring is refilled
            run_pass(std::true_type{});
            mx = fmaxf(mx, other_half(mx));
        }
    }

    // ---- partials per (chunk, query row): m, l, true max ; O[chunk][row][D] (relative to m)
    const float l_tot = l_run + other_half(l_run);
    if (MODE == 0 && h == 0) {
        m_part[prow + n] = m_ref;
        l_part[prow + n] = l_tot;
        x_part[prow + n] = mx;
    }
    if constexpr (WITH_DQ && !PIPELINED) {
        uint4* dst = opart_dst();
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            store_tile(dst, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int D, bool WITH_DQ>
__global__ __launch_bounds__(256, 1) void infonce_flash_kernel(const uint4* __restrict__ qpack,
                                                               const bf16_raw* __restrict__ queue, int B, int K,
                                                               int nbt, int nchunk, int tiles_per_chunk, int Bpad,
                                                               uint4* __restrict__ o_part, float* __restrict__ m_part,
                                                               float* __restrict__ l_part, float* __restrict__ x_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    infonce_flash_body<D, WITH_DQ>(blockIdx.x, smem, qpack, queue, B, K, nbt, nchunk, tiles_per_chunk, Bpad, o_part, m_part,
                                   l_part, x_part);
}

// ---- small batches (B <= 64: the reference's default --batch_size, BASELINE configs[4] per rank) ------------------------------
// At B <= 64 the problem is HBM-bound (4B/2 = 128 flop per queue byte against a ridge of ~312) and the B = 256 decomposition leaves
// two of a workgroup's four waves on pad rows while the other two run the full 2 x 1024 MFMA cycles per 32-key tile.  Here the four
// waves of a workgroup are 2 row blocks x 2 KEY HALVES of every tile: wave (rb, kh) takes keys 16 kh .. 16 kh + 15 of each 32-key
// tile for the query rows 32 rb .. 32 rb + 31 -- half the matrix work, half the LDS reads and half the softmax per wave and tile,
// with the same tile ring, the same LDS image and the same DMA as the one-pass kernel (all four waves still fetch the whole
// tile together).  No exchange between the halves: each wave runs its own online softmax over its keys and leaves its own
// partial (m, l, max, O) as "virtual chunk" 2 * chunk + kh; the combine kernel merges them like any other chunks.
//   scores : X[key, q] by v_mfma_f32_16x16x32_bf16 (A = 16 keys x 32 k by ds_read_b128, B = Q: two MFMAs per k-step for the 32
//            query rows).  Lane l of the results holds A-rows 4 (l>>4) + r of column l & 15; ONE v_permlane16_swap per register
//            pair (X0[r], X1[r]) leaves lane L with query row L & 31 and A-rows 8 (L>>5) + 0..3 (X0) / + 4..7 (X1).  The A-rows are
//            fed from the tile's key rows in the order rho = {0-3, 8-11, 4-7, 12-15}, so that lane (q, h) ends up with keys
//            {4h .. 4h+3} and {8 + 4h .. 8 + 4h + 3} of the wave's half -- exactly the k order key(0,h,j) = 8 (j>>2) + 4h + (j&3)
//            in which the one-pass kernel's transposed reads deliver the B operand of
//   P.K    : O[q, :] += P[q, 16 keys] . K_half  -- ONE v_mfma_f32_32x32x16_bf16 per 32-column tile (the 16 keys are its k).
// Q comes from the same packed image (infonce_qpack_kernel / K1's proj epilogue), gathered per lane in the 16x16x32 B layout.
template <int D>
__global__ __launch_bounds__(256, 1) void infonce_small_kernel(const uint4* __restrict__ qpack, const bf16_raw* __restrict__ queue,
                                                               int B, int K, int tiles_per_chunk, int Bpad,
                                                               uint4* __restrict__ o_part, float* __restrict__ m_part,
                                                               float* __restrict__ l_part, float* __restrict__ x_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16;        // k-steps of the packed Q image
    constexpr int KS2 = D / 32;       // k-steps of the 16x16x32 score product
    constexpr int NCT = D / 32;       // 32-column tiles of O
    constexpr int TILE_BYTES = KT * D * 2;
    constexpr int PPW = D / 64;       // DMA pieces per wave and tile
    static_assert(3 * PPW + 2 * KS2 <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = wave & 1, kh = wave >> 1;
    const int n = lane & 31, h = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const int chunk = blockIdx.x;
    const int ntiles = (K + KT - 1) / KT;
    const int t0 = chunk * tiles_per_chunk;
    const int t1 = min(t0 + tiles_per_chunk, ntiles);
    const DmaLane dl = dma_lane_terms<D>(lane, wave, (unsigned)(D * 2));
    const long prow = (long)chunk * Bpad + rb * 32;
    const bool live = wave_block_live(rb, B);

    // ---- Q fragments (B operand of the 16x16x32 product): lane l holds Q[32 rb + 16 m + (l&15)][32 s + 8 (l>>4) + j]
    //      = element (k-step 2s + (g>>1), lane 16m + (l&15) + 32 (g&1)) of the packed image; inline asm, counted by hand
    bf16x8 qf[2][KS2];
    {
        const char* qb = reinterpret_cast<const char*>(qpack + (long)rb * KS * 64) + (g >> 1) * 1024 + (g & 1) * 512 + i16 * 16;
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[m][s]) : "v"(qb + (s >> 1) * 4096), "n"((s & 1) * 2048 + m * 256) : "memory");
        }
    }
    auto slot = [&](int t) __attribute__((always_inline)) -> char* { return smem + ((unsigned)(t - t0) & (NBUF - 1)) * TILE_BYTES; };
    const int npro = min(t1 - t0, NBUF - 1);                  // tiles requested up front: t0 .. t0+2 (t0+3 follows in tile t0's shadow)
    // (requests and first wait branch on the same nested wave-uniform predicates, as in the one-pass kernel: scripts/audit_isa.py)
    const bool has1 = npro > 1, has2 = npro > 2;
    auto issue_ring = [&]() __attribute__((always_inline)) {
        int tb = t0;
        asm volatile("" : "+s"(tb));
        auto req = [&](int j) __attribute__((always_inline)) {
            dma_tile<D, MOMA_K2_SMALL_AUX>(dl, queue, (long)(tb + j) * KT, K, smem + j * TILE_BYTES, wave, lane);
        };
        static_assert(NBUF - 1 == 3, "three tiles up front");
        req(0);
        if (has1) {
            req(1);
            if (has2) req(2);
        }
    };
    issue_ring();

    f32x16 O[NCT];
    float l_run = 0.f, mx = NEG_BIG, m_ref = NEG_BIG;
    int ovf = 0;

    // per-lane LDS offsets.  Row read (A operand, 16x16x32): key row rho(l & 15) + 16 kh, 16-B chunk 4 s + g of the row:
    //   chunk ((s&3) << 2 | g) ^ swz(row) = lane part ^ ((s&3) << 6 in bytes); segment s >> 2 as the immediate.
    const int arow = ((i16 & 3) | ((i16 & 4) << 1) | ((i16 & 8) >> 1)) + 16 * kh;          // rho: A-rows 0-3, 4-7, 8-11, 12-15 <- keys 0-3, 8-11, 4-7, 12-15
    const int a_off = arow * 256 + (((swz(arow) & 12) | ((swz(arow) & 3) ^ g)) << 4);
    // transposed read (B operand of P.K), as in the one-pass kernel; k-step 0 of the tile's half: + 16 kh rows
    int b_off;
    {
        const int q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
        const int e = (2 * g2 + (p >> 1)) ^ h;
        b_off = (4 * h + q4) * 256 + 8 * (p & 1) + (((q4 << 2) | e) << 4) + kh * 4096;
    }
    // exchange words of the two key halves of a row block (behind the ring and its 4 overflow words): row maxima
    const unsigned xch_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(smem + NBUF * TILE_BYTES + 16);
    const unsigned xch_mine = xch_base + ((rb * 2 + kh) * 32 + n) * 4, xch_other = xch_base + ((rb * 2 + (kh ^ 1)) * 32 + n) * 4;
    constexpr int RD = 4;                                                                 // row reads in flight (k-steps)
    constexpr int PF = 2;                                                                 // column tiles of transposed reads in flight
    f32x4 kf[RD];
    unsigned aa[4];
    auto rd = [&](int s) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[s % RD]) : "v"(aa[s & 3]), "i"((s >> 2) * 8192) : "memory");
    };
    s16x4 kb[PF + 1][2];
    unsigned ba[4][2];
    auto issue_tr = [&](int c) __attribute__((always_inline)) {
        s16x4* k2 = kb[c % (PF + 1)];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k2[0]) : "v"(ba[c & 3][0]), "i"((c >> 2) * 8192) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k2[1]) : "v"(ba[c & 3][1]), "i"((c >> 2) * 8192 + 2048) : "memory");
    };
    auto wait_lgkm = [&](int c) __attribute__((always_inline)) {
        switch (c) {
            case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        }
    };
    auto wait_tiles_in_flight = [&](int j) __attribute__((always_inline)) {
        if (j >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (j == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    typedef __attribute__((ext_vector_type(8))) short s16x8;

    // One pass over the chunk; `repass`: the references are the true row maxima of the first pass (cannot overflow).
    auto run_pass = [&](auto repass_tag) __attribute__((always_inline)) {
        constexpr bool repass = decltype(repass_tag)::value;
        if constexpr (repass) issue_ring();
        {
            bf16x8 zq = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            asm volatile("" : "+v"(zq));
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                asm volatile("" : "+v"(zq));
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zq, zq, z, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run = 0.f;
        mx = NEG_BIG;
        ovf = 0;
        if (has1) {                                           // Q and tile t0 have landed
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KS2; ++s) { asm volatile("" : "+v"(qf[0][s])); asm volatile("" : "+v"(qf[1][s])); }
        __builtin_amdgcn_s_barrier();
        // ---- scores of this wave's 16 keys x 32 query rows of the tile in `buf`.  refill_tag: 0 = no refill, 1 = a full tile, 2 =
        //      the queue's last (partial) tile into `rbuf` -- chosen once per tile, not per DMA piece
        auto score = [&](auto refill_tag, const char* buf, long rkey0, char* rbuf, f32x4& x0, f32x4& x1) __attribute__((always_inline)) {
            constexpr int REFILL = decltype(refill_tag)::value;
            {
                const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)buf + a_off;
#pragma unroll
                for (int c = 0; c < 4; ++c) aa[c] = a0 ^ (c << 6);
#pragma unroll
                for (int s = 0; s < RD; ++s) rd(s);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                // (older LDS requests -- the first transposed reads of P.K, issued ahead of this product -- complete first: the
                //  counter only has to leave room for the younger row reads)
                const int ahead = (KS2 - 1 - s) < (RD - 1) ? (KS2 - 1 - s) : (RD - 1);
                wait_lgkm(ahead);
                __builtin_amdgcn_sched_barrier(0);
                // inline asm (the fragments are asm-loaded registers: nothing may copy them before the wait above); the two
                // accumulate chains alternate; the last MFMA carries the wait states in front of the VALU readers (8-pass: 12)
                if (s == 0) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(x0) : "v"(kf[s % RD]), "v"(qf[0][s]));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(x1) : "v"(kf[s % RD]), "v"(qf[1][s]));
                } else if (s == KS2 - 1) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(x0) : "v"(kf[s % RD]), "v"(qf[0][s]));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 11" : "+v"(x1) : "v"(kf[s % RD]), "v"(qf[1][s]));
                } else {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(x0) : "v"(kf[s % RD]), "v"(qf[0][s]));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(x1) : "v"(kf[s % RD]), "v"(qf[1][s]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s + RD < KS2) rd(s + RD);
                if constexpr (REFILL != 0) {
                    if ((s & 1) == 1) dma_piece<D, REFILL == 2, MOMA_K2_SMALL_AUX>(s >> 1, dl, queue, rkey0, K, rbuf, wave, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto score_tile = [&](const char* buf, bool refill, int rtile, f32x4& x0, f32x4& x1) __attribute__((always_inline)) {
            const long rkey0 = (long)rtile * KT;
            char* rbuf = slot(rtile);
            if (!refill) score(std::integral_constant<int, 0>{}, buf, rkey0, rbuf, x0, x1);
            else if (rkey0 + KT > K) score(std::integral_constant<int, 2>{}, buf, rkey0, rbuf, x0, x1);
            else score(std::integral_constant<int, 1>{}, buf, rkey0, rbuf, x0, x1);
        };
        // first transposed reads of P.K on the tile in `buf` (issued AHEAD of whatever hides their latency)
        auto pv_begin = [&](const char* buf) __attribute__((always_inline)) {
            const unsigned b0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)buf + b_off;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                ba[c4][0] = b0 ^ (c4 << 6);
                ba[c4][1] = b0 ^ ((c4 << 6) | 32);
            }
#pragma unroll
            for (int c = 0; c < PF; ++c) issue_tr(c);
            __builtin_amdgcn_sched_barrier(0);
        };
        // O[q, cols] += P[q, 16 keys] . K_half[16 keys, cols]; `between(c)` runs in the shadow of column tile c's MFMA
        auto pv = [&](const bf16x8& pa, auto&& between) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int ahead = (NCT - 1 - c) < (PF - 1) ? (NCT - 1 - c) : (PF - 1);
                wait_lgkm(2 * ahead);
                __builtin_amdgcn_sched_barrier(0);
                s16x4* k2 = kb[c % (PF + 1)];
                const s16x8 kk = __builtin_shufflevector(k2[0], k2[1], 0, 1, 2, 3, 4, 5, 6, 7);
                O[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, __builtin_bit_cast(bf16x8, kk), O[c], 0, 0, 0);
                if (c + PF < NCT) issue_tr(c + PF);
                between(c);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // lane (q = L & 31, h) after the swaps: keys 4h + r (e[r]) and 8 + 4h + r (e[4 + r]) of the wave's half
        auto swap_pair = [&](const f32x4& x0, const f32x4& x1, int r, float (&e)[8]) __attribute__((always_inline)) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0[r]), __float_as_uint(x1[r]), false, false);
            e[r] = __uint_as_float(sw[0]);
            e[4 + r] = __uint_as_float(sw[1]);
        };
        auto mask_tail = [&](float (&e)[8], int t) __attribute__((always_inline)) {
            if ((t + 1) * KT > K) {                           // keys past K (the queue's last tile)
                const int kbase = t * KT + 16 * kh + 4 * h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kbase + r >= K) e[r] = NEG_BIG;
                    if (kbase + 8 + r >= K) e[4 + r] = NEG_BIG;
                }
            }
        };
        auto pack8 = [&](const float (&e)[8]) __attribute__((always_inline)) -> bf16x8 {
            return bf16x8{(__bf16)e[0], (__bf16)e[1], (__bf16)e[2], (__bf16)e[3], (__bf16)e[4], (__bf16)e[5], (__bf16)e[6], (__bf16)e[7]};
        };

        // ---- software pipeline (one wave per SIMD: nothing else hides the softmax):
        //   prologue     X(t0), softmax all at once (fixes the shared reference of the two key halves)
        //   iteration t  X(t+1) = scores of tile t+1 [DMA pieces of tile t+3 in its shadow]
        //                O += P(t) . K(t)   ||   P(t+1) = softmax numerators of X(t+1), one value per column tile's MFMA
        //   epilogue     O += P(t1-1) . K(t1-1)
        // Live ring slots: t, t+1; in flight: t+2, t+3 (tile t+3 goes into the slot of tile t-1, free since the last barrier).
        f32x4 x0, x1;
        float e[8];
        bf16x8 pa;
        score_tile(slot(t0), false, t0, x0, x1);
        {
#pragma unroll
            for (int r = 0; r < 4; ++r) swap_pair(x0, x1, r, e);
            mask_tail(e, t0);
            float tmax = fmaxf(fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])), fmaxf(fmaxf(e[4], e[5]), fmaxf(e[6], e[7])));
            tmax = fmaxf(tmax, other_half(tmax));
            mx = fmaxf(mx, tmax);
            if constexpr (!repass) {
                // the two key halves of a row block share ONE fixed reference (the 32-key tile's row maximum + the margin, as the
                // one-pass kernel): their O partials then add up without rescaling when the halves are merged behind the loop
                // (inline asm + raw barrier: __syncthreads() would drain vmcnt, i.e. wait for the whole tile ring)
                float other;
                asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "v"(xch_mine), "v"(tmax) : "memory");
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(other) : "v"(xch_other) : "memory");
                const float both = fmaxf(tmax, other);
                m_ref = both > 0.5f * NEG_BIG ? both + REF_MARGIN : 0.f;
            }
            ovf |= (tmax - m_ref > OVERFLOW_THR) ? 1 : 0;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                e[r] = __builtin_amdgcn_exp2f(e[r] - m_ref);
                psum += e[r];
            }
            l_run += psum;
            pa = pack8(e);
        }
        // tile t0+1 must have landed before the loop's first score (t0+2 may stay in flight)
        wait_tiles_in_flight(max(min(t0 + 2, t1 - 1) - (t0 + 1), 0));
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = t0; t < t1 - 1; ++t) {
            pv_begin(slot(t));                                // (older than every read of the score product below)
            score_tile(slot(t + 1), t + NBUF - 1 < t1, t + NBUF - 1, x0, x1);
            float tmax = NEG_BIG, psum = 0.f;
            pv(pa, [&](int c) __attribute__((always_inline)) {
                // the softmax of tile t+1 in the shadows of P.K(t): the swaps behind the first two MFMAs (the score product's last
                // MFMA already carries the wait states its readers need), then one value per column tile
                if (c == 0) { swap_pair(x0, x1, 0, e); swap_pair(x0, x1, 1, e); }
                if (c == 1) { swap_pair(x0, x1, 2, e); swap_pair(x0, x1, 3, e); mask_tail(e, t + 1); }
                if (c >= 2 && c < 10 && c - 2 < 8) {
                    const int r = c - 2;
                    tmax = fmaxf(tmax, e[r]);
                    e[r] = __builtin_amdgcn_exp2f(e[r] - m_ref);
                    psum += e[r];
                }
            });
            if constexpr (NCT < 10) {                         // narrow rows: fewer column tiles than softmax steps
#pragma unroll
                for (int r = (NCT > 2 ? NCT - 2 : 0); r < 8; ++r) {
                    if (NCT <= 1 && r == 0) { swap_pair(x0, x1, 2, e); swap_pair(x0, x1, 3, e); mask_tail(e, t + 1); }
                    tmax = fmaxf(tmax, e[r]);
                    e[r] = __builtin_amdgcn_exp2f(e[r] - m_ref);
                    psum += e[r];
                }
            }
            // tile t+2 must have landed (t+3 may stay in flight); every wave is done with slot t
            if (t + NBUF - 1 < t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tmax = fmaxf(tmax, other_half(tmax));
            mx = fmaxf(mx, tmax);
            ovf |= (tmax - m_ref > OVERFLOW_THR) ? 1 : 0;
            l_run += psum;
            pa = pack8(e);
        }
        pv_begin(slot(t1 - 1));
        pv(pa, [&](int) __attribute__((always_inline)) {});
    };
    run_pass(std::false_type{});
    {
        int* wflag = reinterpret_cast<int*>(smem + NBUF * TILE_BYTES);         // 4 words behind the ring
        const int any = __any(ovf) ? 1 : 0;
        if (lane == 0) wflag[wave] = any;
        __syncthreads();
        if ((wflag[0] | wflag[1] | wflag[2] | wflag[3]) != 0) {
            float* xch = reinterpret_cast<float*>(smem + NBUF * TILE_BYTES + 16);
            if (h == 0) xch[(rb * 2 + kh) * 32 + n] = mx;                        // (mx: already merged over the lane halves per tile)
            __syncthreads();                                                     // flags read, maxima written, ring idle
            m_ref = fmaxf(mx, xch[(rb * 2 + (kh ^ 1)) * 32 + n]);               // true row maxima over BOTH key halves
            if (m_ref < 0.5f * NEG_BIG) m_ref = 0.f;
            __syncthreads();
            run_pass(std::true_type{});
        }
    }
    // ---- merge the key halves (same reference: plain sums) through the idle ring: each wave of a pair gives the other one half
    //      of its column tiles -- rounded to bf16, the storage format of the partials, in store order -- and sums / stores the half
    //      it keeps, so both waves share the tail (one wave merging and storing everything was 1.5 us slower)
    float l_tot = l_run + other_half(l_run);
    constexpr int HCT = NCT / 2;
    static_assert(NCT % 2 == 0, "column tiles split evenly over the two key halves");
    uint4* xo_mine = reinterpret_cast<uint4*>(smem) + (rb * 2 + kh) * (HCT * 2 * 64);         // 16 KiB per wave at D = 512
    const uint4* xo_other = reinterpret_cast<const uint4*>(smem) + (rb * 2 + (kh ^ 1)) * (HCT * 2 * 64);
    float* xs = reinterpret_cast<float*>(smem + 4 * HCT * 2 * 64 * 16) + rb * 64;             // l [32], max [32] of kh = 1
    auto packed = [&](int c, int g2) __attribute__((always_inline)) -> uint4 {
        uint4 v;
        v.x = pack_bf16x2(O[c][8 * g2 + 0], O[c][8 * g2 + 1]);
        v.y = pack_bf16x2(O[c][8 * g2 + 2], O[c][8 * g2 + 3]);
        v.z = pack_bf16x2(O[c][8 * g2 + 4], O[c][8 * g2 + 5]);
        v.w = pack_bf16x2(O[c][8 * g2 + 6], O[c][8 * g2 + 7]);
        return v;
    };
    auto add_packed = [&](int c, int g2, const uint4& v) __attribute__((always_inline)) {
        O[c][8 * g2 + 0] += __uint_as_float(v.x << 16); O[c][8 * g2 + 1] += __uint_as_float(v.x & 0xffff0000u);
        O[c][8 * g2 + 2] += __uint_as_float(v.y << 16); O[c][8 * g2 + 3] += __uint_as_float(v.y & 0xffff0000u);
        O[c][8 * g2 + 4] += __uint_as_float(v.z << 16); O[c][8 * g2 + 5] += __uint_as_float(v.z & 0xffff0000u);
        O[c][8 * g2 + 6] += __uint_as_float(v.w << 16); O[c][8 * g2 + 7] += __uint_as_float(v.w & 0xffff0000u);
    };
    uint4* dst = o_part + ((long)chunk * (Bpad / 32) + rb) * (long)(NCT * 2 * 64) + lane;
    // keep_tag: the half of the column tiles this wave keeps (0: tiles 0 .. HCT-1, 1: HCT .. NCT-1) -- compile-time register indices
    auto give = [&](auto keep_tag) __attribute__((always_inline)) {
        constexpr int G0 = decltype(keep_tag)::value ? 0 : HCT;
#pragma unroll
        for (int cc = 0; cc < HCT; ++cc) {
            xo_mine[(cc * 2 + 0) * 64 + lane] = packed(G0 + cc, 0);
            xo_mine[(cc * 2 + 1) * 64 + lane] = packed(G0 + cc, 1);
            __builtin_amdgcn_sched_barrier(0);       // (tile by tile: left alone hipcc reads all 256 accumulators out first and spills)
        }
    };
    auto take = [&](auto keep_tag) __attribute__((always_inline)) {
        constexpr int K0 = decltype(keep_tag)::value ? HCT : 0;
#pragma unroll
        for (int c2 = 0; c2 < HCT; c2 += 2) {                                          // four requests, then their sums and stores
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = xo_other[(c2 * 2 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                add_packed(K0 + c2 + (u >> 1), u & 1, v[u]);
                dst[((K0 + c2 + (u >> 1)) * 2 + (u & 1)) * 64] = packed(K0 + c2 + (u >> 1), u & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();                                                                   // every wave is done with the ring
    if (live) {
        if (kh == 0) give(std::false_type{});
        else give(std::true_type{});
    }
    if (kh == 1 && h == 0) { xs[n] = l_tot; xs[32 + n] = mx; }
    __syncthreads();
    if (kh == 0) {
        l_tot += xs[n];
        mx = fmaxf(mx, xs[32 + n]);
        if (h == 0) {
            m_part[prow + n] = m_ref;
            l_part[prow + n] = l_tot;
            x_part[prow + n] = mx;
        }
    }
    if (live) {
        if (kh == 0) take(std::false_type{});
        else take(std::true_type{});
    }
}

// ---- wide queues (d > 512): slab passes ------------------------------------------------------------------------
template <int D, int MODE>
__global__ __launch_bounds__(256, 1) void infonce_slab_kernel(const uint4* __restrict__ qpack, const bf16_raw* __restrict__ queue,
                                                              int B, int K, int nbt, int nchunk, int tiles_per_chunk, int Bpad,
                                                              uint4* __restrict__ o_part, SlabArgs slab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    infonce_flash_body<D, MODE == 2, MODE>(blockIdx.x, smem, qpack, queue, B, K, nbt, nchunk, tiles_per_chunk, Bpad,
                                           o_part, nullptr, nullptr, nullptr, slab);
}

// ---- wide queues, scores in ONE pass (d = NSEG x 128 columns, 512 < d <= 1536) ---------------------------------------------
// The score product needs no O accumulator, so the whole 32 x d Q block of a wave fits the register file as the B operand:
// the first 64 fragments (1024 columns) in the 256 AGPRs, the rest in VGPRs (MFMA A/B operands may be AGPRs).  A key tile
// (32 keys x d, up to 96 KiB) does not fit a ring, so the LDS-DMA unit is a PHASE of PSEG segments (32 keys x PSEG*128 columns,
// 3 phase buffers): x accumulates over the NSEG/PSEG phases of a tile; behind the tile's last phase the wave updates its
// online (max, sum) per lane half and stores P~ = bf16(2^(x - r)) in A-operand order for the P.K pass (below: `sweep`).
// Leaves per (chunk, row): m = true maximum, l = sum 2^(x - m), max = m -- the partial format of the one-pass kernel -- and r.
// QP = 2 (rows of 2 * NSEG segments, e.g. d = 2048 = 2 x 8): the Q block of a wave is twice the register file, so the workgroup
// sweeps its key chunk once per HALF of the columns -- sweep A with Q[:, :d/2] resident leaves the partial scores (fp32, tile
// register order, 4 KiB per wave and tile) in the scratch `xq`, sweep B with Q[:, d/2:] resident starts every tile's accumulator
// from them (requested one phase ahead) and finishes the tile as above.  The partials are written and re-read by the same wave
// (L2-resident: 16 KiB per workgroup and tile against 128 KiB of keys); the keys cross L2 -> CU once, as for QP = 1.
template <int NSEG, int PSEG, int QP = 1>
__global__ __launch_bounds__(256, 1) void infonce_wide_scores_kernel(const uint4* __restrict__ qpack,
                                                                     const bf16_raw* __restrict__ queue, int B, int K, int nbt,
                                                                     int nchunk, int tiles_per_chunk, int Bpad,
                                                                     uint4* __restrict__ ps, float* __restrict__ m_part,
                                                                     float* __restrict__ l_part, float* __restrict__ x_part,
                                                                     float* __restrict__ r_part, uint4* __restrict__ xq) {
    static_assert(NSEG % PSEG == 0 && PSEG <= 6, "phases tile the row; 3 phase buffers fit 160 KiB");
    static_assert(QP == 1 || QP == 2, "one or two register passes of Q");
    constexpr int KS = NSEG * 8;                    // k-steps of a complete score tile
    constexpr int NA = KS < 64 ? KS : 64;           // Q fragments kept in AGPRs
    constexpr int NV = KS - NA;                     // ... and in VGPRs
    constexpr int NPH = NSEG / PSEG;                // phases per key tile
    constexpr int PSTEPS = PSEG * 8;                // k-steps per phase
    constexpr int PPP = PSEG * 2;                   // LDS-DMA pieces per wave and phase
    constexpr int PH_BYTES = PSEG * 8192;
    constexpr int RD = 4;                           // A fragments in flight
    static_assert(NV <= 32 && PSTEPS % 4 == 0 && PPP == PSTEPS / 4, "one DMA piece per 4 k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
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
    const unsigned pitch = NSEG * QP * 256;
    const DmaLane dl = dma_lane_terms<128>(lane, wave, pitch);
    const long prow = (long)chunk * Bpad + bt * QROWS_WG + wave * 32;

    // ---- Q (of column half `qp`): inline-asm loads (counted by hand: LDS-DMA and ordinary loads share vmcnt), AGPR part first
    bf16x8 qa[NA];
    bf16x8 qv[NV > 0 ? NV : 1];
    auto load_q = [&](int qp) __attribute__((always_inline)) {
        const char* qb = reinterpret_cast<const char*>(qpack + ((long)(bt * 4 + wave) * (KS * QP) + qp * KS) * 64 + lane);
#pragma unroll
        for (int f = 0; f < KS; ++f) {
            if (f < NA)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(qa[f < NA ? f : 0]) : "v"(qb + (f >> 2) * 4096), "n"((f & 3) * 1024) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qv[f >= NA ? f - NA : 0]) : "v"(qb + (f >> 2) * 4096), "n"((f & 3) * 1024) : "memory");
        }
    };
    auto pin_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NA; ++f) asm volatile("" : "+a"(qa[f]));
#pragma unroll
        for (int f = 0; f < NV; ++f) asm volatile("" : "+v"(qv[f]));
    };
    load_q(0);
    int colb = 0;                                    // byte offset of the resident column half in a key row (wave-uniform)
    // ---- partial scores of tile t (QP = 2): 4 x 16 B per lane, lane-contiguous kilobytes
    f32x4 xn[4];
    auto xq_ptr = [&](int t) __attribute__((always_inline)) {
        return reinterpret_cast<char*>(xq) + (((long)(bt * 4 + wave) * ((K + KT - 1) / KT) + t) * 4096 + lane * 16);
    };
    auto load_x = [&](int t) __attribute__((always_inline)) {
        const char* xb = xq_ptr(t);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xn[i]) : "v"(xb), "n"(i * 1024) : "memory");
    };
    // ---- LDS-DMA of phase (tile, p) into a phase buffer: piece i of this wave = segment i>>1, row group (i&1)*4 + wave
    auto dma_piece_ph = [&](int i, int tile, int p, char* buf) __attribute__((always_inline)) {
        const int sg = i >> 1, rg = (i & 1) * 4 + wave;
        const long key0 = (long)tile * KT;
        const char* src = reinterpret_cast<const char*>(queue) + key0 * (long)pitch + (colb + (p * PSEG + sg) * 256);
        unsigned off = dl.term;
        if (key0 + KT > K) {                        // queue's last, partial tile: clamp rows past K (masked later)
            const int rl = lane >> 4;
            const int row = min(rg * 4 + rl, (int)(K - 1 - key0));
            off = off - (unsigned)rl * pitch + (unsigned)row * pitch;
        } else {
            src += rg * 4 * (long)pitch;
        }
        char* dst = buf + sg * 8192 + rg * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // global phase index g = (tile - t0) * NPH + p lives in buffer g % 3
    const int nph_total = (t1 - t0) * NPH;
    auto issue_first_phases = [&]() __attribute__((always_inline)) {          // phases 0 and 1 up front
        int tb = t0;
        asm volatile("" : "+s"(tb));                   // (opaque: the two sweeps do not share hoisted source addresses)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g < nph_total) {
#pragma unroll
                for (int i = 0; i < PPP; ++i) dma_piece_ph(i, tb + g / NPH, g % NPH, smem + g * PH_BYTES);
            }
        }
    };
    issue_first_phases();
    const int a_off = n * 256 + ((swz(n) ^ h) << 4);
    f32x4 kf[RD];
    unsigned aa[8];
    auto rd = [&](int j) __attribute__((always_inline)) {           // k-step j of the phase
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[j % RD]) : "v"(aa[j & 7]), "i"((j >> 3) * 8192) : "memory");
    };
    auto wait_lgkm = [&](int c) __attribute__((always_inline)) {
        switch (c) {
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        }
    };
    float m_run = NEG_BIG, l_run = 0.f;
    // Q (and, before sweep B, the first tile's partial scores) and phase 0 have landed (phase 1 may stay in flight)
    auto first_phase_landed = [&]() __attribute__((always_inline)) {
        if (nph_total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    first_phase_landed();
    pin_q();
    __builtin_amdgcn_s_barrier();

    // The scores leave as P~ = bf16(2^(x - r)) in MFMA A-operand order (half the bytes of fp32 scores, and pass 2 needs no
    // exponential): r = an INTEGER reference per (chunk, row) -- ceil(row max of the chunk's first tile) + REF_MARGIN -- so that pass 2
    // moves a tile to its chunk group's reference by an exact power of two.  A score more than OVERFLOW_THR above r (P~ could
    // overflow) makes the workgroup repeat its chunk with r = ceil(true row maximum), which cannot (rare, workgroup-uniform).
    float r_ref = 0.f;
    int ovf = 0;
    // KIND 0: sweep A of QP = 2 (partial scores -> xq);  1: the (first) finishing sweep;  2: its repeat with r = true maxima
    auto sweep = [&](auto kind_tag) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool FIRST = KIND == 1;
    constexpr bool FROM_X = QP == 2 && KIND != 0;    // accumulators start from the other half's partial scores
    int bi = 0;                                      // buffer of the current phase
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        f32x16 x;
#pragma unroll
        for (int p = 0; p < NPH; ++p) {
            const char* buf = smem + bi * PH_BYTES;
            const int bfree = bi == 0 ? 2 : bi - 1;  // buffer of the previous phase: free since the last barrier
            char* rbuf = smem + bfree * PH_BYTES;
            const int rt = t + (p + 2) / NPH, rp = (p + 2) % NPH;
            // (ONE scalar for the request and for the wait that counts it: hipcc otherwise carries the flag through a vector
            //  register and re-derives it, and scripts/audit_isa.py can no longer tell that the two branches agree)
            const bool refill = __builtin_amdgcn_readfirstlane((int)(rt < t1)) != 0;
            {
                const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)buf + a_off;
#pragma unroll
                for (int c = 0; c < 8; ++c) aa[c] = a0 ^ (c << 5);
            }
#pragma unroll
            for (int j = 0; j < RD; ++j) rd(j);
            // (sweep B: the next tile's partial scores, requested in front of this phase's refill so that the wait at the phase's
            //  end covers them; their registers were consumed by this tile's first MFMA a phase ago)
            if (FROM_X && p == NPH - 1 && t + 1 < t1) load_x(t + 1);
            // refill requested at the TOP of the phase: the kernel is bound by (bytes in flight) / (memory latency), and pieces of
            // phase g+2 spread over the MFMAs of phase g are in flight half a phase less (measured: 69 -> 64 us forward-only)
            if (refill) {
#pragma unroll
                for (int i = 0; i < PPP; ++i) dma_piece_ph(i, rt, rp, rbuf);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PSTEPS; ++j) {
                const int f = p * PSTEPS + j;
                const int ahead = (PSTEPS - 1 - j) < (RD - 1) ? (PSTEPS - 1 - j) : (RD - 1);
                wait_lgkm(ahead);
                __builtin_amdgcn_sched_barrier(0);
                if (f == 0 && FROM_X) {
                    const f32x16 xin = {xn[0][0], xn[0][1], xn[0][2], xn[0][3], xn[1][0], xn[1][1], xn[1][2], xn[1][3],
                                        xn[2][0], xn[2][1], xn[2][2], xn[2][3], xn[3][0], xn[3][1], xn[3][2], xn[3][3]};
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(x) : "v"(kf[j % RD]), "a"(qa[0]), "v"(xin));
                } else if (f == 0)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(x) : "v"(kf[j % RD]), "a"(qa[0]));
                else if (f < NA)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[j % RD]), "a"(qa[f < NA ? f : 0]));
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[j % RD]), "v"(qv[f >= NA ? f - NA : 0]));
                if (j + RD < PSTEPS) rd(j + RD);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p == NPH - 1) {
                // ---- the tile's scores are complete (the MFMA result needs its wait states before a VALU reads it)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x));
                if constexpr (KIND == 0) {
                    char* xb = xq_ptr(t);
                    const f32x4 v0 = {x[0], x[1], x[2], x[3]}, v1 = {x[4], x[5], x[6], x[7]}, v2 = {x[8], x[9], x[10], x[11]},
                                v3 = {x[12], x[13], x[14], x[15]};
                    // ONE statement, closed by the two wait states a 16-byte store needs before its data registers may be
                    // written again: hipcc pads that hazard only behind stores it emitted itself (scripts/audit_isa.py, check c)
                    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:1024\n\t"
                                 "global_store_dwordx4 %0, %3, off offset:2048\n\tglobal_store_dwordx4 %0, %4, off offset:3072\n\ts_nop 1"
                                 :: "v"(xb), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
                } else {
                if ((t + 1) * KT > K) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + (r & 3) + 8 * (r >> 2) + 4 * h >= K) x[r] = NEG_BIG;
                }
                float tm = x[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, x[r]);
                if constexpr (FIRST) {
                    if (t == t0) r_ref = ceilf(fmaxf(tm, other_half(tm))) + REF_MARGIN;       // (row-uniform: both lane halves)
                    ovf |= (tm - r_ref > OVERFLOW_THR) ? 1 : 0;
                    const float mn = fmaxf(m_run, tm);
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(x[r] - mn);
                    l_run = l_run * __builtin_amdgcn_exp2f(m_run - mn) + sum;
                    m_run = mn;
                }
                uint4 w0, w1;
                {
                    unsigned pk[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = __builtin_amdgcn_exp2f(x[2 * i] - r_ref), b = __builtin_amdgcn_exp2f(x[2 * i + 1] - r_ref);
                        pk[i] = pack_bf16x2(a, b);
                    }
                    w0 = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    w1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                }
                uint4* pd = ps + (((long)(bt * 4 + wave) * ntiles + t) * 64 + lane) * 2;
                pd[0] = w0;
                pd[1] = w1;
                }
            }
            // the next phase must have landed.  Certainly younger than its pieces: this phase's refill pieces and, behind a tile's
            // last phase, the 4 score stores just issued (the previous tile's stores may be younger too: not counted, i.e.
            // waited for -- they are a phase old)
            if (refill) {
                if (p == NPH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPP + (KIND == 0 ? 4 : 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPP) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (FROM_X && p == NPH - 1) asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]));
            __builtin_amdgcn_s_barrier();
            bi = bi == 2 ? 0 : bi + 1;
        }
    }
    };
    // (sweep B's start: the other half of Q over the same registers, the first tile's partial scores, the first two phases)
    auto start_sweep_b = [&](bool reload_q) __attribute__((always_inline)) {
        if (reload_q) load_q(1);
        load_x(t0);
        issue_first_phases();
        first_phase_landed();
        if (reload_q) pin_q();
        asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]));
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (QP == 2) {
        sweep(std::integral_constant<int, 0>{});     // (ends with vmcnt(0) + barrier: the partial scores are written)
        colb = NSEG * 256;
        asm volatile("" : "+s"(colb));
        start_sweep_b(true);
    }
    sweep(std::integral_constant<int, 1>{});
    // ---- per (chunk, row): the two lane halves hold disjoint keys of the same row
    const float mo = other_half(m_run), lo = other_half(l_run);
    const float M = fmaxf(m_run, mo);
    const float L = l_run * __builtin_amdgcn_exp2f(m_run - M) + lo * __builtin_amdgcn_exp2f(mo - M);
    {
        int* wflag = reinterpret_cast<int*>(smem + 3 * PH_BYTES);               // 4 words behind the phase buffers
        const int any = __any(ovf) ? 1 : 0;
        if (lane == 0) wflag[wave] = any;
        __syncthreads();
        if ((wflag[0] | wflag[1] | wflag[2] | wflag[3]) != 0) {
            r_ref = ceilf(M);                                                    // true row maximum of the chunk
            __syncthreads();
            if constexpr (QP == 2) {
                start_sweep_b(false);                                            // (Q[:, d/2:] is still resident, xq still holds sweep A)
            } else {
                issue_first_phases();
                first_phase_landed();
                __builtin_amdgcn_s_barrier();
            }
            sweep(std::integral_constant<int, 2>{});
        }
    }
    if (h == 0) {
        m_part[prow + n] = M;
        l_part[prow + n] = L;
        x_part[prow + n] = M;
        r_part[prow + n] = r_ref;
    }
}

// ---- wide queues, P.K by COLUMN RANGES ---------------------------------------------------------------------------------------
// Workgroup = 8 row-waves (256 query rows) x ONE range of 256 columns (2 segments): the keys of the range cross L2 -> CU once per
// 256 rows, the scores once per range; wave w owns O[32 rows x 256 columns] = 128 accumulator registers (two waves per SIMD).
// 32-key tiles in the standard swizzled image (2 segments x 8 KiB per ring slot, 6 slots), so one workgroup barrier serves 16
// MFMAs per wave; keys (2 LDS-DMA pieces per wave) and scores (4 x 16 B per lane, register ring) are requested SD tiles ahead by
// inline asm and counted by hand.  P arrives as bf16 relative to its CHUNK's integer reference (pass 1); the workgroup covers a
// GROUP of cg chunks and accumulates relative to R = the largest reference of the group, so a tile of chunk c is scaled by the
// exact power of two 2^(r_c - R) <= 1 (unpack, multiply, repack: no exponential, no rounding) -- skipped when the whole wave has
// r_c == R.  P of tile t+1 is prepared in the shadow of tile t's MFMAs; the transposed reads run two column tiles ahead of their
// MFMAs (hand-counted lgkmcnt).  Partials per (group, wave block) in the column-slab layout the combine kernel reads.
template <int NSEG>
__global__ __launch_bounds__(512, 1) void infonce_wide_pv2_kernel(const bf16_raw* __restrict__ queue, int B, int K, int nchunk,
                                                                  int tiles_per_chunk, int cg, int Bpad,
                                                                  const uint4* __restrict__ ps, const float* __restrict__ r_part,
                                                                  uint4* __restrict__ o_part, long slab_stride) {
    constexpr int NCR = (NSEG + 1) / 2;             // column ranges of the row
    constexpr int SLOT = 2 * 8192;                  // 32 keys x 256 columns bf16
    constexpr int SD = MOMA_K2_WPV_SD, NB = SD + 1;   // tiles requested ahead (P: 8 registers per tile) and ring slots
    constexpr int OPT = 4;                          // vector-memory operations per wave and tile: 2 loads of P + 2 pieces
    constexpr int CGMAX = 8;                        // chunks per group (the launcher keeps cg <= CGMAX)
    constexpr int PF = 2;                           // column tiles of transposed reads in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int ngroups = (nchunk + cg - 1) / cg;
    const int nrb = (Bpad / 32 + 7) / 8;
    const int nj = nrb * NCR;
    // block id -> (key group, member): blocks go round-robin to the 8 XCDs, so XCD x = id % 8 takes the groups g = 8i + x and all
    // nj members (row block, column range) of a group sit on ONE XCD -- the scores are fetched into its L2 once, not once per
    // range (measured without it: FETCH_SIZE 492 MB instead of 235).  The grid is padded to 8 * ceil(ngroups / 8) * nj blocks.
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int grp = (kk / nj) * 8 + xcd, jj = kk % nj;
    if (grp >= ngroups) return;
    const int rb = jj / NCR, cr = jj % NCR;
    const bool live = rb * 8 + w < Bpad / 32;
    const int wb = min(rb * 8 + w, Bpad / 32 - 1);  // (a wave past the rows repeats the last block and stores nothing)
    const int ntiles = (K + KT - 1) / KT;
    const int c0 = grp * cg, c1 = min(c0 + cg, nchunk);
    const int t0 = c0 * tiles_per_chunk, t1 = min(c1 * tiles_per_chunk, ntiles);
    const int cb = cr * 2, ncs = min(2, NSEG - cb);
    const unsigned pitch = NSEG * 256;

    // scale factors 2^(r_c - R) of the group's chunks, per lane (= per query row), kept in LDS behind the ring
    float* ftab = reinterpret_cast<float*>(smem + NB * SLOT) + w * (CGMAX * 64);
    {
        float R = NEG_BIG;
        for (int c = c0; c < c1; ++c) R = fmaxf(R, r_part[(long)c * Bpad + wb * 32 + n]);
        for (int c = c0; c < c1; ++c) ftab[(c - c0) * 64 + lane] = __builtin_amdgcn_exp2f(r_part[(long)c * Bpad + wb * 32 + n] - R);
    }

    // ---- P~ of tile t: 16 bf16 per lane in A-operand order (8 packed registers)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 sx[SD][2];
    const unsigned xlane = lane * 32;
    // Every request of this kernel is UNCONDITIONAL, so that every wait has a constant count (what scripts/audit_isa.py can
    // prove on the compiled code: a count that follows how many tiles are left cannot be checked path by path).  A request
    // past the group's last tile (`keep` = 0) re-reads ONE 16-byte piece of the last tile's P and keys -- every lane at the same
    // address: one cache line per instruction -- into registers and a ring slot that nothing reads any more.
    auto load_scores = [&](int t, u32x4 (&d)[2], unsigned keep) __attribute__((always_inline)) {
        const char* xb = reinterpret_cast<const char*>(ps) + ((long)wb * ntiles + t) * 2048;          // wave-uniform
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                     : "=&v"(d[0]), "=&v"(d[1]) : "v"(xlane & keep), "s"(xb) : "memory");
    };
    // ---- keys of tile t: wave w issues row group w (4 rows) of both segments (of segment 0 twice when the range has one)
    const int rl = lane >> 4, s16 = lane & 15;
    const unsigned term0 = (unsigned)(rl * pitch + ((s16 ^ (rl << 2)) << 4));
    const unsigned voff = term0 ^ (unsigned)((w & 3) << 4);
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
    auto dma_tile = [&](int t, unsigned lds_slot, unsigned keep) __attribute__((always_inline)) {
        const long key0 = (long)t * KT;
        const unsigned dst = lds_slot + w * 1024;
        const int seg1 = ncs == 2 ? 256 : 0;        // (one-segment range: the second piece repeats the first into the unused half)
        if (key0 + KT <= K) {
            const char* src = reinterpret_cast<const char*>(queue) + key0 * (long)pitch + (cb * 256 + w * 4 * (int)pitch);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff & keep), "s"(src), "s"(dst) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff & keep), "s"(src + seg1), "s"(dst + 8192) : "memory");
        } else {                                    // the queue's last, partial tile: clamp rows past K (their P is 0)
            const int row = min(w * 4 + rl, (int)(K - 1 - key0));
            const unsigned off = voff - (unsigned)rl * pitch + (unsigned)row * pitch;
            const char* src = reinterpret_cast<const char*>(queue) + key0 * (long)pitch + cb * 256;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off & keep), "s"(src), "s"(dst) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(off & keep), "s"(src + seg1), "s"(dst + 8192) : "memory");
        }
    };
    static_assert(SD >= 2, "tile t+1 is awaited with SD - 1 younger requests in flight");
    __builtin_assume(t0 < t1);                          // (no zero-trip copy of the tile loop: grp < ngroups has tiles)
    auto request = [&](int t, u32x4 (&d)[2], unsigned lds_slot) __attribute__((always_inline)) {
        const bool real = t < t1;
        const unsigned keep = real ? ~0u : 0u;
        const int te = real ? t : t1 - 1;               // (t0 < t1: the group has at least one tile)
        load_scores(te, d, keep);
        dma_tile(te, lds_slot, keep);
    };
#pragma unroll
    for (int j = 0; j < SD; ++j) request(t0 + j, sx[j], lds0 + j * SLOT);

    f32x16 O[2][4];
    {
        bf16x8 zq = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                asm volatile("" : "+v"(zq));
                O[s][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zq, zq, z, 0, 0, 0);
            }
    }
    int b_off;
    {
        const int i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3, g2 = (lane >> 4) & 1;
        const int e = (2 * g2 + (p >> 1)) ^ h;
        b_off = (4 * h + q4) * 256 + 8 * (p & 1) + (((q4 << 2) | e) << 4);
    }
    // P of a tile: its packed words ARE the two A fragments (k-step s = words 4s..4s+3); moved to the group's reference by the
    // exact factor f = 2^(r_c - R) of the tile's chunk unless the whole wave has f == 1
    auto make_p = [&](const u32x4 (&d)[2], int t, bf16x8 (&pa)[2]) __attribute__((always_inline)) {
        const float f = ftab[(t / tiles_per_chunk - c0) * 64 + lane];
        unsigned wv[8] = {d[0][0], d[0][1], d[0][2], d[0][3], d[1][0], d[1][1], d[1][2], d[1][3]};
        if (__any(f != 1.f)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float lo = __uint_as_float(wv[i] << 16) * f, hi = __uint_as_float(wv[i] & 0xffff0000u) * f;
                wv[i] = pack_bf16x2(lo, hi);
            }
        }
        pa[0] = __builtin_bit_cast(bf16x8, u32x4{wv[0], wv[1], wv[2], wv[3]});
        pa[1] = __builtin_bit_cast(bf16x8, u32x4{wv[4], wv[5], wv[6], wv[7]});
    };
    bf16x8 pa[2];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SD - 1) * OPT) : "memory");         // tile t0 landed
    asm volatile("" : "+v"(sx[0][0]), "+v"(sx[0][1]));
    make_p(sx[0], t0, pa);
    __builtin_amdgcn_s_barrier();

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    int slot_cur = 0;
    // one tile; `jc` = its place in the score-register ring (indexed statically: the loop below is unrolled SD times)
    auto tile_step = [&](const int t, auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            {
                // tile t+SD: scores into the registers consumed last iteration, keys into the slot freed by the last barrier
                {
                    const int slot_free = slot_cur >= NB - SD ? slot_cur - (NB - SD) : slot_cur + SD;
                    request(t + SD, sx[j], lds0 + slot_free * SLOT);
                }
                // O += P(t) . K_tile: column tile ct = 4*seg + c; its 4 transposed reads run PF column tiles ahead
                const unsigned lb = lds0 + slot_cur * SLOT + b_off;
                s16x4 kb[PF + 1][4];
                auto issue = [&](int ct) __attribute__((always_inline)) {
                    const int sgi = ct >> 2, c = ct & 3;
                    s16x4* k4 = kb[ct % (PF + 1)];
                    const unsigned a0 = lb ^ (unsigned)(c << 6), a1 = a0 ^ 32u;
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[0]) : "v"(a0), "i"(0) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[1]) : "v"(a1), "i"(2048) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[2]) : "v"(a0), "i"(4096) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[3]) : "v"(a1), "i"(6144) : "memory");
                    (void)sgi;
                };
                // (the second segment's reads take their 8 KiB from the immediate offset: separate statements per segment)
                auto issue_seg = [&](int ct) __attribute__((always_inline)) {
                    if ((ct >> 2) == 0) issue(ct);
                    else {
                        const int c = ct & 3;
                        s16x4* k4 = kb[ct % (PF + 1)];
                        const unsigned a0 = lb ^ (unsigned)(c << 6), a1 = a0 ^ 32u;
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[0]) : "v"(a0), "i"(8192) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[1]) : "v"(a1), "i"(8192 + 2048) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[2]) : "v"(a0), "i"(8192 + 4096) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(k4[3]) : "v"(a1), "i"(8192 + 6144) : "memory");
                    }
                };
                bf16x8 pn[2];
                // ALWAYS 8 column tiles: the last range of an odd NSEG has one segment, its second half of the slot holds a repeat
                // of the first (dma_tile) and the four extra column tiles are computed and not stored -- a read-ahead count that
                // follows a run-time number of column tiles cannot be checked path by path (scripts/audit_isa.py), and a second
                // copy of the sweep does not fit the 256 registers of two waves per SIMD.
                auto sweep_cols = [&](auto nct_c) __attribute__((always_inline)) {
                    constexpr int NCTC = decltype(nct_c)::value;
#pragma unroll
                    for (int ct = 0; ct < PF; ++ct) issue_seg(ct);
                    // tile t+1 (keys and scores, requested two iterations ago) has landed by now; the tiles requested after it stay
                    // in flight.  P(t+1) is formed in the shadow of this tile's MFMAs, 4 (or 2) scores per column tile.
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SD - 1) * OPT) : "memory");
                    u32x4 (&nx)[2] = sx[(j + 1) % SD];
                    asm volatile("" : "+v"(nx[0]), "+v"(nx[1]));
                    if (t + 1 < t1) make_p(nx, t + 1, pn);                 // (VALU work under this tile's MFMAs; no tail masking:
                                                                           //  pass 1 stored P~ = 0 for keys past K)
#pragma unroll
                    for (int ct = 0; ct < NCTC; ++ct) {
                        if (ct + PF < NCTC) issue_seg(ct + PF);
                        // in flight behind column tile ct: min(PF, NCTC - 1 - ct) tiles of 4 reads
                        const int ahead = (NCTC - 1 - ct) < PF ? (NCTC - 1 - ct) : PF;
                        if (ahead >= 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                        else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        s16x4* k4 = kb[ct % (PF + 1)];
                        asm volatile("" : "+v"(k4[0]), "+v"(k4[1]), "+v"(k4[2]), "+v"(k4[3]));
                        const s16x8 k0 = __builtin_shufflevector(k4[0], k4[1], 0, 1, 2, 3, 4, 5, 6, 7);
                        const s16x8 k1 = __builtin_shufflevector(k4[2], k4[3], 0, 1, 2, 3, 4, 5, 6, 7);
                        f32x16& o = O[ct >> 2][ct & 3];
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], __builtin_bit_cast(bf16x8, k0), o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], __builtin_bit_cast(bf16x8, k1), o, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                sweep_cols(std::integral_constant<int, 8>{});
                pa[0] = pn[0];
                pa[1] = pn[1];
                __builtin_amdgcn_s_barrier();                              // every wave is done with slot t
                slot_cur = slot_cur == NB - 1 ? 0 : slot_cur + 1;
            }
    };
    // (the steps of one trip are NESTED behind each other's guards -- `break`, not three independent `if (t < t1)`: the audit
    //  walks every path of the compiled code and cannot know that t + 1 >= t1 rules out t + 2 < t1)
    static_assert(SD == 3, "three steps per trip");
#pragma unroll 1
    for (int tb = t0; tb < t1; tb += SD) {
        tile_step(tb, std::integral_constant<int, 0>{});
        if (tb + 1 >= t1) break;
        tile_step(tb + 1, std::integral_constant<int, 1>{});
        if (tb + 2 >= t1) break;
        tile_step(tb + 2, std::integral_constant<int, 2>{});
    }
    // the requests past the group's last tile have landed too: from here on their registers are free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    if (live) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < ncs) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int C = (cb + s) * 4 + c;                         // column tile of the full row
                    const int slab = C >> 4, cl = C & 15;
                    const int nct = min(16, NSEG * 4 - 16 * slab);
                    uint4* dst = o_part + slab * slab_stride + ((long)grp * (Bpad / 32) + wb) * (long)(nct * 2 * 64) + lane;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        uint4 v;
                        v.x = pack_bf16x2(O[s][c][8 * g + 0], O[s][c][8 * g + 1]);
                        v.y = pack_bf16x2(O[s][c][8 * g + 2], O[s][c][8 * g + 3]);
                        v.z = pack_bf16x2(O[s][c][8 * g + 4], O[s][c][8 * g + 5]);
                        v.w = pack_bf16x2(O[s][c][8 * g + 6], O[s][c][8 * g + 7]);
                        dst[(cl * 2 + g) * 64] = v;
                    }
                }
            }
        }
    }
}

// softmax statistics of the finished score scratch: wave = (row-wave rw, split of the key tiles); online max per lane,
// halves merged at the end; writes (m, l, true max) in the partial format of the one-pass kernel (chunk = split)
__global__ __launch_bounds__(256) void infonce_slab_stats_kernel(const float* __restrict__ xs, int K, int ntiles, int Bpad,
                                                                 int nsplit, int per, float* __restrict__ m_part,
                                                                 float* __restrict__ l_part, float* __restrict__ x_part) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nrw = Bpad / 32;
    if (item >= nrw * nsplit) return;
    const int rw = item / nsplit, split = item - rw * nsplit;
    const int tb = split * per, te = min(tb + per, ntiles);       // per = tiles_per_chunk: split == key chunk of the slab passes
    float m = NEG_BIG, l = 0.f;
    for (int t = tb; t < te; ++t) {
        const float4* xa = reinterpret_cast<const float4*>(xs + (((long)rw * ntiles + t) * 64 + lane) * 16);
        float x[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v = xa[g4];
            x[4 * g4] = v.x; x[4 * g4 + 1] = v.y; x[4 * g4 + 2] = v.z; x[4 * g4 + 3] = v.w;
        }
        if ((t + 1) * KT > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + (r & 3) + 8 * (r >> 2) + 4 * h >= K) x[r] = NEG_BIG;
        }
        float tm = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, x[r]);
        const float mn = fmaxf(m, tm);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += exp2f(x[r] - mn);
        l = l * exp2f(m - mn) + ps;
        m = mn;
    }
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float M = fmaxf(m, mo);
    const float L = l * exp2f(m - M) + lo * exp2f(mo - M);
    if (h == 0) {
        const long o = (long)split * Bpad + rw * 32 + n;
        m_part[o] = M;                          // (an empty split leaves M = NEG_BIG, L = 0: weight 0 in the merge)
        l_part[o] = L;
        x_part[o] = M;
    }
}

// merge the key chunks of the 8 query rows of one (wave block, g, h) group -- they share the packed 16-B O words -- for one
// group of 64 columns: add the positive logit (exact fp32), emit loss / lse / top-1 (column group 0 only) and dq.
// Scalars: 32 threads per row.  dq: the 256 threads are 8 chunk-groups x 32 columns, one 16-B load (8 rows of a column) per
// thread and chunk; the chunk-groups are summed through LDS in a fixed order (bitwise reproducible).
constexpr int COMBINE_MAX_CHUNKS = 1024;
constexpr int SMALL_B_MAX = 64;      // batches up to this take infonce_small_kernel (with dq)
template <int TB = 1>
__device__ __forceinline__ void infonce_combine_body(const float* __restrict__ q, const float* __restrict__ k,
                                                     int B, int D, float inv_T, int nchunk, int Bpad,
                                                     const uint4* __restrict__ o_part,
                                                     const float* __restrict__ m_part,
                                                     const float* __restrict__ l_part,
                                                     const float* __restrict__ x_part,
                                                     float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                     int32_t* __restrict__ top1, float* __restrict__ dq,
                                                     long slab_stride, int cg, int tpb,
                                                     const float* __restrict__ ref_part, const int by) {
    // by: the block's row in the combine grid (blockIdx.y, less the rows that carry an enqueue in front of it)
    // ref_part: per (chunk, row) reference of the O partials when it is not m_part (wide rows: the integer references of pass 1)
    // tpb: column tiles per block (grid.y = ceil(D/32 / tpb)).  Every block repeats the row statistics of its 8 rows: 16x at
    // d = 512 with tpb = 1 (cheap); wide rows take 4 tiles per block (10x instead of 40x at d = 1280) in the same single launch.
    __shared__ __attribute__((aligned(16))) float wts[COMBINE_MAX_CHUNKS][8];
    __shared__ float accs[8][8][32];
    __shared__ float rowc[8][2];                     // per row: 1/L, p0/L - 1
    const int tid = threadIdx.x;
    const int wb = blockIdx.x >> 2, g = (blockIdx.x >> 1) & 1, h = blockIdx.x & 1;
    if (!group_live(wb, g, h, B)) return;                                // all 8 rows of this block lie past B (pad rows: their
                                                                         // partials are not even written by the passes over the queue)
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    auto row_of = [&](int i) { return wb * 32 + opart_row(g, h, i >> 1, i & 1); };
    auto sum32 = [](float v) { return half32_sum(v); };          // (DPP + one permlane swap: no LDS round trips)
    auto max32 = [](float v) { return half32_max(v); };
    const int nparts = (nchunk + cg - 1) / cg;                            // partial buffers: one per chunk (group)
    // (requesting the first batch of O partials ahead of the statistics -- they do not depend on them -- was slower, 14.4 vs
    //  10.3 us: 64 more live registers halve the waves per CU of a kernel that lives on bytes in flight)
    {
        const int rr = tid >> 5, l32 = tid & 31;
        const int b = row_of(rr);
        const bool live = b < B;
        const int bb = live ? b : B - 1;
        // The statistics are three dependent sweeps (positive logit; maxima; weights and sum): with up to 128 chunks every
        // partial a lane needs is requested up front, next to the q / k rows, so the block pays ONE global round trip, not three
        // (the kernel is latency-bound: 6.5 us of its 13.9 were this prologue).
        constexpr int MC = 8;                      // (256 chunks -- the small-batch plan -- still take the one-round-trip path)
        const bool small = nchunk <= 32 * MC;
        // Round 5: the (chunk, row) statistics come in by COALESCED loads.  Lane = (row, 32 chunks) gathers -- one 4-byte value per
        // lane, 1 KB apart: 32 sectors per wave instruction, 48 instructions per block, every sector asked for by all four waves --
        // took 3.2 us of the block's 9.7 (in-kernel stamps, scripts/diag_combine_stamps.py).  The 8 rows of a block are two runs of
        // 4 consecutive rows (opart_row: base + {0..3} and base + {8..11}), so a chunk's 8 values are TWO float4: 6 * nchunk loads
        // per block, 3 per thread at 128 chunks, transposed through the part of wts[] that 256 chunks leave unused.
        float* stat = &wts[256][0];                // [3 arrays][8 rows][256 chunks] floats = rows 256 .. 1023 of wts (small only)
        if (small) {
            const int rowbase = wb * 32 + 16 * g + 4 * h;                 // = row_of(0); rows +0..3 and +8..11
            const int per = 2 * nchunk;
            for (int idx = tid; idx < 3 * per; idx += 256) {
                const int a = idx / per, rem = idx - a * per, c = rem >> 1, hf = rem & 1;
                const float* src = (a == 0 ? m_part : a == 1 ? x_part : l_part) + (long)c * Bpad + rowbase + 8 * hf;
                const float4 v = *reinterpret_cast<const float4*>(src);
                float* dst = stat + (a * 8 + 4 * hf) * 256 + c;
                dst[0] = v.x; dst[256] = v.y; dst[512] = v.z; dst[768] = v.w;
            }
        }
        float s = 0.f;                                                   // positive logit
#pragma unroll 4
        for (int c = l32 * 4; c < D; c += 128) {
            const float4 qa = *reinterpret_cast<const float4*>(q + (long)bb * D + c);
            const float4 ka = *reinterpret_cast<const float4*>(k + (long)bb * D + c);
            s = fmaf(qa.x, ka.x, s); s = fmaf(qa.y, ka.y, s); s = fmaf(qa.z, ka.z, s); s = fmaf(qa.w, ka.w, s);
        }
        if (small) __syncthreads();                                      // (block-uniform) the transposed statistics are in LDS
        float mv[MC], xv[MC], lv[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int c = l32 + 32 * i;
            const bool ok = small && c < nchunk;
            const int o = ok ? c : 0;
            // (rows past B inside a live group read their own -- unused -- entries; nothing of them reaches an output)
            mv[i] = ok ? stat[(0 * 8 + rr) * 256 + o] : NEG_BIG;
            xv[i] = ok ? stat[(1 * 8 + rr) * 256 + o] : NEG_BIG;
            lv[i] = ok ? stat[(2 * 8 + rr) * 256 + o] : 0.f;
        }
        const float s0 = sum32(s) * inv_T;
        const float s0l = s0 * LOG2E;
        float M = NEG_BIG, X = NEG_BIG;
        if (small) {
#pragma unroll
            for (int i = 0; i < MC; ++i) { M = fmaxf(M, mv[i]); X = fmaxf(X, xv[i]); }
        } else {
            for (int c = l32; c < nchunk; c += 32) {
                M = fmaxf(M, m_part[(long)c * Bpad + bb]);
                X = fmaxf(X, x_part[(long)c * Bpad + bb]);
            }
        }
        M = fmaxf(max32(M), s0l);
        X = max32(X);
        float L = 0.f;
        if (small) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int c = l32 + 32 * i;
                if (c < nchunk) {
                    const float w = exp2f(mv[i] - M);
                    if (cg == 1 && ref_part == m_part) wts[c][rr] = w;
                    L += w * lv[i];
                }
            }
        } else {
            for (int c = l32; c < nchunk; c += 32) {
                const float w = exp2f(m_part[(long)c * Bpad + bb] - M);
                if (cg == 1 && ref_part == m_part) wts[c][rr] = w;
                L += w * l_part[(long)c * Bpad + bb];
            }
        }
        if (cg > 1 || ref_part != m_part) {
            // O partials per GROUP of cg key chunks (wide P.K pass), relative to the largest reference of the group
            for (int j = l32; j * cg < nchunk; j += 32) {
                float mg = NEG_BIG;
                for (int c = j * cg; c < min((j + 1) * cg, nchunk); ++c) mg = fmaxf(mg, ref_part[(long)c * Bpad + bb]);
                wts[j][rr] = exp2f(mg - M);
            }
        }
        const float p0u = exp2f(s0l - M);
        L = sum32(L) + p0u;
        const float lse = (M + log2f(L)) * LN2;
        if (l32 == 0) {
            rowc[rr][0] = 1.f / L;
            rowc[rr][1] = p0u / L - 1.f;
            if (live && by == 0) {
                lse_out[b] = lse;
                loss_rows[b] = lse - s0;
                top1[b] = (s0l >= X) ? 1 : 0;
            }
        }
    }
    if (dq == nullptr) return;
    __syncthreads();                                                     // wts[][], rowc[][] complete
    // dq: one 32-column tile of O per block; the 256 threads are 8 chunk-groups x 32 columns, every thread keeps up to 16 loads
    // of 16 B in flight (the kernel is one dependent sweep over 32 MB of partials: what matters is bytes in flight per CU)
    const int col = tid & 31, cgrp = tid >> 5;
    // wide queues (D > 512) keep one partial buffer per column slab of 512 (the last one narrower), `slab_stride` uint4 apart
    const int n = col;
    if constexpr (TB > 1) {
        // ---- wide rows: the TB column tiles of a block with ALL their partial loads in flight at once instead of one dependent
        //      round trip per tile (own instantiation: the d <= 512 combine keeps its register count).  Measured (round 5, d = 1280):
        //      16.4 -> 15.6 - 15.9 us per combine -- the tiles were not what the wide-row combine waits for; its per-block row
        //      statistics are (1 tile per block, 4x the blocks: 22.3 us; 8 tiles: 21.7).
        if (tpb == TB && nparts <= 64) {
            const int c0 = by * TB;
            uint4 v[TB][8];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int c = min(c0 + t, D / 32 - 1);                   // (a block's surplus tiles re-read the last one; not stored)
                const int sl = c >> 4, cl = c & 15;
                const int nct = min(16, D / 32 - 16 * sl);
                const long wb_stride = (long)nct * 2 * 64;
                const long chunk_stride = (long)(Bpad / 32) * wb_stride;
                const uint4* src = o_part + sl * slab_stride + (long)wb * wb_stride + (cl * 2 + g) * 64 + h * 32 + n;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[t][u] = (cgrp + 8 * u < nparts) ? src[(long)(cgrp + 8 * u) * chunk_stride] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int c = c0 + t;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ck = cgrp + 8 * u;
                    if (ck < nparts) {
                        const float4 w0 = *reinterpret_cast<const float4*>(&wts[ck][0]);
                        const float4 w1 = *reinterpret_cast<const float4*>(&wts[ck][4]);
                        const uint4 x = v[t][u];
                        acc[0] = fmaf(w0.x, __uint_as_float(x.x << 16), acc[0]);
                        acc[1] = fmaf(w0.y, __uint_as_float(x.x & 0xffff0000u), acc[1]);
                        acc[2] = fmaf(w0.z, __uint_as_float(x.y << 16), acc[2]);
                        acc[3] = fmaf(w0.w, __uint_as_float(x.y & 0xffff0000u), acc[3]);
                        acc[4] = fmaf(w1.x, __uint_as_float(x.z << 16), acc[4]);
                        acc[5] = fmaf(w1.y, __uint_as_float(x.z & 0xffff0000u), acc[5]);
                        acc[6] = fmaf(w1.z, __uint_as_float(x.w << 16), acc[6]);
                        acc[7] = fmaf(w1.w, __uint_as_float(x.w & 0xffff0000u), acc[7]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) accs[cgrp][i][col] = acc[i];
                __syncthreads();
                {
                    const int i = cgrp;
                    const int b = row_of(i);
                    if (b < B && c < D / 32) {
                        float a = accs[0][i][col];
#pragma unroll
                        for (int u = 1; u < 8; ++u) a += accs[u][i][col];   // (the order of the sequential walk: same bits)
                        const long o = (long)b * D + c * 32 + col;
                        dq[o] = (rowc[i][1] * k[o] + a * rowc[i][0]) * inv_T;
                    }
                }
                __syncthreads();
            }
            return;
        }
    }
    for (int c = by * tpb; c < min((by + 1) * tpb, D / 32); ++c) {
    const int sl = c >> 4, cl = c & 15;
    const int nct = min(16, D / 32 - 16 * sl);
    const long wb_stride = (long)nct * 2 * 64;                           // uint4 per wave block
    const long chunk_stride = (long)(Bpad / 32) * wb_stride;
    const uint4* src = o_part + sl * slab_stride + (long)wb * wb_stride + (cl * 2 + g) * 64 + h * 32 + n;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fma8 = [&](const uint4& v, int ck) {
        const float4 w0 = *reinterpret_cast<const float4*>(&wts[ck][0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&wts[ck][4]);
        acc[0] = fmaf(w0.x, __uint_as_float(v.x << 16), acc[0]);
        acc[1] = fmaf(w0.y, __uint_as_float(v.x & 0xffff0000u), acc[1]);
        acc[2] = fmaf(w0.z, __uint_as_float(v.y << 16), acc[2]);
        acc[3] = fmaf(w0.w, __uint_as_float(v.y & 0xffff0000u), acc[3]);
        acc[4] = fmaf(w1.x, __uint_as_float(v.z << 16), acc[4]);
        acc[5] = fmaf(w1.y, __uint_as_float(v.z & 0xffff0000u), acc[5]);
        acc[6] = fmaf(w1.z, __uint_as_float(v.w << 16), acc[6]);
        acc[7] = fmaf(w1.w, __uint_as_float(v.w & 0xffff0000u), acc[7]);
    };
    int ck = cgrp;
    for (; ck + 120 < nparts; ck += 128) {                               // 16 loads in flight per lane
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(long)(ck + 8 * u) * chunk_stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) fma8(v[u], ck + 8 * u);
    }
    for (; ck < nparts; ck += 64) {                                      // the rest: up to 8 loads in flight, predicated
        uint4 v[8];                                                      // (one at a time this tail was a chain of dependent
#pragma unroll                                                           //  round trips: 43 chunk groups at d = 1280 -> 6 of them)
        for (int u = 0; u < 8; ++u)
            v[u] = (ck + 8 * u < nparts) ? src[(long)(ck + 8 * u) * chunk_stride] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ck + 8 * u < nparts) fma8(v[u], ck + 8 * u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) accs[cgrp][i][col] = acc[i];
    __syncthreads();
    {
        const int i = cgrp;                                              // 8 rows x 32 columns of output, one per thread
        const int b = row_of(i);
        if (b < B) {
            float a = accs[0][i][col];
#pragma unroll
            for (int u = 1; u < 8; ++u) a += accs[u][i][col];           // fixed order: bitwise reproducible
            const long o = (long)b * D + c * 32 + col;
            dq[o] = (rowc[i][1] * k[o] + a * rowc[i][0]) * inv_T;
        }
    }
    __syncthreads();                                                     // accs[] is reused by the next column tile
    }
}

// K3 inside the combine launch (round 6).  The enqueue must follow the LAST read of the queue by this K2 call (read old, then
// enqueue: MoMA/mem_moco.py:89-99) -- and the combine launch sits behind the passes over the queue on the stream, every workgroup
// of which has finished: extra rows of the combine grid (blockIdx.y < ey) copy the B key rows into the ring -- fp32 rows into an
// fp32 queue, rounded to bf16 into a bf16 queue / mirror, both from one read -- instead of a launch of their own (4.4 us of pure
// dispatch in the step).  Semantics of enqueue_kernel (queue.hip): slot (index + i) mod K; with n > K a slot rewritten by a later
// row is skipped, so the result is the serial index_copy_ without a write race.
__device__ __forceinline__ void enqueue_rows(const EnqueueJob& e, int blk, int nblk) {
    const int d = e.d;
    bf16_raw* q16 = reinterpret_cast<bf16_raw*>(e.queue16);
    const bool vec = d % 4 == 0 && (((uintptr_t)e.rows | (uintptr_t)e.queue32) & 15) == 0 && ((uintptr_t)e.queue16 & 7) == 0;
    for (int i = blk; i < e.n; i += nblk) {
        if ((int64_t)i + e.K < e.n) continue;
        const int64_t slot = (e.index + i) % e.K;
        const float* src = e.rows + (int64_t)i * d;
        if (vec) {
            for (int c = threadIdx.x * 4; c < d; c += 256 * 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                if (e.queue32) *reinterpret_cast<float4*>(e.queue32 + slot * d + c) = v;
                if (q16) {
                    ushort4 o;
                    o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
                    *reinterpret_cast<ushort4*>(q16 + slot * d + c) = o;
                }
            }
        } else {
            for (int c = threadIdx.x; c < d; c += 256) {
                if (e.queue32) e.queue32[slot * d + c] = src[c];
                if (q16) q16[slot * d + c] = f32_to_bf16(src[c]);
            }
        }
    }
}

template <int TB>
__global__ __launch_bounds__(256) void infonce_combine_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               int B, int D, float inv_T, int nchunk, int Bpad,
                                                               const uint4* __restrict__ o_part,
                                                               const float* __restrict__ m_part,
                                                               const float* __restrict__ l_part,
                                                               const float* __restrict__ x_part,
                                                               float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                               int32_t* __restrict__ top1, float* __restrict__ dq,
                                                               long slab_stride, int cg, int tpb,
                                                               const float* __restrict__ ref_part, EnqueueJob enq, int ey) {
    // the first `ey` rows of the grid carry the enqueue (FIRST in dispatch order: they run beside the combine's own workgroups;
    // behind them they were a 3 us tail of their own -- a dependent load -> store chain with the chip already drained)
    if ((int)blockIdx.y < ey) {                      // (block-uniform; ey = 0 when nothing rides along)
        enqueue_rows(enq, (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, ey * (int)gridDim.x);
        return;
    }
    infonce_combine_body<TB>(q, k, B, D, inv_T, nchunk, Bpad, o_part, m_part, l_part, x_part, loss_rows, lse_out, top1, dq, slab_stride,
                             cg, tpb, ref_part, (int)blockIdx.y - ey);
}

// rows of the combine grid that carry the enqueue: about 64 workgroups for the B key rows (a row of 512 floats is half a pass of a
// 256-thread workgroup), none without a job
static int enqueue_grid_rows(const EnqueueJob* e, int gx) {
    if (e == nullptr || e->n <= 0) return 0;
    int want = e->n < 64 ? e->n : 64;
    return (want + gx - 1) / gx;
}

// ---- several InfoNCE terms in ONE sweep (the dual-queue memories MoCoST / MoCoSSTT, reference MoMA/mem_moco.py:165-253:
// (q, k, memory_s), (q, k_t, memory_t) [, (q_t, k, memory_s), (q_t, k_t, memory_t)]): one pre-pack launch over the distinct
// query sets, one launch of the one-pass kernel whose grid covers every (term, query tile, key chunk) -- sized to one
// workgroup per compute unit over ALL terms, so the queues are streamed back to back by the same launch --, one combine.
constexpr int MULTI_MAX_TERMS = 4;
struct MultiTerm {
    const float* q;
    const float* k;
    const bf16_raw* queue;
    float* loss_rows;
    float* lse;
    int32_t* top1;
    float* dq;
    int qset;                 // index of the term's query among the distinct query sets
};
struct MultiArgs {
    MultiTerm t[MULTI_MAX_TERMS];
    const float* qsets[MULTI_MAX_TERMS];
    int n, nq;
};
// (a run-time index into a by-value kernel argument sends the struct to scratch: select with compile-time indices)
__device__ __forceinline__ MultiTerm pick_term(const MultiArgs& a, int term) {
    MultiTerm t = a.t[0];
#pragma unroll
    for (int i = 1; i < MULTI_MAX_TERMS; ++i)
        if (term == i) t = a.t[i];
    return t;
}

template <int D>
__global__ __launch_bounds__(256) void infonce_qpack_multi_kernel(MultiArgs a, int B, float scale_log2, uint4* __restrict__ qpack,
                                                                  int n_row_tiles, long qset_stride) {
    constexpr int KS = D / 16;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);          // (row tile, k-step)
    if (item >= n_row_tiles * KS) return;
    const float* q = a.qsets[0];
#pragma unroll
    for (int i = 1; i < MULTI_MAX_TERMS; ++i)
        if ((int)blockIdx.y == i) q = a.qsets[i];
    const int rt = item / KS, ks = item % KS;
    const int row = rt * 32 + (lane & 31), h = lane >> 5;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
    if (row < B) {
        const float* qp = q + (long)row * D + 16 * ks + 8 * h;
        x = *reinterpret_cast<const float4*>(qp);
        y = *reinterpret_cast<const float4*>(qp + 4);
    }
    const bf16x8 f = bf16x8{(__bf16)(x.x * scale_log2), (__bf16)(x.y * scale_log2), (__bf16)(x.z * scale_log2),
                            (__bf16)(x.w * scale_log2), (__bf16)(y.x * scale_log2), (__bf16)(y.y * scale_log2),
                            (__bf16)(y.z * scale_log2), (__bf16)(y.w * scale_log2)};
    qpack[(long)blockIdx.y * qset_stride + (long)item * 64 + lane] = __builtin_bit_cast(uint4, f);
}

template <int D, bool WITH_DQ>
__global__ __launch_bounds__(256, 1) void infonce_flash_multi_kernel(MultiArgs a, const uint4* __restrict__ qpack, long qset_stride,
                                                                     int B, int K, int nbt, int nchunk, int tiles_per_chunk, int Bpad,
                                                                     uint4* __restrict__ o_part, long o_term_stride,
                                                                     float* __restrict__ m_part, float* __restrict__ l_part,
                                                                     float* __restrict__ x_part, long term_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per = nbt * nchunk;
    const int term = blockIdx.x / per, id = blockIdx.x - term * per;
    const MultiTerm t = pick_term(a, term);
    infonce_flash_body<D, WITH_DQ>(id, smem, qpack + t.qset * qset_stride, t.queue, B, K, nbt, nchunk, tiles_per_chunk, Bpad,
                                   o_part + term * o_term_stride, m_part + term * term_rows, l_part + term * term_rows,
                                   x_part + term * term_rows);
}

__global__ __launch_bounds__(256) void infonce_combine_multi_kernel(MultiArgs a, int B, int D, float inv_T, int nchunk, int Bpad,
                                                                     const uint4* __restrict__ o_part, long o_term_stride,
                                                                     const float* __restrict__ m_part,
                                                                     const float* __restrict__ l_part,
                                                                     const float* __restrict__ x_part, long term_rows) {
    const int term = blockIdx.z;
    const MultiTerm t = pick_term(a, term);
    infonce_combine_body(t.q, t.k, B, D, inv_T, nchunk, Bpad, o_part + term * o_term_stride, m_part + term * term_rows,
                         l_part + term * term_rows, x_part + term * term_rows, t.loss_rows, t.lse, t.top1, t.dq, 0L, 1, 1,
                         m_part + term * term_rows, (int)blockIdx.y);
}

struct FlashPlan {
    int nbt, nchunk, tiles_per_chunk, Bpad;
};

// Workgroups a pass over the queue is cut into: one per compute unit.  Plan sweeps (scripts/sweep_k2_plan.sh) override it through
// moma_debug_set_k2_target_wg() -- a DEBUG entry point, not an environment variable: the library reads none (include/moma_hip.h).
// Every caller of plan() -- the workspace queries included -- sees the value that is set at the time of ITS call: set it once,
// before the first query, and leave it (the entry point says so).  8 .. COMBINE_MAX_CHUNKS, 0 = the product's own plan.
std::atomic<int> g_target_override{0};
int target_override() { return g_target_override.load(std::memory_order_relaxed); }
int target_workgroups() { return target_override() ? target_override() : 256; }

// Short queues (the reference's default --nce_k 16384, train_student_moma.py:103) and narrow rows (--feat_dim 128 / 256): a
// workgroup's pass is a fixed ~10 us (dispatch, Q + first tile, partial stores) plus its tiles, and every chunk is one more partial
// the combine reads back (combine: 5 / 7 / 10.5 / 18+ us at 32 / 64 / 128 / 256 chunks).  When a workgroup's tile loop would run
// for less than ~4 us the pass is cut into HALF as many chunks instead (repeatedly, down to 32): measured, one-pass kernel + combine
// (profiles/r05_k2_plan_sweep.txt): (B, d, K) = (64, 512, 16384) 21.8 -> 18.3 us, (256, 128, 16384) 17.9 -> 15.4, (256, 256, 16384)
// 21.4 -> 20.0; (256, 512, 16384) and every K = 65536 shape keep one workgroup per CU (halving them costs 1 - 13 us).
// tile_ns: duration of one 32-key tile of the loop at d = 512 (1.63 us one-pass kernel, 1.44 us small-batch kernel), linear in d.
template <typename F>
void coarsen_short_passes(int& want, int& tpc, int ntiles, int d, long tile_ns_512, long tile_ns_fixed, F&& round) {
    if (d > 512) return;                                   // (wide rows: two passes with their own grouping; a tile is long anyway)
    if (target_override()) return;                         // (a sweep asks for exactly that many workgroups)
    const long tile_ns = tile_ns_512 * d / 512 + tile_ns_fixed;
    while (want > 32 && (long)tpc * tile_ns < 4000) {
        want = round(want / 2);
        tpc = (ntiles + want - 1) / want;
    }
}

FlashPlan plan(int B, int K, int d, int nterms = 1) {
    FlashPlan p;
    p.nbt = (B + QROWS_WG - 1) / QROWS_WG;
    p.Bpad = p.nbt * QROWS_WG;
    const int ntiles = (K + KT - 1) / KT;
    int want = target_workgroups() / (p.nbt * nterms);         // ~1 workgroup per CU over all terms
    if (want < 8) want = 8;
    if (want > 1024) want = 1024;
    want = (want / 8) * 8;
    int tpc = (ntiles + want - 1) / want;
    coarsen_short_passes(want, tpc, ntiles, d, 1630, 150, [](int w) { return w < 8 ? 8 : (w / 8) * 8; });
    if (tpc < 1) tpc = 1;
    p.tiles_per_chunk = tpc;
    p.nchunk = (ntiles + tpc - 1) / tpc;       // no empty chunk by construction
    return p;
}

// small batches (infonce_small_kernel): workgroups of 64 padded rows, two virtual chunks each
struct SmallPlan {
    int nwg, tiles_per_wg;
};
SmallPlan small_plan(int K, int d) {
    // one workgroup per CU; short queues (fewer than 8 tiles per workgroup: K < 65536) take half as many -- measured
    // (profiles/r05_k2_plan_sweep.txt, kernel + combine): K = 16384: d = 128 / 256 / 512 16.3 / 18.9 / 21.8 us at 256 workgroups,
    // 13.3 / 14.6 / 18.3 at 128, 13.8 / 15.1 / 20.7 at 64; K = 65536: 19.7 / 21.8 / 29.4 at 256, 19.7 / 22.9 / 32.2 at 128
    (void)d;
    const int ntiles = (K + KT - 1) / KT;
    int want = target_workgroups();
    if (want > COMBINE_MAX_CHUNKS) want = COMBINE_MAX_CHUNKS;       // (one partial per workgroup: the combine's chunk table)
    int tpc = (ntiles + want - 1) / want;
    if (!target_override() && tpc < 8 && want > 128) {
        want = 128;
        tpc = (ntiles + want - 1) / want;
    }
    if (tpc < 1) tpc = 1;
    return SmallPlan{(ntiles + tpc - 1) / tpc, tpc};
}

}  // namespace

int set_k2_target_wg(int n) {                          // moma_debug_set_k2_target_wg (api.hip): -> the previous value, -1 = refused
    if (n != 0 && (n < 8 || n > COMBINE_MAX_CHUNKS)) return -1;
    return g_target_override.exchange(n, std::memory_order_relaxed);
}

constexpr int WIDE_PV2_LDS = (MOMA_K2_WPV_SD + 1) * 16384 + 8 * 8 * 64 * 4;     // ring of (32 keys x 256 columns) slots + the scale-factor table
static bool one_pass_dim(int d) { return d == 128 || d == 256 || d == 384 || d == 512; }
static bool two_qpass_dim(int d) { return d == 2048; }                           // scores in two register passes of Q (8 + 8 segments)
static bool slab_dim(int d) { return d > 512 && d <= 4096 && d % 128 == 0; }     // column slabs of 512 / 384 / 256 / 128

bool infonce_flash_supported(int B, int d, int K, int qdtype, int prec) {
    if (prec != MOMA_PREC_BF16 || qdtype != MOMA_DT_BF16) return false;
    if (!one_pass_dim(d) && !slab_dim(d)) return false;
    return B >= 1 && K >= 1;
}

size_t infonce_flash_workspace_bytes(int B, int d, int K) {
    const FlashPlan p = plan(B, K, d);
    size_t rows = (size_t)p.nchunk * p.Bpad;
    if (B <= SMALL_B_MAX && d <= 512) {                      // (the small-batch path lays out 64-row partials of its own plan)
        const size_t rs = (size_t)small_plan(K, d).nwg * 64;     // its rows: 5 float arrays + the O partials (4 + O in the formula below)
        if (rs + rs / 4 + 64 > rows) rows = rs + rs / 4 + 64;
    }
    const int ds = d > 512 ? 512 : d;                        // widest slab
    const int nslab = (d + 511) / 512;                       // one partial buffer per column slab
    size_t bytes = (size_t)nslab * rows * ds * 2 + 4 * rows * sizeof(float) + (size_t)p.Bpad * d * 2 + 1024;
    if (d > 512) bytes += (size_t)p.Bpad * ((K + KT - 1) / KT) * KT * sizeof(float) + 512;      // score / P scratch
    if (two_qpass_dim(d)) bytes += (size_t)p.Bpad * ((K + KT - 1) / KT) * KT * 2 + 512;         // P behind the partial scores
    return bytes;
}

namespace {
// dynamic-LDS opt-in of every instantiation, once per process (hipFuncSetAttribute is not a stream operation)
std::once_flag g_lds_attr_once;
void set_lds_attrs() {
    const int mx = NBUF * KT * 512 * 2 + 16 + 512;
#define MOMA_SET_LDS(DD)                                                                                                        \
    (void)hipFuncSetAttribute((const void*)infonce_flash_kernel<DD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);     \
    (void)hipFuncSetAttribute((const void*)infonce_flash_kernel<DD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);    \
    (void)hipFuncSetAttribute((const void*)infonce_flash_multi_kernel<DD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);   \
    (void)hipFuncSetAttribute((const void*)infonce_flash_multi_kernel<DD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);  \
    (void)hipFuncSetAttribute((const void*)infonce_small_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);           \
    (void)hipFuncSetAttribute((const void*)infonce_slab_kernel<DD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);         \
    (void)hipFuncSetAttribute((const void*)infonce_slab_kernel<DD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx)
    MOMA_SET_LDS(512); MOMA_SET_LDS(384); MOMA_SET_LDS(256); MOMA_SET_LDS(128);
#undef MOMA_SET_LDS
#define MOMA_SET_WIDE(NS, PS) (void)hipFuncSetAttribute((const void*)infonce_wide_scores_kernel<NS, PS>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PS * 8192 + 16)
    MOMA_SET_WIDE(5, 5); MOMA_SET_WIDE(6, 6); MOMA_SET_WIDE(8, 4); MOMA_SET_WIDE(10, 5); MOMA_SET_WIDE(12, 6);
#undef MOMA_SET_WIDE
    (void)hipFuncSetAttribute((const void*)infonce_wide_scores_kernel<8, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 8192 + 16);
#define MOMA_SET_WPV(NS) (void)hipFuncSetAttribute((const void*)infonce_wide_pv2_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_PV2_LDS)
    MOMA_SET_WPV(5); MOMA_SET_WPV(6); MOMA_SET_WPV(8); MOMA_SET_WPV(10); MOMA_SET_WPV(12); MOMA_SET_WPV(16);
#undef MOMA_SET_WPV
}
}  // namespace

size_t infonce_qpack_bytes(int B, int d) {
    if (B < 1 || !one_pass_dim(d)) return 0;
    return (size_t)plan(B, 1, d).Bpad * d * 2;
}

// q_packed (nullable, one-pass widths only): q * inv_T * log2(e) as bf16 in infonce_qpack_kernel's layout, written by the
// producer of q (K1's proj epilogue, k1_fast.hip) -- the pre-pack launch is skipped.
hipError_t launch_infonce_flash(const float* q, const float* k, const void* queue, int B, int d, int K, float inv_T,
                                float* loss_rows, float* lse, int32_t* top1, float* dq, void* ws, int qdtype,
                                hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end, const void* q_packed, hipEvent_t ev_call_end,
                                const EnqueueJob* enq) {
    const EnqueueJob no_job{nullptr, nullptr, nullptr, 0, 0, 1, d};
    const EnqueueJob job = (enq && enq->n > 0) ? *enq : no_job;
    if (B <= SMALL_B_MAX && dq != nullptr && one_pass_dim(d)) {
        // ---- small batches: key-half split (infonce_small_kernel), two virtual chunks per workgroup, 64 padded rows
        const SmallPlan sp = small_plan(K, d);
        const int tpc = sp.tiles_per_wg, nwg = sp.nwg;
        if (nwg > COMBINE_MAX_CHUNKS) return hipErrorInvalidValue;         // (as the general path below: the combine's wts[] table)
        const int Bp = 64;
        const size_t rows = (size_t)nwg * Bp;                              // (infonce_flash_workspace_bytes covers this layout)
        float* m_part = (float*)ws;
        float* l_part = m_part + rows;
        float* x_part = l_part + rows;
        uint4* o_part = (uint4*)(((uintptr_t)(x_part + 2 * rows) + 255) & ~(uintptr_t)255);
        uint4* qpack = (uint4*)(((uintptr_t)((char*)o_part + rows * d * 2) + 255) & ~(uintptr_t)255);
        std::call_once(g_lds_attr_once, set_lds_attrs);
        const float scale_log2 = inv_T * 1.4426950408889634f;
        const bf16_raw* qu = (const bf16_raw*)queue;
        const size_t lds = (size_t)NBUF * KT * d * 2 + 16 + 512;          // ring, overflow words, exchange words of the key halves
        if (q_packed != nullptr) qpack = (uint4*)q_packed;
#define MOMA_SMALL_LAUNCH(DD)                                                                                                  \
        do {                                                                                                                   \
            if (q_packed == nullptr) hipLaunchKernelGGL((infonce_qpack_kernel<DD>), dim3(((Bp / 32) * (DD / 16) + 3) / 4), dim3(256), 0, st, q, B, d, 0, scale_log2, qpack, Bp / 32); \
            hipExtLaunchKernelGGL((infonce_small_kernel<DD>), dim3(nwg), dim3(256), lds, st, ev_begin, ev_end, 0, qpack, qu, B, K, tpc, Bp, o_part, m_part, l_part, x_part); \
        } while (0)
        if (d == 512) MOMA_SMALL_LAUNCH(512);
        else if (d == 384) MOMA_SMALL_LAUNCH(384);
        else if (d == 256) MOMA_SMALL_LAUNCH(256);
        else MOMA_SMALL_LAUNCH(128);
#undef MOMA_SMALL_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const int ny = d / 32, ey = enqueue_grid_rows(&job, Bp / 8);
        hipExtLaunchKernelGGL((infonce_combine_kernel<1>), dim3(Bp / 8, ny + ey), dim3(256), 0, st, (hipEvent_t) nullptr, ev_call_end, 0, q, k, B,
                              d, inv_T, nwg, Bp, o_part, m_part, l_part, x_part, loss_rows, lse, top1, dq, 0L, 1, 1, m_part, job, ey);
        return hipGetLastError();
    }
    const FlashPlan p = plan(B, K, d);
    if (p.nchunk > COMBINE_MAX_CHUNKS) return hipErrorInvalidValue;
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    const int dsl = d > 512 ? 512 : d;
    float* m_part = (float*)ws;
    float* l_part = m_part + rows;
    float* x_part = l_part + rows;
    float* r_part = x_part + rows;                             // wide rows: integer references of the stored P (pass 1)
    uint4* o_part = (uint4*)(((uintptr_t)(r_part + rows) + 255) & ~(uintptr_t)255);
    uint4* qpack = (uint4*)(((uintptr_t)((char*)o_part + rows * dsl * 2) + 255) & ~(uintptr_t)255);
    std::call_once(g_lds_attr_once, set_lds_attrs);
    const float scale_log2 = inv_T * 1.4426950408889634f;
    const bf16_raw* qu = (const bf16_raw*)queue;
    const dim3 grid(p.nbt * p.nchunk), block(256);
    if (d > 512) {
        // ---- wide queue.  Scores of every (row, key) -> fp32 scratch + per-chunk (max, sum): ONE pass with the whole Q row block
        //      in registers when d is 5 / 6 / 8 / 10 / 12 segments of 128 columns (d <= 1536), else one pass per column slab
        //      (the slab adds its partial scores to the scratch) and a statistics pass.  Then, for dq, per column slab of <= 512:
        //      O = sum 2^(x - chunk max) key over the chunk (P.K on the slab's columns), and ONE combine launch for loss / lse /
        //      top-1 / dq over all slabs.
        const int nslab = (d + 511) / 512;
        const size_t slab_bytes = rows * 512 * 2;
        qpack = (uint4*)(((uintptr_t)((char*)o_part + (size_t)nslab * slab_bytes) + 255) & ~(uintptr_t)255);
        float* xs = (float*)(((uintptr_t)((char*)qpack + (size_t)p.Bpad * d * 2) + 255) & ~(uintptr_t)255);
        const int ntiles = (K + KT - 1) / KT;
        // (two register passes of Q: the fp32 partial scores take the scratch, P~ sits behind them)
        uint4* pscr = two_qpass_dim(d) ? (uint4*)(((uintptr_t)(xs + (size_t)p.Bpad * ntiles * KT) + 255) & ~(uintptr_t)255) : (uint4*)xs;
        auto slab_width = [&](int col0) { const int rem = d - col0; return rem >= 512 ? 512 : rem; };   // 512.., then 384/256/128
        // (the measurement events span the passes over the queue -- scores and P.K -- like the one-pass path, where they span the
        //  flash kernel: the Q pre-pack in front and the combine behind are outside in both)
        const int nseg = d / 128;
        bool ev_on_dispatch = false;                           // the two-pass kernels carry the measurement events on their dispatches
#define MOMA_WIDE_SCORES(NS, PS, DD, QPS)                                                                                        \
        do {                                                                                                                  \
            hipLaunchKernelGGL((infonce_qpack_kernel<DD>), dim3(((p.Bpad / 32) * (DD / 16) + 3) / 4), dim3(256), 0, st, q, B, d, 0, scale_log2, qpack, p.Bpad / 32); \
            hipExtLaunchKernelGGL((infonce_wide_scores_kernel<NS, PS, QPS>), grid, block, 3 * PS * 8192 + 16, st, ev_begin, dq ? (hipEvent_t) nullptr : ev_end, 0, qpack, qu, B, K, p.nbt, p.nchunk, p.tiles_per_chunk, p.Bpad, pscr, m_part, l_part, x_part, r_part, (uint4*)xs); \
            ev_on_dispatch = true;                                                                                            \
        } while (0)
        if (nseg == 5) MOMA_WIDE_SCORES(5, 5, 640, 1);
        else if (nseg == 6) MOMA_WIDE_SCORES(6, 6, 768, 1);
        else if (nseg == 8) MOMA_WIDE_SCORES(8, 4, 1024, 1);
        else if (nseg == 10) MOMA_WIDE_SCORES(10, 5, 1280, 1);
        else if (nseg == 12) MOMA_WIDE_SCORES(12, 6, 1536, 1);
        else if (nseg == 16) MOMA_WIDE_SCORES(8, 4, 2048, 2);
        else {
            if (ev_begin) (void)hipEventRecord(ev_begin, st);      // (slab passes: a pre-pack per slab sits between the passes)
            for (int col0 = 0; col0 < d; col0 += slab_width(col0)) {
                const int D = slab_width(col0);
                const SlabArgs sa{xs, nullptr, (unsigned)(d * 2), col0 == 0 ? 1 : 0};
                const size_t slds = (size_t)NBUF * KT * D * 2 + 16;
#define MOMA_SLAB_SCORES(DD)                                                                                             \
                do {                                                                                                         \
                    hipLaunchKernelGGL((infonce_qpack_kernel<DD>), dim3(((p.Bpad / 32) * (DD / 16) + 3) / 4), dim3(256), 0, st, q, B, d, col0, scale_log2, qpack, p.Bpad / 32); \
                    hipLaunchKernelGGL((infonce_slab_kernel<DD, 1>), grid, block, slds, st, qpack, qu + col0, B, K, p.nbt, p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, sa); \
                } while (0)
                if (D == 512) MOMA_SLAB_SCORES(512); else if (D == 384) MOMA_SLAB_SCORES(384); else if (D == 256) MOMA_SLAB_SCORES(256); else MOMA_SLAB_SCORES(128);
#undef MOMA_SLAB_SCORES
            }
            hipLaunchKernelGGL(infonce_slab_stats_kernel, dim3(((p.Bpad / 32) * p.nchunk + 3) / 4), dim3(256), 0, st, xs, K, ntiles,
                               p.Bpad, p.nchunk, p.tiles_per_chunk, m_part, l_part, x_part);
        }
#undef MOMA_WIDE_SCORES
        int cg = 1;
        const float* wide_ref = m_part;                        // reference of the O partials (the combine kernel's weights)
        if (dq) {
            const bool wide_pv = nseg == 5 || nseg == 6 || nseg == 8 || nseg == 10 || nseg == 12 || nseg == 16;
            if (wide_pv) {
                // column ranges of 256 columns x 256 rows per workgroup; key chunks grouped (<= 8 per group) so that the grid is
                // about one workgroup per CU
                const int ncr = (nseg + 1) / 2;
                const int nj = ((p.Bpad / 32 + 7) / 8) * ncr;
                int want = 256 / nj;
                if (want < 1) want = 1;
                cg = (p.nchunk + want - 1) / want;
                if (cg > 8) cg = 8;
                const int ngroups = (p.nchunk + cg - 1) / cg;
                wide_ref = r_part;
#define MOMA_WIDE_PV(NS) hipExtLaunchKernelGGL((infonce_wide_pv2_kernel<NS>), dim3(8 * ((ngroups + 7) / 8) * nj), dim3(512), WIDE_PV2_LDS, st, (hipEvent_t) nullptr, ev_end, 0, qu, B, K, p.nchunk, p.tiles_per_chunk, cg, p.Bpad, (const uint4*)pscr, r_part, o_part, (long)(slab_bytes / 16))
                if (nseg == 5) MOMA_WIDE_PV(5); else if (nseg == 6) MOMA_WIDE_PV(6); else if (nseg == 8) MOMA_WIDE_PV(8);
                else if (nseg == 10) MOMA_WIDE_PV(10); else if (nseg == 12) MOMA_WIDE_PV(12); else MOMA_WIDE_PV(16);
#undef MOMA_WIDE_PV
            } else {
                int sl = 0;
                for (int col0 = 0; col0 < d; col0 += slab_width(col0), ++sl) {
                    const int D = slab_width(col0);
                    const SlabArgs sa{xs, m_part, (unsigned)(d * 2), 0};
                    const size_t slds = (size_t)NBUF * KT * D * 2 + 16;
                    uint4* op = (uint4*)((char*)o_part + (size_t)sl * slab_bytes);
#define MOMA_SLAB_PV(DD) hipLaunchKernelGGL((infonce_slab_kernel<DD, 2>), grid, block, slds, st, qpack, qu + col0, B, K, p.nbt, p.nchunk, p.tiles_per_chunk, p.Bpad, op, sa)
                    if (D == 512) MOMA_SLAB_PV(512); else if (D == 384) MOMA_SLAB_PV(384); else if (D == 256) MOMA_SLAB_PV(256); else MOMA_SLAB_PV(128);
#undef MOMA_SLAB_PV
                }
            }
        }
        if (ev_end && !ev_on_dispatch) (void)hipEventRecord(ev_end, st);      // (slab passes: recorded behind the last pass)
        {
            // column tiles per block (measured at d = 1280, B = 256: 1 / 2 / 4 / 8 -> 157 / 152 / 150 / 151 us per call; B = 64, where 4
            // leaves only 80 live workgroups: 1 -> 22.3 us for the combine against 16.5 at 4 -- every block repeats the row
            // statistics, and those, not the tiles, are what the wide-row combine spends its time on)
            int tpb = 4;
            const int nty = dq ? (d / 32 + tpb - 1) / tpb : 1;
            const int ey = enqueue_grid_rows(&job, p.Bpad / 8);
            hipExtLaunchKernelGGL((infonce_combine_kernel<4>), dim3(p.Bpad / 8, nty + ey), dim3(256), 0, st, (hipEvent_t) nullptr, ev_call_end, 0, q, k, B,
                                  d, inv_T, p.nchunk, p.Bpad, o_part, m_part, l_part, x_part, loss_rows, lse, top1, dq,
                                  (long)(slab_bytes / 16), cg, tpb, wide_ref, job, ey);
        }
        return hipGetLastError();
    }
    const size_t lds = (size_t)NBUF * KT * d * 2 + 16;      // ring + the 4 overflow words
#define MOMA_FLASH_ARGS qpack, qu, B, K, p.nbt, p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, m_part, l_part, x_part
    if (q_packed != nullptr) qpack = (uint4*)q_packed;
#define MOMA_FLASH_LAUNCH(DD)                                                                             \
    do {                                                                                                  \
        if (q_packed == nullptr) hipLaunchKernelGGL((infonce_qpack_kernel<DD>), dim3(((p.Bpad / 32) * (DD / 16) + 3) / 4), dim3(256), 0, st, q, B, d, 0, scale_log2, qpack, p.Bpad / 32); \
        /* measurement events ride on the dispatch itself (kernel begin / end timestamps, what rocprofv3 reports): events   */ \
        /* recorded around the launch add the latency of two event packets, ~3 us on a 36 us kernel                        */ \
        if (dq) hipExtLaunchKernelGGL((infonce_flash_kernel<DD, true>), grid, block, lds, st, ev_begin, ev_end, 0, MOMA_FLASH_ARGS);  \
        else hipExtLaunchKernelGGL((infonce_flash_kernel<DD, false>), grid, block, lds, st, ev_begin, ev_end, 0, MOMA_FLASH_ARGS);    \
    } while (0)
    if (d == 512) MOMA_FLASH_LAUNCH(512);
    else if (d == 384) MOMA_FLASH_LAUNCH(384);
    else if (d == 256) MOMA_FLASH_LAUNCH(256);
    else MOMA_FLASH_LAUNCH(128);
#undef MOMA_FLASH_LAUNCH
#undef MOMA_FLASH_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // (ev_call_end rides on the combine's dispatch: ev_begin .. ev_call_end spans the call's kernels, dispatch to dispatch)
    const int ny = dq ? d / 32 : 1, ey = enqueue_grid_rows(&job, p.Bpad / 8);
    hipExtLaunchKernelGGL((infonce_combine_kernel<1>), dim3(p.Bpad / 8, ny + ey), dim3(256), 0, st, (hipEvent_t) nullptr, ev_call_end, 0, q,
                          k, B, d, inv_T, p.nchunk, p.Bpad, o_part, m_part, l_part, x_part, loss_rows, lse, top1, dq, 0L, 1, 1, m_part, job, ey);
    return hipGetLastError();
}

size_t infonce_multi_workspace_bytes(int n_terms, int B, int d, int K) {
    const FlashPlan p = plan(B, K, d, n_terms);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    return (size_t)n_terms * (rows * d * 2 + 3 * rows * sizeof(float)) + (size_t)n_terms * p.Bpad * d * 2 + 2048;
}

bool infonce_multi_supported(int n_terms, int B, int d, int K, int qdtype, int prec) {
    return n_terms >= 1 && n_terms <= MULTI_MAX_TERMS && prec == MOMA_PREC_BF16 && qdtype == MOMA_DT_BF16 && one_pass_dim(d) && B >= 1 &&
           K >= 1 && plan(B, K, d, n_terms).nchunk <= COMBINE_MAX_CHUNKS;
}

hipError_t launch_infonce_multi(const moma_infonce_term_t* terms, int n_terms, int B, int d, int K, float inv_T, void* ws,
                                hipStream_t st) {
    const FlashPlan p = plan(B, K, d, n_terms);
    const size_t rows = (size_t)p.nchunk * p.Bpad;
    MultiArgs a{};
    a.n = n_terms;
    a.nq = 0;
    bool with_dq = false;
    for (int i = 0; i < n_terms; ++i) {
        int qs = -1;
        for (int j = 0; j < a.nq; ++j)
            if (a.qsets[j] == terms[i].q) qs = j;
        if (qs < 0) { qs = a.nq; a.qsets[a.nq++] = terms[i].q; }
        a.t[i] = MultiTerm{terms[i].q, terms[i].k, (const bf16_raw*)terms[i].queue, terms[i].loss_rows, terms[i].lse, terms[i].top1,
                           terms[i].dq, qs};
        with_dq = with_dq || terms[i].dq != nullptr;
    }
    for (int i = 0; i < n_terms; ++i)
        if (with_dq && terms[i].dq == nullptr) return hipErrorInvalidValue;      // all terms with dq, or none
    float* m_part = (float*)ws;
    float* l_part = m_part + (size_t)n_terms * rows;
    float* x_part = l_part + (size_t)n_terms * rows;
    uint4* o_part = (uint4*)(((uintptr_t)(x_part + (size_t)n_terms * rows) + 255) & ~(uintptr_t)255);
    const long o_term_stride = (long)(rows * d * 2 / 16);
    uint4* qpack = (uint4*)(((uintptr_t)((char*)o_part + (size_t)n_terms * rows * d * 2) + 255) & ~(uintptr_t)255);
    const long qset_stride = (long)p.Bpad * d * 2 / 16;
    std::call_once(g_lds_attr_once, set_lds_attrs);
    const float scale_log2 = inv_T * 1.4426950408889634f;
    const size_t lds = (size_t)NBUF * KT * d * 2 + 16;
    const dim3 grid(p.nbt * p.nchunk * n_terms), block(256);
#define MOMA_MULTI_LAUNCH(DD)                                                                                                   \
    do {                                                                                                                        \
        hipLaunchKernelGGL((infonce_qpack_multi_kernel<DD>), dim3(((p.Bpad / 32) * (DD / 16) + 3) / 4, a.nq), dim3(256), 0, st, a, B, \
                           scale_log2, qpack, p.Bpad / 32, qset_stride);                                                        \
        if (with_dq) hipLaunchKernelGGL((infonce_flash_multi_kernel<DD, true>), grid, block, lds, st, a, qpack, qset_stride, B, K, p.nbt, \
                                        p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, o_term_stride, m_part, l_part, x_part, (long)rows); \
        else hipLaunchKernelGGL((infonce_flash_multi_kernel<DD, false>), grid, block, lds, st, a, qpack, qset_stride, B, K, p.nbt,    \
                                p.nchunk, p.tiles_per_chunk, p.Bpad, o_part, o_term_stride, m_part, l_part, x_part, (long)rows);  \
    } while (0)
    if (d == 512) MOMA_MULTI_LAUNCH(512);
    else if (d == 384) MOMA_MULTI_LAUNCH(384);
    else if (d == 256) MOMA_MULTI_LAUNCH(256);
    else MOMA_MULTI_LAUNCH(128);
#undef MOMA_MULTI_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(infonce_combine_multi_kernel, dim3(p.Bpad / 8, with_dq ? d / 32 : 1, n_terms), dim3(256), 0, st, a, B, d, inv_T,
                       p.nchunk, p.Bpad, o_part, o_term_stride, m_part, l_part, x_part, (long)rows);
    return hipGetLastError();
}

}  // namespace moma
