// cfm_attn_kernels.h -- Cross-frame Feature Mining (CFM): fused windowed multi-head cross-attention.
//
// Reference semantics: WindowAttention3d3.forward, cffm_transformer.py:364-606 -- 49 target-window
// queries against 289 assembled keys (49 own | 132 cyclic 3-px ring :389-418 | 25 pooled-target
// :426-468 | 49+25+9 pooled reference-frame cells :470-518), 6 additive position biases + unfold
// padding masks (:536-587), softmax (:597), attn @ V (:601).  SURVEY.md A.3-A.8.
//
// MI355X design (the reference materialises k_all/v_all (48 MB/clip) and the 37 MB attention matrix):
//  * nothing is materialised: a workgroup owns one (clip, window, head); its 289 K/V rows are gathered
//    straight from the q/k/v GEMM output through a host-built key table (roll / unfold / cat / masks of
//    the reference become one int32 row index per key, -1 = unfold zero padding) as 64-byte f16 row
//    segments (one (token, head) slice; buffer loads: an absent key is an out-of-range offset that reads as
//    zeros) and staged in LDS as ROWS;
//  * blockIdx.x % 8 == head, so (as dispatched today) one XCD serves one head: its L2 holds the 64-ch
//    slice of every token and the per-head bias table, and neighbouring windows' overlapping
//    ring / pooled keys hit in that L2;
//  * QK^T is computed transposed (S^T = K Q^T, one 16x16x32 f16 MFMA per 16 keys since hd = 32 is
//    exactly one K-step) so the softmax row lives in the registers of 4 lanes: the reduction is
//    76 in-register ops + 2 wave shuffles, and P (un-normalised, <= 1) feeds the PV MFMA as the B
//    operand straight from registers, accumulating in f32; the A operand V^T is read out of the V ROWS
//    with the LDS transpose read (`ds_read_b64_tr_b16`, att_tr_frag; the k-slot <-> key bijection of P
//    is built into its row arithmetic) -- no transposed image is written by any kernel of this file.
//    f16 operands / f32 accumulate keep the block output within ~1e-4 of the fp32 reference (bf16
//    would miss the 1e-3 contract: SURVEY.md fact 10).
//  * LDS rows are 64 B with the 16-byte chunk index XOR-ed by 2*((row>>3)&1) (ATT_ROW): a ds_read_b128
//    is served in non-contiguous 16-lane groups, which no row padding can make conflict-free.
#pragma once
#include "cffa_kernels.h"

#define ATT_KS_STRIDE 32   // halfs per K/V/Q/dO row in LDS: 64 bytes, no padding, 16-byte chunks XOR-swizzled (ATT_ROW)
// element offset of 16-byte chunk `chunk` (0..3) of row `row`.  Two read patterns must both be conflict-free on the 64 x 4-byte
// banks (a 64-byte row covers 16 of them, so rows r and r+4 collide unless their chunks are permuted differently):
//  * MFMA fragment reads (ds_read_b128, lane (l15, g) reads chunk g of row l15): served in four NON-contiguous groups of 16
//    lanes ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS table), each mixing rows 0-3 / 12-15 with chunk g and rows 4-11
//    with chunk g^1 -- so padding the rows cannot help (80-byte rows: 45 % of the LDS cycles were bank conflicts);
//  * transposed reads (ds_read_b64_tr_b16 via att_tr_frag: 32 lanes read 32 contiguous bytes -- chunks {0,1} or {2,3} -- of 8
//    consecutive rows): rows r and r+4 need chunk sets that differ in bit 1.  (Round 1's swizzle, 2*((row>>3)&1), satisfied only
//    the first pattern: the V^T / K^T / Q^T / dO^T reads ran 2-way conflicted -- 48 % of the attention kernels' LDS cycles.)
// f(row) = [0,2,3,1][(row>>2)&3] XOR-ed into the chunk index satisfies both (checked exhaustively in tests/test_geometry.py).
#define ATT_SWZ(row) ((0x78 >> (((row) >> 1) & 6)) & 3)
#define ATT_ROW(row, chunk) ((row) * ATT_KS_STRIDE + 8 * ((chunk) ^ ATT_SWZ(row)))

// the 4 key-validity flags (0 / -inf) of this lane's keys 16t + 4g .. +3
__device__ __forceinline__ f32x4 vflag4(const float* vflag, int o) { return *(const f32x4*)(vflag + o); }
#define CFFM_FIRST_POOLED_KEY 181   // keys 0..180 = own window + ring: always present; 181.. = pooled cells (may fall off the grid)
#define ATT_FWD_LDS (2 * CFFM_NKEY_PAD * ATT_KS_STRIDE * sizeof(f16) + CFFM_NKEY_PAD * 4)
// MFMA A-operand fragment of the TRANSPOSED view of a row image `img` (ATT_ROW layout): A[i = column c0 + (lane & 15)]
// [k-slots 8 (lane >> 4) + j] = img[row r0 + 4 (lane >> 4) + j (+16 for j >= 4)][that column] -- the k-slot <-> row map of the
// S^T / S tiles held in registers (4 (lane >> 4) + r of two consecutive 16-row tiles).  Two LDS transpose reads.
// NROWS > 0: the image has only NROWS rows; a second-half row past it is read from NROWS - 16 + (its offset) instead -- any
// finite data do, the other operand's entries for those slots are exact zeros (keys 304..319 of the last PV / dQ k-step).
template <int NROWS = 0>
__device__ __forceinline__ f16x8 att_tr_frag(const f16* img, int r0, int c0, int lane) {
    const int i = lane & 15, row = r0 + 4 * (lane >> 4) + (i >> 2), col = c0 + 4 * (i & 3);
    int row2 = row + 16;
    if (NROWS > 0 && row2 >= NROWS) row2 -= 16;
    const f16x4 a = lds_tr4(img + ATT_ROW(row, col >> 3) + (col & 7));
    const f16x4 b = lds_tr4(img + ATT_ROW(row2, col >> 3) + (col & 7));
    return cat_f16x4(a, b);
}

// The same with the lane-dependent part of the address precomputed: for r0 a multiple of 16 the row swizzle depends only on the lane
// (ATT_SWZ looks at row bits 2..3, r0 contributes bits >= 4), so element offset = 32 r0 + att_tr_lane(c0, lane) (+ 512 for the second
// half).  The persistent kernels keep the two per-lane constants (c0 = 0, 16) in registers instead of re-deriving -- or, worse,
// hoisting -- one address per tile.
__device__ __forceinline__ int att_tr_lane(int c0, int lane) {
    const int i = lane & 15, row = 4 * (lane >> 4) + (i >> 2), col = c0 + 4 * (i & 3);
    return ATT_ROW(row, col >> 3) + (col & 7);
}
__device__ __forceinline__ f16x8 att_tr_frag_at(const f16* img_r0, int lane_off, bool clamp_second = false) {
    const f16x4 a = lds_tr4(img_r0 + lane_off);
    const f16x4 b = lds_tr4(img_r0 + lane_off + (clamp_second ? 0 : 16 * ATT_KS_STRIDE));
    return cat_f16x4(a, b);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
// Position bias on the matrix pipe.  `biasH` (k_bias_assemble): the head's additive bias as f16 B-operand fragments,
// [8 heads][4 waves][10 key-tile pairs][64 lanes][8]: lane 16 g + j holds bias(query 16 wave + j, keys 16 t + 8 (g & 1) .. + 7) of
// tile t = 2 p + (g >> 1) -- k-slots 0..15 of the fragment carry tile 2 p, k-slots 16..31 tile 2 p + 1 (round 4; rounds 2-3 kept one
// tile per fragment and let the upper 32 lanes read zeros: twice the registers for the same bytes).  With the two constant
// selector matrices Sel0[i][k] = [k == i], Sel1[i][k] = [k == 16 + i] (16 x 32, one register quad per lane each) the product
// Sel_(t & 1) * B is the [16 keys x 16 queries] bias tile of key tile t in exactly the C layout of the S^T = K Q^T tiles, so
// S^T = mfma(K, Q, mfma(Sel, B, mask)): the bias costs 512 bytes per (wave, tile) instead of 1 KB of fp32 C-in and no VALU
// instruction (the mask of the pooled tiles rides in as the first C-in), and a wave's whole bias is 10 fragments = 40 VGPRs.
// Round 1 read a query-major fp32 [8][64][304] table, 16 cache lines per quarter-wave; the fragment-ordered fp32 table of the first
// round-2 version halved the kernel's texture-address work but still moved 78 KB per workgroup, two thirds of its L2 traffic.
__device__ __forceinline__ f16x8 bias_sel_frag(int lane, int odd) {
    const int g = lane >> 4, l15 = lane & 15;
    f16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = ((g >> 1) == odd && 8 * (g & 1) + e == l15) ? (f16)1.f : (f16)0.f;
    return a;
}
__device__ __forceinline__ buf_t biash_rsrc(const h16* biasH) { return buf_make(biasH, (uint32_t)(BIASH_HALFS * 2)); }
__device__ __forceinline__ uint32_t biash_voff(int lane) { return 16u * (uint32_t)lane; }
__device__ __forceinline__ uint32_t biash_soff(int h, int wave, int p) { return (uint32_t)(((h * 4 + wave) * 10 + p) * 1024); }

// A window's K / V rows on their way from the f16 q|k|v rows to LDS, held in registers so that the gather of window w+1
// can be in flight while window w is multiplied (kv_load: key-table entries, then the 16-byte row segments; kv_store:
// K, V rows and the key-validity flags, which come from the same entries).
// Lane mapping: 4 ADJACENT lanes fetch the four 16-byte chunks of one (token, head) slice, i.e. a wave-wide gather touches
// 16 rows x 64 contiguous bytes.  (Round 1 gave a row's chunks to lanes 16 apart: every quarter-wave then addressed 16
// different cache lines for 256 bytes, and the gathers ran at the texture-address rate -- ~64 clocks per wave-instruction --
// instead of the data rate.)  Thread t owns chunk t & 3 of rows (t >> 2) + (NTHREADS / 4) * it.
template <int NTHREADS>
struct KvRegs {
    static constexpr int RPB = NTHREADS / 4, NIT = (CFFM_NKEY_PAD + RPB - 1) / RPB;   // rows per batch, batches (256 threads: 64, 5)
    int src[NIT];
    f16x8 k[NIT], v[NIT];
};
// The gather is two dependent loads (key-table entry, then the row it names).  A wave issues in order, so fetching both in
// one go parks it for a full memory latency between them; the persistent kernels therefore fetch the TABLE entries two
// windows ahead (KvTab) and the rows one window ahead.
template <int NTHREADS>
struct KvTab { int s[KvRegs<NTHREADS>::NIT]; };
template <int NTHREADS>
__device__ __forceinline__ void kv_tab_load(KvTab<NTHREADS>& t, const int* __restrict__ ksrc, int tid) {
    constexpr int RPB = KvRegs<NTHREADS>::RPB, NIT = KvRegs<NTHREADS>::NIT;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 2) + RPB * it;
        t.s[it] = row < CFFM_NKEY_PAD ? ksrc[row] : -1;
    }
}
// rows named by the table entries, through a buffer resource over the whole q|k|v array: one 32-bit offset per gather, a
// "no such key" entry (-1) becomes an out-of-range offset that reads as zeros -- no branches, no 64-bit address arithmetic.
// soff_k = byte offset of (clip b, row 0, this head's K slice): ((b*RC)*768 + 256 + h*32) * 2; V is 512 bytes further.
// one of the NIT batches (2 gathers): the persistent kernels spread the batches over their compute loop, so the texture
// path works through the gathers in the background instead of stalling the wave on a full request queue
template <int NTHREADS>
__device__ __forceinline__ void kv_rows_load_it(KvRegs<NTHREADS>& r, const KvTab<NTHREADS>& t, buf_t rs_qkv, uint32_t soff_k, int tid, int it) {
    r.src[it] = t.s[it];
    const uint32_t o = t.s[it] >= 0 ? (uint32_t)t.s[it] * 1536u + (uint32_t)(tid & 3) * 16u : BUF_OOB;
    r.k[it] = buf_ld_h8(rs_qkv, o, soff_k);
    r.v[it] = buf_ld_h8(rs_qkv, o, soff_k + 512);
}
template <int NTHREADS>
__device__ __forceinline__ void kv_rows_load(KvRegs<NTHREADS>& r, const KvTab<NTHREADS>& t, buf_t rs_qkv, uint32_t soff_k, int tid) {
#pragma unroll
    for (int it = 0; it < KvRegs<NTHREADS>::NIT; ++it) kv_rows_load_it<NTHREADS>(r, t, rs_qkv, soff_k, tid, it);
}
template <int NTHREADS>
__device__ __forceinline__ void kv_load(KvRegs<NTHREADS>& r, buf_t rs_qkv, uint32_t soff_k, const int* __restrict__ ksrc, int tid) {
    KvTab<NTHREADS> t;
    kv_tab_load<NTHREADS>(t, ksrc, tid);
    kv_rows_load<NTHREADS>(r, t, rs_qkv, soff_k, tid);
}
__device__ __forceinline__ buf_t qkv_rsrc(const Geo& G, const h16* qkv) { return buf_make(qkv, (uint32_t)((long)G.B * G.RC * 768 * 2)); }
__device__ __forceinline__ uint32_t qkv_soff_k(const Geo& G, int b, int h) { return (uint32_t)(((long)b * G.RC * 768 + 256 + h * CFFM_HD) * 2); }
// registers -> LDS: K and V rows (ATT_ROW layout) and the key-validity flags.  8 adjacent lanes write 2 whole rows: no bank
// conflicts under any swizzle.  (No transposed image: the kernels read the transposed views they need with the LDS
// transpose read, att_tr_frag.)
template <int NTHREADS>
__device__ __forceinline__ void kv_store(const KvRegs<NTHREADS>& r, f16* Ks, f16* Vs, float* vflag, int tid) {
    constexpr int RPB = KvRegs<NTHREADS>::RPB, NIT = KvRegs<NTHREADS>::NIT;
    const int c = tid & 3;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = (tid >> 2) + RPB * it;
        if (row < CFFM_NKEY_PAD) {
            if (c == 0) vflag[row] = r.src[it] >= 0 ? 0.f : -INFINITY;
            *(f16x8*)(Ks + ATT_ROW(row, c)) = r.k[it];
            *(f16x8*)(Vs + ATT_ROW(row, c)) = r.v[it];
        }
    }
}
template <int NTHREADS>
__device__ __forceinline__ void stage_kv(buf_t rs_qkv, uint32_t soff_k, const int* __restrict__ ksrc, f16* Ks, f16* Vs, float* vflag,
                                         int tid) {
    KvRegs<NTHREADS> r;
    kv_load<NTHREADS>(r, rs_qkv, soff_k, ksrc, tid);
    kv_store<NTHREADS>(r, Ks, Vs, vflag, tid);
}

// grid: B*nW*8 workgroups (head fastest), 256 threads
// (K / V staged through registers, not by LDS-DMA: 14.5 vs 15.0 us -- the staging is bound by the L2 -> CU burst of all resident workgroups,
//  not by the ds_write pass; the LDS-DMA leg, the ablation switches and the shader-clock stamps of rounds 2-5 live in the repository's
//  history: scripts/README.md, "profiling legs")
#define FWD_OCC 4   // workgroups per CU: 39.2 KB of LDS and <= 128 VGPRs each
__global__ void __launch_bounds__(256, FWD_OCC) k_cfm_attn_fwd(Geo G, const h16* __restrict__ qkv,
                                                       const int* __restrict__ key_src, const int* __restrict__ q_dst,
                                                       const h16* __restrict__ biasH, float* __restrict__ ao,
                                                       float* __restrict__ lse_out) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
    float* vflag = (float*)(Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE);

    const int h = blockIdx.x & 7, wb = blockIdx.x >> 3, w = wb % G.nW, b = wb / G.nW;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int qcol = 16 * wave + (lane & 15), g = lane >> 4, l15 = lane & 15;
    // ---- stage ----
    // Global loads go out in three batches, each complete before anything waits on it: the key-table entries of this thread's 5
    // rows (+ the destination pixel the epilogue needs), then the 10 gathered 16-byte K/V segments and this lane's Q fragment
    // (straight into the MFMA operand: Q never touches LDS), then the wave's 10 bias fragments.
    // A (token, head) slice is 64 B of f16 = four 16-byte chunks.  K and V are both kept as ROWS: the PV step reads V
    // through the LDS transpose read (lds_tr4), so no transposed image is written.
    // (One workgroup per (window, head), NOT persistent over windows: round 6 measured the loop form with the bias fragments kept
    // across windows at 14.3-15.7 us against 13.9-14.4 -- they would have to live through the PV phase next to s[19], which spills at
    // four workgroups per CU and costs the fourth workgroup otherwise: profiles/r06_attn_fwd_persistent_ab.txt.)
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const int qdst = (qcol < CFFM_WA) ? q_dst[w * CFFM_WA + qcol] : -1;
    KvRegs<256> kv;
    kv_load<256>(kv, rs_qkv, qkv_soff_k(G, b, h), key_src + w * CFFM_NKEY_PAD, tid);
    const f16x8 qfrag = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)(w * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                                  (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
    const buf_t rs_bias = biash_rsrc(biasH);
    const uint32_t bvoff = biash_voff(lane);
    f16x8 bh[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) bh[t] = buf_ld_h8(rs_bias, bvoff, biash_soff(h, wave, t));
    kv_store<256>(kv, Ks, Vs, vflag, tid);
    __syncthreads();
    const f16x8 sel0 = bias_sel_frag(lane, 0), sel1 = bias_sel_frag(lane, 1);
    f16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (f16)1.f;

    // ---- S^T = K Q^T (+bias, +mask), softmax over the 289 keys of each query column -----------------
    // VALU budget: per 16-key tile and lane 2 v_max3 + 4 (v_fma + v_exp) + 2 v_cvt_pk.  The position bias arrives on the matrix
    // pipe (Sel * B, see bias_sel_frag) with the mask of the tiles that can hold an absent key (the pooled groups, keys >= 181:
    // own and ring keys always exist) as ITS C-in; exp(s - m) is 2^(s log2e - m log2e), one FMA feeding v_exp_f32; the row sums
    // come out of the matrix pipe too (a constant all-ones A operand next to V^T: 10 more MFMAs instead of 76 adds).
    f32x4 s[19];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 19; ++t) {
        const f16x8 kf = *(const f16x8*)(Ks + ATT_ROW((16 * t + l15), g));
        const f32x4 c0 = (16 * t + 15 >= CFFM_FIRST_POOLED_KEY) ? vflag4(vflag, 16 * t + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 acc = mfma16x16x32_f16(kf, qfrag, mfma16x16x32_f16((t & 1) ? sel1 : sel0, bh[t >> 1], c0));   // K Q^T + bias + mask
        s[t] = acc;
        m = fmaxf(fmaxf(m, acc[0]), fmaxf(fmaxf(acc[1], acc[2]), acc[3]));    // (two v_max3_f32)
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float m2 = m * CFFM_LOG2E;

    // ---- O^T = V^T P^T : A = V^T[d][key slots] read transposed out of the V rows, B = P^T from registers; the third
    //      accumulator (A = ones) is the softmax denominator of the f16-rounded weights the product really uses ----------
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, osum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
        f16x4 ph[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * kt + u;
            if (t < 19) {
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = fast_exp2(fmaf(s[t][r], CFFM_LOG2E, -m2));
                ph[u] = to_f16x4(p);
            } else {
                ph[u] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            }
        }
        const f16x8 pf = cat_f16x4(ph[0], ph[1]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            o[mt] = mfma16x16x32_f16(att_tr_frag<CFFM_NKEY_PAD>(Vs, 32 * kt, 16 * mt, lane), pf, o[mt]);
        osum = mfma16x16x32_f16(ones, pf, osum);
    }
    const float l = osum[0];      // every row of the ones-product is the column sum: l of query l15, in all four lane groups

    // ---- epilogue: normalise, un-window, drop padded pixels (cffm_transformer.py:812-821) ------------
    if (g == 0) lse_out[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + qcol] = (qcol < CFFM_WA) ? m + logf(l) : 0.f;
    if (qdst >= 0) {
        const float inv = 1.f / l;
        float* orow = ao + ((long)b * G.HW + qdst) * CFFM_C + h * CFFM_HD + 4 * g;
        *(f32x4*)(orow) = o[0] * inv;
        *(f32x4*)(orow + 16) = o[1] * inv;
    }
}

// =====================================================================================================
// Fused backward (round 4: key-split form; rounds 2-3 ran 4 waves x 16 queries in the S^T orientation and exchanged P and dS through
// LDS per 32-key chunk, 12 barriers per window -- scripts/r03_attn_bwd_chunked/).  grid (8 heads, NG window groups).
//   * ONE workgroup of 12 waves per CU walks its window group; waves 0..7 own key tiles 2w, 2w+1 of every window, waves 8..10
//     tiles 16..18 (5 / 5 / 5 / 4 tiles per SIMD) against ALL 64 queries, in the S = Q K^T orientation (C rows = queries, C columns = keys, a lane
//     = one key): P and dS of a tile contract over QUERIES -- dV^T = dO^T P, dK^T = Q^T dS -- straight from the C registers (B operand,
//     the k-slot <-> query map of att_tr_frag on the Q / dO rows), so P never leaves the registers and the head's bias gradient of
//     the wave's two tiles accumulates in 32 registers over the whole group;
//   * only dS crosses LDS, once per window: two [320 keys][32 queries] images (ATT_ROW rows, the lane's 4 queries = 8 bytes) that
//     the query phase -- after ONE barrier per window -- reads back transposed for dQ^T = K^T dS^T (8 of the waves: one (query tile,
//     channel half) each, 10 MFMAs);
//   * K / V / Q rows of window i+1 arrive by LDS-DMA into the other image while window i is multiplied (the table slices that
//     address them by LDS-DMA a window before that); the dO / O / LSE rows of window i+1 land, fp32 as fetched, in a parking
//     area during window i's key phase and are converted (window-wide power-of-two scale, D from the rounded values) by two waves
//     during its query phase.  No staging registers, no VALU staging pass; waits are counted (vmcnt(N)), never vmcnt(0).
// Two barriers per window instead of twelve, one exp per (query, key) as before.
// =====================================================================================================
#define ATT_BK_THREADS 768
#define ATT_BK_IMG (2 * CFFM_NKEY_PAD * ATT_KS_STRIDE)      // halfs of one K | V image
#define ATT_BK_DSROWS 320                                   // dS image rows: 10 pairs x 32 keys (rows 304.. stay zero)
#define ATT_BK_TAB (320 + 64)                               // ints of one table buffer: the window's key-table slice, its q_dst slice
#define ATT_BK_RAW (2 * 64 * CFFM_HD + 64)                  // floats of the parking area: dO rows, O rows, the LSE row
#define ATT_BK_LDS ((2 * ATT_BK_IMG + 4 * 64 * ATT_KS_STRIDE + 2 * ATT_BK_DSROWS * ATT_KS_STRIDE) * 2 + (ATT_BK_RAW + 2 * CFFM_NKEY_PAD + 4 * 64 + 4 + 8 + 3 * ATT_BK_TAB) * 4)
__device__ __forceinline__ void wait_vm0() {   // every outstanding global access of this wave -- LDS-DMA included -- has completed
#ifndef CFFM_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), gfx9 encoding
#endif
}
// all but the N youngest global accesses of this wave have completed (gfx9: loads, LDS-DMA and stores share one in-order counter;
// vmcnt bits [3:0] and [15:14], the other counters left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vm() {
#ifndef CFFM_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
#endif
}
__global__ void __launch_bounds__(ATT_BK_THREADS) k_cfm_attn_bwd(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                                     const int* __restrict__ q_dst, const h16* __restrict__ biasH,
                                                                     const float* __restrict__ ao, const float* __restrict__ dao,
                                                                     const float* __restrict__ lse_in, float* __restrict__ dqkv,
                                                                     float* __restrict__ dbias_part, float* __restrict__ dkv_part, int per_group) {
    CFFM_DYN_SMEM(smem);
    f16* img = (f16*)smem;                                   // [2][K rows | V rows]
    f16* Qs = img + 2 * ATT_BK_IMG;                          // [2][64 query rows]
    f16* dOs = Qs + 2 * 64 * ATT_KS_STRIDE;                  // [2][64 query rows], dO * sc
    f16* DS = dOs + 2 * 64 * ATT_KS_STRIDE;                  // [2 query halves][320 key rows][32 queries]
    float* raw = (float*)(DS + 2 * ATT_BK_DSROWS * ATT_KS_STRIDE);   // the NEXT window's dO [64][32] | O [64][32] | LSE [64], fp32 as fetched
    float* vfl = raw + ATT_BK_RAW;                           // [2][304] key validity (0 / -inf)
    float* lse2 = vfl + 2 * CFFM_NKEY_PAD;                   // [2][64] LSE * log2(e)
    float* sD = lse2 + 128;                                  // [2][64] rowsum(dO_h * O)
    float* sisc = sD + 128;                                  // [2] 1 / sc of the window
    float* smax = sisc + 4;                                  // [8] |dO| maxima of the parked rows, one per fetching wave
    int* tabs = (int*)(smax + 8);                            // [3][key-table slice 320 | q_dst slice 64] of the windows to come

    const int h = blockIdx.x, grp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const float scale = 0.17677669529663687f;
    const int wb0 = grp * per_group;
    const int wb1 = (wb0 + per_group < G.B * G.nW) ? wb0 + per_group : G.B * G.nW;
    const buf_t rs_bias = biash_rsrc(biasH);
    const buf_t rs_part = buf_make(dkv_part, (uint32_t)((long)G.B * G.nW * CFFM_NKEY_PAD * 512 * 2));
    float* part_scale = dkv_part + (long)G.B * G.nW * CFFM_NKEY_PAD * 256;
    const dma_t dm_qkv = dma_make(qkv, (uint32_t)((long)G.B * G.RC * 768 * 2));
    const dma_t dm_ao = dma_make(ao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const dma_t dm_dao = dma_make(dao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const dma_t dm_lse = dma_make(lse_in, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * CFFM_NQ_PAD * 4));
    const dma_t dm_tab = dma_make(key_src, (uint32_t)(G.nW * CFFM_NKEY_PAD * 4));
    const dma_t dm_qd = dma_make(q_dst, (uint32_t)(G.nW * CFFM_WA * 4));
    const int drow = lane >> 2, sc4 = lane & 3;              // K / V / Q DMA role: row of the wave's tile, chunk position
    const int rrow = 8 * wave + (lane >> 3);                 // dO / O DMA role (waves 0..7): query row, 16-byte chunk lane & 7 of its 128 bytes
    // key-phase role: waves 0..7 own key tiles 2 w, 2 w + 1, waves 8..10 tiles 16..18, wave 11 none -- a workgroup's waves go to the four
    // SIMDs cyclically, so every SIMD runs 5 tiles (one 4): with ten waves x 2 tiles two SIMDs ran 6 and 5 tiles against 4 on the
    // others and the barrier waited ~2 k cycles per window for the third wave of the fullest SIMD (shader-clock stamps)
    const int ntile = wave < 8 ? 2 : (wave < 11 ? 1 : 0), t0 = wave < 8 ? 2 * wave : 8 + wave;
    // per-lane LDS address parts (element offsets; tiles / pairs add multiples of 512)
    const int lrow = ATT_ROW(l15, g);                        // row-fragment reads: row 16 t + l15, chunk g
    const int ltr0 = att_tr_lane(0, lane), ltr1 = att_tr_lane(16, lane);
    // transposed reads of the Q / dO rows for the key-owner products with the CHANNELS permuted: the LDS transpose read hands lane i the
    // column its source lanes address, so letting quad c of a 16-lane group read columns 8 c + 4 mt .. + 3 makes C row 4 g + r of
    // product mt channel 8 g + 4 mt + r -- a lane then owns 8 CONSECUTIVE channels of its key over mt = 0, 1: one 16-byte store
    const int ltp0 = ATT_ROW(4 * g + (l15 >> 2), l15 & 3), ltp1 = ltp0 + 4;
    const int dsw0 = ATT_ROW(l15, (g >> 1)) + 4 * (g & 1), dsw1 = ATT_ROW(l15, 2 + (g >> 1)) + 4 * (g & 1);   // dS image writes
    // the selector of the wave's first / second tile inside its pair's bias fragment
    const f16x8 sel0 = bias_sel_frag(lane, wave_uniform(wave < 8 ? 0 : (t0 & 1))), sel1 = bias_sel_frag(lane, 1);

    // the wave's bias fragments: the pair of its tiles for the four query tiles (A operands: rows = queries)
    f16x8 bT[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) bT[qt] = buf_ld_h8(rs_bias, biash_voff(lane), biash_soff(h, qt, t0 >> 1));
    f32x4 dB[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) dB[u][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid < 128) {   // dS rows 304..319 (the missing tile 19) of both halves: zeros, never written again
        f16x8 z8;
        for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
        *(f16x8*)(DS + ((tid >> 6) * ATT_BK_DSROWS + CFFM_NKEY_PAD) * ATT_KS_STRIDE + 8 * (tid & 63)) = z8;
    }

    // what a lane needs of the tables to address its DMAs: the key-table entries of its two K / V rows and the destination pixel of its
    // dO / O row (-1: padding).  First window: ordinary loads; afterwards the table slices themselves travel by LDS-DMA a window ahead
    // (tabs), so that the loop holds NO ordinary global load: the compiler, which cannot see the DMAs, would put an s_waitcnt vmcnt(0)
    // in front of the first use of any load result -- the round trip of every DMA and store in flight (measured: 4 such stalls per
    // window in the first version of this kernel).
    struct Ahead { int src[2]; int qd; };
    struct Win { int wb, w, b; };       // window, its index inside the clip, the clip (kept incrementally: no division in the loop)
    auto win_next = [&](const Win& x) { Win y = {x.wb + 1, x.w + 1, x.b}; if (y.w == G.nW) { y.w = 0; y.b += 1; } return y; };
    auto ahead_load = [&](int w, Ahead& a) {
        const int* ksrc = key_src + w * CFFM_NKEY_PAD;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it;
            a.src[it] = i < 19 ? ksrc[16 * i + drow] : -1;
        }
        a.qd = (wave < 8 && rrow < CFFM_WA) ? q_dst[w * CFFM_WA + rrow] : -1;
    };
    auto ahead_lds = [&](int tb, Ahead& a) {
        const int* t = tabs + tb * ATT_BK_TAB;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it;
            a.src[it] = i < 19 ? t[16 * i + drow] : -1;
        }
        a.qd = (wave < 8 && rrow < CFFM_WA) ? t[320 + rrow] : -1;
    };
    // table slices of window wb -> buffer tb: waves 7..11 one 64-entry piece of the key table each (the last one runs 16 entries into the
    // next window's slice or off the table: never used), wave 6 the q_dst slice
    auto tab_issue = [&](int w, int tb) {
        if (wave >= 7) dma_ld4(dm_tab, 4u * (uint32_t)(64 * (wave - 7) + lane), (uint32_t)(w * CFFM_NKEY_PAD * 4), tabs + tb * ATT_BK_TAB + 64 * (wave - 7));
        else if (wave == 6) dma_ld4(dm_qd, 4u * (uint32_t)lane, (uint32_t)(w * CFFM_WA * 4), tabs + tb * ATT_BK_TAB + 320);
        sched_fence();
    };
    // A window's DMAs: dma_prep turns table entries into byte offsets (plain registers: a register copy of a load result would cost an
    // s_waitcnt vmcnt(0), i.e. the round trip of every access in flight); dma_rows (step 1, right behind B1): the key-validity flags,
    // the dO / O rows (waves 0..7: 8 rows x 128 bytes each) and the LSE row (wave 8) -> the parking area; dma_piece (step 2, inside the
    // key phase): K / V tiles wave, wave + 12 and (waves 0..3) 16 Q rows -> image bi.
    struct DmaOff { uint32_t kv[2], q, rows; };
    auto dma_prep = [&](int w, const Ahead& a, DmaOff& o) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it, row = 16 * i + drow;
            o.kv[it] = a.src[it] >= 0 ? (uint32_t)a.src[it] * 1536u + 16u * (uint32_t)(sc4 ^ ATT_SWZ(row)) : BUF_OOB;
        }
        const int qrow = 16 * wave + drow;
        o.q = (wave < 4 && qrow < CFFM_WA) ? (uint32_t)(w * CFFM_WA + qrow) * 1536u + 16u * (uint32_t)(sc4 ^ ATT_SWZ(qrow)) : BUF_OOB;
        o.rows = a.qd >= 0 ? (uint32_t)a.qd * (CFFM_C * 4u) + 16u * (uint32_t)(lane & 7) : BUF_OOB;
    };
    // step 1 (top of the window): the key-validity flags, the dO / O / LSE rows
    auto dma_rows = [&](const Win& x, const DmaOff& o, int bi) {
        const int wb = x.wb, b = x.b;
        const uint32_t ps = (uint32_t)(((long)b * G.HW * CFFM_C + h * CFFM_HD) * 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it;
            if (i < 19 && sc4 == 0) vfl[bi * CFFM_NKEY_PAD + 16 * i + drow] = o.kv[it] != BUF_OOB ? 0.f : -INFINITY;
        }
        sched_fence();
        if (wave < 8) {
            dma_ld16(dm_dao, o.rows, ps, raw + 8 * wave * CFFM_HD);
            dma_ld16(dm_ao, o.rows, ps, raw + 64 * CFFM_HD + 8 * wave * CFFM_HD);
        } else if (wave == 8) {
            dma_ld4(dm_lse, 4u * (uint32_t)lane, (uint32_t)((((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD) * 4), raw + 2 * 64 * CFFM_HD);
        }
        sched_fence();
    };
    // step 2, piece k = 0..4: K, V of tile wave; K, V of tile wave + 12; (waves 0..3) 16 Q rows.  The pieces go out one at a time between the
    // chains of the wave's first key tile: all waves issuing their five to seven DMAs at once right behind the barrier kept every wave
    // of the CU in the texture addresser's queue for ~2 k cycles (shader-clock stamps, scripts/r04_ks_timing.py).
    auto dma_piece = [&](int b, const DmaOff& o, int bi, int k) {
        const uint32_t soff_k = qkv_soff_k(G, b, h);
        f16* Ks = img + bi * ATT_BK_IMG;
        f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
        if (k < 4) {
            const int it = k >> 1, i = wave + 12 * it;
            if (i < 19) dma_ld16(dm_qkv, o.kv[it], soff_k + 512 * (k & 1), ((k & 1) ? Vs : Ks) + 16 * i * ATT_KS_STRIDE);
        } else if (wave < 4) {
            dma_ld16(dm_qkv, o.q, (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2), Qs + (bi * 64 + 16 * wave) * ATT_KS_STRIDE);
        }
    };
    // |dO| maximum of the 8 rows this wave fetched (they have landed: its own wait) -> smax[wave]
    auto rows_max = [&]() {
        const f32x4 r = *(const f32x4*)(raw + 8 * wave * CFFM_HD + 4 * lane);
        const float am = wave_max(fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3]))));
        if (lane == 0) smax[wave] = am;
    };
    // parked rows -> dO_h = f16(dO * sc) rows of image bi, D = rowsum(dO_h * O), LSE * log2(e), 1 / sc.  dO is rescaled per window by
    // a power of two so that every f16 gradient operand sits near 1 (training-size gradients of 1e-6 would flush to zero in f16);
    // results are scaled back in f32.  D comes from the ROUNDED dO: with dP = dO_h V^T the kernel then sees sum_n P_n (dP_n - D) = 0
    // exactly, i.e. the exact softmax backward of a dO perturbed by 2^-12 per element; with D from the unrounded dO the rounding error
    // of dP met an exact D in the cancelling difference dP - D (stage test: 7e-4 of max|dq| against 2.8e-4).  Waves 8..11 (the other eight run the query phase): lane l of wave 8 + v owns 8 channels l & 3 of row 16 v + (l >> 2).
    auto rows_convert = [&](int wb, int bi) {
        const f32x4 m0 = *(const f32x4*)(smax), m1 = *(const f32x4*)(smax + 4);
        const float am = fmaxf(fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3])), fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3])));
        int ex = 0;
        if (am > 0.f) frexpf(am, &ex);
        const float sc = (am > 0.f) ? ldexpf(1.f, 1 - ex) : 1.f, isc = 1.f / sc;   // max|dO * sc| in [1,2) over the window
        {
            const int row = 16 * (wave - 8) + (lane >> 2), c = lane & 3;
            const float* p = raw + row * CFFM_HD + 8 * c;
            const f32x4 r0 = *(const f32x4*)(p), r1 = *(const f32x4*)(p + 4);
            const f32x4 o0 = *(const f32x4*)(p + 64 * CFFM_HD), o1 = *(const f32x4*)(p + 64 * CFFM_HD + 4);
            f16x8 dh;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dh[e] = (f16)(r0[e] * sc); dh[4 + e] = (f16)(r1[e] * sc);
                d = fmaf((float)dh[e], o0[e], d);
                d = fmaf((float)dh[4 + e], o1[e], d);
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if (c == 0) sD[bi * 64 + row] = d;
            *(f16x8*)(dOs + bi * 64 * ATT_KS_STRIDE + ATT_ROW(row, c)) = dh;
        }
        if (wave == 8) {
            lse2[bi * 64 + lane] = raw[2 * 64 * CFFM_HD + lane] * CFFM_LOG2E;
            if (lane == 0) { part_scale[(long)wb * CFFM_HEADS + h] = isc; sisc[bi] = isc; }
        }
    };

    // loop-carried: the current window and the DMA offsets of the next one (plain arithmetic results, computed in the tail of the previous
    // key phase from table slices that landed a window before: nothing in flight crosses the back edge, nothing is prepared behind B1)
    Win cur = {wb0, wave_uniform(wb0 % G.nW), wave_uniform(wb0 / G.nW)};
    DmaOff onx = {{BUF_OOB, BUF_OOB}, BUF_OOB, BUF_OOB};
    if (wb0 < wb1) {
        const Win n1 = win_next(cur), n2 = win_next(n1);
        Ahead a;
        DmaOff o;
        ahead_load(cur.w, a);
        dma_prep(cur.w, a, o);
        dma_rows(cur, o, 0);
#pragma unroll
        for (int k = 0; k < 5; ++k) dma_piece(cur.b, o, 0, k);
        sched_fence();
        if (n1.wb < wb1) { ahead_load(n1.w, a); dma_prep(n1.w, a, onx); }
        if (n2.wb < wb1) tab_issue(n2.w, 2);
        wait_vm0();
        if (wave < 8) rows_max();
        // (the bias fragments are complete: say so to the compiler, which otherwise waits for them -- vmcnt(0) -- at their first use
        // in EVERY window, behind the DMAs issued there)
#ifndef CFFM_EMU
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) asm volatile("" : "+v"(bT[qt]));
#endif
        __syncthreads();
        if (wave >= 8) rows_convert(wb0, 0);
    }
    int r3 = 0;   // (wb - wb0) % 3: table buffer r3 is free (it held window wb's slices), buffer (r3 + 2) % 3 holds window wb + 2's
    for (int wb = wb0; wb < wb1; ++wb) {
        const int bi = (wb - wb0) & 1, w = cur.w, b = cur.b;
        const Win n1 = win_next(cur), n2 = win_next(n1), n3 = win_next(n2);
        // The K / V / Q DMAs of this window were issued a window ago; the only accesses younger than them are the wave's stores of the
        // previous window (2 partial-row stores per key tile and at most one more): waiting for those as well would expose their round
        // trip to HBM at every window.  (Not so for the first window: its DMAs were issued just now.)
        if (wb == wb0 || ntile == 0) wait_vm0(); else if (ntile == 1) wait_vm<2>(); else wait_vm<4>();
        __syncthreads();   // B1: window wb's K / V / Q rows are in image bi, its dO_h rows, D, LSE, 1 / sc are written; image bi ^ 1, the dS
                           //     images and the parking area are free
        const bool more = wb + 1 < wb1;
        if (more) dma_rows(n1, onx, bi ^ 1);   // flags and row DMAs of window wb + 1; the rest goes out inside the key phase
        const f16* Ks = img + bi * ATT_BK_IMG;
        const f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
        const f16* Q = Qs + bi * 64 * ATT_KS_STRIDE;
        const f16* dO = dOs + bi * 64 * ATT_KS_STRIDE;
        const float isc = sisc[bi];
        // ---- key phase: the wave's key tiles against all 64 queries
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t0 + u;
            if (u < ntile) {
                sched_fence();
                const f16x8 kf = *(const f16x8*)(Ks + 512 * t + lrow);
                const f16x8 vf = *(const f16x8*)(Vs + 512 * t + lrow);
                const float mk = t >= 11 ? vfl[bi * CFFM_NKEY_PAD + 16 * t + l15] : 0.f;   // only keys >= 181 can be absent
                const f32x4 c0 = (f32x4){mk, mk, mk, mk};
                f16x4 ph[4], dsh[4];
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    if (u == 0 && more) dma_piece(n1.b, onx, bi ^ 1, qt);
                    const f16x8 qf = *(const f16x8*)(Q + 512 * qt + lrow);
                    const f16x8 dof = *(const f16x8*)(dO + 512 * qt + lrow);
                    const f32x4 sv = mfma16x16x32_f16(qf, kf, mfma16x16x32_f16(bT[qt], u ? sel1 : sel0, c0));   // Q K^T + bias + mask
                    const f32x4 dp = mfma16x16x32_f16(dof, vf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    const f32x4 lq = *(const f32x4*)(lse2 + bi * 64 + 16 * qt + 4 * g), Dq = *(const f32x4*)(sD + bi * 64 + 16 * qt + 4 * g);
                    f32x4 pr, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pr[r] = fast_exp2(fmaf(sv[r], CFFM_LOG2E, -lq[r]));
                        ds[r] = pr[r] * (dp[r] - Dq[r]);
                    }
                    dB[u][qt] += ds * isc;
                    ph[qt] = to_f16x4(pr);
                    dsh[qt] = to_f16x4(ds);
                    // dS image: row = key, 8 bytes = queries 16 qt + 4 g .. + 3
                    *(f16x4*)(DS + (qt >> 1) * ATT_BK_DSROWS * ATT_KS_STRIDE + 512 * t + ((qt & 1) ? dsw1 : dsw0)) = dsh[qt];
                }
                if (u == 0 && more) {   // the last pieces: Q rows, the table slices of window wb + 3 -- all DMAs precede the wave's stores
                    dma_piece(n1.b, onx, bi ^ 1, 4);
                    if (wb + 3 < wb1) tab_issue(n3.w, r3);
                }
                f32x4 aK[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, aV[2] = {aK[0], aK[0]};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f16x8 pf = cat_f16x4(ph[2 * a], ph[2 * a + 1]), dsf = cat_f16x4(dsh[2 * a], dsh[2 * a + 1]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        aV[mt] = mfma16x16x32_f16(att_tr_frag_at(dO + 1024 * a, mt ? ltp1 : ltp0), pf, aV[mt]);
                        aK[mt] = mfma16x16x32_f16(att_tr_frag_at(Q + 1024 * a, mt ? ltp1 : ltp0), dsf, aK[mt]);
                    }
                }
                // tiles [channel 8 g + 4 mt + r][key = l15]: 16 bytes of the key's partial row (8 heads x (K 32 | V 32) halfs) per store,
                // a quarter-wave covers the head's whole 64-byte K (V) slice; an absent key's stores go out of range and are dropped
                const uint32_t so = (uint32_t)(((long)wb * CFFM_NKEY_PAD + 16 * t) * 1024);
                const uint32_t vo = mk == 0.f ? (uint32_t)(l15 * 1024 + (h * 2 * CFFM_HD + 8 * g) * 2) : BUF_OOB;
                buf_st16_pair(rs_part, __builtin_bit_cast(f32x4, cat_f16x4(to_f16x4(aK[0]), to_f16x4(aK[1]))),
                              __builtin_bit_cast(f32x4, cat_f16x4(to_f16x4(aV[0]), to_f16x4(aV[1]))), vo, so, so + 64);
            }
        }
        // the next window's dO / O / LSE rows have landed: of this wave's accesses only its K / V DMAs, the Q or table DMA and the
        // partial-row stores are younger than the row DMAs
        if (more && ntile == 0) {   // wave 11 has no key tile to issue its DMA pieces from
#pragma unroll
            for (int k = 0; k < 5; ++k) dma_piece(n1.b, onx, bi ^ 1, k);
            if (wb + 3 < wb1) tab_issue(n3.w, r3);
        }
        if (wb + 2 < wb1) {   // the offsets of window wb + 2 (its slices landed before B1), behind the last use of window wb + 1's
            Ahead a;
            sched_fence();
            ahead_lds(r3 + 2 >= 3 ? r3 - 1 : r3 + 2, a);
            dma_prep(n2.w, a, onx);
        }
        if (more && wave < 9) {   // younger than the row DMAs: waves 0..6 >= 9 accesses, wave 7 (one K / V tile to fetch) 7, wave 8 (one key tile) 5
            if (wave < 7) wait_vm<8>(); else if (wave == 7) wait_vm<6>(); else wait_vm<4>();
            if (wave < 8) rows_max();
        }
        __syncthreads();   // B2: the dS images of window wb and the parked rows of window wb + 1 are complete
        if (wave >= 8) {
            if (more) rows_convert(wb + 1, bi ^ 1);
        } else {
            // ---- query phase: dQ^T[16 channels mt][16 queries qt] = K^T dS^T over the 10 key pairs
            const int qt = wave & 3, mt = wave >> 2;
            const f16* D = DS + (qt >> 1) * ATT_BK_DSROWS * ATT_KS_STRIDE;
            const int la = mt ? ltr1 : ltr0, lb = (qt & 1) ? ltr1 : ltr0;
            f32x4 dq = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 10; ++p)
                dq = mfma16x16x32_f16(att_tr_frag_at(Ks + 1024 * p, la, p == 9), att_tr_frag_at(D + 1024 * p, lb), dq);
            const int q = 16 * qt + l15;
            if (q < CFFM_WA)
                *(f32x4*)(dqkv + ((long)b * G.RC + w * CFFM_WA + q) * 768 + h * CFFM_HD + 16 * mt + 4 * g) = dq * (scale * isc);
        }
        cur = n1;
        r3 = r3 == 2 ? 0 : r3 + 1;
    }
    // the group's bias gradient: one plain [304 keys][64 queries] tile per (group, head); k_sum_splits adds the groups
    const buf_t rs_dbp = buf_make(dbias_part + (((long)grp * CFFM_HEADS + h) * CFFM_NKEY_PAD) * CFFM_NQ_PAD,
                                  (uint32_t)(CFFM_NKEY_PAD * CFFM_NQ_PAD * 4));
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int qt = 0; qt < 4; qt += 2)
            if (u < ntile) {
                const uint32_t so = (uint32_t)(((16 * (t0 + u)) * CFFM_NQ_PAD + 16 * qt) * 4);
                buf_st16_pair(rs_dbp, dB[u][qt], dB[u][qt + 1], (uint32_t)((l15 * CFFM_NQ_PAD + 4 * g) * 4), so, so + 64);
            }
}

// dqkv[b][row][256..767] = sum over the (window, key slot) pairs that read `row` of the partial rows dkv_part[b*nW + window][slot]
// (512 halfs each: 8 heads x (K 32 | V 32), in units of part_scale[window][head]); inv_ptr [RC+1], inv_idx [nnz] = CSR inverse of key_src (per clip).
// One wave per token row, a lane owns 8 channels of one head; pooled rows also get their (unused) q third zeroed so the qkv
// weight/bias gradient GEMMs see zeros there.  grid (ceil(RC/4), B).
__global__ void __launch_bounds__(256) k_dkv_gather(Geo G, const int* __restrict__ inv_ptr, const int* __restrict__ inv_idx,
                                                     const float* __restrict__ dkv_part, float* __restrict__ dqkv) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (row >= G.RC) return;
    const int e0 = inv_ptr[row], e1 = inv_ptr[row + 1];
    const h16* part = (const h16*)dkv_part + (long)b * G.nW * CFFM_NKEY_PAD * 512 + 8 * lane;     // partial row: 8 heads x (K 32 | V 32)
    const float* scl = dkv_part + (long)G.B * G.nW * CFFM_NKEY_PAD * 256 + (long)b * G.nW * CFFM_HEADS + (lane >> 3);
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int eb = e0; eb < e1; eb += 64) {          // a row has at most 49 readers; the loop is for generality
        const int n = (e1 - eb < 64) ? e1 - eb : 64;
        const int mine = (lane < n) ? inv_idx[eb + lane] : 0;   // the whole reader list in one load
        int e = 0;
        for (; e + 3 < n; e += 4) {                 // 4 independent 16-B loads (+ their scales) in flight per lane
            const int i0 = __shfl(mine, e, 64), i1 = __shfl(mine, e + 1, 64), i2 = __shfl(mine, e + 2, 64), i3 = __shfl(mine, e + 3, 64);
            const h16x8 x0 = *(const h16x8*)(part + (long)i0 * 512), x1 = *(const h16x8*)(part + (long)i1 * 512);
            const h16x8 x2 = *(const h16x8*)(part + (long)i2 * 512), x3 = *(const h16x8*)(part + (long)i3 * 512);
            const float s0 = scl[(i0 / CFFM_NKEY_PAD) * CFFM_HEADS], s1 = scl[(i1 / CFFM_NKEY_PAD) * CFFM_HEADS];
            const float s2 = scl[(i2 / CFFM_NKEY_PAD) * CFFM_HEADS], s3 = scl[(i3 / CFFM_NKEY_PAD) * CFFM_HEADS];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a0[c] += ((float)x0[c] * s0 + (float)x1[c] * s1) + ((float)x2[c] * s2 + (float)x3[c] * s3);
                a1[c] += ((float)x0[4 + c] * s0 + (float)x1[4 + c] * s1) + ((float)x2[4 + c] * s2 + (float)x3[4 + c] * s3);
            }
        }
        for (; e < n; ++e) {
            const int i0 = __shfl(mine, e, 64);
            const h16x8 x0 = *(const h16x8*)(part + (long)i0 * 512);
            const float s0 = scl[(i0 / CFFM_NKEY_PAD) * CFFM_HEADS];
#pragma unroll
            for (int c = 0; c < 4; ++c) { a0[c] += (float)x0[c] * s0; a1[c] += (float)x0[4 + c] * s0; }
        }
    }
    float* drow = dqkv + ((long)b * G.RC + row) * 768;
    const int dcol = 256 + 256 * ((lane >> 2) & 1) + CFFM_HD * (lane >> 3) + 8 * (lane & 3);   // lane = (head, K / V, 8 channels)
    *(f32x4*)(drow + dcol) = a0;
    *(f32x4*)(drow + dcol + 4) = a1;
    if (row >= CFFM_WA * G.nW) *(f32x4*)(drow + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
}
