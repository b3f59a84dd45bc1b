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
// element offset of 16-byte chunk `chunk` (0..3) of row `row`.  A ds_read_b128 is served in four NON-contiguous groups of 16
// lanes ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS table), each mixing 8 rows of chunk g with 8 rows of chunk g+1, so
// padding the rows cannot make the MFMA fragment reads conflict-free (80-byte rows: 45 % of the LDS cycles were bank
// conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE); flipping chunk bit 1 on rows 8..15 of every 16 does.
#define ATT_ROW(row, chunk) ((row) * ATT_KS_STRIDE + 8 * ((chunk) ^ (2 * (((row) >> 3) & 1))))

// the 4 key-validity flags (0 / -inf) of this lane's keys 16t + 4g .. +3
#ifndef VFLAG_RD
#define VFLAG_RD 0
#endif
__device__ __forceinline__ f32x4 vflag4(const float* vflag, int o) {
#if VFLAG_RD == 1
    return (f32x4){vflag[o], vflag[o + 1], vflag[o + 2], vflag[o + 3]};
#elif VFLAG_RD == 2
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a = *(const f32x2*)(vflag + o), b = *(const f32x2*)(vflag + o + 2);
    return (f32x4){a[0], a[1], b[0], b[1]};
#else
    return *(const f32x4*)(vflag + o);
#endif
}
#define ATT_VROWS (CFFM_NKEY_PAD + 16)   // rows of an image that is read transposed 32 keys at a time: 16 zero rows past key 303
#define ATT_FWD_LDS (2 * CFFM_NKEY_PAD * ATT_KS_STRIDE * sizeof(f16) + CFFM_NKEY_PAD * 4)
#define ATT_FWP_LDS_ ((CFFM_NKEY_PAD + ATT_VROWS) * ATT_KS_STRIDE * sizeof(f16) + CFFM_NKEY_PAD * 4)
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

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }

// grid: B*nW*8 workgroups (head fastest), 256 threads = 4 waves x 16 queries.
// (k_cfm_attn_fwd is defined below, after the staging helpers it shares with the backward kernels)

// =====================================================================================================
// Backward: two kernels + a gather pass.
//  k_cfm_attn_bwd_q  ("query owners", S^T orientation as in the forward): grid (8 heads, NG groups), 256 threads = 4 waves x
//     16 queries; every workgroup walks the windows of its group so the position-bias gradient (19 tiles x 4 regs per lane)
//     stays in registers and is added to dbiasT once at the end; dQ is written directly (each query row has one owner).
//     Q / dO fragments come straight from global memory (16 / 32 B per lane); LDS holds K, V and K^T: 70 KB -> 2 workgroups
//     per CU overlap each other's gather latency.
//  k_cfm_attn_bwd_kv ("key owners", S orientation): one workgroup per (clip, window, head), 4 waves x 16-key tiles x all 64
//     queries -> dK^T, dV^T of the window's 289 key slots, written to per-window partial rows; 69 KB LDS -> 2 per CU.
//  k_dkv_gather sums, for every token row, the slots of all windows that read it (ring / pooled keys are shared by up to 49
//     windows) through a host-built inverse of the key table -- deterministic, no atomics on shared rows.
// dO is rescaled by a power of two (per wave / per window) so every f16 gradient operand sits near 1 (training-size gradients
// of 1e-6 would otherwise flush to zero in f16); results are scaled back in f32.
// =====================================================================================================
#ifndef BWK_OCC
#define BWK_OCC 3
#endif
#ifndef BWK_ABLATE
#define BWK_ABLATE 0   // profiling builds only: 1 staging only (no key-tile loop), 2 no partial-row stores
#endif
#ifndef BWQ_ABLATE
#define BWQ_ABLATE 0   // profiling builds only: 1 no dB flush, 2 no exp, 4 no dQ product, 8 no dP product
#endif
#ifdef BWQ_TIMING   // profiling builds only: shader-clock stamps of workgroup (0,0), wave 0, per window and phase
__device__ long long g_bwq_t[8 * 8];
#define BWQ_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && wb - wb0 < 8) g_bwq_t[(wb - wb0) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BWQ_STAMP(i)
#endif
#ifndef BWQ_OCC
#define BWQ_OCC 1  // workgroups per CU of the query-owner backward kernel: 1 = 512-register budget, no spills (measured faster than 2)
#endif
#define ATT_BWQ_LDS ((CFFM_NKEY_PAD + ATT_VROWS) * ATT_KS_STRIDE * sizeof(f16) + CFFM_NKEY_PAD * 4)
#define ATT_BWK_LDS ((2 * 64 * ATT_KS_STRIDE + 2 * CFFM_NKEY_PAD * ATT_KS_STRIDE) * sizeof(f16) + \
                     CFFM_NKEY_PAD * 4 + 64 * 4 + 64 * 4 + 16 * 4)

// A window's K / V rows on their way from the f16 q|k|v rows to LDS, held in registers so that the gather of window w+1
// can be in flight while window w is multiplied (kv_load: key-table entries, then the 16-byte row segments; kv_store:
// K, V rows and the key-validity flags, which come from the same entries).
template <int NTHREADS>
struct KvRegs {
    static constexpr int NW = NTHREADS / 64, NIT = (CFFM_NKEY_PAD / 2 + NW * 16 - 1) / (NW * 16);
    int src0[NIT], src1[NIT];
    f16x8 k0[NIT], k1[NIT], v0[NIT], v1[NIT];
};
// The gather is two dependent loads (key-table entry, then the row it names).  A wave issues in order, so fetching both in
// one go parks it for a full memory latency between them; the persistent kernels therefore fetch the TABLE entries two
// windows ahead (KvTab) and the rows one window ahead.
template <int NTHREADS>
struct KvTab { int s0[KvRegs<NTHREADS>::NIT], s1[KvRegs<NTHREADS>::NIT]; };
template <int NTHREADS>
__device__ __forceinline__ void kv_tab_load(KvTab<NTHREADS>& t, const int* __restrict__ ksrc, int tid) {
    constexpr int NW = KvRegs<NTHREADS>::NW, NIT = KvRegs<NTHREADS>::NIT;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int prk = (it * NW + wave) * 16 + (lane & 15);
        const bool ok = prk < CFFM_NKEY_PAD / 2;
        t.s0[it] = ok ? ksrc[2 * prk] : -1;
        t.s1[it] = ok ? ksrc[2 * prk + 1] : -1;
    }
}
// rows named by the table entries, through a buffer resource over the whole q|k|v array: one 32-bit offset per gather, a
// "no such key" entry (-1) becomes an out-of-range offset that reads as zeros -- no branches, no 64-bit address arithmetic.
// soff_k = byte offset of (clip b, row 0, this head's K slice): ((b*RC)*768 + 256 + h*32) * 2; V is 512 bytes further.
// one of the NIT row-pair batches (4 gathers): the persistent kernels spread the batches over their compute loop, so the
// texture path works through the gathers in the background instead of stalling the wave on a full request queue
template <int NTHREADS>
__device__ __forceinline__ void kv_rows_load_half(KvRegs<NTHREADS>& r, const KvTab<NTHREADS>& t, buf_t rs_qkv, uint32_t soff_k, int tid, int it, int odd) {
    const uint32_t c16 = (uint32_t)((tid & 63) >> 4) * 16;
    if (!odd) {
        r.src0[it] = t.s0[it];
        const uint32_t o0 = t.s0[it] >= 0 ? (uint32_t)t.s0[it] * 1536u + c16 : BUF_OOB;
        r.k0[it] = buf_ld_h8(rs_qkv, o0, soff_k);
        r.v0[it] = buf_ld_h8(rs_qkv, o0, soff_k + 512);
    } else {
        r.src1[it] = t.s1[it];
        const uint32_t o1 = t.s1[it] >= 0 ? (uint32_t)t.s1[it] * 1536u + c16 : BUF_OOB;
        r.k1[it] = buf_ld_h8(rs_qkv, o1, soff_k);
        r.v1[it] = buf_ld_h8(rs_qkv, o1, soff_k + 512);
    }
}
template <int NTHREADS>
__device__ __forceinline__ void kv_rows_load_it(KvRegs<NTHREADS>& r, const KvTab<NTHREADS>& t, buf_t rs_qkv, uint32_t soff_k, int tid, int it) {
    kv_rows_load_half<NTHREADS>(r, t, rs_qkv, soff_k, tid, it, 0);
    kv_rows_load_half<NTHREADS>(r, t, rs_qkv, soff_k, tid, it, 1);
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
// registers -> LDS: K and V rows (ATT_ROW layout) and the key-validity flags.  (No transposed image: the kernels read the
// transposed views they need with the LDS transpose read, att_tr_frag.)
template <int NTHREADS>
__device__ __forceinline__ void kv_store(const KvRegs<NTHREADS>& r, f16* Ks, f16* Vs, float* vflag, int tid) {
    constexpr int NW = KvRegs<NTHREADS>::NW, NIT = KvRegs<NTHREADS>::NIT;
    const int lane = tid & 63, wave = tid >> 6, c4 = lane >> 4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int prk = (it * NW + wave) * 16 + (lane & 15);
        if (prk < CFFM_NKEY_PAD / 2) {
            const int n0 = 2 * prk;
            if (c4 == 0) { vflag[n0] = r.src0[it] >= 0 ? 0.f : -INFINITY; vflag[n0 + 1] = r.src1[it] >= 0 ? 0.f : -INFINITY; }
            *(f16x8*)(Ks + ATT_ROW(n0, c4)) = r.k0[it];
            *(f16x8*)(Ks + ATT_ROW((n0 + 1), c4)) = r.k1[it];
            *(f16x8*)(Vs + ATT_ROW(n0, c4)) = r.v0[it];
            *(f16x8*)(Vs + ATT_ROW((n0 + 1), c4)) = r.v1[it];
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

// the query-owner lane's own operands of one window: Q fragment, dO / O (8 channels), LSE, destination pixel
struct QLaneRegs {
    f16x8 qfrag;
    f32x4 do0, do1, o0, o1;
    float lq;
};
__device__ __forceinline__ int qlane_dst(const Geo& G, const int* __restrict__ q_dst, int wb, int qcol) {
    return (qcol < CFFM_WA) ? q_dst[(wb % G.nW) * CFFM_WA + qcol] : -1;
}
struct QLaneSrc { buf_t qkv, ao, dao, lse; };
__device__ __forceinline__ void qlane_load(QLaneRegs& r, const Geo& G, const QLaneSrc& S, int dst, int wb, int h, int qcol, int g) {
    const int w = wb % G.nW, b = wb / G.nW;
    const bool own = qcol < CFFM_WA;
    // Q fragment: row (b, w*49 + qcol) of the q third; LSE of (wb, h, qcol); dO / O: pixel `dst` of clip b (-1: padding)
    r.qfrag = buf_ld_h8(S.qkv, own ? (uint32_t)(w * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                        (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
    r.lq = buf_ld4(S.lse, own ? 4u * qcol : BUF_OOB, (uint32_t)(((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD * 4));
    const uint32_t po = dst >= 0 ? (uint32_t)dst * (CFFM_C * 4u) + 32u * g : BUF_OOB;
    const uint32_t ps = (uint32_t)(((long)b * G.HW * CFFM_C + h * CFFM_HD) * 4);
    r.do0 = buf_ld16(S.dao, po, ps); r.do1 = buf_ld16(S.dao, po, ps + 16);
    r.o0 = buf_ld16(S.ao, po, ps); r.o1 = buf_ld16(S.ao, po, ps + 16);
}

__global__ void __launch_bounds__(256, BWQ_OCC) k_cfm_attn_bwd_q(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                            const int* __restrict__ q_dst, const float* __restrict__ bias,
                                                            const float* __restrict__ ao, const float* __restrict__ dao,
                                                            const float* __restrict__ lse_in, float* __restrict__ dqkv,
                                                            float* __restrict__ dbias_part, int per_group) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + ATT_VROWS * ATT_KS_STRIDE;   // K rows are also read transposed (dQ = dS K): 16 zero rows past key 303
    float* vflag = (float*)(Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE);

    const int h = blockIdx.x, grp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int qcol = 16 * wave + l15;
    const float scale = 0.17677669529663687f;
    const int wb0 = grp * per_group;
    const int wb1 = (wb0 + per_group < G.B * G.nW) ? wb0 + per_group : G.B * G.nW;
    const float* brow = bias + ((long)h * CFFM_NQ_PAD + qcol) * CFFM_NKEY_PAD + 4 * g;

    // The head's bias tiles (this lane's query, 4 keys per 16-key tile) do not depend on the window: loaded once, they stay
    // in registers next to their gradient for every window of the group (2 x 76 of the 512 registers one workgroup per CU has).
    f32x4 dB[19], bT[19];
#pragma unroll
    for (int t = 0; t < 19; ++t) { dB[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; bT[t] = ld4(brow + 16 * t); }

    // Software pipeline over the group's windows: while window wb is multiplied, the K/V gather and this lane's Q / dO / O
    // operands of window wb+1 are already in flight in registers (the 512-register budget of one workgroup per CU pays
    // for it), so the per-window gather latency is off the critical path.
    if (tid < 64) {
        f16x8 z8;
        for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
        *(f16x8*)(Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE + 8 * tid) = z8;
    }
    KvRegs<256> kv;
    KvTab<256> tabn;   // key-table entries / destination pixel of the window after the one whose rows are in flight
    QLaneRegs ql;
    int dstn = -1;
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    QLaneSrc qsrc;
    qsrc.qkv = rs_qkv;
    qsrc.ao = buf_make(ao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    qsrc.dao = buf_make(dao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    qsrc.lse = buf_make(lse_in, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * CFFM_NQ_PAD * 4));
    if (wb0 < wb1) {
        kv_load<256>(kv, rs_qkv, qkv_soff_k(G, wb0 / G.nW, h), key_src + (wb0 % G.nW) * CFFM_NKEY_PAD, tid);
        qlane_load(ql, G, qsrc, qlane_dst(G, q_dst, wb0, qcol), wb0, h, qcol, g);
    }
    if (wb0 + 1 < wb1) {
        kv_tab_load<256>(tabn, key_src + ((wb0 + 1) % G.nW) * CFFM_NKEY_PAD, tid);
        dstn = qlane_dst(G, q_dst, wb0 + 1, qcol);
    }
    for (int wb = wb0; wb < wb1; ++wb) {
        const int w = wb % G.nW, b = wb / G.nW;
        BWQ_STAMP(0);
        kv_store<256>(kv, Ks, Vs, vflag, tid);
        BWQ_STAMP(1);
        const f16x8 qfrag = ql.qfrag;
        const f32x4 do0 = ql.do0, do1 = ql.do1, o0 = ql.o0, o1 = ql.o1;
        const float lq = ql.lq;
        __syncthreads();
        BWQ_STAMP(2);
        // (the gathers of window wb+1 and the table entries of window wb+2 are issued inside the key loop below)
        const bool pre1 = wb + 1 < wb1, pre2 = wb + 2 < wb1;
        const uint32_t soff_n = qkv_soff_k(G, (wb + 1) / G.nW, h);

        float Dq = (do0[0] * o0[0] + do0[1] * o0[1]) + (do0[2] * o0[2] + do0[3] * o0[3]) + (do1[0] * o1[0] + do1[1] * o1[1]) +
                   (do1[2] * o1[2] + do1[3] * o1[3]);
        Dq += __shfl_xor(Dq, 16, 64);
        Dq += __shfl_xor(Dq, 32, 64);
        float am = 0.f;
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fmaxf(fabsf(do0[e]), fabsf(do1[e])));
        am = wave_max(am);
        int ex = 0;
        if (am > 0.f) frexpf(am, &ex);
        const float sc = (am > 0.f) ? ldexpf(1.f, 1 - ex) : 1.f, isc = 1.f / sc;   // max|dO * sc| in [1,2) for this wave
        Dq *= sc;
        f16x8 dofrag;
        for (int e = 0; e < 4; ++e) { dofrag[e] = (f16)(do0[e] * sc); dofrag[4 + e] = (f16)(do1[e] * sc); }

        f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        BWQ_STAMP(3);
#pragma unroll
        for (int kt = 0; kt < 10; ++kt) {
            f16x4 dsh[2];
            // one slice of the next window's loads per key-tile pair (4-7 loads at a time: all 25 at once cost ~2.3 k cycles of request-queue
            // stall per window, 2 at a time measured slower again): rows (their table entries arrived during the previous window), this lane's
            // Q / dO / O / LSE, then the table entries of the window after
            if (kt < 3 && pre1) kv_rows_load_it<256>(kv, tabn, rs_qkv, soff_n, tid, kt);
            if (kt == 3 && pre1) qlane_load(ql, G, qsrc, dstn, wb + 1, h, qcol, g);
            if (kt == 4 && pre2) {
                kv_tab_load<256>(tabn, key_src + ((wb + 2) % G.nW) * CFFM_NKEY_PAD, tid);
                dstn = qlane_dst(G, q_dst, wb + 2, qcol);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kt + u;
                if (t < 19) {
                    const f16x8 kf = *(const f16x8*)(Ks + ATT_ROW((16 * t + l15), g));
                    const f16x8 vf = *(const f16x8*)(Vs + ATT_ROW((16 * t + l15), g));
                    f32x4 sv = mfma16x16x32_f16(kf, qfrag, bT[t < 19 ? t : 0] + vflag4(vflag, 16 * t + 4 * g));
                    const f32x4 dp = (BWQ_ABLATE & 8) ? (f32x4){(float)vf[0], (float)vf[1], (float)vf[2], (float)dofrag[0]} : mfma16x16x32_f16(vf, dofrag, (f32x4){0.f, 0.f, 0.f, 0.f});
                    f32x4 ds;
                    for (int r = 0; r < 4; ++r) ds[r] = ((BWQ_ABLATE & 2) ? (sv[r] - lq) : fast_exp(sv[r] - lq)) * (dp[r] - Dq);
                    dB[t < 19 ? t : 0] += ds * isc;
                    dsh[u] = to_f16x4(ds);
                } else {
                    dsh[u] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                }
            }
            const f16x8 dsf = cat_f16x4(dsh[0], dsh[1]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f16x8 ka = att_tr_frag(Ks, 32 * kt, 16 * mt, lane);
                if (!(BWQ_ABLATE & 4)) dq[mt] = mfma16x16x32_f16(ka, dsf, dq[mt]);
                else dq[mt] += (f32x4){(float)dsf[0], (float)dsf[1], (float)dsf[2], (float)dsf[3]};
            }
        }
        BWQ_STAMP(4);
        if (qcol < CFFM_WA) {
            float* drow = dqkv + ((long)b * G.RC + w * CFFM_WA + qcol) * 768 + h * CFFM_HD + 4 * g;
            *(f32x4*)(drow) = dq[0] * (scale * isc);      // d(raw q): the stored q carries the 32^-0.5 factor
            *(f32x4*)(drow + 16) = dq[1] * (scale * isc);
        }
        BWQ_STAMP(5);
        __syncthreads();  // LDS is restaged for the next window
        BWQ_STAMP(6);
    }
    // the group's bias gradient: one plain [304 keys][64 queries] tile per (group, head); k_sum_splits adds the groups
    // (32 contended atomicAdds per element cost a quarter of this kernel; rows of padded queries / keys are exact zeros)
    if (!(BWQ_ABLATE & 1)) {
        float* dst = dbias_part + (((long)grp * CFFM_HEADS + h) * CFFM_NKEY_PAD) * CFFM_NQ_PAD + qcol;
#pragma unroll
        for (int t = 0; t < 19; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(16 * t + 4 * g + r) * CFFM_NQ_PAD] = dB[t][r];
    }
}

// grid: B*nW*8 workgroups (head fastest), 256 threads
#ifndef FWD_OCC
#define FWD_OCC 4   // workgroups per CU: 39.1 KB of LDS and <= 128 VGPRs each
#endif
__global__ void __launch_bounds__(256, FWD_OCC) k_cfm_attn_fwd(Geo G, const h16* __restrict__ qkv,
                                                       const int* __restrict__ key_src, const int* __restrict__ q_dst,
                                                       const float* __restrict__ bias, float* __restrict__ ao,
                                                       float* __restrict__ lse_out) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
    float* vflag = (float*)(Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE);

    const int h = blockIdx.x & 7, wb = blockIdx.x >> 3, w = wb % G.nW, b = wb / G.nW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qcol = 16 * wave + (lane & 15), g = lane >> 4, l15 = lane & 15;

    // ---- stage ----
    // Global loads go out in three batches, each complete before anything waits on it: the key-table entries of this
    // thread's 3 row pairs (+ the destination pixel the epilogue needs), then the 12 gathered 16-byte K/V segments and
    // this lane's Q fragment (straight into the MFMA operand: Q never touches LDS), then the wave's 19 bias tiles.
    // A (token, head) slice is 64 B of f16 = four 16-byte chunks.  K and V are both kept as ROWS: the PV step reads V
    // through the LDS transpose read (lds_tr4), so no transposed image is written.
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const int qdst = (qcol < CFFM_WA) ? q_dst[w * CFFM_WA + qcol] : -1;
    KvRegs<256> kv;
    kv_load<256>(kv, rs_qkv, qkv_soff_k(G, b, h), key_src + w * CFFM_NKEY_PAD, tid);
    const f16x8 qfrag = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)(w * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                                  (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
    kv_store<256>(kv, Ks, Vs, vflag, tid);
    // the wave's 19 bias tiles (L2-resident table) fly across the barrier and land in the MFMA C operands
    const float* brow = bias + ((long)h * CFFM_NQ_PAD + qcol) * CFFM_NKEY_PAD + 4 * g;
    f32x4 s[19];
#pragma unroll
    for (int t = 0; t < 19; ++t) s[t] = ld4(brow + 16 * t);
    __syncthreads();

    // ---- S^T = K Q^T (+bias, +mask), softmax over the 289 keys of each query column -----------------
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 19; ++t) {
        const f16x8 kf = *(const f16x8*)(Ks + ATT_ROW((16 * t + l15), g));
        const f32x4 acc = mfma16x16x32_f16(kf, qfrag, s[t] + vflag4(vflag, 16 * t + 4 * g));   // C-in = bias + mask
        s[t] = acc;
        m = fmaxf(m, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 19; ++t) {
        f32x4 p;
        p[0] = fast_exp(s[t][0] - m); p[1] = fast_exp(s[t][1] - m); p[2] = fast_exp(s[t][2] - m); p[3] = fast_exp(s[t][3] - m);
        s[t] = p;
        l += (p[0] + p[1]) + (p[2] + p[3]);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);

    // ---- O^T = V^T P^T : A = V^T[d][key slots] read transposed out of the V rows, B = P^T from registers ----------
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
        const f16x4 lo = to_f16x4(s[2 * kt]);
        const f16x4 hi = (2 * kt + 1 < 19) ? to_f16x4(s[(2 * kt + 1 < 19) ? 2 * kt + 1 : 0]) : (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        const f16x8 pf = cat_f16x4(lo, hi);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            o[mt] = mfma16x16x32_f16(att_tr_frag<CFFM_NKEY_PAD>(Vs, 32 * kt, 16 * mt, lane), pf, o[mt]);
    }

    // ---- epilogue: normalise, un-window, drop padded pixels (cffm_transformer.py:812-821) ------------
    if (g == 0) lse_out[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + qcol] = (qcol < CFFM_WA) ? m + logf(l) : 0.f;
    if (qdst >= 0) {
        const float inv = 1.f / l;
        float* orow = ao + ((long)b * G.HW + qdst) * CFFM_C + h * CFFM_HD + 4 * g;
        *(f32x4*)(orow) = o[0] * inv;
        *(f32x4*)(orow + 16) = o[1] * inv;
    }
}

// ---- forward, split-key form ------------------------------------------------------------------------------------------
// The same mathematics with the 304 key slots taken in two halves (160 + 144) through ONE half-size LDS buffer and a
// running (online) softmax: 21 KB of LDS and ~1/3 fewer registers than k_cfm_attn_fwd (which keeps all 19 S^T tiles of a
// query column in registers for a two-pass softmax), so FWS_OCC = 6 workgroups fit on a CU and the whole grid of a
// 2-clip batch (1296 workgroups) is resident at once instead of running as 1024 + 272.  The second half's gathers are in
// flight while the first half is multiplied.  CFFM_ATTN_FWD=split selects it.
// MEASURED SLOWER (kept selectable, parity-tested): 34.3 us at FWS_OCC 6 (80 VGPRs, 120 B of scratch), 28.3 us at 5 (96 VGPRs,
// 56 B of scratch) against 22.6 us for k_cfm_attn_fwd (event intervals, B = 2): the barrier / store / barrier between the
// halves serialises every workgroup once more, the running softmax adds 20 cross-lane maxima, and the compiler still needs
// ~105 registers for the software-pipelined loads -- more than the resident-grid effect (the 272-workgroup tail) gives back.
#ifndef FWS_OCC
#define FWS_OCC 6
#endif
#define FWS_ROWS 160                   // rows of the LDS buffer: keys 0..159, then keys 160..303 (144 rows)
#define ATT_FWS_LDS (2 * FWS_ROWS * ATT_KS_STRIDE * sizeof(f16) + FWS_ROWS * 4)
struct FwsRegs { int src[3]; f16x8 k[3], v[3]; };
// this thread's 16-byte chunk (tid & 3) of rows (tid >> 2) + 64 it of the half starting at key `base` (nrows rows)
__device__ __forceinline__ void fws_tab(FwsRegs& r, const int* __restrict__ ksrc, int base, int nrows, int tid) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = (tid >> 2) + 64 * it;
        r.src[it] = row < nrows ? ksrc[base + row] : -1;
    }
}
__device__ __forceinline__ void fws_rows(FwsRegs& r, buf_t rs_qkv, uint32_t soff_k, int tid) {
    const uint32_t c16 = (uint32_t)(tid & 3) * 16;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const uint32_t o = r.src[it] >= 0 ? (uint32_t)r.src[it] * 1536u + c16 : BUF_OOB;
        r.k[it] = buf_ld_h8(rs_qkv, o, soff_k);
        r.v[it] = buf_ld_h8(rs_qkv, o, soff_k + 512);
    }
}
__device__ __forceinline__ void fws_store(const FwsRegs& r, f16* Ks, f16* Vs, float* vflag, int nrows, int tid) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = (tid >> 2) + 64 * it, c = tid & 3;
        if (row < nrows) {
            if (c == 0) vflag[row] = r.src[it] >= 0 ? 0.f : -INFINITY;
            *(f16x8*)(Ks + ATT_ROW(row, c)) = r.k[it];
            *(f16x8*)(Vs + ATT_ROW(row, c)) = r.v[it];
        }
    }
}
// one half: NT 16-key tiles, taken in pairs (one PV k-step of 32 keys); bias tiles from `brow` (this lane's query row, first
// key of the half), running maximum m, running sum l (this lane's share), output accumulators o
template <int NT>
__device__ __forceinline__ void fws_half(const f16* Ks, const f16* Vs, const float* vflag, const float* __restrict__ brow, f16x8 qfrag,
                                         int lane, float& m, float& l, f32x4 (&o)[2]) {
    const int g = lane >> 4, l15 = lane & 15;
    f32x4 b0 = ld4(brow), b1 = NT > 1 ? ld4(brow + 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
        const bool two = 2 * p + 1 < NT;
        f32x4 n0 = b0, n1 = b1;
        if (2 * p + 2 < NT) n0 = ld4(brow + 16 * (2 * p + 2));          // the next pair's bias tiles are in flight meanwhile
        if (2 * p + 3 < NT) n1 = ld4(brow + 16 * (2 * p + 3));
        const f16x8 kf0 = *(const f16x8*)(Ks + ATT_ROW((32 * p + l15), g));
        f32x4 s0 = mfma16x16x32_f16(kf0, qfrag, b0 + vflag4(vflag, 32 * p + 4 * g));   // C-in = bias + mask
        f32x4 s1 = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (two) {
            const f16x8 kf1 = *(const f16x8*)(Ks + ATT_ROW((32 * p + 16 + l15), g));
            s1 = mfma16x16x32_f16(kf1, qfrag, b1 + vflag4(vflag, 32 * p + 16 + 4 * g));
        }
        float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float mu = mn == -INFINITY ? 0.f : mn;                   // (every key so far masked: exponentials of -inf are 0)
        const float alpha = fast_exp(m - mu);
        f32x4 p0, p1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p0[j] = fast_exp(s0[j] - mu); p1[j] = fast_exp(s1[j] - mu); }
        l = l * alpha + ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
        m = mn;
        const f16x8 pf = cat_f16x4(to_f16x4(p0), to_f16x4(p1));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            o[mt] = mfma16x16x32_f16(att_tr_frag<FWS_ROWS>(Vs, 32 * p, 16 * mt, lane), pf, o[mt] * alpha);
        b0 = n0; b1 = n1;
    }
}
__global__ void __launch_bounds__(256, FWS_OCC) k_cfm_attn_fwd_s(Geo G, const h16* __restrict__ qkv,
                                                           const int* __restrict__ key_src, const int* __restrict__ q_dst,
                                                           const float* __restrict__ bias, float* __restrict__ ao,
                                                           float* __restrict__ lse_out) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + FWS_ROWS * ATT_KS_STRIDE;
    float* vflag = (float*)(Vs + FWS_ROWS * ATT_KS_STRIDE);
    const int h = blockIdx.x & 7, wb = blockIdx.x >> 3, w = wb % G.nW, b = wb / G.nW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qcol = 16 * wave + (lane & 15), g = lane >> 4;
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const uint32_t soff_k = qkv_soff_k(G, b, h);
    const int* ksrc = key_src + w * CFFM_NKEY_PAD;
    const int qdst = (qcol < CFFM_WA) ? q_dst[w * CFFM_WA + qcol] : -1;
    FwsRegs ra, rb;
    fws_tab(ra, ksrc, 0, FWS_ROWS, tid);
    fws_tab(rb, ksrc, FWS_ROWS, CFFM_NKEY_PAD - FWS_ROWS, tid);
    fws_rows(ra, rs_qkv, soff_k, tid);
    const f16x8 qfrag = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)(w * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                                  (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
    fws_store(ra, Ks, Vs, vflag, FWS_ROWS, tid);
    fws_rows(rb, rs_qkv, soff_k, tid);                     // the second half's rows fly while the first half is multiplied
    __syncthreads();
    const float* brow = bias + ((long)h * CFFM_NQ_PAD + qcol) * CFFM_NKEY_PAD + 4 * g;
    float m = -INFINITY, l = 0.f;
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    fws_half<FWS_ROWS / 16>(Ks, Vs, vflag, brow, qfrag, lane, m, l, o);
    __syncthreads();
    fws_store(rb, Ks, Vs, vflag, CFFM_NKEY_PAD - FWS_ROWS, tid);
    __syncthreads();
    fws_half<(CFFM_NKEY_PAD - FWS_ROWS) / 16>(Ks, Vs, vflag, brow + FWS_ROWS, qfrag, lane, m, l, o);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (g == 0) lse_out[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + qcol] = (qcol < CFFM_WA) ? m + logf(l) : 0.f;
    if (qdst >= 0) {
        const float inv = 1.f / l;
        float* orow = ao + ((long)b * G.HW + qdst) * CFFM_C + h * CFFM_HD + 4 * g;
        *(f32x4*)(orow) = o[0] * inv;
        *(f32x4*)(orow + 16) = o[1] * inv;
    }
}

// ---- forward, persistent form ------------------------------------------------------------------------------------------
// grid (8 heads, NG window groups), FWP_OCC workgroups per CU.  The same mathematics as k_cfm_attn_fwd; what changes is where
// the latencies go: the head's 19 bias tiles are loaded ONCE per workgroup and stay in registers (the one-shot kernel pulls
// 78 KB of bias per (window, head) through the CU's 64 B/clk texture path -- more than the K/V gathers), the key-table
// entries arrive two windows ahead and the K/V rows one window ahead (KvTab / KvRegs), Q comes straight from global into
// the MFMA fragment, and with two workgroups per CU one stages (LDS writes, gather issue) while the other multiplies.
#ifndef FWP_OCC
#define FWP_OCC 2
#endif
#define ATT_FWP_LDS ATT_FWP_LDS_
__global__ void __launch_bounds__(256, FWP_OCC) k_cfm_attn_fwd_p(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                                const int* __restrict__ q_dst, const float* __restrict__ bias,
                                                                float* __restrict__ ao, float* __restrict__ lse_out, int per_group) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
    float* vflag = (float*)(Vs + ATT_VROWS * ATT_KS_STRIDE);
    const int h = blockIdx.x, grp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int qcol = 16 * wave + l15;
    const int wb0 = grp * per_group;
    const int wb1 = (wb0 + per_group < G.B * G.nW) ? wb0 + per_group : G.B * G.nW;
    const float* brow = bias + ((long)h * CFFM_NQ_PAD + qcol) * CFFM_NKEY_PAD + 4 * g;
    f32x4 bT[19];
#pragma unroll
    for (int t = 0; t < 19; ++t) bT[t] = ld4(brow + 16 * t);
    if (tid < 64) {
        f16x8 z8;
        for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
        *(f16x8*)(Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE + 8 * tid) = z8;
    }
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    KvRegs<256> kv;
    KvTab<256> tabn;
    f16x8 qn;          // Q fragment of the window whose rows are in flight
    int dstc = -1, dstn = -1;
    if (wb0 < wb1) {
        kv_load<256>(kv, rs_qkv, qkv_soff_k(G, wb0 / G.nW, h), key_src + (wb0 % G.nW) * CFFM_NKEY_PAD, tid);
        qn = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)((wb0 % G.nW) * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                       (uint32_t)(((long)(wb0 / G.nW) * G.RC * 768 + h * CFFM_HD) * 2));
        dstc = qlane_dst(G, q_dst, wb0, qcol);
    }
    if (wb0 + 1 < wb1) {
        kv_tab_load<256>(tabn, key_src + ((wb0 + 1) % G.nW) * CFFM_NKEY_PAD, tid);
        dstn = qlane_dst(G, q_dst, wb0 + 1, qcol);
    }
    for (int wb = wb0; wb < wb1; ++wb) {
        const int b = wb / G.nW;
        kv_store<256>(kv, Ks, Vs, vflag, tid);
        const f16x8 qfrag = qn;
        const int dst = dstc;
        __syncthreads();
        if (wb + 1 < wb1) {
            const int wn = (wb + 1) % G.nW, bn = (wb + 1) / G.nW;
            kv_rows_load<256>(kv, tabn, rs_qkv, qkv_soff_k(G, bn, h), tid);
            qn = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)(wn * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                           (uint32_t)(((long)bn * G.RC * 768 + h * CFFM_HD) * 2));
            dstc = dstn;
        }
        if (wb + 2 < wb1) {
            kv_tab_load<256>(tabn, key_src + ((wb + 2) % G.nW) * CFFM_NKEY_PAD, tid);
            dstn = qlane_dst(G, q_dst, wb + 2, qcol);
        }
        // S^T = K Q^T (+bias, +mask), softmax over the 289 keys of each query column
        f32x4 s[19];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            const f16x8 kf = *(const f16x8*)(Ks + ATT_ROW((16 * t + l15), g));
            s[t] = mfma16x16x32_f16(kf, qfrag, bT[t] + vflag4(vflag, 16 * t + 4 * g));
            m = fmaxf(m, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            f32x4 pv;
            pv[0] = fast_exp(s[t][0] - m); pv[1] = fast_exp(s[t][1] - m); pv[2] = fast_exp(s[t][2] - m); pv[3] = fast_exp(s[t][3] - m);
            s[t] = pv;
            l += (pv[0] + pv[1]) + (pv[2] + pv[3]);
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        // O^T = V^T P^T : A = V^T[d][key-slots] from LDS, B = P^T from registers
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < 10; ++kt) {
            const f16x4 lo = to_f16x4(s[2 * kt]);
            const f16x4 hi = (2 * kt + 1 < 19) ? to_f16x4(s[(2 * kt + 1 < 19) ? 2 * kt + 1 : 0]) : (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            const f16x8 pf = cat_f16x4(lo, hi);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) o[mt] = mfma16x16x32_f16(att_tr_frag(Vs, 32 * kt, 16 * mt, lane), pf, o[mt]);
        }
        // epilogue: normalise, un-window, drop padded pixels (cffm_transformer.py:812-821)
        if (g == 0) lse_out[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + qcol] = (qcol < CFFM_WA) ? m + logf(l) : 0.f;
        if (dst >= 0) {
            const float inv = 1.f / l;
            float* orow = ao + ((long)b * G.HW + dst) * CFFM_C + h * CFFM_HD + 4 * g;
            *(f32x4*)(orow) = o[0] * inv;
            *(f32x4*)(orow + 16) = o[1] * inv;
        }
        __syncthreads();   // LDS is restaged for the next window
    }
}

__global__ void __launch_bounds__(256, BWK_OCC) k_cfm_attn_bwd_kv(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                             const int* __restrict__ q_dst, const float* __restrict__ biasT,
                                                             const float* __restrict__ ao, const float* __restrict__ dao,
                                                             const float* __restrict__ lse_in, float* __restrict__ dkv_part) {
    CFFM_DYN_SMEM(smem);
    f16* Qs = (f16*)smem;
    f16* dOs = Qs + 64 * ATT_KS_STRIDE;
    f16* Ks = dOs + 64 * ATT_KS_STRIDE;
    f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
    float* vflag = (float*)(Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE);
    float* slse = vflag + CFFM_NKEY_PAD;
    float* sD = slse + 64;
    float* smax = sD + 64;

    const int h = blockIdx.x & 7, wb = blockIdx.x >> 3, w = wb % G.nW, b = wb / G.nW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int* ksrc = key_src + w * CFFM_NKEY_PAD;
    const h16* base = qkv + (long)b * G.RC * 768 + h * CFFM_HD;

    // Q: threads 0..127 own (query pair, 16-byte chunk); dO / O: every thread owns (query pair, 4-channel chunk)
    const int pr = tid >> 3, c = tid & 7, i0 = 2 * pr, i1 = i0 + 1;
    const int qp = tid >> 2, qc = tid & 3;
    const int t0 = (i0 < CFFM_WA) ? q_dst[w * CFFM_WA + i0] : -1, t1 = (i1 < CFFM_WA) ? q_dst[w * CFFM_WA + i1] : -1;
    f16x8 z8;
    for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
    f16x8 q0 = z8, q1 = z8;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 r0 = z4, r1 = z4, o0 = z4, o1 = z4;
    if (tid < 128) {
        if (2 * qp < CFFM_WA) q0 = ld_h8(base + (long)(w * CFFM_WA + 2 * qp) * 768 + 8 * qc);
        if (2 * qp + 1 < CFFM_WA) q1 = ld_h8(base + (long)(w * CFFM_WA + 2 * qp + 1) * 768 + 8 * qc);
    }
    if (t0 >= 0) { const long off = ((long)b * G.HW + t0) * CFFM_C + h * CFFM_HD + 4 * c; r0 = ld4(dao + off); o0 = ld4(ao + off); }
    if (t1 >= 0) { const long off = ((long)b * G.HW + t1) * CFFM_C + h * CFFM_HD + 4 * c; r1 = ld4(dao + off); o1 = ld4(ao + off); }
    if (tid < 64) slse[tid] = lse_in[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + tid];
    stage_kv<256>(qkv_rsrc(G, qkv), qkv_soff_k(G, b, h), ksrc, Ks, Vs, vflag, tid);
    if (tid < 128) {   // Q rows and the transposed Q image
        *(f16x8*)(Qs + ATT_ROW((2 * qp), qc)) = q0;
        *(f16x8*)(Qs + ATT_ROW((2 * qp + 1), qc)) = q1;
    }
    // D = rowsum(dO * O) (the 8 chunks of a query sit in 8 adjacent lanes) and the window's |dO| maximum
    float d0 = r0[0] * o0[0] + r0[1] * o0[1] + r0[2] * o0[2] + r0[3] * o0[3];
    float d1 = r1[0] * o1[0] + r1[1] * o1[1] + r1[2] * o1[2] + r1[3] * o1[3];
    float amax = 0.f;
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(r0[e]), fabsf(r1[e])));
    d0 += __shfl_xor(d0, 1, 64); d0 += __shfl_xor(d0, 2, 64); d0 += __shfl_xor(d0, 4, 64);
    d1 += __shfl_xor(d1, 1, 64); d1 += __shfl_xor(d1, 2, 64); d1 += __shfl_xor(d1, 4, 64);
    amax = wave_max(amax);
    if (lane == 0) smax[wave] = amax;
    __syncthreads();
    const float am = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    int ex = 0;
    if (am > 0.f) frexpf(am, &ex);
    const float sc = (am > 0.f) ? ldexpf(1.f, 1 - ex) : 1.f, isc = 1.f / sc;   // max|dO * sc| in [1,2)
    r0 *= sc; r1 *= sc;
    if (c == 0) { sD[i0] = d0 * sc; sD[i1] = d1 * sc; }
    *(f16x4*)(dOs + ATT_ROW(i0, c >> 1) + 4 * (c & 1)) = to_f16x4(r0);
    *(f16x4*)(dOs + ATT_ROW(i1, c >> 1) + 4 * (c & 1)) = to_f16x4(r1);
    __syncthreads();

    f32x4 bt[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) bt[mt] = ld4(biasT + ((long)h * CFFM_NKEY_PAD + 16 * wave + l15) * CFFM_NQ_PAD + 4 * g + 16 * mt);
    for (int t = wave; t < ((BWK_ABLATE & 1) ? 0 : 19); t += 4) {
        const int key = 16 * t + l15;
        f32x4 bn[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) bn[mt] = bt[mt];
        if (t + 4 < 19) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) bn[mt] = ld4(biasT + ((long)h * CFFM_NKEY_PAD + key + 64) * CFFM_NQ_PAD + 4 * g + 16 * mt);
        }
        const f16x8 kfrag = *(const f16x8*)(Ks + ATT_ROW(key, g));
        const f16x8 vfrag = *(const f16x8*)(Vs + ATT_ROW(key, g));
        const float vf = vflag[key];
        f16x4 ph[4], dsh[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f16x8 qa = *(const f16x8*)(Qs + ATT_ROW((16 * mt + l15), g));
            const f16x8 da = *(const f16x8*)(dOs + ATT_ROW((16 * mt + l15), g));
            const f32x4 sv = mfma16x16x32_f16(qa, kfrag, bt[mt] + vf);
            const f32x4 dp = mfma16x16x32_f16(da, vfrag, (f32x4){0.f, 0.f, 0.f, 0.f});
            const f32x4 lq = *(const f32x4*)(slse + 16 * mt + 4 * g), Dq = *(const f32x4*)(sD + 16 * mt + 4 * g);
            f32x4 p, ds;
            for (int r = 0; r < 4; ++r) { p[r] = fast_exp(sv[r] - lq[r]); ds[r] = p[r] * (dp[r] - Dq[r]); }
            ph[mt] = to_f16x4(p);
            dsh[mt] = to_f16x4(ds);
        }
        f32x4 dv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        f32x4 dk[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 pf = cat_f16x4(ph[2 * ks], ph[2 * ks + 1]);
            const f16x8 sf = cat_f16x4(dsh[2 * ks], dsh[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {   // dO^T / Q^T operands: transposed reads of the dO / Q rows
                dv[dt] = mfma16x16x32_f16(att_tr_frag(dOs, 32 * ks, 16 * dt, lane), pf, dv[dt]);
                dk[dt] = mfma16x16x32_f16(att_tr_frag(Qs, 32 * ks, 16 * dt, lane), sf, dk[dt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) bt[mt] = bn[mt];
        // dK^T/dV^T tiles sit as [d = 16dt+4g+r][key = l15]: every lane owns 16 contiguous bytes of a key row of this
        // window's slot in the partial buffer
        if (vf == 0.f && (!(BWK_ABLATE & 2) || isc == 12345.f)) {   // valid key (flag 0, -inf otherwise)
            float* prow = dkv_part + ((long)wb * CFFM_NKEY_PAD + key) * 512 + h * CFFM_HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                *(f32x4*)(prow + 16 * dt) = dk[dt] * isc;
                *(f32x4*)(prow + 256 + 16 * dt) = dv[dt] * isc;
            }
        }
    }
}

// dqkv[b][row][256..767] = sum over the (window, key slot) pairs that read `row` of dkv_part[b*nW + window][slot][512];
// inv_ptr [RC+1], inv_idx [nnz] = CSR inverse of key_src (per clip).  One wave per token row; pooled rows also get
// their (unused) q third zeroed so the qkv weight/bias gradient GEMMs see zeros there.  grid (ceil(RC/4), B).
__global__ void __launch_bounds__(256) k_dkv_gather(Geo G, const int* __restrict__ inv_ptr, const int* __restrict__ inv_idx,
                                                     const float* __restrict__ dkv_part, float* __restrict__ dqkv) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (row >= G.RC) return;
    const int e0 = inv_ptr[row], e1 = inv_ptr[row + 1];
    const float* part = dkv_part + (long)b * G.nW * CFFM_NKEY_PAD * 512 + 8 * lane;
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int eb = e0; eb < e1; eb += 64) {          // a row has at most 49 readers; the loop is for generality
        const int n = (e1 - eb < 64) ? e1 - eb : 64;
        const int mine = (lane < n) ? inv_idx[eb + lane] : 0;   // the whole reader list in one load
        int e = 0;
        for (; e + 3 < n; e += 4) {                 // 8 independent 16-B loads in flight per lane
            const float* p0 = part + (long)__shfl(mine, e, 64) * 512;
            const float* p1 = part + (long)__shfl(mine, e + 1, 64) * 512;
            const float* p2 = part + (long)__shfl(mine, e + 2, 64) * 512;
            const float* p3 = part + (long)__shfl(mine, e + 3, 64) * 512;
            const f32x4 x0 = ld4(p0), y0 = ld4(p0 + 4), x1 = ld4(p1), y1 = ld4(p1 + 4);
            const f32x4 x2 = ld4(p2), y2 = ld4(p2 + 4), x3 = ld4(p3), y3 = ld4(p3 + 4);
            a0 += (x0 + x1) + (x2 + x3);
            a1 += (y0 + y1) + (y2 + y3);
        }
        for (; e < n; ++e) {
            const float* p = part + (long)__shfl(mine, e, 64) * 512;
            a0 += ld4(p);
            a1 += ld4(p + 4);
        }
    }
    float* drow = dqkv + ((long)b * G.RC + row) * 768;
    *(f32x4*)(drow + 256 + 8 * lane) = a0;
    *(f32x4*)(drow + 256 + 8 * lane + 4) = a1;
    if (row >= CFFM_WA * G.nW) *(f32x4*)(drow + 4 * lane) = (f32x4){0.f, 0.f, 0.f, 0.f};
}
