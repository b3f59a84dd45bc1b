// cfm_attn_bwd_kernels.h -- backward of the Cross-frame Feature Mining attention (round 4 decomposition).
//
// Reference semantics: autograd of WindowAttention3d3.forward, cffm_transformer.py:364-606: dq of the 49
// window queries, dk / dv of every token and pooled cell summed over all the (window, slot) pairs that read it (own window,
// ring readers incl. the 12 duplicated positions and the cyclic wrap :389-418, up to kk^2 windows per pooled cell :426-518),
// and the gradients of the six position-bias tables (:536-587).
//
// Rounds 2-3 did this with one query-owner kernel that also produced dK / dV: per window it exchanged P and dS through LDS
// (10 barriers), wrote 304 partial key rows per (window, head) (50 MB per launch at B = 2) and a gather kernel summed them per
// token row.  Here nothing is exchanged and no partial row exists.  The work is split the way FlashAttention-2 splits it, by
// who OWNS the output -- S and dP are recomputed by both roles (MFMA time is not what bounds these kernels):
//   * query-owner role (attn_bwd_q_role): a workgroup walks windows of one head; S^T = K Q^T + bias, dP^T = V dO^T, P from
//     the saved LSE, dS = P (dP - D); dQ^T += K^T dS^T out of the C registers; the head's bias gradient stays in registers
//     over all its windows.  No barrier inside a window.
//   * key-owner role (attn_bwd_k_role): a workgroup owns up to 64 key rows of one (clip, head) -- a window's 49 tokens plus
//     its 13 pooled cells of frames t-6 / t-3, or 16 cells of the pooled-target / frame t-9 grids -- and its four waves walk the
//     windows that read them (host-built pass lists, geometry.ko_tables).  In the S = Q K^T orientation the C registers hold
//     4 consecutive queries of one key per lane, which is exactly the B operand of the contractions over queries:
//     dV^T += dO^T P, dK^T += Q^T dS with the A operands read transposed out of the reader's Q / dO rows (att_tr_frag).
//     The accumulators of all 64 keys live in registers (64 VGPRs) for the whole unit; the four waves' sums are added in a fixed
//     order through LDS and every dk / dv row is written once, in place, in fp32.  Deterministic, no atomics.
//   * k_attn_bwd_prep: dO as f16 rows per (window, head), rescaled by a power of two per (window, head) so that training-size
//     gradients survive f16, D = rowsum(dO * O) in the same units, and the scale.  The key-owner role brings the windows of a
//     (clip, head) to one common scale while it stages them (exact: powers of two).
// dS feeds its MFMAs as an f16 hi + lo pair (2 MFMAs instead of 1): with single-f16 dS the worst parameter gradient sat at
// 1.15e-3 of the reference, above the 1e-3 contract.
#pragma once
#include <type_traits>
#include "cfm_attn_kernels.h"

#ifndef BWD_DS_LO
#define BWD_DS_LO 0    // 1: dS feeds dK^T += Q^T dS as an f16 hi + lo pair (stage test: dk 3.4e-4 -> 1.9e-4 of its maximum; 10 more VALU instructions per tile)
#endif
#define KO_TILES_MAX 4
#ifndef BWD_Q_ABLATE
#define BWD_Q_ABLATE 0   // profiling builds only (query-owner kernel): 1 no K / V gather, 2 no multiplication, 4 no barrier wait for the gather
#endif
#ifndef BWD_K_ABLATE
#define BWD_K_ABLATE 0   // profiling builds only (key-owner kernel): 1 no multiplication, 2 no reader staging, 4 no reduction / stores
#endif
// 4 stored halfs (8 bytes, as loaded) -> floats
__device__ __forceinline__ f32x4 h4_to_f32x4(f32x2 raw) {
    h16 v[4];
    __builtin_memcpy(v, &raw, 8);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// wait until every outstanding global load of this wave -- LDS-DMA included -- has landed (vmcnt = 0; gfx9 encoding of s_waitcnt)
__device__ __forceinline__ void wait_vm0() {
#ifndef CFFM_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
}

// ---- prep: dO rows (f16, window-major, per head), D, scale ----------------------------------------------------------------------
// grid (B * nW * 4), 256 threads: workgroup = (window, head pair), thread t = row (query) t >> 2, 16 channels (t & 3) of the pair's 64.
//   doh [B*nW][8][64][32] f16 = dO * sc   (rows of padded pixels / rows 49..63: zeros)
//   dsc [B*nW][8][64]     f32 = rowsum(doh * O)  (D in the units of doh, from the rounded values: see below)
//   scl [B*nW][8]         f32 = 1 / sc (0 for a window whose dO is all zero), sc = the power of two with max|dO * sc| in [1, 2)
__global__ void __launch_bounds__(256) k_attn_bwd_prep(Geo G, const int* __restrict__ q_dst, const float* __restrict__ ao,
                                                        const float* __restrict__ dao, h16* __restrict__ doh, float* __restrict__ dsc,
                                                        float* __restrict__ scl) {
    __shared__ float smax[2][4];
    const int wb = blockIdx.x >> 2, part = blockIdx.x & 3, w = wb % G.nW, b = wb / G.nW;
    const int tid = threadIdx.x, row = tid >> 2, sub = tid & 3, hh = sub >> 1, head = 2 * part + hh;
    const int qd = row < CFFM_WA ? q_dst[w * CFFM_WA + row] : -1;
    const buf_t rs_ao = buf_make(ao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const buf_t rs_dao = buf_make(dao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const uint32_t po = qd >= 0 ? (uint32_t)qd * (CFFM_C * 4u) + 256u * part + 64u * sub : BUF_OOB;
    const uint32_t ps = (uint32_t)((long)b * G.HW * CFFM_C * 4);
    f32x4 r[4], o[4];
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r[c] = buf_ld16(rs_dao, po, ps + 16 * c);
        o[c] = buf_ld16(rs_ao, po, ps + 16 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(r[c][e]));
    }
    // max over the head's 32 channels x 64 rows: lanes of equal (sub >> 1) inside the wave, then the four waves through LDS
    am = fmaxf(am, __shfl_xor(am, 1, 64));
#pragma unroll
    for (int s = 4; s < 64; s <<= 1) am = fmaxf(am, __shfl_xor(am, s, 64));
    if ((tid & 63) < 4 && (sub & 1) == 0) smax[hh][tid >> 6] = am;
    __syncthreads();
    const float amx = fmaxf(fmaxf(smax[hh][0], smax[hh][1]), fmaxf(smax[hh][2], smax[hh][3]));
    int ex = 0;
    if (amx > 0.f) frexpf(amx, &ex);
    const float sc = (amx > 0.f) ? ldexpf(1.f, 1 - ex) : 1.f;
    h16* dst = doh + (((long)wb * CFFM_HEADS + head) * 64 + row) * CFFM_HD + 16 * (sub & 1);
    // D from the ROUNDED dO: with dP = V dO_h^T both roles then see sum_n P_n (dP_n - D) = 0 exactly, i.e. the exact softmax
    // backward of a dO perturbed by 2^-12 per element -- with D from the unrounded dO the rounding error of dP met an exact D in
    // the cancelling difference dP - D (measured on the stage test: 7e-4 of max|dq| against 2.8e-4)
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        h16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (h16)(r[2 * c][e] * sc);
            v[4 + e] = (h16)(r[2 * c + 1][e] * sc);
            d = fmaf((float)v[e], o[2 * c][e], d);
            d = fmaf((float)v[4 + e], o[2 * c + 1][e], d);
        }
        *(h16x8*)(dst + 8 * c) = v;
    }
    d += __shfl_xor(d, 1, 64);
    if ((sub & 1) == 0) dsc[((long)wb * CFFM_HEADS + head) * 64 + row] = d;
    if (row == 0 && (sub & 1) == 0) scl[(long)wb * CFFM_HEADS + head] = (amx > 0.f) ? 1.f / sc : 0.f;
}

// ---- query-owner kernel -------------------------------------------------------------------------------------------------------------
// A persistent workgroup of 12 waves walks `per_group` windows of one head.  Wave (qt, kh) = (wave & 3, wave >> 2) owns the 16
// queries 16 qt .. + 15 and the key tiles 0..6 / 7..12 / 13..18: what never changes stays in registers -- its 4 bias fragments
// (loaded once) and its <= 7 bias-gradient tiles -- and 12 waves fit a CU at <= 168 registers each (16 waves x 128 spilled) (rounds 2-3 and the
// first round-4 forms: 4 waves x all 19 tiles = 76 + 76 registers of bias and bias gradient per wave, two waves per SIMD).  The
// kernel is bound by VALU issue (exp, the softmax-backward arithmetic), not by the matrix pipe: three waves per SIMD keep it issuing.
// The K / V rows of window i+1 are gathered by LDS-DMA (no staging registers) into the second of two row images while window i is
// multiplied; the table entries that address the gather and the Q / dO fragments, LSE, D and scale of the next window are ordinary
// loads issued BEFORE the DMA and consumed only after the next top-of-loop wait (no arithmetic on them before: a use of a load
// result waits for the load and, loads returning in order, for the DMA behind it), so nothing inside the multiplication waits on
// memory.  One barrier per window: the dQ partial sums of the kh = 1, 2 waves cross to their kh = 0 partner through a
// double-buffered LDS tile and are added one window later.
// LDS: 2 x (K rows | V rows) of 304 x 64 B | 2 x 128 validity flags (keys 176..303: only pooled keys can be absent) | 2 x 2 dQ partials
#define ATT_BWD_Q_THREADS 768
#define ATT_BWD_Q_IMG (2 * CFFM_NKEY_PAD * ATT_KS_STRIDE)     // halfs per (K | V) image pair
#define ATT_BWD_Q_LDS (2 * ATT_BWD_Q_IMG * (int)sizeof(f16) + 2 * 128 * 4 + 2 * 2 * 64 * CFFM_HD * 4)
struct QdRegs { f16x8 q, d; float lq, Dq, isc; };
__global__ void __launch_bounds__(ATT_BWD_Q_THREADS, 1) k_cfm_attn_bwd_q(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                                      const h16* __restrict__ biasH, const h16* __restrict__ doh,
                                                                      const float* __restrict__ dsc, const float* __restrict__ scl,
                                                                      const float* __restrict__ lse_in, float* __restrict__ dqkv,
                                                                      float* __restrict__ dbias_part, int per_group) {
    CFFM_DYN_SMEM(smem);
    const int h = blockIdx.x & 7, grp = blockIdx.x >> 3;
    f16* img = (f16*)smem;
    float* vfl = (float*)(img + 2 * ATT_BWD_Q_IMG);
    f32x4* dqx = (f32x4*)(vfl + 2 * 128);                   // [2 buffers][2 partners][2 channel tiles][4 query tiles][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int qt = wave & 3, kh = wave >> 2, t0 = kh == 0 ? 0 : 6 * kh + 1, cnt = kh == 0 ? 7 : 6, pbase = t0 >> 1;
    const int qcol = 16 * qt + l15;
    const int srow = lane >> 2, sc4 = lane & 3;           // DMA role: row srow of the wave's key tiles, 16-byte chunk sc4
    const int lrow = ATT_ROW(l15, g), ltr0 = att_tr_lane(0, lane), ltr1 = att_tr_lane(16, lane);   // per-lane parts of the LDS addresses
    const float scale = 0.17677669529663687f;
    const int wb0 = grp * per_group;
    const int wb1 = (wb0 + per_group < G.B * G.nW) ? wb0 + per_group : G.B * G.nW;
    const buf_t rs_bias = biash_rsrc(biasH);
    const f16x8 sel0 = bias_sel_frag(lane, 0), sel1 = bias_sel_frag(lane, 1);
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const dma_t dm_qkv = dma_make(qkv, (uint32_t)((long)G.B * G.RC * 768 * 2));
    const buf_t rs_doh = buf_make(doh, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * 64 * CFFM_HD * 2));

    f16x8 bT[4];
    f32x4 dB[7];
#pragma unroll
    for (int p = 0; p < 4; ++p) bT[p] = buf_ld_h8(rs_bias, biash_voff(lane), biash_soff(h, qt, pbase + p));
#pragma unroll
    for (int t = 0; t < 7; ++t) dB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tab_load = [&](int wb, int (&src)[2]) {
        const int* ksrc = key_src + (wb % G.nW) * CFFM_NKEY_PAD;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it;
            src[it] = i < 19 ? ksrc[16 * i + srow] : -1;
        }
    };
    auto dma_issue = [&](int wb, const int (&src)[2], int bi) {
        const uint32_t soff_k = qkv_soff_k(G, wb / G.nW, h);
        f16* Ks = img + bi * ATT_BWD_Q_IMG;
        f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
        uint32_t off[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it, row = 16 * i + srow;
            off[it] = src[it] >= 0 ? (uint32_t)src[it] * 1536u + 16u * (uint32_t)(sc4 ^ ATT_SWZ(row)) : BUF_OOB;
            if (i < 19 && sc4 == 0 && row >= 176) vfl[bi * 128 + row - 176] = src[it] >= 0 ? 0.f : -INFINITY;
        }
        sched_fence();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = wave + 12 * it;
            if (i < 19) {
                dma_ld16(dm_qkv, off[it], soff_k, Ks + 16 * i * ATT_KS_STRIDE);
                dma_ld16(dm_qkv, off[it], soff_k + 512, Vs + 16 * i * ATT_KS_STRIDE);
            }
        }
    };
    auto qd_load = [&](int wb, QdRegs& r) {
        const int w = wb % G.nW, b = wb / G.nW;
        const long wh = (long)wb * CFFM_HEADS + h;
        r.q = buf_ld_h8(rs_qkv, qcol < CFFM_WA ? (uint32_t)(w * CFFM_WA + qcol) * 1536u + 16u * g : BUF_OOB,
                        (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
        r.d = buf_ld_h8(rs_doh, (uint32_t)(qcol * 64 + 16 * g), (uint32_t)(wh * 4096));
        r.lq = lse_in[wh * CFFM_NQ_PAD + qcol];
        r.Dq = dsc[wh * 64 + qcol];
        r.isc = scl[wh];
    };
    // dQ of window wbp (this wave's key quarter in dq) + the partners' quarters from exchange buffer xb -> global (kh = 0 waves)
    auto dq_flush = [&](int wbp, const f32x4 (&dq)[2], float f, int xb) {
        if (kh == 0 && qcol < CFFM_WA) {
            float* drow = dqkv + ((long)(wbp / G.nW) * G.RC + (wbp % G.nW) * CFFM_WA + qcol) * 768 + h * CFFM_HD + 4 * g;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 s = dq[mt];
#pragma unroll
                for (int pn = 0; pn < 2; ++pn) s += dqx[(((xb * 2 + pn) * 2 + mt) * 4 + qt) * 64 + lane];
                *(f32x4*)(drow + 16 * mt) = s * f;      // d(raw q): the stored q carries the 32^-0.5 factor
            }
        }
    };

    int src[2];
    QdRegs nxt;
    if (wb0 < wb1) {
        tab_load(wb0, src);
        qd_load(wb0, nxt);
        dma_issue(wb0, src, 0);
        if (wb0 + 1 < wb1) tab_load(wb0 + 1, src);
    }
    f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float fprev = 0.f;
    for (int wb = wb0; wb < wb1; ++wb) {
        const int bi = (wb - wb0) & 1;
        if (!(BWD_Q_ABLATE & 4)) wait_vm0();
        __syncthreads();                    // window wb's rows are in image bi; image bi ^ 1 and exchange buffer bi are free; the
                                            // partners' partial sums of window wb - 1 are in exchange buffer bi ^ 1
        if (wb > wb0) dq_flush(wb - 1, dq, fprev, bi ^ 1);
        const QdRegs cur = nxt;
        if (wb + 1 < wb1) {
            int src_n[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) src_n[it] = src[it];
            if (wb + 2 < wb1) tab_load(wb + 2, src);
            qd_load(wb + 1, nxt);
            if (!(BWD_Q_ABLATE & 1)) dma_issue(wb + 1, src_n, bi ^ 1);
        }
        const f16* Ks = img + bi * ATT_BWD_Q_IMG + 16 * ATT_KS_STRIDE * t0;      // the wave's first tile
        const f16* Vs = Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE;
        const float* vflag = vfl + bi * 128 + 4 * g;  // flags of keys 176..303
        const float lq2 = cur.lq * CFFM_LOG2E, Dq = cur.Dq, isc = cur.isc;
        dq[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dq[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the wave's tiles t0 .. t0 + cnt - 1 against its three bias fragments (tile pairs pbase ..): which fragment and which selector
        // a tile takes depends on the parity of t0 only, so the body exists twice with compile-time register indices
        auto window = [&](auto odd_c) {
            constexpr int ODD = decltype(odd_c)::value;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                if (2 * kp < cnt) {
                    sched_fence();      // bounds the live ranges: the scheduler otherwise hoists the LDS reads of the later tile pairs (spills)
                    f16x4 dsh[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int lt = 2 * kp + u, t = t0 + lt;
                        if (lt < cnt) {
                            const f16x8 kf = *(const f16x8*)(Ks + 16 * ATT_KS_STRIDE * lt + lrow);
                            const f16x8 vf = *(const f16x8*)(Vs + 16 * ATT_KS_STRIDE * lt + lrow);
                            // only tiles >= 11 (keys >= 176) can hold an absent key
                            const f32x4 c0 = t >= 11 ? vflag4(vflag, 16 * t - 176) : (f32x4){0.f, 0.f, 0.f, 0.f};
                            const f32x4 sv = mfma16x16x32_f16(kf, cur.q, mfma16x16x32_f16(((lt + ODD) & 1) ? sel1 : sel0, bT[(lt + ODD) >> 1], c0));
                            const f32x4 dp = mfma16x16x32_f16(vf, cur.d, (f32x4){0.f, 0.f, 0.f, 0.f});
                            f32x4 ds;
#pragma unroll
                            for (int r = 0; r < 4; ++r) ds[r] = fast_exp2(fmaf(sv[r], CFFM_LOG2E, -lq2)) * (dp[r] - Dq);
                            dB[lt] += ds * isc;
                            dsh[u] = to_f16x4(ds);
                        } else {
                            dsh[u] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                        }
                    }
                    const f16x8 dsf = cat_f16x4(dsh[0], dsh[1]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)       // (a tile pair whose second tile does not exist reads the first twice: its dS half is zeros)
                        dq[mt] = mfma16x16x32_f16(att_tr_frag_at(Ks + 32 * ATT_KS_STRIDE * kp, mt ? ltr1 : ltr0, 2 * kp + 1 >= cnt), dsf, dq[mt]);
                }
            }
        };
        if (BWD_Q_ABLATE & 2) { dq[0][0] = cur.q[0] + cur.d[0] + lq2; }
        else if (t0 & 1) window(std::integral_constant<int, 1>{}); else window(std::integral_constant<int, 0>{});
        fprev = scale * isc;
        if (kh > 0) {
            dqx[(((bi * 2 + kh - 1) * 2 + 0) * 4 + qt) * 64 + lane] = dq[0];
            dqx[(((bi * 2 + kh - 1) * 2 + 1) * 4 + qt) * 64 + lane] = dq[1];
        }
    }
    if (wb0 < wb1) {
        __syncthreads();
        dq_flush(wb1 - 1, dq, fprev, (wb1 - 1 - wb0) & 1);
    }
    // the group's bias gradient: one plain [304 keys][64 queries] tile per (group, head); k_sum_splits adds the groups
    // (rows of padded queries / keys are exact zeros).  Through a buffer resource: one 32-bit per-lane offset, the row offsets are scalars.
    const buf_t rs_dbp = buf_make(dbias_part + (((long)grp * CFFM_HEADS + h) * CFFM_NKEY_PAD) * CFFM_NQ_PAD,
                                  (uint32_t)(CFFM_NKEY_PAD * CFFM_NQ_PAD * 4));
    const uint32_t dvoff = (uint32_t)((4 * g * CFFM_NQ_PAD + qcol) * 4);
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (t < cnt) buf_st4(rs_dbp, dB[t][r], dvoff, (uint32_t)((16 * (t0 + t) + r) * CFFM_NQ_PAD * 4));
}

// ---- key-owner kernel ---------------------------------------------------------------------------------------------------------------
// grid.x = 8 * stride, blockIdx.x % 8 == head (one XCD's L2 serves one head's slice of every token row); 4 waves per workgroup,
// four workgroups per CU (<= 128 registers, 34 KB of LDS): like the query-owner kernel this one is bound by VALU issue, and what
// keeps the SIMDs issuing is waves, not instruction-level tricks.  A persistent workgroup walks units of one head (unit index first,
// first + stride, ...; the host sorts them long-first).  Per unit (geometry.ko_tables): each wave holds ONE tile of 16 key rows as K /
// V fragments and its dK^T / dV^T accumulators in registers (16 + 16); the readers come in intervals of up to two windows, staged by
// LDS-DMA into the second of two buffers while the first is multiplied (Q rows, dO rows, LSE, D: no staging registers, no arithmetic
// on the way -- the common dO scale of a clip is applied to P and dS instead).  Per interval a wave has at most one job: its tile
// against one staged reader, one or both query halves -- 2 or 4 independent S / dP / exp chains in one basic block, then
// dV^T += dO^T P, dK^T += Q^T dS with the C registers of the S = Q K^T orientation as B operands.  The position bias of the job (the
// slot under which the reader sees each key: ko_slot) is an ordinary 8-byte load per chain, issued one interval ahead, before the
// DMA, and consumed after the next top-of-loop wait; the slots themselves two intervals ahead.
// Main units (wave v owns tile v): every dk / dv row is written straight from the owner's registers.  Pooled units (one tile, the
// four waves share its readers and query halves): the four partial sums are added in a fixed order through LDS.  Deterministic.
#define ATT_BWD_K_THREADS 256
#define KO_SLOT_LDS (2 * 64 * ATT_KS_STRIDE * (int)sizeof(f16) + 2 * 64 * 4)      // one staged reader: Q rows | dO rows | LSE | D
#define ATT_BWD_K_LDS (2 * 2 * KO_SLOT_LDS)                                      // two buffers of two readers
#define KO_MAX_CLIPS 64
__global__ void __launch_bounds__(ATT_BWD_K_THREADS, 4) k_cfm_attn_bwd_k(Geo G, const h16* __restrict__ qkv, const h16* __restrict__ biasKT,
                                                                      const h16* __restrict__ doh, const float* __restrict__ dsc,
                                                                      const float* __restrict__ scl, const float* __restrict__ lse_in,
                                                                      const int* __restrict__ ko, const int16_t* __restrict__ ko_slot,
                                                                      float* __restrict__ dqkv, int stride) {
    CFFM_DYN_SMEM(smem);
    __shared__ float iscc[KO_MAX_CLIPS];
    const int h = blockIdx.x & 7, first = blockIdx.x >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int lrow = ATT_ROW(l15, g), ltr0 = att_tr_lane(0, lane), ltr1 = att_tr_lane(16, lane);   // per-lane parts of the LDS addresses
    const int srow = lane >> 2, sc4 = lane & 3;           // DMA role: row srow of a 16-row group, 16-byte chunk sc4

    const int nu = ko[0], total = nu * G.B;
    const int* units = ko + ko[1];
    const int* ints = ko + ko[2];
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const dma_t dm_qkv = dma_make(qkv, (uint32_t)((long)G.B * G.RC * 768 * 2));
    const dma_t dm_doh = dma_make(doh, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * 64 * CFFM_HD * 2));
    const dma_t dm_lse = dma_make(lse_in, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * 64 * 4));
    const dma_t dm_dsc = dma_make(dsc, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * 64 * 4));
    const buf_t rs_bkt = buf_make(biasKT, (uint32_t)(BIASKT_HALFS * 2));
    const uint32_t bkt_soff = (uint32_t)(h * CFFM_NKEY_PAD * CFFM_NQ_PAD * 2);

    // once per workgroup: the common dO scale of every clip (the largest 1 / sc of its windows; P and dS of a reader are brought to
    // it by the exact power-of-two ratio scl / isc_c)
    for (int bb = wave; bb < G.B; bb += 4) {
        float m = 0.f;
        for (int i = lane; i < G.nW; i += 64) m = fmaxf(m, scl[((long)bb * G.nW + i) * CFFM_HEADS + h]);
        m = wave_max(m);
        if (lane == 0) iscc[bb] = m;
    }
    __syncthreads();

    for (int ui = first; ui < total; ui += stride) {
        const int u = ui / G.B, b = ui % G.B;
        const int* urec = units + 8 * u;
        const int kind = urec[2], i0 = urec[3], i1 = urec[4], sbase = urec[5];
        const int* krows = ko + urec[0];
        const int T = kind ? 0 : wave;                       // the tile this wave holds
        const float isc_c = iscc[b], sc_c = isc_c > 0.f ? 1.f / isc_c : 0.f;
        const long wh0 = (long)b * G.nW * CFFM_HEADS + h;   // + reader * 8
        const uint32_t soff_q = (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2);
        // the wave's tile as B-operand fragments, straight from the q|k|v rows (consumed after the first wait below)
        const int myrow = krows[16 * T + l15];
        const uint32_t ko_ = myrow >= 0 ? (uint32_t)myrow * 1536u + 16u * (uint32_t)g : BUF_OOB;
        const f16x8 kf = buf_ld_h8(rs_qkv, ko_, qkv_soff_k(G, b, h));
        const f16x8 vf = buf_ld_h8(rs_qkv, ko_, qkv_soff_k(G, b, h) + 512);

        // Everything an interval needs besides the staged rows comes through per-lane VECTOR loads of wave-uniform addresses (records,
        // slots, scales, bias): they count on vmcnt like the DMA, are issued before it, and their results are first touched right
        // after the next top-of-loop wait -- a scalar load would make the LDS reads of the multiplication wait on lgkmcnt for it, and
        // a use (even a register copy) of an in-flight result behind the DMA issue would wait for the DMA.
        auto job_load = [&](int i, int (&j)[2]) {
            const int* r = ints + 16 * i + 4 + 3 * wave;
            j[0] = r[0]; j[1] = 0;
        };
        auto rd_load = [&](int i, int (&rd)[2]) { rd[0] = ints[16 * i]; rd[1] = ints[16 * i + 1]; };
        // the two layers of this lane's key: slot of the first reading | slot of the second << 16 (-1: none)
        auto slot_load = [&](int i) { return *(const int*)(ko_slot + sbase + ((i - i0) * 4 + wave) * 32 + 2 * l15); };
        // a job's bias: per chain (query tile c) the 4 queries 16 c + 4 g .. + 3 of this lane's key slot; an absent key reads out of
        // range (zeros) and is masked where it is used
        auto bias_load = [&](int slot2, f32x2 (&bb)[4], f32x2 (&bb2)[4]) {
            const int s0 = (int)(int16_t)(slot2 & 0xFFFF), s1 = slot2 >> 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bb[c] = buf_ld8(rs_bkt, s0 >= 0 ? (uint32_t)(s0 * (CFFM_NQ_PAD * 2) + (16 * c + 4 * g) * 2) : BUF_OOB, bkt_soff);
                bb2[c] = buf_ld8(rs_bkt, s1 >= 0 ? (uint32_t)(s1 * (CFFM_NQ_PAD * 2) + (16 * c + 4 * g) * 2) : BUF_OOB, bkt_soff);   // (mostly out of range: no traffic)
            }
        };
        auto f_load = [&](const int (&rd)[2], float (&f)[2]) {
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) f[rs] = rd[rs] >= 0 ? scl[wh0 + (long)rd[rs] * CFFM_HEADS] : 0.f;
        };
        // readers rd -> buffer bi: per reader 4 + 4 row groups and the two 256-byte rows, spread over the 4 waves
        auto dma_issue = [&](const int (&rd)[2], int bi) {
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) {
                const int wr = wave_uniform(rd[rs]);
                if (wr < 0) continue;
                char* sb = smem + (bi * 2 + rs) * KO_SLOT_LDS;
                const long wh = wh0 + (long)wr * CFFM_HEADS;
                const int row = 16 * wave + srow;            // this wave's row group of the reader's 64 query rows
                const uint32_t sw = 16u * (uint32_t)(sc4 ^ ATT_SWZ(row));
                dma_ld16(dm_qkv, row < CFFM_WA ? (uint32_t)(wr * CFFM_WA + row) * 1536u + sw : BUF_OOB, soff_q, sb + 1024 * wave);
                dma_ld16(dm_doh, (uint32_t)(row * 64) + sw, (uint32_t)(wh * 4096), sb + 4096 + 1024 * wave);
                if (wave == 2 * rs) dma_ld4(dm_lse, 4u * (uint32_t)lane, (uint32_t)(wh * 256), sb + 8192);
                if (wave == 2 * rs + 1) dma_ld4(dm_dsc, 4u * (uint32_t)lane, (uint32_t)(wh * 256), sb + 8192 + 256);
            }
        };

        f32x4 accK[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, accV[2] = {accK[0], accK[0]};
        // prologue (synchronous, once per unit): the state "next = interval i0", its rows in flight
        int n_job[2], n_rd[2], nn_rd[2] = {-1, -1}, n_slot, nn_slot = -1;
        f32x2 n_bias[4], n_bias2[4];
        float n_f[2];
        job_load(i0, n_job);
        rd_load(i0, n_rd);
        n_slot = slot_load(i0);
        f_load(n_rd, n_f);
        bias_load(n_slot, n_bias, n_bias2);
        if (i0 + 1 < i1) { rd_load(i0 + 1, nn_rd); nn_slot = slot_load(i0 + 1); }
        __syncthreads();                               // the previous unit's reduction / reads of the buffers are over
        dma_issue(n_rd, 0);
        for (int i = i0; i < i1; ++i) {
            const int bi = (i - i0) & 1;
            wait_vm0();
            __syncthreads();                           // interval i is staged in buffer bi; buffer bi ^ 1 is free
            // everything loaded so far has arrived: "next" becomes "current" (plain register moves here, never further down)
            const int flags = wave_uniform(n_job[0]);
            const int sl_c = (int)(int16_t)(n_slot & 0xFFFF), sl_c2 = n_slot >> 16;
            const float f0 = n_f[0], f1 = n_f[1];
            f32x2 bc[4], bc2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { bc[c] = n_bias[c]; bc2[c] = n_bias2[c]; }
            if (i + 1 < i1) {
                n_rd[0] = nn_rd[0]; n_rd[1] = nn_rd[1];
                n_slot = nn_slot;
                job_load(i + 1, n_job);
                f_load(n_rd, n_f);
                bias_load(n_slot, n_bias, n_bias2);
                if (i + 2 < i1) { rd_load(i + 2, nn_rd); nn_slot = slot_load(i + 2); }
                if (!(BWD_K_ABLATE & 2)) dma_issue(n_rd, bi ^ 1);
            }
            // ---- multiply interval i ----
            if (flags && !(BWD_K_ABLATE & 1)) {
                const int rs = flags & 3, qpm = (flags >> 2) & 3, nl = flags >> 4;
                const char* sb = smem + (bi * 2 + rs) * KO_SLOT_LDS;
                const f16* Qb = (const f16*)sb;
                const f16* dOb = Qb + 64 * ATT_KS_STRIDE;
                const float* lsb = (const float*)(sb + 8192);
                const float f = (rs ? f1 : f0) * sc_c;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    if (!((qpm >> qp) & 1)) continue;
                    f16x4 ph[2], dsh[2];
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu) {
                        const int c = 2 * qp + uu;
                        const f16x8 qf = *(const f16x8*)(Qb + 16 * ATT_KS_STRIDE * c + lrow);
                        const f16x8 dof = *(const f16x8*)(dOb + 16 * ATT_KS_STRIDE * c + lrow);
                        const f32x4 lq = *(const f32x4*)(lsb + 16 * c + 4 * g), Dq = *(const f32x4*)(lsb + 64 + 16 * c + 4 * g);
                        const f32x4 raw = mfma16x16x32_f16(qf, kf, (f32x4){0.f, 0.f, 0.f, 0.f});
                        const f32x4 dp = mfma16x16x32_f16(dof, vf, (f32x4){0.f, 0.f, 0.f, 0.f});
                        const f32x4 b4 = h4_to_f32x4(bc[c]);
                        f32x4 p;
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[r] = sl_c >= 0 ? fast_exp2(fmaf(raw[r], CFFM_LOG2E, (b4[r] - lq[r]) * CFFM_LOG2E)) : 0.f;
                        if (nl > 1) {       // some key of the tile is read twice by this window: the second readings (no load in here)
                            const f32x4 e4 = h4_to_f32x4(bc2[c]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) p[r] += sl_c2 >= 0 ? fast_exp2(fmaf(raw[r], CFFM_LOG2E, (e4[r] - lq[r]) * CFFM_LOG2E)) : 0.f;
                        }
                        f32x4 ds;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { p[r] *= f; ds[r] = p[r] * (dp[r] - Dq[r]); }
                        ph[uu] = to_f16x4(p);
                        dsh[uu] = to_f16x4(ds);
                    }
                    const f16x8 pf = cat_f16x4(ph[0], ph[1]), dsf = cat_f16x4(dsh[0], dsh[1]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        accV[mt] = mfma16x16x32_f16(att_tr_frag_at(dOb + 32 * ATT_KS_STRIDE * qp, mt ? ltr1 : ltr0), pf, accV[mt]);
                        accK[mt] = mfma16x16x32_f16(att_tr_frag_at(Qb + 32 * ATT_KS_STRIDE * qp, mt ? ltr1 : ltr0), dsf, accK[mt]);
                    }
                }
            }
        }
        // ---- the unit's dk / dv rows: every row once, fp32, in place ----
        if (BWD_K_ABLATE & 4) {
            if (accK[0][0] == 12345.f) dqkv[0] = accV[0][0] + accK[1][1] + accV[1][2];
        } else if (kind == 0) {
            if (myrow >= 0) {
                float* drow = dqkv + ((long)b * G.RC + myrow) * 768 + h * CFFM_HD + 4 * g;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    *(f32x4*)(drow + 256 + 16 * mt) = accK[mt] * isc_c;
                    *(f32x4*)(drow + 512 + 16 * mt) = accV[mt] * isc_c;
                    // a pooled row has no query: its q third is zeroed so that the q|k|v weight / bias gradient GEMMs see zeros there
                    if (myrow >= CFFM_WA * G.nW) *(f32x4*)(drow + 16 * mt) = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
            __syncthreads();                           // the last interval's reads of the buffers are over
            f32x4* red = (f32x4*)smem;                 // [4 waves][K0 K1 V0 V1][64 lanes]
            red[(wave * 4 + 0) * 64 + lane] = accK[0]; red[(wave * 4 + 1) * 64 + lane] = accK[1];
            red[(wave * 4 + 2) * 64 + lane] = accV[0]; red[(wave * 4 + 3) * 64 + lane] = accV[1];
            __syncthreads();
            if (myrow >= 0) {                          // wave v sums and stores quarter v (K0, K1, V0, V1) of every row
                const f32x4 s = ((red[(0 * 4 + wave) * 64 + lane] + red[(1 * 4 + wave) * 64 + lane]) + red[(2 * 4 + wave) * 64 + lane]) +
                                red[(3 * 4 + wave) * 64 + lane];
                float* drow = dqkv + ((long)b * G.RC + myrow) * 768 + h * CFFM_HD + 4 * g;
                *(f32x4*)(drow + 256 * (1 + (wave >> 1)) + 16 * (wave & 1)) = s * isc_c;
                if (wave < 2) *(f32x4*)(drow + 16 * wave) = (f32x4){0.f, 0.f, 0.f, 0.f};     // (every row of a pooled unit is a pooled row)
            }
        }
    }
}
