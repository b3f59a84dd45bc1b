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
#define BWD_K_ABLATE 0   // profiling builds only (key-owner role): 1 no multiplication, 2 no reduction / stores, 4 no row loads, 8 no bias reads
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
                buf_ld16_lds(rs_qkv, off[it], soff_k, Ks + 16 * i * ATT_KS_STRIDE);
                buf_ld16_lds(rs_qkv, off[it], soff_k + 512, Vs + 16 * i * ATT_KS_STRIDE);
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
// grid.x = 8 * stride, blockIdx.x % 8 == head (one XCD's L2 serves one head's slice of every token row).  A persistent workgroup of 8
// waves walks key-owner units of one head (unit index first, first + stride, ...; the host sorts them long-first).  A pass = (reader window, one half of its 64 padded queries); the host spreads a unit's passes over the 8 waves.
// LDS: the unit's 64 K rows and 64 V rows | per wave: the pass's 32 Q rows, 32 dO rows, LSE * log2(e), D | the head's bias, key-major
// [289 slots][64 queries] f16 with the 16-byte chunks of a row XOR-ed by (slot >> 1) & 7 (lanes of a wave read 16 different slots at
// the same query offset) | the common dO scale of every clip | the reduction area of the 8 waves' accumulators
#define KO_WAVES 8
#define KO_UNIT_REC 12
#define KO_WAVE_LDS (2 * 32 * ATT_KS_STRIDE * (int)sizeof(f16) + 2 * 32 * 4)
#define KO_BIAS_LDS ((CFFM_NKEY + 1) * CFFM_NQ_PAD * (int)sizeof(f16))   // row 289: -inf, what an absent key reads
#define KO_MAX_CLIPS 64
#define KO_RED_LDS (KO_WAVES * KO_TILES_MAX * 2 * 64 * 16)
#define ATT_BWD_K_LDS (2 * 64 * ATT_KS_STRIDE * (int)sizeof(f16) + KO_WAVES * KO_WAVE_LDS + KO_BIAS_LDS + KO_MAX_CLIPS * 4 + KO_RED_LDS)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct KoPre {
    f16x8 q[2], d[2];
    float lse, D, f;
    u32x2 sl;       // the pass's layer-0 slots of this lane's key (4 tiles x int16)
    int nlp, so, qp;
};
__device__ __forceinline__ int ko_slot_of(u32x2 sl, int T) {
    const uint32_t wd = sl[T >> 1];
    return (int)(int16_t)((T & 1) ? (wd >> 16) : (wd & 0xFFFFu));
}
#define ATT_BWD_K_THREADS 512
__global__ void __launch_bounds__(ATT_BWD_K_THREADS, 1) k_cfm_attn_bwd_k(Geo G, const h16* __restrict__ qkv, const h16* __restrict__ biasKT,
                                                                      const h16* __restrict__ doh, const float* __restrict__ dsc,
                                                                      const float* __restrict__ scl, const float* __restrict__ lse_in,
                                                                      const int* __restrict__ ko, const int16_t* __restrict__ ko_slot,
                                                                      float* __restrict__ dqkv, int stride) {
    CFFM_DYN_SMEM(smem);
    const int h = blockIdx.x & 7, first = blockIdx.x >> 3;
    f16* Ks = (f16*)smem;
    f16* Vs = Ks + 64 * ATT_KS_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int lrow = ATT_ROW(l15, g), ltr0 = att_tr_lane(0, lane), ltr1 = att_tr_lane(16, lane);   // per-lane parts of the LDS addresses
    char* wbase = smem + 2 * 64 * ATT_KS_STRIDE * sizeof(f16);
    f16* Qw = (f16*)(wbase + wave * KO_WAVE_LDS);
    f16* dOw = Qw + 32 * ATT_KS_STRIDE;
    float* lsw = (float*)(dOw + 32 * ATT_KS_STRIDE);
    float* Dw = lsw + 32;
    char* bsm = wbase + KO_WAVES * KO_WAVE_LDS;
    float* iscc = (float*)(bsm + KO_BIAS_LDS);
    f32x4* red = (f32x4*)(iscc + KO_MAX_CLIPS);

    const int nu = ko[0], total = nu * G.B;
    const int* units = ko + ko[1];
    const int* passes = ko + ko[2];
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const buf_t rs_doh = buf_make(doh, (uint32_t)((long)G.B * G.nW * CFFM_HEADS * 64 * CFFM_HD * 2));

    // once per workgroup: the head's bias table and the common dO scale of every clip (the largest 1 / sc of its windows: every
    // window is brought to it while staged, exact power-of-two ratios)
    {
        const h16* src = biasKT + (long)h * CFFM_NKEY_PAD * CFFM_NQ_PAD;
        for (int i = tid; i < CFFM_NKEY * 8; i += 512) {
            const int slot = i >> 3, c = i & 7;
            *(f32x4*)(bsm + slot * 128 + 16 * (c ^ ((slot >> 1) & 7))) = *(const f32x4*)(src + slot * CFFM_NQ_PAD + 8 * c);
        }
        if (tid < 64) ((h16*)(bsm + CFFM_NKEY * 128))[tid] = (h16)(-INFINITY);
        for (int bb = wave; bb < G.B; bb += KO_WAVES) {
            float m = 0.f;
            for (int i = lane; i < G.nW; i += 64) m = fmaxf(m, scl[((long)bb * G.nW + i) * CFFM_HEADS + h]);
            m = wave_max(m);
            if (lane == 0) iscc[bb] = m;
        }
    }
    // own K / V rows: threads 0..255 = (row t >> 2, 16-byte chunk t & 3)
    auto kv_load = [&](int ui, f16x8& kr, f16x8& vr) {
        const int u = ui / G.B, b = ui % G.B, row = (tid & 255) >> 2, c = tid & 3;
        const int* urec = units + KO_UNIT_REC * u;
        const int src = (tid < 256 && row < 16 * urec[1]) ? ko[urec[0] + row] : -1;
        const uint32_t o = src >= 0 ? (uint32_t)src * 1536u + 16u * (uint32_t)c : BUF_OOB;
        const uint32_t soff_k = qkv_soff_k(G, b, h);
        kr = buf_ld_h8(rs_qkv, o, soff_k);
        vr = buf_ld_h8(rs_qkv, o, soff_k + 512);
    };
    // the record of a pass (reader window, layers per tile, slot block, query half) is loaded one pass before the rows it addresses:
    // fetched together they are two dependent round trips, and the second sat in front of every pass (measured: 1.5 us per step)
    struct KoRec { int w, nlp, so, qp; };
    auto rec_load = [&](int pi, KoRec& r) {
        const int* pr = passes + 4 * pi;
        r.w = pr[0]; r.nlp = pr[1]; r.so = pr[2]; r.qp = pr[3];
    };
    auto prefetch = [&](const KoRec& r, int b, KoPre& P) {
        const int wr = wave_uniform(r.w);
        P.nlp = wave_uniform(r.nlp); P.so = wave_uniform(r.so); P.qp = wave_uniform(r.qp);
        const long wh = ((long)b * G.nW + wr) * CFFM_HEADS + h;
        const uint32_t soff_q = (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 32 * P.qp + 16 * j + (lane >> 2);
            P.q[j] = buf_ld_h8(rs_qkv, (row < CFFM_WA && !(BWD_K_ABLATE & 4)) ? (uint32_t)(wr * CFFM_WA + row) * 1536u + 16u * (uint32_t)(lane & 3) : BUF_OOB, soff_q);
            P.d[j] = buf_ld_h8(rs_doh, (BWD_K_ABLATE & 4) ? BUF_OOB : (uint32_t)(2048 * P.qp + 1024 * j + 16 * lane), (uint32_t)(wh * 4096));
        }
        const int ql = 32 * P.qp + (lane & 31);
        P.lse = lse_in[wh * CFFM_NQ_PAD + ql];
        P.D = dsc[wh * 64 + ql];
        P.f = scl[wh];                                  // (raw: arithmetic on a load result here would make the prefetch synchronous)
        P.sl = *(const u32x2*)(ko_slot + P.so + 4 * l15);
    };
    auto stage = [&](const KoPre& P, float sc_c) {
        const float f = P.f * sc_c;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * j + (lane >> 2), c = lane & 3;
            *(f16x8*)(Qw + ATT_ROW(row, c)) = P.q[j];
            f16x8 dv;
#pragma unroll
            for (int e = 0; e < 8; ++e) dv[e] = (f16)((float)P.d[j][e] * f);
            *(f16x8*)(dOw + ATT_ROW(row, c)) = dv;
        }
        if (lane < 32) {
            lsw[lane] = 32 * P.qp + lane < CFFM_WA ? P.lse * CFFM_LOG2E : INFINITY;     // a padded query row: P = 2^(-inf) = 0
            Dw[lane] = P.D * f;
        }
    };

    int ui = first;
    f16x8 kpre, vpre;
    KoPre cur;
    if (ui < total) kv_load(ui, kpre, vpre);
    __syncthreads();                                    // bias table, clip scales
    KoRec rec;
    if (ui < total) {
        const int* urec = units + KO_UNIT_REC * (ui / G.B);
        const int pb = wave_uniform(urec[2 + wave]), pe = wave_uniform(urec[3 + wave]);
        if (pb < pe) { rec_load(pb, rec); prefetch(rec, ui % G.B, cur); }
        if (pb + 1 < pe) rec_load(pb + 1, rec);
    }
    while (ui < total) {
        const int u = ui / G.B, b = ui % G.B;
        const int* urec = units + KO_UNIT_REC * u;
        const int ntile = urec[1];
        const int* krows = ko + urec[0];
        const int p_begin = wave_uniform(urec[2 + wave]), p_end = wave_uniform(urec[3 + wave]);
        const float isc_c = iscc[b], sc_c = isc_c > 0.f ? 1.f / isc_c : 0.f;
        if (tid < 256) {
            *(f16x8*)(Ks + ATT_ROW(tid >> 2, tid & 3)) = kpre;
            *(f16x8*)(Vs + ATT_ROW(tid >> 2, tid & 3)) = vpre;
        }
        const int nxt = ui + stride;
        if (nxt < total) kv_load(nxt, kpre, vpre);     // the next unit's rows: consumed at the top of the next iteration
        __syncthreads();                               // the unit's K / V rows are in place (and the previous unit's dV sums are read)

        f32x4 accK[KO_TILES_MAX][2], accV[KO_TILES_MAX][2];
#pragma unroll
        for (int T = 0; T < KO_TILES_MAX; ++T)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { accK[T][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accV[T][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        for (int pi = p_begin; pi < p_end; ++pi) {
            wave_lds_sync();                           // the previous pass's reads of the wave's LDS region are done
            stage(cur, sc_c);
            const int nlp = cur.nlp, so = cur.so, qp = cur.qp;
            const u32x2 sl = cur.sl;
            wave_lds_sync();
            if (pi + 1 < p_end) prefetch(rec, b, cur);         // the next pass's rows fly while this one is multiplied
            if (pi + 2 < p_end) rec_load(pi + 2, rec);         // ... and the record of the one after (consumed at the next pass)
            f16x8 qf[2], dof[2], qT[2], doT[2];
            f32x4 lq[2], Dq[2];
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                qf[uu] = *(const f16x8*)(Qw + 16 * ATT_KS_STRIDE * uu + lrow);
                dof[uu] = *(const f16x8*)(dOw + 16 * ATT_KS_STRIDE * uu + lrow);
                lq[uu] = *(const f32x4*)(lsw + 16 * uu + 4 * g);
                Dq[uu] = *(const f32x4*)(Dw + 16 * uu + 4 * g);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                qT[mt] = att_tr_frag_at(Qw, mt ? ltr1 : ltr0);
                doT[mt] = att_tr_frag_at(dOw, mt ? ltr1 : ltr0);
            }
#pragma unroll
            for (int T = 0; T < KO_TILES_MAX; ++T) {
                const int nl = (BWD_K_ABLATE & 1) ? 0 : (nlp >> (4 * T)) & 15;
                if (nl == 0) continue;
                const f16x8 kf = *(const f16x8*)(Ks + 16 * ATT_KS_STRIDE * T + lrow);
                const f16x8 vf = *(const f16x8*)(Vs + 16 * ATT_KS_STRIDE * T + lrow);
                const int slot0 = ko_slot_of(sl, T);
                f16x4 ph[2], dsh[2], dsl[2];
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const int c8 = 4 * (2 * qp + uu) + g;        // this lane's 8-byte unit (queries 4 c8 .. + 3) of a bias row
                    const f32x4 raw = mfma16x16x32_f16(qf[uu], kf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    const f32x4 dp = mfma16x16x32_f16(dof[uu], vf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    f32x4 p = (f32x4){0.f, 0.f, 0.f, 0.f};
                    int slot = slot0;
                    for (int l = 0; l < nl; ++l) {
                        if (l > 0) slot = ko_slot[so + 64 * l + 4 * l15 + T];
                        const int sidx = slot >= 0 ? slot : CFFM_NKEY;      // (an absent key reads the -inf row: no select per element)
                        const f32x4 bh4 = (BWD_K_ABLATE & 8) ? (f32x4){0.f, 0.f, 0.f, 0.f} : h4_to_f32x4(*(const f32x2*)(bsm + sidx * 128 + 16 * ((c8 >> 1) ^ ((sidx >> 1) & 7)) + 8 * (c8 & 1)));
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[r] += fast_exp2(fmaf(bh4[r], CFFM_LOG2E, fmaf(raw[r], CFFM_LOG2E, -lq[uu][r])));
                    }
                    f32x4 ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = p[r] * (dp[r] - Dq[uu][r]);
                    ph[uu] = to_f16x4(p);
                    dsh[uu] = to_f16x4(ds);
#if BWD_DS_LO
                    f32x4 rem;
#pragma unroll
                    for (int r = 0; r < 4; ++r) rem[r] = ds[r] - (float)dsh[uu][r];
                    dsl[uu] = to_f16x4(rem);
#endif
                }
                const f16x8 pf = cat_f16x4(ph[0], ph[1]), dsf = cat_f16x4(dsh[0], dsh[1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    accV[T][mt] = mfma16x16x32_f16(doT[mt], pf, accV[T][mt]);
                    accK[T][mt] = mfma16x16x32_f16(qT[mt], dsf, accK[T][mt]);
                }
#if BWD_DS_LO
                const f16x8 dsf_lo = cat_f16x4(dsl[0], dsl[1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) accK[T][mt] = mfma16x16x32_f16(qT[mt], dsf_lo, accK[T][mt]);
#endif
            }
        }
        // the next unit's first pass, so that its rows fly during the reduction below
        if (nxt < total) {
            const int* nrec = units + KO_UNIT_REC * (nxt / G.B);
            const int pb = wave_uniform(nrec[2 + wave]), pe = wave_uniform(nrec[3 + wave]);
            if (pb < pe) { rec_load(pb, rec); prefetch(rec, nxt % G.B, cur); }
            if (pb + 1 < pe) rec_load(pb + 1, rec);
        }
        // ---- the eight waves' sums, added in a fixed order; every dk / dv row of the unit is written once (fp32, in place) -----
#pragma unroll
        for (int part = 0; part < ((BWD_K_ABLATE & 2) ? 0 : 2); ++part) {
            if (part) __syncthreads();                 // the dK sums are read
#pragma unroll
            for (int T = 0; T < KO_TILES_MAX; ++T)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) red[(wave * KO_TILES_MAX * 2 + T * 2 + mt) * 64 + lane] = part ? accV[T][mt] : accK[T][mt];
            __syncthreads();
            const int e = tid, T = e >> 7, mt = (e >> 6) & 1, ln = e & 63;
            const int row = T < ntile ? krows[16 * T + (ln & 15)] : -1;
            if (row >= 0) {
                f32x4 s = red[e];
#pragma unroll
                for (int v = 1; v < KO_WAVES; ++v) s += red[v * KO_TILES_MAX * 2 * 64 + e];
                float* drow = dqkv + ((long)b * G.RC + row) * 768 + h * CFFM_HD + 16 * mt + 4 * (ln >> 4);
                *(f32x4*)(drow + (part ? 512 : 256)) = s * isc_c;
                // a pooled row has no query: its q third is zeroed so that the q|k|v weight / bias gradient GEMMs see zeros there
                if (part == 0 && row >= CFFM_WA * G.nW) *(f32x4*)drow = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        ui = nxt;
    }
}
