// EXPERIMENT RECORD, not part of the library: the round-2/3 form of the fused attention backward (4 waves x 16 queries, S^T
// orientation, P and dS exchanged through LDS per 32-key chunk, 12 barriers per window, register staging), replaced in round 4 by the
// key-split kernel of csrc/cfm_attn_kernels.h.  Same interface; drop it next to that kernel to compare (scripts/r04_ks.sh did:
// 39.6 us against 38.4 us for the stage alone, 51.1 us against 41.2 us inside the training step under rocprofv3).
// =====================================================================================================
// Fused backward (round 2): ONE kernel does what k_cfm_attn_bwd_q + k_cfm_attn_bwd_kv did with two stagings, two S / dP
// recomputations and two exp passes.  grid (8 heads, NG window groups), 256 threads = 4 waves x 16 queries, two workgroups
// per CU (66 KB of LDS, <= 256 registers).  Per window, the 304 key slots are walked in 10 chunks of 32 keys:
//   query-owner half (S^T orientation: C rows = keys, C columns = queries; a wave owns 16 queries):
//       S^T = K Q^T + bias (+mask), dP^T = V dO^T, P = 2^(S log2e - LSE log2e), dS = P (dP - D);
//       the head's bias gradient accumulates in registers over the whole window group (19 x 4 per lane);
//       dQ^T += K^T dS^T with dS^T straight from the C registers (contraction over keys = C rows);
//       P and dS (f16) are also written to a [64 queries][32 keys] exchange image in LDS;
//   key-owner half (after ONE barrier; the exchange image is double-buffered): contraction over QUERIES, which the C layout of
//       the S^T orientation cannot feed from registers -- the exchange image read back through the LDS transpose read can:
//       wave (u, which): key tile 2 kt + u, dV^T = dO^T P (which = 0) or dK^T = Q^T dS (which = 1), both operands via
//       att_tr_frag so that their k-slot <-> query maps agree; the finished 16-key x 32-channel tile goes to the window's
//       partial rows (k_dkv_gather sums them per token row, deterministic).
// dO is rescaled per window by a power of two so that every f16 gradient operand sits near 1 (training-size gradients of 1e-6
// would flush to zero in f16); results are scaled back in f32.
// =====================================================================================================
#define ATT_BWD_XROWS 64
#define ATT_BWD_LDS ((ATT_VROWS + CFFM_NKEY_PAD + 2 * 64 + 4 * ATT_BWD_XROWS) * ATT_KS_STRIDE * sizeof(f16) + CFFM_NKEY_PAD * 4 + 64 * 4 + 64 * 4 + 16 * 4)
#ifdef BWD_TIMING   // profiling builds only: shader-clock stamps of wave 0 of workgroups (head 0, group 0 / 7 / 30), first window
__device__ long long g_bwd_t[3 * 16];
#define BWD_STAMP(i) do { if (tid == 0 && blockIdx.x == 0 && wb == wb0 && (grp == 0 || grp == 7 || grp == 30)) g_bwd_t[(grp == 0 ? 0 : grp == 7 ? 1 : 2) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BWD_STAMP(i)
#endif
#ifndef BWD_ABLATE
#define BWD_ABLATE 0   // profiling builds only: 1 no key-owner half, 2 no partial-row stores, 4 no exp
#endif
__global__ void __launch_bounds__(256, 2) k_cfm_attn_bwd(Geo G, const h16* __restrict__ qkv, const int* __restrict__ key_src,
                                                           const int* __restrict__ q_dst, const h16* __restrict__ biasH,
                                                           const float* __restrict__ ao, const float* __restrict__ dao,
                                                           const float* __restrict__ lse_in, float* __restrict__ dqkv,
                                                           float* __restrict__ dbias_part, float* __restrict__ dkv_part, int per_group) {
    CFFM_DYN_SMEM(smem);
    f16* Ks = (f16*)smem;                                   // K rows (+16 zero rows: read transposed 32 keys at a time)
    f16* Vs = Ks + ATT_VROWS * ATT_KS_STRIDE;
    f16* Qs = Vs + CFFM_NKEY_PAD * ATT_KS_STRIDE;           // 64 query rows
    f16* dOs = Qs + 64 * ATT_KS_STRIDE;
    f16* Xs = dOs + 64 * ATT_KS_STRIDE;                     // exchange images: [buffer 2][P | dS][64 queries][32 keys]
    float* vflag = (float*)(Xs + 4 * ATT_BWD_XROWS * ATT_KS_STRIDE);
    float* slse = vflag + CFFM_NKEY_PAD;                    // LSE * log2(e) per query
    float* sD = slse + 64;                                  // rowsum(dO * O) * sc per query
    float* smax = sD + 64;

    const int h = blockIdx.x, grp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int qcol = 16 * wave + l15;
    const float scale = 0.17677669529663687f;
    const int wb0 = grp * per_group;
    const int wb1 = (wb0 + per_group < G.B * G.nW) ? wb0 + per_group : G.B * G.nW;
    // the bias tiles and the partial rows go through buffer resources: one 32-bit per-lane offset each, everything else is a
    // scalar offset (with plain pointers the compiler hoists one 64-bit per-lane address per tile out of the unrolled loops
    // and spills: 85 registers in the first version of this kernel)
    const buf_t rs_bias = biash_rsrc(biasH);
    const uint32_t bias_soff = biash_soff(h, wave, 0), bias_voff = biash_voff(lane);
    const f16x8 sel0 = bias_sel_frag(lane, 0), sel1 = bias_sel_frag(lane, 1);
    // partial rows as f16 (row = 512 halfs: K | V x 8 heads x 32) in units of the window's power-of-two dO scale, which goes to
    // part_scale[window][head]: half the bytes of fp32 rows on the way out and in k_dkv_gather
    const buf_t rs_part = buf_make(dkv_part, (uint32_t)((long)G.B * G.nW * CFFM_NKEY_PAD * 512 * 2));
    float* part_scale = dkv_part + (long)G.B * G.nW * CFFM_NKEY_PAD * 256;
    const buf_t rs_qkv = qkv_rsrc(G, qkv);
    const buf_t rs_ao = buf_make(ao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const buf_t rs_dao = buf_make(dao, (uint32_t)((long)G.B * G.HW * CFFM_C * 4));
    const int srow = tid >> 2, sc4 = tid & 3;               // staging role: row (query) srow, 16-byte chunk sc4 of Q / 8 channels of dO, O

    f32x4 dB[19];
#pragma unroll
    for (int t = 0; t < 19; ++t) dB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid < 64) {
        f16x8 z8;
        for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
        *(f16x8*)(Ks + CFFM_NKEY_PAD * ATT_KS_STRIDE + 8 * tid) = z8;
    }
    KvTab<256> tab;
    if (wb0 < wb1) kv_tab_load<256>(tab, key_src + (wb0 % G.nW) * CFFM_NKEY_PAD, tid);

    for (int wb = wb0; wb < wb1; ++wb) {
        const int w = wb % G.nW, b = wb / G.nW;
        BWD_STAMP(0);
        // ---- stage: K / V rows (table entries were fetched during the previous window), Q rows, dO / O rows -> D, |dO| maximum
        KvRegs<256> kv;
        kv_rows_load<256>(kv, tab, rs_qkv, qkv_soff_k(G, b, h), tid);
        const int qd = (srow < CFFM_WA) ? q_dst[w * CFFM_WA + srow] : -1;
        const f16x8 qrow = buf_ld_h8(rs_qkv, srow < CFFM_WA ? (uint32_t)(w * CFFM_WA + srow) * 1536u + 16u * sc4 : BUF_OOB,
                                     (uint32_t)(((long)b * G.RC * 768 + h * CFFM_HD) * 2));
        const uint32_t po = qd >= 0 ? (uint32_t)qd * (CFFM_C * 4u) + 32u * sc4 : BUF_OOB;
        const uint32_t ps = (uint32_t)(((long)b * G.HW * CFFM_C + h * CFFM_HD) * 4);
        f32x4 r0 = buf_ld16(rs_dao, po, ps), r1 = buf_ld16(rs_dao, po, ps + 16);
        const f32x4 o0 = buf_ld16(rs_ao, po, ps), o1 = buf_ld16(rs_ao, po, ps + 16);
        if (tid < 64) slse[tid] = lse_in[((long)wb * CFFM_HEADS + h) * CFFM_NQ_PAD + tid] * CFFM_LOG2E;
        if (wb + 1 < wb1) kv_tab_load<256>(tab, key_src + ((wb + 1) % G.nW) * CFFM_NKEY_PAD, tid);   // next window's entries
        BWD_STAMP(1);
        kv_store<256>(kv, Ks, Vs, vflag, tid);
        *(f16x8*)(Qs + ATT_ROW(srow, sc4)) = qrow;
        BWD_STAMP(2);
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(r0[e]), fabsf(r1[e])));
        amax = wave_max(amax);
        if (lane == 0) smax[wave] = amax;
        __syncthreads();
        const float am = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        int ex = 0;
        if (am > 0.f) frexpf(am, &ex);
        const float sc = (am > 0.f) ? ldexpf(1.f, 1 - ex) : 1.f, isc = 1.f / sc;   // max|dO * sc| in [1,2) over the window
        {
            // D = rowsum(dO * O) from the ROUNDED dO (round 4): with dP = V dO_h^T the kernel then sees sum_n P_n (dP_n - D) = 0 exactly,
            // i.e. the exact softmax backward of a dO perturbed by 2^-12 per element.  With D from the unrounded dO the rounding error of
            // dP met an exact D in the cancelling difference dP - D (two-pass experiment, stage test: 7e-4 of max|dq| against 2.8e-4).
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
            if (sc4 == 0) sD[srow] = d;
            *(f16x8*)(dOs + ATT_ROW(srow, sc4)) = dh;
        }
        __syncthreads();

        BWD_STAMP(3);
        const f16x8 qfrag = *(const f16x8*)(Qs + ATT_ROW(qcol, g));
        const f16x8 dofrag = *(const f16x8*)(dOs + ATT_ROW(qcol, g));
        const float lq2 = slse[qcol], Dq = sD[qcol];
        f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        const int ku = wave & 1, kwhich = wave >> 1;          // key-owner role of this wave
        const f16* kimg = kwhich ? Qs : dOs;
        const uint32_t part_voff = (uint32_t)(l15 * 1024 + (h * 2 * CFFM_HD + (kwhich ? 0 : CFFM_HD) + 4 * g) * 2);
        if (tid == 0) part_scale[(long)wb * CFFM_HEADS + h] = isc;

        // Software pipeline over the 10 chunks: between two barriers a wave runs the key-owner half of chunk kt AND the
        // query-owner half of chunk kt + 1 -- two independent dependency chains the scheduler interleaves (one chain alone
        // leaves the wave parked on LDS / MFMA / exp latencies: the first version, one chain per barrier interval, ran at 45 %
        // issue utilisation).  The bias tiles of a chunk are loaded one interval ahead, BEFORE the previous interval's
        // partial-row stores (a load older than the stores never waits for them: gfx9's vmcnt counts both, in order).
        f16x8 cb = buf_ld_h8(rs_bias, bias_voff, bias_soff);             // chunk kt = tile pair kt: one fragment (cfm_attn_kernels.h, bias_sel_frag)
        f16x8 nb = buf_ld_h8(rs_bias, bias_voff, bias_soff + 1024);
#define BWD_QHALF(KT)                                                                                                                  \
        {                                                                                                                               \
            f16* Px_ = Xs + ((KT) & 1) * 2 * ATT_BWD_XROWS * ATT_KS_STRIDE;                                                             \
            f16* Sx_ = Px_ + ATT_BWD_XROWS * ATT_KS_STRIDE;                                                                             \
            f16x4 dsh[2];                                                                                                               \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                             \
                const int t = 2 * (KT) + u;                                                                                             \
                if (t < 19) {                                                                                                           \
                    const f16x8 kf = *(const f16x8*)(Ks + ATT_ROW((16 * t + l15), g));                                                  \
                    const f16x8 vf = *(const f16x8*)(Vs + ATT_ROW((16 * t + l15), g));                                                  \
                    const f32x4 c0 = (16 * t + 15 >= CFFM_FIRST_POOLED_KEY) ? vflag4(vflag, 16 * t + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f}; \
                    const f32x4 sv = mfma16x16x32_f16(kf, qfrag, mfma16x16x32_f16(u ? sel1 : sel0, cb, c0));                             \
                    const f32x4 dp = mfma16x16x32_f16(vf, dofrag, (f32x4){0.f, 0.f, 0.f, 0.f});                                         \
                    f32x4 pr, ds;                                                                                                       \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                     \
                        pr[r] = (BWD_ABLATE & 4) ? fmaf(sv[r], CFFM_LOG2E, -lq2) : fast_exp2(fmaf(sv[r], CFFM_LOG2E, -lq2));            \
                        ds[r] = pr[r] * (dp[r] - Dq);                                                                                   \
                    }                                                                                                                   \
                    dB[t < 19 ? t : 0] += ds * isc;                                                                                     \
                    dsh[u] = to_f16x4(ds);                                                                                              \
                    /* exchange images: row = query, 8 bytes = keys 16u + 4g .. +3 of this chunk */                                     \
                    *(f16x4*)(Px_ + ATT_ROW(qcol, 2 * u + (g >> 1)) + 4 * (g & 1)) = to_f16x4(pr);                                      \
                    *(f16x4*)(Sx_ + ATT_ROW(qcol, 2 * u + (g >> 1)) + 4 * (g & 1)) = dsh[u];                                            \
                } else {                                                                                                                \
                    dsh[u] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};                                                           \
                }                                                                                                                       \
            }                                                                                                                           \
            const f16x8 dsf = cat_f16x4(dsh[0], dsh[1]);                                                                                \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                            \
                dq[mt] = mfma16x16x32_f16(att_tr_frag(Ks, 32 * (KT), 16 * mt, lane), dsf, dq[mt]);                                      \
        }
        BWD_QHALF(0)
        BWD_STAMP(4);
#pragma unroll
        for (int kt = 0; kt < 10; ++kt) {
            const f16* Px = Xs + (kt & 1) * 2 * ATT_BWD_XROWS * ATT_KS_STRIDE;
            const f16* Sx = Px + ATT_BWD_XROWS * ATT_KS_STRIDE;
            __syncthreads();   // chunk kt's P / dS images are complete (double-buffered: this buffer is rewritten only after the next barrier)
            BWD_STAMP(5 + kt);
            sched_fence();
            // bias tiles: chunk kt + 1's become current, chunk kt + 2's go in flight
            cb = nb;
            if (kt + 2 < 10) nb = buf_ld_h8(rs_bias, bias_voff, bias_soff + 1024 * (kt + 2));
            // ---- query-owner half of the NEXT chunk (independent of the key-owner half below: interleaved by the scheduler)
            if (kt + 1 < 10) BWD_QHALF(kt + 1)
            // ---- key-owner half: wave (ku, kwhich) finishes key tile 2 kt + ku for dV (kwhich 0) or dK (kwhich 1)
            const int tk = 2 * kt + ku;
            if (tk < 19 && !(BWD_ABLATE & 1)) {
                const f16* X = kwhich ? Sx : Px;
                f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f16x8 xb = att_tr_frag(X, 32 * ks, 16 * ku, lane);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) acc[dt] = mfma16x16x32_f16(att_tr_frag(kimg, 32 * ks, 16 * dt, lane), xb, acc[dt]);
                }
                // tile [d = 16 dt + 4 g + r][key = l15]: every lane owns 16 contiguous bytes of a key row of this window's slot;
                // present keys only (flag 0, -inf otherwise): an absent key's store goes out of range and is dropped
                const int key = 16 * tk + l15;
                if (!(BWD_ABLATE & 2)) {
                    const uint32_t so = (uint32_t)(((long)wb * CFFM_NKEY_PAD + 16 * tk) * 1024);
                    const uint32_t vo = vflag[key] == 0.f ? part_voff : BUF_OOB;
                    buf_st8(rs_part, __builtin_bit_cast(f32x2, to_f16x4(acc[0])), vo, so);
                    buf_st8(rs_part, __builtin_bit_cast(f32x2, to_f16x4(acc[1])), vo, so + 32);
                }
            }
        }
#undef BWD_QHALF
        sched_fence();
        BWD_STAMP(15);
        if (qcol < CFFM_WA) {
            float* drow = dqkv + ((long)b * G.RC + w * CFFM_WA + qcol) * 768 + h * CFFM_HD + 4 * g;
            *(f32x4*)(drow) = dq[0] * (scale * isc);      // d(raw q): the stored q carries the 32^-0.5 factor
            *(f32x4*)(drow + 16) = dq[1] * (scale * isc);
        }
        __syncthreads();  // LDS is restaged for the next window
    }
    // the group's bias gradient: one plain [304 keys][64 queries] tile per (group, head); k_sum_splits adds the groups
    // (rows of padded queries / keys are exact zeros)
    float* dst = dbias_part + (((long)grp * CFFM_HEADS + h) * CFFM_NKEY_PAD) * CFFM_NQ_PAD + qcol;
#pragma unroll
    for (int t = 0; t < 19; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * t + 4 * g + r) * CFFM_NQ_PAD] = dB[t][r];
}

