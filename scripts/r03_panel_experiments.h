// Rejected variants of the fused row-panel Mlp kernels (round 3: wave-specialised stores and friends), moved out of
// vss_cffm_amd/csrc/panel_kernels.h in round 4.  Include AFTER panel_kernels.h in a -DCFFM_EXPERIMENTS build (scripts/r03_mlp_bench.hip).
#pragma once
#ifdef CFFM_EXPERIMENTS   // measured and rejected (DESIGN.md section 3, round-3 dead ends): kept for scripts/r03_mlp_bench.hip only
// =====================================================================================================================
// Wave-specialised forms of the two fused kernels (round 3, second step): 8 compute waves + 4 STORE waves per workgroup.
// Why: on gfx9-family parts a wave's loads and stores complete through ONE in-order counter (vmcnt).  A compute wave that stores its
// epilogue (hraw, act, x1, ...: 81 MB per launch of the fused forward) cannot see any B-fragment load it issued AFTER a store until that
// store has been acknowledged -- with every CU storing, the acknowledge latency is the HBM write queue (~4 TB/s): measured 44.6 us for
// the fused forward against 24.4 us with the stores removed, whether they are issued in bursts or spread one per k-step.  Here the
// compute waves never store to global memory: results go to LDS (the hi / lo images the next product reads anyway, plus an fp32 tile for
// what is kept in full precision), and four extra waves copy them out as whole 1 KiB rows while the compute waves are in the next
// product; their own vmcnt stalls cost nothing.  All 12 waves run the same barrier sequence; a buffer a store wave reads is rewritten
// two barriers later at the earliest.  VGPRs: the kernel-wide allocation must allow 3 waves per SIMD (<= 168): ring depth 2.
#define PNL_WS_THREADS 768
#define PNL_WS_STORE_WAVES 4
// fp32 tile [rows][256], 16-byte chunk c of row r at chunk position c ^ (r & 15): conflict-free ds_write_b128 from the C^T layout
// (16 rows x one chunk per quarter wave) and ds_read_b128 of whole rows
__device__ __forceinline__ int pnl_tile_off(int row, int n) { return row * 256 + ((((n >> 2) ^ row) & 15) << 2) + ((n >> 2) & ~15) * 4; }
#define PNL_WS_FWD_LDS(MT) (6 * PNL_IMG(MT) * 2 + 2 * 16 * (MT) * 256 * 4)              // P + 2 ACT + 2 OUT tiles (reduction scratch inside OUT1)
#define PNL_WS_BWD_LDS(MT) (6 * PNL_IMG(MT) * 2 + 16 * (MT) * 256 * 4 + 2 * PNL_WAVES * 16 * (MT) * 4 + (2 * 256 + 1024) * 4)

// store waves: an fp32 tile -> rows m0.. of dst[., ld] at column col0 (whole 1 KiB row segments)
template <int MT>
__device__ __forceinline__ void pnl_flush_f32(const float* __restrict__ tile, float* __restrict__ dst, long ld, int col0, long m0, int NP,
                                              int sw, int lane) {
#pragma unroll
    for (int k = 0; k < 16 * MT / PNL_WS_STORE_WAVES; ++k) {
        const int r = sw + PNL_WS_STORE_WAVES * k;
        const long m = m0 + r;
        const f32x4 v = *(const f32x4*)(tile + r * 256 + 4 * lane);
        const int c = (lane & ~15) | ((lane ^ r) & 15);              // the logical chunk stored at physical position `lane`
        if (m < NP && !(PNL_ABLATE & 4)) *(f32x4*)(dst + m * ld + col0 + 4 * c) = v;
    }
}
// store waves: split-4 rows ({hi x4, lo x4} per 4 floats) out of a hi / lo image pair
template <int MT>
__device__ __forceinline__ void pnl_flush_split4(const bf16* __restrict__ hi, const bf16* __restrict__ lo, float* __restrict__ dst, long ld,
                                                 int col0, long m0, int NP, int sw, int lane) {
#pragma unroll
    for (int k = 0; k < 16 * MT / PNL_WS_STORE_WAVES; ++k) {
        const int r = sw + PNL_WS_STORE_WAVES * k;
        const long m = m0 + r;
        const int o = pnl_off(r, lane >> 1) + 4 * (lane & 1);
        const bf16x4 h = *(const bf16x4*)(hi + o), l = *(const bf16x4*)(lo + o);
        if (m < NP && !(PNL_ABLATE & 4)) *(f32x4*)(dst + m * ld + col0 + 4 * lane) = pnl_pack_hl(h, l);
    }
}

template <int MT, int D>
__global__ void __launch_bounds__(PNL_WS_THREADS) k_mlp_fwd_ws(MlpFwdArgs a) {
    CFFM_DYN_SMEM(smem);
    bf16* P = (bf16*)smem;                              // ao panel, later the z2 panel
    bf16* ACT = P + 2 * PNL_IMG(MT);                    // two hidden-chunk images
    float* OUT = (float*)(ACT + 4 * PNL_IMG(MT));       // two fp32 tiles [16 MT][256]
    float* red = OUT + 16 * MT * 256;                   // [2][8 waves][16 MT]: inside OUT1 (first written by the first hidden chunk)
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const long m0 = (long)blockIdx.x * 16 * MT;
    const int NP = a.NP;
    if (wave >= PNL_WAVES) {
        // ---------------------------------------------------------------- store waves: same barriers, copies in between
        const int sw = wave - PNL_WAVES;
        pnl_lds_barrier();      // B0 ao staged
        pnl_lds_barrier();      // B1 row sums
        pnl_lds_barrier();      // B2 row square sums
        pnl_lds_barrier();      // B3 x1 tile + z2 image complete
        pnl_flush_f32<MT>(OUT, a.x1, 256, 0, m0, NP, sw, lane);
        pnl_flush_split4<MT>(P, P + PNL_IMG(MT), a.z2s, 256, 0, m0, NP, sw, lane);
        for (int c = 0; c < 4; ++c) {
            pnl_lds_barrier();  // B4+c: hraw tile + act image of chunk c complete
            pnl_flush_f32<MT>(OUT + ((c + 1) & 1) * 16 * MT * 256, a.hraw, 1024, 256 * c, m0, NP, sw, lane);
            const bf16* Ah = ACT + (c & 1) * 2 * PNL_IMG(MT);
            if (a.acts) pnl_flush_split4<MT>(Ah, Ah + PNL_IMG(MT), a.acts, 1024, 256 * c, m0, NP, sw, lane);
        }
        pnl_lds_barrier();      // B8 x2 tile complete
        pnl_flush_f32<MT>(OUT + 16 * MT * 256, a.x2, 256, 0, m0, NP, sw, lane);
        return;
    }
    // -------------------------------------------------------------------- compute waves
    PnlStream sp, s1, s2;
    sp.rs = buf_make(a.wp, 256u * 256 * 4); sp.voff = lane * 16; sp.KS = 8;
    s1.rs = buf_make(a.w1, 1024u * 256 * 4); s1.voff = lane * 16; s1.KS = 8;
    s2.rs = buf_make(a.w2, 1024u * 256 * 4); s2.voff = lane * 16; s2.KS = 32;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool valid[MT];
    long mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long m = m0 + 16 * i + l15;
        valid[i] = m < NP;
        mrow[i] = m;
    }
    f32x4 xr[2][MT];     // the residual rows, requested first: consumed right behind the proj product
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            xr[t][i] = z4;
            if (valid[i]) {
                const long b = mrow[i] / a.rows_per_batch, rr = mrow[i] % a.rows_per_batch;
                xr[t][i] = *(const f32x4*)(a.xt + b * a.xt_bs + rr * 256 + 32 * wave + 16 * t + 4 * g);
            }
        }
    PnlRing<2, D> ring;
#pragma unroll
    for (int s = 0; s < D - 1; ++s) pnl_ring_load<2, D>(ring, s, sp, 2 * wave, s);
    pnl_pin_vmem();
    {
        PnlStage<MT> sr;
        pnl_stage_load<MT>(sr, buf_make(a.ao, (uint32_t)((long)NP * 256 * 4)), 256, (int)m0, 0, tid);
        pnl_stage_store<MT, false>(sr, P, P + PNL_IMG(MT), tid);
    }
    pnl_lds_barrier();          // B0
    f32x4 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[t][i] = z4;
    pnl_chunk_mma<MT, 2, 2, D>(acc, ring, P, P + PNL_IMG(MT), sp, 2 * wave, 0, l15, g, s1, 2 * wave, 0);
    f32x4 x1v[2][MT];
    float s[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) s[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + 4 * g;
        const f32x4 bv = *(const f32x4*)(a.bp + n);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4 v = acc[t][i] + bv + xr[t][i];
            x1v[t][i] = v;
            *(f32x4*)(OUT + pnl_tile_off(16 * i + l15, n)) = v;
            s[i] += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        s[i] = pnl_sum_g(s[i]);
        if (g == 0) red[wave * 16 * MT + 16 * i + l15] = s[i];
    }
    pnl_lds_barrier();          // B1
    float mu[MT], rs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < PNL_WAVES; ++w) tot += red[w * 16 * MT + 16 * i + l15];
        mu[i] = tot * (1.f / 256);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 d = x1v[t][i] - mu[i];
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        q = pnl_sum_g(q);
        if (g == 0) red[(PNL_WAVES + wave) * 16 * MT + 16 * i + l15] = q;
    }
    pnl_lds_barrier();          // B2
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < PNL_WAVES; ++w) tot += red[(PNL_WAVES + w) * 16 * MT + 16 * i + l15];
        rs[i] = 1.f / sqrtf(tot * (1.f / 256) + CFFM_LN_EPS);
        if (wave == 0 && g == 0 && PNL_ST(valid[i])) { a.mean2[mrow[i]] = mu[i]; a.rstd2[mrow[i]] = rs[i]; }   // (2 x 128 B per workgroup)
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + 4 * g;
        const f32x4 gm = *(const f32x4*)(a.g2 + n), be = *(const f32x4*)(a.be2 + n);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4 zv = (x1v[t][i] - mu[i]) * rs[i] * gm + be;
            bf16x4 h, l;
            split4(zv, h, l);
            pnl_img_put(P, P + PNL_IMG(MT), 16 * i + l15, n, h, l);
        }
    }
    pnl_lds_barrier();          // B3
    f32x4 acc2[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc2[t][i] = z4;
    for (int c = 0; c < 4; ++c) {
        f32x4 h1[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i) h1[t][i] = z4;
        pnl_chunk_mma<MT, 2, 2, D>(h1, ring, P, P + PNL_IMG(MT), s1, 16 * c + 2 * wave, 0, l15, g, s2, 2 * wave, 8 * c);
        bf16* Ah = ACT + (c & 1) * 2 * PNL_IMG(MT);
        float* Ot = OUT + ((c + 1) & 1) * 16 * MT * 256;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nl = 32 * wave + 16 * t + 4 * g;
            const f32x4 bv = *(const f32x4*)(a.b1 + 256 * c + nl);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f32x4 raw = h1[t][i];
                f32x4 av;
                for (int e = 0; e < 4; ++e) av[e] = gelu_erf(raw[e] + bv[e]);
                bf16x4 h, l;
                split4(av, h, l);
                pnl_img_put(Ah, Ah + PNL_IMG(MT), 16 * i + l15, nl, h, l);
                *(f32x4*)(Ot + pnl_tile_off(16 * i + l15, nl)) = raw;
            }
        }
        pnl_lds_barrier();      // B4+c
        const int cn = c < 3 ? c + 1 : 0;
        pnl_chunk_mma<MT, 2, 2, D>(acc2, ring, Ah, Ah + PNL_IMG(MT), s2, 2 * wave, 8 * c, l15, g, s1, 16 * cn + 2 * wave, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + 4 * g;
        const f32x4 bv = *(const f32x4*)(a.b2 + n);
#pragma unroll
        for (int i = 0; i < MT; ++i) *(f32x4*)(OUT + 16 * MT * 256 + pnl_tile_off(16 * i + l15, n)) = x1v[t][i] + acc2[t][i] + bv;
    }
    pnl_lds_barrier();          // B8
}

template <int MT, int D>
__global__ void __launch_bounds__(PNL_WS_THREADS) k_mlp_bwd_ws(MlpBwdArgs a) {
    CFFM_DYN_SMEM(smem);
    bf16* P = (bf16*)smem;                              // dout panel, later the dx1 panel
    bf16* DH = P + 2 * PNL_IMG(MT);                     // two hidden-chunk images of dh; the first one ends as the fp32 tile of dao
    float* OUT = (float*)(DH + 4 * PNL_IMG(MT));        // fp32 tile [16 MT][256]: dx1
    float* red = OUT + 16 * MT * 256;                   // [2][8 waves][16 MT]
    float* recb = red + 2 * PNL_WAVES * 16 * MT;        // [2][256]: column sums of dh of a hidden chunk (double-buffered)
    float* recl = recb + 2 * 256;                       // [4][256]: dgamma2 | dbeta2 | colsum(dout) | colsum(dx1)
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const long m0 = (long)blockIdx.x * 16 * MT;
    const int NP = a.NP;
    if (wave >= PNL_WAVES) {
        const int sw = wave - PNL_WAVES;
        pnl_lds_barrier();      // B0 dout staged
        for (int c = 0; c < 4; ++c) {
            pnl_lds_barrier();  // B1+c: dh image + column sums of chunk c complete
            const bf16* Ah = DH + (c & 1) * 2 * PNL_IMG(MT);
            pnl_flush_split4<MT>(Ah, Ah + PNL_IMG(MT), a.dhs, 1024, 256 * c, m0, NP, sw, lane);
            if (sw == 0 && !(PNL_ABLATE & 4)) *(f32x4*)(a.rec_b1 + (long)blockIdx.x * 1024 + 256 * c + 4 * lane) = *(const f32x4*)(recb + (c & 1) * 256 + 4 * lane);
        }
        pnl_lds_barrier();      // B5 row sums of the LayerNorm backward
        pnl_lds_barrier();      // B6 dx1 tile + norm / bias records complete
        pnl_flush_f32<MT>(OUT, a.dx1, 256, 0, m0, NP, sw, lane);
        if (!(PNL_ABLATE & 4)) *(f32x4*)(a.rec_ln + (long)blockIdx.x * 1024 + 256 * sw + 4 * lane) = *(const f32x4*)(recl + 256 * sw + 4 * lane);
        pnl_lds_barrier();      // B7 dao tile complete
        pnl_flush_f32<MT>((const float*)DH, a.dao, 256, 0, m0, NP, sw, lane);
        return;
    }
    PnlStream s2, s1, sp;
    s2.rs = buf_make(a.w2n, 1024u * 256 * 4); s2.voff = lane * 16; s2.KS = 8;    // out 1024 hidden, contraction 256
    s1.rs = buf_make(a.w1n, 1024u * 256 * 4); s1.voff = lane * 16; s1.KS = 32;   // out 256, contraction 1024
    sp.rs = buf_make(a.wpn, 256u * 256 * 4); sp.voff = lane * 16; sp.KS = 8;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool valid[MT];
    long mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long m = m0 + 16 * i + l15;
        valid[i] = m < NP;
        mrow[i] = m;
    }
    PnlRing<2, D> ring;
#pragma unroll
    for (int s = 0; s < D - 1; ++s) pnl_ring_load<2, D>(ring, s, s2, 2 * wave, s);
    pnl_pin_vmem();
    {
        PnlStage<MT> sr;
        pnl_stage_load<MT>(sr, buf_make(a.dout, (uint32_t)((long)NP * 256 * 4)), 256, (int)m0, 0, tid);
        pnl_stage_store<MT, false>(sr, P, P + PNL_IMG(MT), tid);
    }
    pnl_lds_barrier();          // B0
    f32x4 dz[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) dz[t][i] = z4;
    f32x4 xv[2][MT], dr[2][MT];
    float mu[MT], rs[MT];
    for (int c = 0; c < 4; ++c) {
        f32x4 da[2][MT], hr[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                da[t][i] = z4;
                hr[t][i] = valid[i] ? *(const f32x4*)(a.hraw + mrow[i] * 1024 + 256 * c + 32 * wave + 16 * t + 4 * g) : z4;
            }
        pnl_pin_vmem();
        pnl_chunk_mma<MT, 2, 2, D>(da, ring, P, P + PNL_IMG(MT), s2, 16 * c + 2 * wave, 0, l15, g, s1, 2 * wave, 8 * c);
        bf16* Ah = DH + (c & 1) * 2 * PNL_IMG(MT);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nl = 32 * wave + 16 * t + 4 * g;
            const f32x4 bv = *(const f32x4*)(a.b1 + 256 * c + nl);
            f32x4 cs = z4;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 dh = z4;
                if (valid[i]) {
                    const f32x4 hv = hr[t][i] + bv;
                    for (int e = 0; e < 4; ++e) dh[e] = da[t][i][e] * gelu_erf_grad(hv[e]);
                }
                bf16x4 h, l;
                split4(dh, h, l);
                pnl_img_put(Ah, Ah + PNL_IMG(MT), 16 * i + l15, nl, h, l);
                cs += dh;
            }
            for (int e = 0; e < 4; ++e) cs[e] = row16_sum(cs[e]);
            if (l15 == 0) *(f32x4*)(recb + (c & 1) * 256 + nl) = cs;
        }
        pnl_lds_barrier();      // B1+c
        const bool more = c < 3;
        pnl_chunk_mma<MT, 2, 2, D>(dz, ring, Ah, Ah + PNL_IMG(MT), s1, 2 * wave, 8 * c, l15, g, more ? s2 : sp, more ? 16 * (c + 1) + 2 * wave : 2 * wave, 0);
    }
    // ---- LayerNorm backward + residual (its rows are requested here: holding them across the last product would cost the third wave
    // per SIMD -- 168 VGPRs is the budget)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        mu[i] = valid[i] ? a.mean2[mrow[i]] : 0.f;
        rs[i] = valid[i] ? a.rstd2[mrow[i]] : 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * wave + 16 * t + 4 * g;
            xv[t][i] = valid[i] ? *(const f32x4*)(a.x1 + mrow[i] * 256 + n) : z4;
            dr[t][i] = valid[i] ? *(const f32x4*)(a.dout + mrow[i] * 256 + n) : z4;
        }
    }
    f32x4 xh[2][MT], gz[2][MT];
    float m1[MT], m2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) m1[i] = m2[i] = 0.f;
    f32x4 ag[2], ab[2], ar[2], ax[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + 4 * g;
        const f32x4 gm = *(const f32x4*)(a.g2 + n);
        ag[t] = ab[t] = ar[t] = ax[t] = z4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            xh[t][i] = (xv[t][i] - mu[i]) * rs[i];
            gz[t][i] = dz[t][i] * gm;
            ag[t] += dz[t][i] * xh[t][i];
            ab[t] += dz[t][i];
            ar[t] += dr[t][i];
            const f32x4 p = gz[t][i] * xh[t][i];
            m1[i] += (gz[t][i][0] + gz[t][i][1]) + (gz[t][i][2] + gz[t][i][3]);
            m2[i] += (p[0] + p[1]) + (p[2] + p[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        m1[i] = pnl_sum_g(m1[i]);
        m2[i] = pnl_sum_g(m2[i]);
        if (g == 0) {
            red[wave * 16 * MT + 16 * i + l15] = m1[i];
            red[(PNL_WAVES + wave) * 16 * MT + 16 * i + l15] = m2[i];
        }
    }
    pnl_lds_barrier();          // B5
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < PNL_WAVES; ++w) {
            t1 += red[w * 16 * MT + 16 * i + l15];
            t2 += red[(PNL_WAVES + w) * 16 * MT + 16 * i + l15];
        }
        m1[i] = t1 * (1.f / 256);
        m2[i] = t2 * (1.f / 256);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + 4 * g;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4 dxv = (gz[t][i] - m1[i] - xh[t][i] * m2[i]) * rs[i] + dr[t][i];
            ax[t] += dxv;
            bf16x4 h, l;
            split4(dxv, h, l);
            pnl_img_put(P, P + PNL_IMG(MT), 16 * i + l15, n, h, l);     // (the dout image is dead since the last hidden chunk)
            *(f32x4*)(OUT + pnl_tile_off(16 * i + l15, n)) = dxv;
        }
        for (int e = 0; e < 4; ++e) {
            ag[t][e] = row16_sum(ag[t][e]);
            ab[t][e] = row16_sum(ab[t][e]);
            ar[t][e] = row16_sum(ar[t][e]);
            ax[t][e] = row16_sum(ax[t][e]);
        }
        if (l15 == 0) {
            *(f32x4*)(recl + n) = ag[t];
            *(f32x4*)(recl + 256 + n) = ab[t];
            *(f32x4*)(recl + 512 + n) = ar[t];
            *(f32x4*)(recl + 768 + n) = ax[t];
        }
    }
    pnl_lds_barrier();          // B6
    f32x4 dq[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) dq[t][i] = z4;
    pnl_chunk_mma<MT, 2, 2, D>(dq, ring, P, P + PNL_IMG(MT), sp, 2 * wave, 0, l15, g, sp, 2 * wave, 8);
    // dao -> the first dh image's space as an fp32 tile (last read by the products of hidden chunk 2: two barriers ago)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i) *(f32x4*)((float*)DH + pnl_tile_off(16 * i + l15, 32 * wave + 16 * t + 4 * g)) = dq[t][i];
    pnl_lds_barrier();          // B7
}
#endif  // CFFM_EXPERIMENTS
