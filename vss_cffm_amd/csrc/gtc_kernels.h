// gtc_kernels.h -- CFFM++ global temporal context: per-token cross-attention against K prototype
// ("cluster centre") tokens.  Reference: WindowAttention_cluster.forward,
// pvt/swin_transformer_2d.py:208-262 with only_use_cluster_center_as_context=True (:216): no in-window
// keys, no position bias, no mask, so window partition / padding are numerically no-ops
// (SURVEY.md A "GTC is window-independent") and the op is plain [token, head] x K attention.
// Two forms: K <= 128 (the reference's default is 100) on the matrix pipe -- three-pass bf16 hi / lo products, so the block
// keeps its 1e-4 tolerance: k_gtc_pack_frags + k_gtc_attn_fwd_mfma / _bwd_dq_mfma / _bwd_dkv_mfma below -- and, above that,
// fp32 on the VALU: a lane owns a token, the K centre rows of the head sit in LDS, softmax is online (k_gtc_attn_fwd / _bwd).
#pragma once
#include "cffm_common.h"
#include "gemm_kernels.h"      // mfma16x16x32_bf16, split4

#define GTB_QS_ 36          // floats per row of the q / dO / o tiles in LDS (32 + 4: 16-byte aligned rows on different banks)

// plain row LayerNorm: z = LN(x) (+ stats)
// split: z in split-4 storage (cffm_common.h) -- the A operand of a row-panel GEMM
__global__ void __launch_bounds__(256) k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ z,
                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out, long nrows, int split = 0) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const f32x4 v = *(const f32x4*)(x + row * CFFM_C + 4 * lane);
    const float mu = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / CFFM_C);
    const f32x4 d = v - mu;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / CFFM_C);
    const float rs = 1.f / sqrtf(var + CFFM_LN_EPS);
    const f32x4 zv = d * rs * *(const f32x4*)(gamma + 4 * lane) + *(const f32x4*)(beta + 4 * lane);
    *(f32x4*)(z + row * CFFM_C + 4 * lane) = split ? split4_pack(zv) : zv;
    if (lane == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
}

// Forward (round 5 form): a LANE owns a token -- its q head slice (32 floats) and its output accumulator live in registers, the
// prototype rows are broadcast LDS reads, so s = q . Kc[k] and o += p Vc[k] are 32 + 32 register FMAs per key with NO cross-lane
// reduction; softmax is online (running maximum per lane).  The q / o rows travel through an LDS tile so that global accesses are
// 16-byte pieces of whole 128-byte head slices (8 threads per token row).  Rounds 1-4: 8 lanes per token, three ds_bpermute per key.
// grid (ceil(T/256), 8, B), 256 threads; dynamic LDS gtf_lds(K).  q_raw [B*T,256] (q = (q_raw + bq) * scale), kv_raw [B*K,512],
// o [B*T,256], lse [B*T,8]
#define GTC_TOK 256  // tokens per workgroup
__host__ __device__ constexpr int gtf_lds(int K) { return 4 * (2 * K * CFFM_HD + GTC_TOK * GTB_QS_); }
__global__ void __launch_bounds__(256) k_gtc_attn_fwd(const float* __restrict__ q_raw, const float* __restrict__ bq,
                                                       const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                       float* __restrict__ o, float* __restrict__ lse, int T, int K) {
    CFFM_DYN_SMEM(smem);
    float* Kc = (float*)smem;       // [K][32]
    float* Vc = Kc + K * CFFM_HD;   // [K][32]
    float* Qs = Vc + K * CFFM_HD;   // [256][GTB_QS_]: q rows in, o rows out
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int t0 = blockIdx.x * GTC_TOK;
    const float scale = 0.17677669529663687f;
    for (int e = tid; e < K * CFFM_HD; e += 256) {
        const int k = e >> 5, d = e & 31;
        const float* row = kv_raw + ((long)b * K + k) * 512 + h * CFFM_HD + d;
        Kc[e] = row[0] + bkv[h * CFFM_HD + d];
        Vc[e] = row[256] + bkv[256 + h * CFFM_HD + d];
    }
    for (int e = tid; e < GTC_TOK * 8; e += 256) {
        const int tl = e >> 3, c = e & 7, t = t0 + tl;
        f32x4 qv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t < T) qv = (*(const f32x4*)(q_raw + ((long)b * T + t) * CFFM_C + h * CFFM_HD + 4 * c) + *(const f32x4*)(bq + h * CFFM_HD + 4 * c)) * scale;
        *(f32x4*)(Qs + tl * GTB_QS_ + 4 * c) = qv;
    }
    __syncthreads();
    float q[32], acc[32];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const f32x4 qv = *(const f32x4*)(Qs + tid * GTB_QS_ + 4 * d4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[4 * d4 + e] = qv[e]; acc[4 * d4 + e] = 0.f; }
    }
    float m = -INFINITY, l = 0.f;
    for (int k = 0; k < K; ++k) {
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};      // four independent partial sums: one 32-long FMA chain would run at the FMA latency
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 kc = *(const f32x4*)(Kc + k * CFFM_HD + 4 * d4);      // (broadcast read)
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] += q[4 * d4 + e] * kc[e];
        }
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn), p = expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 vc = *(const f32x4*)(Vc + k * CFFM_HD + 4 * d4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * d4 + e] = acc[4 * d4 + e] * alpha + p * vc[e];
        }
        m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4)      // (a thread overwrites only its own row)
        *(f32x4*)(Qs + tid * GTB_QS_ + 4 * d4) = (f32x4){acc[4 * d4] * inv, acc[4 * d4 + 1] * inv, acc[4 * d4 + 2] * inv, acc[4 * d4 + 3] * inv};
    if (t0 + tid < T) lse[((long)b * T + t0 + tid) * CFFM_HEADS + h] = m + logf(l);
    __syncthreads();
    for (int e = tid; e < GTC_TOK * 8; e += 256) {
        const int tl = e >> 3, c = e & 7, t = t0 + tl;
        if (t < T) *(f32x4*)(o + ((long)b * T + t) * CFFM_C + h * CFFM_HD + 4 * c) = *(const f32x4*)(Qs + tl * GTB_QS_ + 4 * c);
    }
}

// The prototypes of a (clip, head) as MFMA fragments, made ONCE per call (k_gtc_pack_frags) instead of once per wave: every wave of the three
// matrix-pipe kernels below needs the same 16-64 KB of bf16 hi / lo fragments of Kc / Vc, and building them in place -- scalar LDS reads of a
// row-major copy, conversions, for the transposed forms one 2-byte read per element -- cost more than the products they feed (the forward
// spent half of its 17.7 us at K = 100 there).  Layout per (clip, head), 16-byte words, unit = 64 lanes:
//   KF  [KT tiles][hi | lo]      lane (l15, g): Kc[16 t + l15][8 g .. 8 g + 7]                      (A / B operand with keys as rows / columns)
//   VF  [KT tiles][hi | lo]      the same of Vc
//   KTF [2 mt][U][hi | lo]       lane (l15, g): Kc[key(u, g, j)][16 mt + l15], j = 0..7, key(u, g, j) = 32 u + 16 (j >> 2) + 4 g + (j & 3)
//   VTF [2 mt][U][hi | lo]       the same of Vc                                                     (operands with channels as rows, k-slots = keys)
// = 16 U units of 1 KiB per (clip, head): KF at unit 0, VF at 4 U, KTF at 8 U, VTF at 12 U.
__host__ __device__ constexpr int gtm_frag_units(int U) { return 16 * U; }
// lane `lane` of unit pair `pu` of (clip b, head h): pair units [0, 2U) KF, [2U, 4U) VF, [4U, 6U) KTF, [6U, 8U) VTF; straight from kv_raw (8 loads)
template <int U>
__device__ __forceinline__ void gtm_pack_item(const float* __restrict__ kv_raw, const float* __restrict__ bkv, int b, int h, int K, int pu, int lane,
                                              bf16x8& hi, bf16x8& lo) {
    const int l15 = lane & 15, g = lane >> 4;
    const int grp = pu / (2 * U), idx = pu % (2 * U), vofs = (grp & 1) ? 256 : 0;
    const float* base = kv_raw + (long)b * K * 512 + vofs + h * CFFM_HD;
    const float* bias = bkv + vofs + h * CFFM_HD;
    if (grp < 2) {                       // row fragments: 8 consecutive channels of key 16 idx + l15
        const int key = 16 * idx + l15;
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, c = a;
        if (key < K) {
            a = *(const f32x4*)(base + (long)key * 512 + 8 * g) + *(const f32x4*)(bias + 8 * g);
            c = *(const f32x4*)(base + (long)key * 512 + 8 * g + 4) + *(const f32x4*)(bias + 8 * g + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = (bf16)a[j]; lo[j] = (bf16)(a[j] - (float)hi[j]);
            hi[4 + j] = (bf16)c[j]; lo[4 + j] = (bf16)(c[j] - (float)hi[4 + j]);
        }
    } else {                             // transposed fragments: channel 16 mt + l15 of the 8 keys of k-slots (g, j)
        const int mt = idx / U, u = idx % U, ch = 16 * mt + l15;
        const float bv = bias[ch];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = 32 * u + 16 * (j >> 2) + 4 * g + (j & 3);
            const float x = key < K ? base[(long)key * 512 + ch] + bv : 0.f;
            hi[j] = (bf16)x;
            lo[j] = (bf16)(x - (float)hi[j]);
        }
    }
}
// grid (8 heads, B, 2 U): a workgroup makes four {hi, lo} unit pairs, a thread one lane of one pair
template <int U>
__global__ void __launch_bounds__(256) k_gtc_pack_frags(const float* __restrict__ kv_raw, const float* __restrict__ bkv, f32x4* __restrict__ frags, int K) {
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, pu = 4 * blockIdx.z + (tid >> 6);
    bf16x8 hi, lo;
    gtm_pack_item<U>(kv_raw, bkv, b, h, K, pu, lane, hi, lo);
    f32x4* out = frags + (long)(b * CFFM_HEADS + h) * gtm_frag_units(U) * 64;
    out[(2 * pu) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
    out[(2 * pu + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
}
// The same inside a kernel, into LDS (frags == NULL): for U = 1 (K <= 32: 16 units, 4 items per thread) the workgroup makes its own copy --
// two packing launches per step cost more there than they save (gtc_step at K = 8: 0.262 ms with fragments built in the kernels, 0.267-0.273
// with the packing kernel); from U = 2 on the packing kernel wins (K = 100: 0.297-0.300 against 0.287-0.290).  Ends with a barrier.
template <int U>
__device__ __forceinline__ void gtm_pack_local(f32x4* lds, const float* __restrict__ kv_raw, const float* __restrict__ bkv, int b, int h, int K, int tid) {
#pragma unroll
    for (int q = 0; q < 2 * U; ++q) {
        const int item = tid + 256 * q, pu = item >> 6, lane = item & 63;
        bf16x8 hi, lo;
        gtm_pack_item<U>(kv_raw, bkv, b, h, K, pu, lane, hi, lo);
        lds[(2 * pu) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
        lds[(2 * pu + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
    }
    __syncthreads();
}
#define GTM_LOCAL_UNITS(U) ((U) == 1 ? 16 * 64 : 1)      // f32x4 words of the local copy (U = 1 only)

// Forward on the matrix pipe (round 5, third session; K <= 128).  49 * K * 32 MACs per token-head are little, but the VALU form spends them
// at 16 broadcast LDS reads per key and wave (LDS-bound: 41-53 us at the reference's K = 100, cffm_head.py:217).  Here a WAVE owns 16 tokens
// of one head at a time and the prototypes never leave its registers:
//   S^T [keys x tokens] = Kc Q^T        A = Kc fragments (lane (key l15, channels 8 g ..)), loaded ONCE per wave; B = Q^T fragments straight
//                                       from the lane's own q row (token l15, channels 8 g ..: two 16-byte loads);
//   softmax over the keys of a token = over the 4 C registers x KT tiles of a lane, then two shuffles across the four lane groups;
//   O^T [channels x tokens] = Vc^T P^T  A = Vc^T fragments (loaded once per wave), B = P^T from the C
//                                       registers of two key tiles (k-slot (g, j) <-> key 32 u + 16 (j >> 2) + 4 g + (j & 3), the bijection
//                                       of cfm_attn_kernels.h), so P never leaves the registers.
// Every product is the three-pass bf16 hi / lo split of the Linear GEMMs (hi x lo + lo x hi + hi x hi, error ~2^-17): the block keeps its
// 1e-4 tolerance -- no f16 operands here.  Keys are padded to 32 U (U = k-steps of the PV product, template parameter); padded keys carry
// s = -inf.  grid (workgroups, 8 heads, B); a workgroup's four waves walk `tiles_per_wave` consecutive 16-token tiles each.
template <int U>
__global__ void __launch_bounds__(256) k_gtc_attn_fwd_mfma(const float* __restrict__ q_raw, const float* __restrict__ bq, const f32x4* __restrict__ frags,
                                                            const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                            float* __restrict__ o, float* __restrict__ lse, int T, int K, int tiles_per_wave) {
    constexpr int KT = 2 * U;
    __shared__ f32x4 sfr[GTM_LOCAL_UNITS(U)];
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f;
    // the wave's constant fragments (k_gtc_pack_frags, or the workgroup's own copy when frags == NULL): 8 U coalesced 16-byte loads per lane
    const f32x4* fr = frags + (long)(b * CFFM_HEADS + h) * gtm_frag_units(U) * 64 + lane;
    if (U == 1 && !frags) {
        gtm_pack_local<U>(sfr, kv_raw, bkv, b, h, K, tid);
        fr = sfr + lane;
    }
    bf16x8 kh[KT], kl[KT], vh[2][U], vl[2][U];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        kh[t] = __builtin_bit_cast(bf16x8, fr[(2 * t) * 64]);
        kl[t] = __builtin_bit_cast(bf16x8, fr[(2 * t + 1) * 64]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vh[mt][u] = __builtin_bit_cast(bf16x8, fr[(12 * U + 2 * (mt * U + u)) * 64]);
            vl[mt][u] = __builtin_bit_cast(bf16x8, fr[(12 * U + 2 * (mt * U + u) + 1) * 64]);
        }
    const f32x4 bq0 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g), bq1 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g + 4);
    const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
    // the lane's q row of a tile (raw): requested a tile ahead, so that a wave's run of tiles does not pay one memory round trip per tile
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 n0 = z4, n1 = z4;
    {
        const int t = 16 * tile0 + l15;
        if (t < T) {
            const float* qp = q_raw + ((long)b * T + t) * CFFM_C + h * CFFM_HD + 8 * g;
            n0 = *(const f32x4*)qp; n1 = *(const f32x4*)(qp + 4);
        }
    }
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int t0 = 16 * (tile0 + it);
        if (t0 >= T) break;                              // (wave-uniform)
        const int t = t0 + l15;
        const bool live = t < T;
        const long row = (long)b * T + (live ? t : 0);
        const f32x4 q0 = live ? (n0 + bq0) * scale : z4, q1 = live ? (n1 + bq1) * scale : z4;
        if (it + 1 < tiles_per_wave && t + 16 < T) {
            const float* qp = q_raw + (row + 16) * CFFM_C + h * CFFM_HD + 8 * g;
            n0 = *(const f32x4*)qp; n1 = *(const f32x4*)(qp + 4);
        }
        bf16x8 qh, ql;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qh[e] = (bf16)q0[e]; ql[e] = (bf16)(q0[e] - (float)qh[e]);
            qh[4 + e] = (bf16)q1[e]; ql[4 + e] = (bf16)(q1[e] - (float)qh[4 + e]);
        }
        // S^T tiles: lane (token l15) holds keys 16 t + 4 g + r
        f32x4 sv[KT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
            c = mfma16x16x32_bf16(kh[kt], ql, c);
            c = mfma16x16x32_bf16(kl[kt], qh, c);
            c = mfma16x16x32_bf16(kh[kt], qh, c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * g + r >= K) c[r] = -INFINITY;
                m = fmaxf(m, c[r]);
            }
            sv[kt] = c;
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = fast_exp2((sv[kt][r] - m) * 1.4426950408889634f);     // one v_exp_f32 (a padded key: 2^-inf = 0; key 0 always exists, so m is finite)
                sv[kt][r] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        f32x4 ov[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = sv[2 * u][e], c = sv[2 * u + 1][e];
                ph[e] = (bf16)a; pl[e] = (bf16)(a - (float)ph[e]);
                ph[4 + e] = (bf16)c; pl[4 + e] = (bf16)(c - (float)ph[4 + e]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ov[mt] = mfma16x16x32_bf16(vh[mt][u], pl, ov[mt]);
                ov[mt] = mfma16x16x32_bf16(vl[mt][u], ph, ov[mt]);
                ov[mt] = mfma16x16x32_bf16(vh[mt][u], ph, ov[mt]);
            }
        }
        // lane: channels 16 mt + 4 g .. + 3 of token l15
        if (live) {
            const float inv = 1.f / l;
            float* orow = o + row * CFFM_C + h * CFFM_HD + 4 * g;
            *(f32x4*)(orow) = ov[0] * inv;
            *(f32x4*)(orow + 16) = ov[1] * inv;
            if (g == 0) lse[row * CFFM_HEADS + h] = m + logf(l);
        }
    }
}

// Backward (round 5).  dq is per token, but dKc / dVc contract over ALL tokens of a clip: rounds 1-4 did that with one LDS atomic
// per (token, key, channel) plus global atomics -- 180 us at K = 8 and 2.1 ms at K = 100 for 2 x 3600 tokens, and not deterministic.
// Now a workgroup = (clip b, head h, GTB_CHUNKS chunks of 64 tokens), two phases per chunk:
//   1. a LANE owns a token (its q and dO head slices in registers, read from LDS tiles that were loaded coalesced), the four waves
//      split the keys: s, dp by 32-term register dot products against the broadcast prototype rows -- no cross-lane reduction at all --
//      then p = exp(s - lse), ds = p (dp - D); P and dS tiles go to LDS, the waves' partial dq rows are summed through LDS;
//   2. the thread that owns a 4-key x 4-channel block of dKc (= dS^T q) or dVc (= P^T dO) accumulates it over the chunk's tokens
//      from the tiles (two 16-byte LDS reads per 16 FMAs), in registers across the workgroup's chunks.
// Every workgroup leaves one record [K][64] (dKc | dVc rows, unscaled); k_gtc_dkv_sum adds the records of a (clip, head) in a fixed
// order: deterministic, no atomics.
#define GTB_CHUNKS 2        // chunks per workgroup; a chunk = `tok` tokens, one per lane: 64, or 32 when the P / dS tiles of 64 would not fit the LDS (K > 128)
#define GTB_QS GTB_QS_
__host__ __device__ constexpr int gtb_kp(int K) { return (K + 3) / 4 * 4; }             // keys padded to whole 4-blocks
__host__ __device__ constexpr int gtb_ps(int K) { return gtb_kp(K) + 4; }               // floats per row of the P / dS tiles
__host__ __device__ constexpr int gtb_tok(int K) { return K > 128 ? 32 : 64; }
__host__ __device__ constexpr int gtb_lds(int K) {
    return 4 * (2 * gtb_kp(K) * CFFM_HD + 2 * gtb_tok(K) * GTB_QS + 2 * gtb_tok(K) * gtb_ps(K) + 4 * gtb_tok(K) * 33);
}
// grid (ceil(T / (gtb_tok(K) GTB_CHUNKS)), 8, B), 256 threads.  rec [B][8][gridDim.x][K][64]
__global__ void __launch_bounds__(256) k_gtc_attn_bwd(const float* __restrict__ q_raw, const float* __restrict__ bq,
                                                       const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                       const float* __restrict__ o, const float* __restrict__ dout,
                                                       const float* __restrict__ lse, float* __restrict__ dq_raw,
                                                       float* __restrict__ rec, int T, int K) {
    CFFM_DYN_SMEM(smem);
    const int KP = gtb_kp(K), PS = gtb_ps(K), GTB_TOK = gtb_tok(K);
    float* Kc = (float*)smem;                 // [KP][32] (rows >= K zero)
    float* Vc = Kc + KP * CFFM_HD;
    float* Qs = Vc + KP * CFFM_HD;            // [64][GTB_QS]  q (scaled, with bias)
    float* Gs = Qs + GTB_TOK * GTB_QS;        // [64][GTB_QS]  dO
    float* Ps = Gs + GTB_TOK * GTB_QS;        // [64][PS]
    float* Ds = Ps + GTB_TOK * PS;            // [64][PS]      dS
    float* Rq = Ds + GTB_TOK * PS;            // [4 waves][64][33]  partial dq rows
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = 0.17677669529663687f;
    for (int e = tid; e < KP * CFFM_HD; e += 256) {
        const int k = e >> 5, d = e & 31;
        float kc = 0.f, vc = 0.f;
        if (k < K) {
            const float* row = kv_raw + ((long)b * K + k) * 512 + h * CFFM_HD + d;
            kc = row[0] + bkv[h * CFFM_HD + d];
            vc = row[256] + bkv[256 + h * CFFM_HD + d];
        }
        Kc[e] = kc; Vc[e] = vc;
    }
    // the 4 x 4 blocks of dKc (blocks [0, NB)) and dVc ([NB, 2 NB)) this thread owns: block = (key quad kq, channel quad dq4)
    const int NB = (KP / 4) * 8;
    f32x4 accb[4][4];                         // [owned block: K <= 256 gives at most 1024 blocks = 4 per thread][key in block] x 4 channels
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) accb[j][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < GTB_CHUNKS; ++ch) {
        const int t0 = (blockIdx.x * GTB_CHUNKS + ch) * GTB_TOK;
        if (t0 >= T) break;                   // (uniform)
        __syncthreads();                      // the previous chunk's phase 2 is done with the tiles
        // ---- tiles: q, dO (16-byte pieces, 8 threads per token row) and D = sum_d dO * O
        for (int e = tid; e < GTB_TOK * 8; e += 256) {
            const int tl = e >> 3, c = e & 7, t = t0 + tl;
            f32x4 qv = (f32x4){0.f, 0.f, 0.f, 0.f}, gv = qv;
            if (t < T) {
                const long row = (long)b * T + t;
                qv = (*(const f32x4*)(q_raw + row * CFFM_C + h * CFFM_HD + 4 * c) + *(const f32x4*)(bq + h * CFFM_HD + 4 * c)) * scale;
                gv = *(const f32x4*)(dout + row * CFFM_C + h * CFFM_HD + 4 * c);
            }
            *(f32x4*)(Qs + tl * GTB_QS + 4 * c) = qv;
            *(f32x4*)(Gs + tl * GTB_QS + 4 * c) = gv;
        }
        __syncthreads();
        // ---- phase 1: lane = token, wave = every fourth key
        {
            const int t = t0 + lane;
            const bool live = t < T && lane < GTB_TOK;
            const long row = (long)b * T + (live ? t : 0);
            float q[32], g[32], dq[32];
            float D = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 8; ++d4) {
                const int lq = lane < GTB_TOK ? lane : 0;
                const f32x4 qv = *(const f32x4*)(Qs + lq * GTB_QS + 4 * d4), gv = *(const f32x4*)(Gs + lq * GTB_QS + 4 * d4);
                const f32x4 ov = live ? *(const f32x4*)(o + row * CFFM_C + h * CFFM_HD + 4 * d4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) { q[4 * d4 + e] = qv[e]; g[4 * d4 + e] = gv[e]; dq[4 * d4 + e] = 0.f; D += gv[e] * ov[e]; }
            }
            const float ls = live ? lse[row * CFFM_HEADS + h] : 0.f;
            const int ln = lane < GTB_TOK ? lane : 0;          // (tok = 32: the upper half of the wave idles)
            for (int k = wave; k < KP; k += 4) {
                f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f}, p4 = s4;      // (independent partial sums, as in the forward)
#pragma unroll
                for (int d4 = 0; d4 < 8; ++d4) {
                    const f32x4 kc = *(const f32x4*)(Kc + k * CFFM_HD + 4 * d4), vc = *(const f32x4*)(Vc + k * CFFM_HD + 4 * d4);   // (broadcast reads)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s4[e] += q[4 * d4 + e] * kc[e]; p4[e] += g[4 * d4 + e] * vc[e]; }
                }
                const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]), dp = (p4[0] + p4[1]) + (p4[2] + p4[3]);
                const float p = (live && k < K) ? expf(s - ls) : 0.f;
                const float ds = p * (dp - D);
                if (lane < GTB_TOK) { Ps[ln * PS + k] = p; Ds[ln * PS + k] = ds; }
#pragma unroll
                for (int d4 = 0; d4 < 8; ++d4) {
                    const f32x4 kc = *(const f32x4*)(Kc + k * CFFM_HD + 4 * d4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[4 * d4 + e] += ds * kc[e];
                }
            }
#pragma unroll
            for (int d = 0; d < 32; ++d)
                if (lane < GTB_TOK) Rq[(wave * GTB_TOK + lane) * 33 + d] = dq[d];
        }
        __syncthreads();
        // dq rows: the four waves' partial rows added, 16-byte stores (8 threads per token row)
        for (int e = tid; e < GTB_TOK * 8; e += 256) {
            const int tl = e >> 3, c = e & 7, t = t0 + tl;
            if (t < T) {
                f32x4 v;
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    v[x] = (Rq[(0 * GTB_TOK + tl) * 33 + 4 * c + x] + Rq[(1 * GTB_TOK + tl) * 33 + 4 * c + x]) +
                           (Rq[(2 * GTB_TOK + tl) * 33 + 4 * c + x] + Rq[(3 * GTB_TOK + tl) * 33 + 4 * c + x]);
                *(f32x4*)(dq_raw + ((long)b * T + t) * CFFM_C + h * CFFM_HD + 4 * c) = v * scale;
            }
        }
        // ---- phase 2: dKc += dS^T q, dVc += P^T dO over the chunk's 64 tokens, 4 x 4 register blocks
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int blk = tid + 256 * j;
            if (blk >= 2 * NB) break;
            const bool isv = blk >= NB;
            const int bb = isv ? blk - NB : blk, kq = bb >> 3, d4 = bb & 7;
            const float* A = (isv ? Ps : Ds) + 4 * kq;
            const float* Bm = (isv ? Gs : Qs) + 4 * d4;
#pragma unroll 4
            for (int tl = 0; tl < GTB_TOK; ++tl) {
                const f32x4 a = *(const f32x4*)(A + tl * PS), bv = *(const f32x4*)(Bm + tl * GTB_QS);
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[j][r] += a[r] * bv;
            }
        }
    }
    // the record of this workgroup: rows k < K, 64 floats each (dKc 32 | dVc 32)
    float* out = rec + (((long)(b * CFFM_HEADS + h) * gridDim.x + blockIdx.x) * K) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int blk = tid + 256 * j;
        if (blk >= 2 * NB) break;
        const bool isv = blk >= NB;
        const int bb = isv ? blk - NB : blk, kq = bb >> 3, d4 = bb & 7;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kq + r < K) *(f32x4*)(out + (long)(4 * kq + r) * 64 + (isv ? 32 : 0) + 4 * d4) = accb[j][r];
    }
}
// Backward on the matrix pipe (K <= 128), two launches, every product the three-pass bf16 split of the forward:
//   k_gtc_attn_bwd_dq_mfma   a wave per 16-token tile, keys x tokens orientation as in the forward: S^T = Kc Q^T, dP^T = Vc dO^T (A = row fragments of
//                            the prototypes, read from an LDS copy of the packed fragments; B = the lane's own q / dO rows), p = exp(s - lse),
//                            ds = p (dp - D); dQ^T = Kc^T dS with dS as the B operand straight from the C registers (the k-slot <-> key bijection of
//                            the forward) and the Kc^T fragments in registers.  D = sum_keys p dp (= sum_ch dO o) comes out of the same products; it is left in
//                            Dbuf [B*T][8] for the second launch.
//   k_gtc_attn_bwd_dkv_mfma  a wave per PAIR of 16-token tiles, tokens x keys orientation (the operands swapped: S = Q Kc^T, dP = dO Vc^T -- the same
//                            fragments), so that p and ds of a key tile sit in the C registers as B operands of products that contract over the 32
//                            TOKENS: dVc^T += dO^T P, dKc^T += Q^T dS (A = transposed q / dO tiles out of a wave-private LDS copy).  The wave's
//                            accumulators (2 x 32 channels x K keys) live in registers over its run of pairs; the four waves of a workgroup are added
//                            through LDS in wave order and leave ONE record [K][64] (dKc | dVc) -- k_gtc_dkv_sum adds the records in order as before.
// Deterministic, no atomics.  (The VALU kernel above stays for K > 128.)
#define GTM_TLD 36                                                        // floats per row of the wave-private q / dO tiles
__host__ __device__ constexpr int gtm_bwd_lds(int U) { return 8 * U * 1024 + 4 * 2 * 32 * GTM_TLD * 4; }   // the KF | VF units + 4 waves x (q | dO) tiles (the record tiles reuse those)
// the row fragments KF | VF of the head (k_gtc_pack_frags units [0, 8 U)) copied into LDS: a wave then reads fragment (tile t, hi | lo) of Kc
// at unit 2 t (+ 1), of Vc at 4 U + 2 t (+ 1), lane-linear 16-byte reads
template <int U>
__device__ __forceinline__ void gtm_copy_row_frags(f32x4* lds, const f32x4* __restrict__ fr, int tid) {
#pragma unroll
    for (int q = 0; q < 2 * U; ++q) lds[tid + 256 * q] = fr[tid + 256 * q];
}
__device__ __forceinline__ void gtm_split8(f32x4 a, f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16)a[e]; lo[e] = (bf16)(a[e] - (float)hi[e]);
        hi[4 + e] = (bf16)b[e]; lo[4 + e] = (bf16)(b[e] - (float)hi[4 + e]);
    }
}
__device__ __forceinline__ f32x4 gtm_mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
    c = mfma16x16x32_bf16(ah, bl, c);
    c = mfma16x16x32_bf16(al, bh, c);
    return mfma16x16x32_bf16(ah, bh, c);
}

template <int U>
__global__ void __launch_bounds__(256) k_gtc_attn_bwd_dq_mfma(const float* __restrict__ q_raw, const float* __restrict__ bq, const f32x4* __restrict__ frags,
                                                               const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                               const float* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ dq_raw,
                                                               float* __restrict__ Dbuf, int T, int K, int tiles_per_wave) {
    constexpr int KT = 2 * U;
    CFFM_DYN_SMEM(smem);
    __shared__ f32x4 sfr[GTM_LOCAL_UNITS(U)];
    f32x4* RF = (f32x4*)smem;            // KF | VF units
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f;
    const f32x4* fr = frags + (long)(b * CFFM_HEADS + h) * gtm_frag_units(U) * 64;
    if (U == 1 && !frags) {
        gtm_pack_local<U>(sfr, kv_raw, bkv, b, h, K, tid);
        fr = sfr;
    }
    gtm_copy_row_frags<U>(RF, fr, tid);
    // Kc^T fragments: lane (channel 16 mt + l15, k-slot (g, j) <-> key 32 u + 16 (j >> 2) + 4 g + (j & 3))
    bf16x8 kth[2][U], ktl[2][U];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kth[mt][u] = __builtin_bit_cast(bf16x8, fr[(8 * U + 2 * (mt * U + u)) * 64 + lane]);
            ktl[mt][u] = __builtin_bit_cast(bf16x8, fr[(8 * U + 2 * (mt * U + u) + 1) * 64 + lane]);
        }
    __syncthreads();
    const f32x4 bq0 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g), bq1 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g + 4);
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
    f32x4 nq0 = z4, nq1 = z4, ng0 = z4, ng1 = z4;
    float nls = 0.f;
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int t0 = 16 * (tile0 + it);
        if (t0 >= T) break;                              // (wave-uniform)
        const int t = t0 + l15;
        const bool live = t < T;
        const long row = (long)b * T + (live ? t : 0);
        if (it == 0 && live) {                           // (later tiles: requested a tile ahead, below)
            const long base = row * CFFM_C + h * CFFM_HD + 8 * g;
            nq0 = *(const f32x4*)(q_raw + base); nq1 = *(const f32x4*)(q_raw + base + 4);
            ng0 = *(const f32x4*)(dout + base); ng1 = *(const f32x4*)(dout + base + 4);
            nls = lse[row * CFFM_HEADS + h];
        }
        const f32x4 q0 = live ? (nq0 + bq0) * scale : z4, q1 = live ? (nq1 + bq1) * scale : z4, g0 = live ? ng0 : z4, g1 = live ? ng1 : z4;
        const float ls = live ? nls : 0.f;
        if (it + 1 < tiles_per_wave && t + 16 < T) {
            const long base = (row + 16) * CFFM_C + h * CFFM_HD + 8 * g;
            nq0 = *(const f32x4*)(q_raw + base); nq1 = *(const f32x4*)(q_raw + base + 4);
            ng0 = *(const f32x4*)(dout + base); ng1 = *(const f32x4*)(dout + base + 4);
            nls = lse[(row + 16) * CFFM_HEADS + h];
        }
        bf16x8 qh, ql, gh, gl;
        gtm_split8(q0, q1, qh, ql);
        gtm_split8(g0, g1, gh, gl);
        // D = sum_ch dO o = sum_keys p dp: taken from the SAME products as dp, so that ds = p (dp - D) cancels to rounding noise of the
        // fp32 sums, not of the split products, where it must (one prototype: p = 1, dp = D, ds = 0); the saved o is not read at all
        f32x4 ds[KT], dpv[KT];
        float D = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const f32x4 sv = gtm_mma3(__builtin_bit_cast(bf16x8, RF[(2 * kt) * 64 + lane]), __builtin_bit_cast(bf16x8, RF[(2 * kt + 1) * 64 + lane]), qh, ql, z4);
            dpv[kt] = gtm_mma3(__builtin_bit_cast(bf16x8, RF[(4 * U + 2 * kt) * 64 + lane]), __builtin_bit_cast(bf16x8, RF[(4 * U + 2 * kt + 1) * 64 + lane]), gh, gl, z4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (live && 16 * kt + 4 * g + r < K) ? fast_exp2((sv[r] - ls) * 1.4426950408889634f) : 0.f;
                ds[kt][r] = pv;
                D += pv * dpv[kt][r];
            }
        }
        D += __shfl_xor(D, 16, 64);
        D += __shfl_xor(D, 32, 64);
        if (live && g == 0) Dbuf[row * CFFM_HEADS + h] = D;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[kt][r] *= dpv[kt][r] - D;
        f32x4 dq[2] = {z4, z4};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 dh, dl;
            gtm_split8(ds[2 * u], ds[2 * u + 1], dh, dl);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) dq[mt] = gtm_mma3(kth[mt][u], ktl[mt][u], dh, dl, dq[mt]);
        }
        if (live) {      // lane: channels 16 mt + 4 g .. + 3 of token l15
            float* drow = dq_raw + row * CFFM_C + h * CFFM_HD + 4 * g;
            *(f32x4*)(drow) = dq[0] * scale;
            *(f32x4*)(drow + 16) = dq[1] * scale;
        }
    }
}

// grid (records per (clip, head), 8, B), 256 threads; rec [B][8][gridDim.x][K][64]
template <int U>
__global__ void __launch_bounds__(256) k_gtc_attn_bwd_dkv_mfma(const float* __restrict__ q_raw, const float* __restrict__ bq, const f32x4* __restrict__ frags,
                                                                const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                                const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                float* __restrict__ rec, int T, int K, int pairs_per_wave) {
    constexpr int KT = 2 * U;
    CFFM_DYN_SMEM(smem);
    __shared__ f32x4 sfr[GTM_LOCAL_UNITS(U)];
    f32x4* RF = (f32x4*)smem;            // KF | VF units
    float* tiles = (float*)(RF + 8 * U * 64);           // [4 waves][q tile 32 x GTM_TLD | dO tile 32 x GTM_TLD]; later the waves' record tiles
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f;
    {
        const f32x4* fr = frags + (long)(b * CFFM_HEADS + h) * gtm_frag_units(U) * 64;
        if (U == 1 && !frags) {
            gtm_pack_local<U>(sfr, kv_raw, bkv, b, h, K, tid);
            fr = sfr;
        }
        gtm_copy_row_frags<U>(RF, fr, tid);
    }
    __syncthreads();
    float* Tq = tiles + wave * (2 * 32 * GTM_TLD);
    float* Tg = Tq + 32 * GTM_TLD;
    const f32x4 bq0 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g), bq1 = *(const f32x4*)(bq + h * CFFM_HD + 8 * g + 4);
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 aK[2][KT], aV[2][KT];      // dKc^T / dVc^T: lane (channel 16 mt + 4 g + r, key 16 t + l15)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) aK[mt][kt] = aV[mt][kt] = z4;
    const int pair0 = (blockIdx.x * 4 + wave) * pairs_per_wave;
    for (int it = 0; it < pairs_per_wave; ++it) {
        const int t0 = 32 * (pair0 + it);
        if (t0 >= T) break;                              // (wave-uniform)
        bf16x8 qh[2], ql[2], gh[2], gl[2];
        f32x4 ls[2], Dv[2];
        wave_lds_sync();                                 // the previous pair's transposed reads are done
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int t = t0 + 16 * x + l15;
            f32x4 q0 = z4, q1 = z4, g0 = z4, g1 = z4;
            if (t < T) {
                const long base = ((long)b * T + t) * CFFM_C + h * CFFM_HD + 8 * g;
                q0 = (*(const f32x4*)(q_raw + base) + bq0) * scale;
                q1 = (*(const f32x4*)(q_raw + base + 4) + bq1) * scale;
                g0 = *(const f32x4*)(dout + base); g1 = *(const f32x4*)(dout + base + 4);
            }
            gtm_split8(q0, q1, qh[x], ql[x]);
            gtm_split8(g0, g1, gh[x], gl[x]);
            float* tq = Tq + (16 * x + l15) * GTM_TLD + 8 * g;
            float* tg = Tg + (16 * x + l15) * GTM_TLD + 8 * g;
            *(f32x4*)tq = q0; *(f32x4*)(tq + 4) = q1;
            *(f32x4*)tg = g0; *(f32x4*)(tg + 4) = g1;
            // the statistics of this lane's four tokens 4 g + r of the half (C rows of the tokens x keys orientation)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tt = t0 + 16 * x + 4 * g + r;
                const long rr = ((long)b * T + (tt < T ? tt : 0)) * CFFM_HEADS + h;
                ls[x][r] = tt < T ? lse[rr] : INFINITY;          // (exp(s - inf) = 0: a token past the end contributes nothing)
                Dv[x][r] = tt < T ? Dbuf[rr] : 0.f;
            }
        }
        wave_lds_sync();
        // transposed q / dO fragments: lane (channel 16 mt + l15, k-slot (g, j) <-> token 16 (j >> 2) + 4 g + (j & 3) of the pair)
        bf16x8 qth[2], qtl[2], gth[2], gtl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 a0, a1, c0, c1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] = Tq[(4 * g + j) * GTM_TLD + 16 * mt + l15]; a1[j] = Tq[(16 + 4 * g + j) * GTM_TLD + 16 * mt + l15];
                c0[j] = Tg[(4 * g + j) * GTM_TLD + 16 * mt + l15]; c1[j] = Tg[(16 + 4 * g + j) * GTM_TLD + 16 * mt + l15];
            }
            gtm_split8(a0, a1, qth[mt], qtl[mt]);
            gtm_split8(c0, c1, gth[mt], gtl[mt]);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const bf16x8 kh = __builtin_bit_cast(bf16x8, RF[(2 * kt) * 64 + lane]), kl = __builtin_bit_cast(bf16x8, RF[(2 * kt + 1) * 64 + lane]);
            const bf16x8 vh = __builtin_bit_cast(bf16x8, RF[(4 * U + 2 * kt) * 64 + lane]), vl = __builtin_bit_cast(bf16x8, RF[(4 * U + 2 * kt + 1) * 64 + lane]);
            const bool kok = 16 * kt + l15 < K;
            f32x4 pv[2], dsv[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f32x4 sv = gtm_mma3(qh[x], ql[x], kh, kl, z4);     // S[token 4 g + r][key 16 kt + l15]
                const f32x4 dp = gtm_mma3(gh[x], gl[x], vh, vl, z4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = kok ? fast_exp2((sv[r] - ls[x][r]) * 1.4426950408889634f) : 0.f;
                    pv[x][r] = p;
                    dsv[x][r] = p * (dp[r] - Dv[x][r]);
                }
            }
            bf16x8 ph, pl, dh, dl;
            gtm_split8(pv[0], pv[1], ph, pl);
            gtm_split8(dsv[0], dsv[1], dh, dl);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                aV[mt][kt] = gtm_mma3(gth[mt], gtl[mt], ph, pl, aV[mt][kt]);
                aK[mt][kt] = gtm_mma3(qth[mt], qtl[mt], dh, dl, aK[mt][kt]);
            }
        }
    }
    // ---- the record of this workgroup: the four waves' accumulators added in wave order, key tile by key tile
    float* out = rec + (((long)(b * CFFM_HEADS + h) * gridDim.x + blockIdx.x) * K) * 64;
    constexpr int RLD = 68;                               // floats per row of a record tile [16 keys][64]
    float* R = tiles;                                     // [4 waves][16][RLD] (17 KB of the 36 KB tile area)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();                                  // the tile area is free (first trip: every wave is done with its q / dO tiles)
        float* mine = R + wave * 16 * RLD + l15 * RLD + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            *(f32x4*)(mine + 16 * mt) = aK[mt][kt];
            *(f32x4*)(mine + 32 + 16 * mt) = aV[mt][kt];
        }
        __syncthreads();
        const int rowk = tid >> 4, c4 = 4 * (tid & 15), key = 16 * kt + rowk;
        if (key < K) {
            const float* src = R + rowk * RLD + c4;
            *(f32x4*)(out + (long)key * 64 + c4) = (*(const f32x4*)(src) + *(const f32x4*)(src + 16 * RLD)) + (*(const f32x4*)(src + 32 * RLD) + *(const f32x4*)(src + 48 * RLD));
        }
    }
}

// dkv [B*K][512] (dKc at column h*32, dVc at 256 + h*32; dKc carries the q scale already) = the records of each (clip, head) added in
// order.  grid (K, 8, B), 64 threads
__global__ void __launch_bounds__(64) k_gtc_dkv_sum(const float* __restrict__ rec, int nrec, int K, float* __restrict__ dkv) {
    const int k = blockIdx.x, h = blockIdx.y, b = blockIdx.z, e = threadIdx.x;
    const float* p = rec + ((long)(b * CFFM_HEADS + h) * nrec * K + k) * 64 + e;
    float s0 = 0.f, s1 = 0.f;
    int r = 0;
    for (; r + 1 < nrec; r += 2) { s0 += p[(long)r * K * 64]; s1 += p[(long)(r + 1) * K * 64]; }
    if (r < nrec) s0 += p[(long)r * K * 64];
    dkv[((long)b * K + k) * 512 + (e < 32 ? 0 : 256) + h * CFFM_HD + (e & 31)] = s0 + s1;
}
