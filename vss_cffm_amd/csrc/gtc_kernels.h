// gtc_kernels.h -- CFFM++ global temporal context: per-token cross-attention against K prototype
// ("cluster centre") tokens.  Reference: WindowAttention_cluster.forward,
// pvt/swin_transformer_2d.py:208-262 with only_use_cluster_center_as_context=True (:216): no in-window
// keys, no position bias, no mask, so window partition / padding are numerically no-ops
// (SURVEY.md A "GTC is window-independent") and the op is plain [token, head] x K attention.
// HBM-bound (reads q, writes o once); 49*K*32 MACs per token-head is far below the MFMA ridge, so
// this stays on the VALU: 8 lanes own one (token, head) (4 of the 32 dims each), the K centre
// rows of the head sit in LDS, softmax is online.
#pragma once
#include "cffm_common.h"

#define GTC_TOK 32  // tokens per workgroup

// plain row LayerNorm: z = LN(x) (+ stats)
__global__ void __launch_bounds__(256) k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ z,
                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out, long nrows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const f32x4 v = *(const f32x4*)(x + row * CFFM_C + 4 * lane);
    const float mu = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / CFFM_C);
    const f32x4 d = v - mu;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / CFFM_C);
    const float rs = 1.f / sqrtf(var + CFFM_LN_EPS);
    *(f32x4*)(z + row * CFFM_C + 4 * lane) = d * rs * *(const f32x4*)(gamma + 4 * lane) + *(const f32x4*)(beta + 4 * lane);
    if (lane == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
}

__device__ __forceinline__ float red8(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

// grid (ceil(T/32), 8, B).  q_raw [B*T,qld] (q = (q_raw + bq) * scale), kv_raw [B*K,512], o [B*T,256], lse [B*T,8]
__global__ void __launch_bounds__(256) k_gtc_attn_fwd(const float* __restrict__ q_raw, const float* __restrict__ bq,
                                                       const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                       float* __restrict__ o, float* __restrict__ lse, int T, int K) {
    CFFM_DYN_SMEM(smem);
    float* Kc = (float*)smem;       // [K][32]
    float* Vc = Kc + K * CFFM_HD;   // [K][32]
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    for (int e = tid; e < K * CFFM_HD; e += 256) {
        const int k = e >> 5, d = e & 31;
        const float* row = kv_raw + ((long)b * K + k) * 512 + h * CFFM_HD + d;
        Kc[e] = row[0] + bkv[h * CFFM_HD + d];
        Vc[e] = row[256] + bkv[256 + h * CFFM_HD + d];
    }
    __syncthreads();
    const int t = blockIdx.x * GTC_TOK + (tid >> 3), c = tid & 7;
    const bool live = t < T;
    const long row = (long)b * T + (live ? t : 0);
    const float scale = 0.17677669529663687f;
    f32x4 q = (*(const f32x4*)(q_raw + row * CFFM_C + h * CFFM_HD + 4 * c) + *(const f32x4*)(bq + h * CFFM_HD + 4 * c)) * scale;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const f32x4 kc = *(const f32x4*)(Kc + k * CFFM_HD + 4 * c);
        const float s = red8(q[0] * kc[0] + q[1] * kc[1] + q[2] * kc[2] + q[3] * kc[3]);
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn), p = expf(s - mn);
        l = l * alpha + p;
        acc = acc * alpha + p * *(const f32x4*)(Vc + k * CFFM_HD + 4 * c);
        m = mn;
    }
    if (live) {
        *(f32x4*)(o + row * CFFM_C + h * CFFM_HD + 4 * c) = acc * (1.f / l);
        if (c == 0) lse[row * CFFM_HEADS + h] = m + logf(l);
    }
}

// dq_raw [B*T,256] (written), dkv [B*K,512] (atomics; pre-zeroed)
__global__ void __launch_bounds__(256) k_gtc_attn_bwd(const float* __restrict__ q_raw, const float* __restrict__ bq,
                                                       const float* __restrict__ kv_raw, const float* __restrict__ bkv,
                                                       const float* __restrict__ o, const float* __restrict__ dout,
                                                       const float* __restrict__ lse, float* __restrict__ dq_raw,
                                                       float* __restrict__ dkv, int T, int K) {
    CFFM_DYN_SMEM(smem);
    float* Kc = (float*)smem;
    float* Vc = Kc + K * CFFM_HD;
    float* dKc = Vc + K * CFFM_HD;
    float* dVc = dKc + K * CFFM_HD;
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    for (int e = tid; e < K * CFFM_HD; e += 256) {
        const int k = e >> 5, d = e & 31;
        const float* row = kv_raw + ((long)b * K + k) * 512 + h * CFFM_HD + d;
        Kc[e] = row[0] + bkv[h * CFFM_HD + d];
        Vc[e] = row[256] + bkv[256 + h * CFFM_HD + d];
        dKc[e] = 0.f;
        dVc[e] = 0.f;
    }
    __syncthreads();
    const int t = blockIdx.x * GTC_TOK + (tid >> 3), c = tid & 7;
    const bool live = t < T;
    const long row = (long)b * T + (live ? t : 0);
    const float scale = 0.17677669529663687f;
    const f32x4 q = (*(const f32x4*)(q_raw + row * CFFM_C + h * CFFM_HD + 4 * c) + *(const f32x4*)(bq + h * CFFM_HD + 4 * c)) * scale;
    f32x4 dov = *(const f32x4*)(dout + row * CFFM_C + h * CFFM_HD + 4 * c);
    const f32x4 ov = *(const f32x4*)(o + row * CFFM_C + h * CFFM_HD + 4 * c);
    if (!live) dov = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float D = red8(dov[0] * ov[0] + dov[1] * ov[1] + dov[2] * ov[2] + dov[3] * ov[3]);
    const float ls = lse[row * CFFM_HEADS + h];
    f32x4 dq = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const f32x4 kc = *(const f32x4*)(Kc + k * CFFM_HD + 4 * c);
        const f32x4 vc = *(const f32x4*)(Vc + k * CFFM_HD + 4 * c);
        const float s = red8(q[0] * kc[0] + q[1] * kc[1] + q[2] * kc[2] + q[3] * kc[3]);
        const float dp = red8(dov[0] * vc[0] + dov[1] * vc[1] + dov[2] * vc[2] + dov[3] * vc[3]);
        const float p = live ? expf(s - ls) : 0.f;
        const float ds = p * (dp - D);
        dq += ds * kc;
        for (int e = 0; e < 4; ++e) {
            atomicAdd(dKc + k * CFFM_HD + 4 * c + e, ds * q[e]);
            atomicAdd(dVc + k * CFFM_HD + 4 * c + e, p * dov[e]);
        }
    }
    if (live) *(f32x4*)(dq_raw + row * CFFM_C + h * CFFM_HD + 4 * c) = dq * scale;
    __syncthreads();
    for (int e = tid; e < K * CFFM_HD; e += 256) {
        const int k = e >> 5, d = e & 31;
        float* row2 = dkv + ((long)b * K + k) * 512 + h * CFFM_HD + d;
        atomicAdd(row2, dKc[e]);
        atomicAdd(row2 + 256, dVc[e]);
    }
}
