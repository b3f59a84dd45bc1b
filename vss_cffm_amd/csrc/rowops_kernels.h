// rowops_kernels.h -- HBM-bound row-wise kernels around the GEMMs of one CFFM block:
//   bias table assembly / scatter      (cffm_transformer.py:536-587)
//   residual + LayerNorm(norm2)        (cffm_transformer.py:823-824)
//   bias + exact-erf GELU              (Mlp, cffm_transformer.py:10-26)
//   final residual                     (cffm_transformer.py:824)
//   column sums for the Linear bias gradients.
// One wave per 256-channel row (1 KiB coalesced f32x4 accesses), 4 rows per workgroup.
#pragma once
#include "cffm_common.h"

// --------------------------------------------------------------------------- position-bias tables
struct BiasTables {
    const float* own;      // relative_position_bias_table [169, 8]
    const float* ring;     // relative_position_bias_table_to_neighbors [1, 8, 49, 132]
    const float* pool[4];  // ..._to_windows.0 [8,121], ..._to_windows_clips.{0,1,2} [8,169],[8,121],[8,81]
};
struct BiasTablesG {
    float* own;
    float* ring;
    float* pool[4];
};

// element offset of bias entry (head h, query q, key n) inside its parameter table; returns table id
// 0 = own, 1 = ring, 2..5 = pooled groups (get_relative_position_index :158-185)
__device__ __forceinline__ int bias_locate(int h, int q, int n, int& off) {
    const int qi = q / 7, qj = q % 7;
    if (n < 49) {
        const int ki = n / 7, kj = n % 7;
        off = ((qi - ki + 6) * 13 + (qj - kj + 6)) * CFFM_HEADS + h;
        return 0;
    }
    if (n < 181) {
        off = (h * 49 + q) * 132 + (n - 49);
        return 1;
    }
    int base, kk, id;
    if (n < 206) { base = 181; kk = 5; id = 2; }
    else if (n < 255) { base = 206; kk = 7; id = 3; }
    else if (n < 280) { base = 255; kk = 5; id = 4; }
    else { base = 280; kk = 3; id = 5; }
    const int a = (n - base) / kk, bb = (n - base) % kk, side = 6 + kk;
    off = h * side * side + (qi - a + kk - 1) * side + (qj - bb + kk - 1);
    return id;
}

// bias [8][64][304] (query-major, for the S^T = K Q^T orientation) and biasT [8][304][64]
// (key-major, for the S = Q K^T orientation of the backward); pad entries are 0.
__global__ void __launch_bounds__(256) k_bias_assemble(BiasTables t, float* __restrict__ bias, float* __restrict__ biasT) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD) return;
    const int n = e % CFFM_NKEY_PAD, q = (e / CFFM_NKEY_PAD) % CFFM_NQ_PAD, h = e / (CFFM_NKEY_PAD * CFFM_NQ_PAD);
    float v = 0.f;
    if (q < CFFM_WA && n < CFFM_NKEY) {
        int off;
        const int id = bias_locate(h, q, n, off);
        v = id == 0 ? t.own[off] : id == 1 ? t.ring[off] : t.pool[id - 2][off];
    }
    bias[e] = v;
    if (biasT) biasT[((long)h * CFFM_NKEY_PAD + n) * CFFM_NQ_PAD + q] = v;
}

// dbiasT [8][304][64] (key-major, as the attention backward accumulates it) -> the six tables
__global__ void __launch_bounds__(256) k_bias_scatter(const float* __restrict__ dbiasT, BiasTablesG g) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= CFFM_HEADS * CFFM_NKEY * CFFM_WA) return;
    const int q = e % CFFM_WA, n = (e / CFFM_WA) % CFFM_NKEY, h = e / (CFFM_WA * CFFM_NKEY);
    const float v = dbiasT[((long)h * CFFM_NKEY_PAD + n) * CFFM_NQ_PAD + q];
    int off;
    const int id = bias_locate(h, q, n, off);
    if (id == 1) g.ring[off] = v;  // the dense ring table is one-to-one
    else atomicAdd((id == 0 ? g.own : g.pool[id - 2]) + off, v);
}

// --------------------------------------------------------------------------- x1 = xt + (yraw + bproj); z2 = LN2(x1)
__global__ void __launch_bounds__(256) k_residual_ln(const float* __restrict__ xt, long xt_bs, int rows_per_batch,
                                                      const float* __restrict__ yraw, const float* __restrict__ bproj,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ x1, float* __restrict__ z2,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, long nrows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const long b = row / rows_per_batch, r = row % rows_per_batch;
    const f32x4 a = *(const f32x4*)(xt + b * xt_bs + r * CFFM_C + 4 * lane);
    const f32x4 y = *(const f32x4*)(yraw + row * CFFM_C + 4 * lane) + *(const f32x4*)(bproj + 4 * lane);
    const f32x4 v = a + y;
    *(f32x4*)(x1 + row * CFFM_C + 4 * lane) = v;
    const float mu = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / CFFM_C);
    const f32x4 d = v - mu;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / CFFM_C);
    const float rs = 1.f / sqrtf(var + CFFM_LN_EPS);
    *(f32x4*)(z2 + row * CFFM_C + 4 * lane) = d * rs * *(const f32x4*)(gamma + 4 * lane) + *(const f32x4*)(beta + 4 * lane);
    if (lane == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
}

// backward of z2 = LN(x1): dx1 = dres + LNbwd(dz2); also dgamma/dbeta partial sums (atomics).
__global__ void __launch_bounds__(256) k_ln_bwd_residual(const float* __restrict__ x1, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                          const float* __restrict__ dz2, const float* __restrict__ dres,
                                                          float* __restrict__ dx1, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, long nrows, int rows_per_block) {
    __shared__ float red[4][2][CFFM_C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 gm = *(const f32x4*)(gamma + 4 * lane);
    f32x4 ag = (f32x4){0.f, 0.f, 0.f, 0.f}, ab = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (long row = r0 + wave; row < r0 + rows_per_block && row < nrows; row += 4) {
        const f32x4 xv = *(const f32x4*)(x1 + row * CFFM_C + 4 * lane);
        const float mu = mean_in[row], rs = rstd_in[row];
        const f32x4 xh = (xv - mu) * rs;
        const f32x4 dz = *(const f32x4*)(dz2 + row * CFFM_C + 4 * lane);
        ag += dz * xh;
        ab += dz;
        const f32x4 gz = dz * gm;
        const float m1 = wave_sum(gz[0] + gz[1] + gz[2] + gz[3]) * (1.f / CFFM_C);
        const float m2 = wave_sum(gz[0] * xh[0] + gz[1] * xh[1] + gz[2] * xh[2] + gz[3] * xh[3]) * (1.f / CFFM_C);
        f32x4 dxv = (gz - m1 - xh * m2) * rs;
        if (dres) dxv += *(const f32x4*)(dres + row * CFFM_C + 4 * lane);
        *(f32x4*)(dx1 + row * CFFM_C + 4 * lane) = dxv;
    }
    *(f32x4*)(&red[wave][0][4 * lane]) = ag;
    *(f32x4*)(&red[wave][1][4 * lane]) = ab;
    __syncthreads();
    const int ch = threadIdx.x;
    atomicAdd(dgamma + ch, red[0][0][ch] + red[1][0][ch] + red[2][0][ch] + red[3][0][ch]);
    atomicAdd(dbeta + ch, red[0][1][ch] + red[1][1][ch] + red[2][1][ch] + red[3][1][ch]);
}

// --------------------------------------------------------------------------- act = gelu(hraw + b1)
__global__ void __launch_bounds__(256) k_bias_gelu(const float* __restrict__ hraw, const float* __restrict__ b1,
                                                    float* __restrict__ act, long n4, int ncol4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)hraw)[e] + ((const f32x4*)b1)[e % ncol4];
        f32x4 o;
        o[0] = gelu_erf(v[0]); o[1] = gelu_erf(v[1]); o[2] = gelu_erf(v[2]); o[3] = gelu_erf(v[3]);
        ((f32x4*)act)[e] = o;
    }
}
// dhraw = dact * gelu'(hraw + b1)   (in place on dact)
__global__ void __launch_bounds__(256) k_gelu_bwd(const float* __restrict__ hraw, const float* __restrict__ b1,
                                                   float* __restrict__ dact, long n4, int ncol4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)hraw)[e] + ((const f32x4*)b1)[e % ncol4];
        f32x4 d = ((f32x4*)dact)[e];
        d[0] *= gelu_erf_grad(v[0]); d[1] *= gelu_erf_grad(v[1]); d[2] *= gelu_erf_grad(v[2]); d[3] *= gelu_erf_grad(v[3]);
        ((f32x4*)dact)[e] = d;
    }
}

// --------------------------------------------------------------------------- out = x1 + (oraw + b2)
__global__ void __launch_bounds__(256) k_residual_out(const float* __restrict__ x1, const float* __restrict__ oraw,
                                                       const float* __restrict__ b2, float* __restrict__ out, long n4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
        ((f32x4*)out)[e] = ((const f32x4*)x1)[e] + ((const f32x4*)oraw)[e] + ((const f32x4*)b2)[e % (CFFM_C / 4)];
}

// --------------------------------------------------------------------------- column sums (Linear bias grads)
// out[c] (+)= sum_r a[r][c];  grid (ncol/256, nslices); atomics across row slices (out pre-zeroed).
__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ a, long nrows, int ncol, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const long per = (nrows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = (r0 + per < nrows) ? r0 + per : nrows;
    if (c >= ncol) return;
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += a[r * ncol + c];
    atomicAdd(out + c, s);
}

// a += b (f32x4)
__global__ void __launch_bounds__(256) k_add_inplace(float* __restrict__ a, const float* __restrict__ b, long n4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
        ((f32x4*)a)[e] += ((const f32x4*)b)[e];
}
