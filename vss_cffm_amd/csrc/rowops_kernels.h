// rowops_kernels.h -- HBM-bound row-wise kernels around the GEMMs of one CFFM block:
//   bias table assembly / scatter      (cffm_transformer.py:536-587)
//   residual + LayerNorm(norm2)        (cffm_transformer.py:823-824)
//   bias + exact-erf GELU              (Mlp, cffm_transformer.py:10-26)
//   final residual                     (cffm_transformer.py:824)
//   column sums for the Linear bias gradients.
// One wave per 256-channel row (1 KiB coalesced f32x4 accesses), 4 rows per workgroup.
#pragma once
#include "cffm_common.h"
#include "cffa_kernels.h"
#include "panel_kernels.h"

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

// The dense [8 heads][64 queries][304 keys] additive bias in two layouts (pad entries are 0):
//   biasH [8][4 waves][10 key-tile pairs][64 lanes][8] f16 -- the B operand of the bias MFMA of the attention kernels: entry
//         (h, wave, p, lane = 16 g + j, e) = bias(h, query 16 wave + j, key 16 (2 p + (g >> 1)) + 8 (g & 1) + e).  One contiguous
//         1 KiB load per (wave, tile pair); the kernels add it to S^T on the matrix pipe (S^T tile t = K Q^T + Sel_(t & 1) B,
//         Sel_o = the 16 x 32 selection matrix that routes k-slot 16 o + i to key row i), so the position bias costs neither
//         fp32 bytes nor VALU instructions (round 4: tile pairs; rounds 2-3 kept one tile per fragment, upper 32 lanes zero).  f16 rounding of the bias moves the layer output by 3e-5 of its maximum (the oracle
//         with the tables rounded to f16), a fifth of what the f16 Q / K / P / V operands already contribute;
//   bias  [8][64][304] fp32, query-major (stage-level checks only; NULL inside the block).
__device__ __forceinline__ long biash_index(int h, int q, int n) {      // n < 320: key tile 19 (keys 304..319) is the zero half of pair 9
    const int t = n >> 4;
    return ((long)((h * 4 + (q >> 4)) * 10 + (t >> 1)) * 64 + (((t & 1) * 2 + ((n >> 3) & 1)) * 16 + (q & 15))) * 8 + (n & 7);
}
__device__ __forceinline__ void bias_assemble_body(const BiasTables& t, float* __restrict__ bias, h16* __restrict__ biasH, int bx) {
    const int e = bx * 256 + threadIdx.x;
    if (e >= CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD) return;
    const int n = e % CFFM_NKEY_PAD, q = (e / CFFM_NKEY_PAD) % CFFM_NQ_PAD, h = e / (CFFM_NKEY_PAD * CFFM_NQ_PAD);
    float v = 0.f;
    if (q < CFFM_WA && n < CFFM_NKEY) {
        int off;
        const int id = bias_locate(h, q, n, off);
        v = id == 0 ? t.own[off] : id == 1 ? t.ring[off] : t.pool[id - 2][off];
    }
    if (bias) bias[e] = v;
    if (biasH) {
        biasH[biash_index(h, q, n)] = (h16)v;
        if (n >= CFFM_NKEY_PAD - 16) biasH[biash_index(h, q, n + 16)] = (h16)0.f;    // the unused second half of the last tile pair
    }
}
__global__ void __launch_bounds__(256) k_bias_assemble(BiasTables t, float* __restrict__ bias, h16* __restrict__ biasH) {
    bias_assemble_body(t, bias, biasH, blockIdx.x);
}
// Everything a layer's blocks derive from parameters alone (dense bias tiles, composed pooling matrices), for up to
// PREP_MAXD blocks in one launch: grid (bias workgroups + 1, blocks), the last workgroup of a row builds the pooling matrix.
#define PREP_MAXD 4
// ... and (pack != 0) a copy of the four Linear weights in split-4 storage (qkv 768x256 | proj 256x256 | fc1 1024x256 |
// fc2 256x1024, one after the other): the weight operand of every forward / input-gradient GEMM is then staged without
// arithmetic -- each weight element is otherwise re-split by every row panel of the GEMM (81x for the q|k|v Linear).
#define PREP_WFLOATS (768 * 256 + 256 * 256 + 1024 * 256 + 256 * 1024)
#define PREP_WBLOCKS (PREP_WFLOATS / 4 / 256)
// ... and the same four weights in MFMA-fragment order for the row-panel kernels (panel_kernels.h), forward (NT) forms first,
// input-gradient (NN) forms behind them, each set laid out qkv | proj | fc1 | fc2 like the split-4 copy: 2 x PREP_WFLOATS floats
#define PREP_FBLOCKS (2 * PREP_WFLOATS / 8 / 256)
struct PrepArgs {
    BiasTables t[PREP_MAXD];
    PoolW pw[PREP_MAXD];
    float* bias[PREP_MAXD];    // the f16 fragment table (biasH)
    float* M[PREP_MAXD];
    const float* w[PREP_MAXD][4];
    float* w_s[PREP_MAXD];
    float* w_f[PREP_MAXD];
    int nbias, pack;
};
__device__ __forceinline__ void param_prep_body(const PrepArgs& a, int bx, int d) {
    if (bx < a.nbias) { bias_assemble_body(a.t[d], nullptr, (h16*)a.bias[d], bx); return; }
    if (bx == a.nbias) { pool_matrix_body(a.pw[d], a.M[d]); return; }
    if (!a.pack) return;
    const long n4[4] = {768 * 256 / 4, 256 * 256 / 4, 1024 * 256 / 4, 256 * 1024 / 4};
    if (bx >= a.nbias + 1 + PREP_WBLOCKS) {   // fragment-ordered copies: one thread per 8 weights
        long it = (long)(bx - a.nbias - 1 - PREP_WBLOCKS) * 256 + threadIdx.x;
        const int form = it >= PREP_WFLOATS / 8;
        if (form) it -= PREP_WFLOATS / 8;
        const int N[4] = {768, 256, 1024, 256}, K[4] = {256, 256, 256, 1024};
        long o8 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (it >= o8 && it < o8 + n4[k] / 2)
                pnl_pack_weight(a.w[d][k], N[k], K[k], form != 0, (f32x4*)(a.w_f[d] + (long)form * PREP_WFLOATS + o8 * 8), it - o8);
            o8 += n4[k] / 2;
        }
        return;
    }
    if (a.pack == 2) return;                                   // (split-4 copy not wanted: see prep_args)
    long e = (long)(bx - a.nbias - 1) * 256 + threadIdx.x;   // float4 index into the concatenated weights
    long off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e >= off && e < off + n4[k]) ((f32x4*)a.w_s[d])[e] = split4_pack(((const f32x4*)a.w[d][k])[e - off]);
        off += n4[k];
    }
}
__global__ void __launch_bounds__(256) k_param_prep(PrepArgs a) { param_prep_body(a, blockIdx.x, blockIdx.y); }
// The layer forward's input transpose and the parameter prep of its blocks as ONE launch (round 3): the prep used to run beside the
// transpose on a side stream, and the join of that stream in front of the first q|k|v GEMM cost the chain ~5 us under graph replay
// although the prep had long finished.  1-D grid: the transpose's workgroups first, then pnx x (blocks) of the prep.
struct TrArgs { const float* src; float* dst; int rows, cols; long src_bs, dst_bs; int add_mod, add_skip; float* copy_dst; int gx, gy, gz; };
__global__ void __launch_bounds__(256) k_transpose_prep(TrArgs T, PrepArgs a, int pnx) {
    const int nt = T.gx * T.gy * T.gz;
    int b = blockIdx.x;
    if (b < nt) {
        transpose_body(T.src, T.dst, T.rows, T.cols, T.src_bs, T.dst_bs, nullptr, T.add_mod, T.add_skip, T.copy_dst, b % T.gx, (b / T.gx) % T.gy,
                       b / (T.gx * T.gy));
        return;
    }
    b -= nt;
    param_prep_body(a, b % pnx, b / pnx);
}

// dbiasT [8][304][64] (key-major, as the attention backward accumulates it) -> the six tables.
// Gather form (no atomics): one thread per table entry sums the <= 49 (query, key) pairs that index it.
__global__ void __launch_bounds__(256) k_bias_scatter(const float* __restrict__ dbiasT, BiasTablesG g) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int n_own = 169 * CFFM_HEADS, n_ring = CFFM_HEADS * 49 * 132;
    if (e < n_ring) {  // dense ring table [1,8,49,132]: one-to-one
        const int n = e % 132, q = (e / 132) % 49, h = e / (132 * 49);
        g.ring[e] = dbiasT[((long)h * CFFM_NKEY_PAD + 49 + n) * CFFM_NQ_PAD + q];
    }
    if (blockIdx.x * 256 + 255 < n_ring) return;   // (whole workgroups only: the row sums below are wave-collective)
    // shared tables: 16 lanes per entry (n_ring is a multiple of 16, so the groups are DPP rows), lane j takes the queries
    // j, j+16, j+32, j+48 -- four independent loads instead of a 49-step serial loop -- and the row sum combines them
    const int j = (e - n_ring) & 15;
    int r = (e - n_ring) >> 4;
    float acc = 0.f;
    float* dst = nullptr;
    if (e < n_ring) {
        r = -1;       // ring lanes of the one mixed workgroup: no entry
    } else if (r < n_own) {  // own table [169,8]: idx = (qi-ki+6)*13 + (qj-kj+6)
        const int h = r % CFFM_HEADS, idx = r / CFFM_HEADS, di = idx / 13 - 6, dj = idx % 13 - 6;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = j + 16 * u, ki = q / 7 - di, kj = q % 7 - dj;
            if (q < 49 && ki >= 0 && ki < 7 && kj >= 0 && kj < 7) acc += dbiasT[((long)h * CFFM_NKEY_PAD + ki * 7 + kj) * CFFM_NQ_PAD + q];
        }
        dst = g.own + r;
    } else {
        r -= n_own;
        const int kks[4] = {5, 7, 5, 3}, bases[4] = {181, 206, 255, 280};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = kks[t], side = 6 + kk, n_t = CFFM_HEADS * side * side;
            if (!dst && r >= 0 && r < n_t) {  // pooled tables [8, side*side]: idx = (qi-a+kk-1)*side + (qj-b+kk-1)
                const int h = r / (side * side), idx = r % (side * side), di = idx / side - (kk - 1), dj = idx % side - (kk - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = j + 16 * u, a = q / 7 - di, bb = q % 7 - dj;
                    if (q < 49 && a >= 0 && a < kk && bb >= 0 && bb < kk)
                        acc += dbiasT[((long)h * CFFM_NKEY_PAD + bases[t] + a * kk + bb) * CFFM_NQ_PAD + q];
                }
                dst = g.pool[t] + r;
            }
            r -= n_t;
        }
    }
    acc = row16_sum(acc);   // every lane of the wave takes part (lanes past the last entry carry zeros)
    if (dst && j == 0) *dst = acc;
}
#define BIAS_SCATTER_THREADS (CFFM_HEADS * 49 * 132 + 16 * (169 * CFFM_HEADS + CFFM_HEADS * (121 + 169 + 121 + 81)))

// --------------------------------------------------------------------------- x1 = xt + (yraw + bproj); z2 = LN2(x1)
__global__ void __launch_bounds__(256) k_residual_ln(const float* __restrict__ xt, long xt_bs, int rows_per_batch,
                                                      const float* __restrict__ yraw, const float* __restrict__ bproj,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ x1, float* __restrict__ z2,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, long nrows,
                                                      int split /* z2 in split-4 storage */) {
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
    const f32x4 zv = d * rs * *(const f32x4*)(gamma + 4 * lane) + *(const f32x4*)(beta + 4 * lane);
    *(f32x4*)(z2 + row * CFFM_C + 4 * lane) = split ? split4_pack(zv) : zv;
    if (lane == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
}

// backward of z2 = LN(x1): dx1 = dres + LNbwd(dz2).  Every workgroup writes one partial record
// part[blk][1024] = dgamma | dbeta | colsum(dres) | colsum(dx1) (the last two are the fc2 / proj bias
// gradients, folded in because this kernel streams those rows anyway); k_reduce_partials sums them.
#define LNB_ROWS 16   // rows per workgroup of k_ln_bwd_residual (4 per wave)
__global__ void __launch_bounds__(256) k_ln_bwd_residual(const float* __restrict__ x1, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                          const float* __restrict__ dz2, const float* __restrict__ dres,
                                                          float* __restrict__ dx1, float* __restrict__ part, long nrows,
                                                          int rows_per_block) {
    __shared__ float red[4][4][CFFM_C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 gm = *(const f32x4*)(gamma + 4 * lane);
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ag = z4, ab = z4, ar = z4, ax = z4;
    const long r0 = (long)blockIdx.x * LNB_ROWS;
    // the wave's LNB_ROWS / 4 rows are all requested before the first is consumed (rows r0 + wave + 4 k)
    constexpr int RW = LNB_ROWS / 4;
    f32x4 xr[RW], dzr[RW], drr[RW];
    float mur[RW], rsr[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const long row = r0 + wave + 4 * k;
        xr[k] = dzr[k] = drr[k] = z4;
        mur[k] = rsr[k] = 0.f;
        if (row < nrows) {
            xr[k] = *(const f32x4*)(x1 + row * CFFM_C + 4 * lane);
            dzr[k] = *(const f32x4*)(dz2 + row * CFFM_C + 4 * lane);
            if (dres) drr[k] = *(const f32x4*)(dres + row * CFFM_C + 4 * lane);
            mur[k] = mean_in[row];
            rsr[k] = rstd_in[row];
        }
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const long row = r0 + wave + 4 * k;
        if (row >= nrows) break;
        const float rs = rsr[k];
        const f32x4 xh = (xr[k] - mur[k]) * rs;
        const f32x4 dz = dzr[k];
        ag += dz * xh;
        ab += dz;
        const f32x4 gz = dz * gm;
        const float m1 = wave_sum(gz[0] + gz[1] + gz[2] + gz[3]) * (1.f / CFFM_C);
        const float m2 = wave_sum(gz[0] * xh[0] + gz[1] * xh[1] + gz[2] * xh[2] + gz[3] * xh[3]) * (1.f / CFFM_C);
        const f32x4 dxv = (gz - m1 - xh * m2) * rs + drr[k];
        ar += drr[k];
        ax += dxv;
        *(f32x4*)(dx1 + row * CFFM_C + 4 * lane) = dxv;
    }
    *(f32x4*)(&red[wave][0][4 * lane]) = ag;
    *(f32x4*)(&red[wave][1][4 * lane]) = ab;
    *(f32x4*)(&red[wave][2][4 * lane]) = ar;
    *(f32x4*)(&red[wave][3][4 * lane]) = ax;
    __syncthreads();
    const int ch = threadIdx.x;
    for (int k = 0; k < 4; ++k)
        part[(long)blockIdx.x * 1024 + k * CFFM_C + ch] = red[0][k][ch] + red[1][k][ch] + red[2][k][ch] + red[3][k][ch];
}

// Second stage of every block-partial reduction, ONE launch per record set: column c of the records
// part[b][stride] (b < nblk) is summed and written (or accumulated) into the output segment that owns it.
// A workgroup owns 64 columns; its 16 waves each sum every 16th record, LDS combines.  Deterministic.
#define RED_MAXSEG 8
struct RedSegs {
    int nseg;
    int off[RED_MAXSEG];     // first record column of the segment
    int width[RED_MAXSEG];
    int accumulate[RED_MAXSEG];
    float* out[RED_MAXSEG];
};
// One 64-column slab of one record set (round 5): a lane owns FOUR consecutive columns (16-byte loads; `part`, `stride` and the column
// offsets of every caller are multiples of 4 floats, and rows are at least ceil4(total) long), a quarter-wave one record, so the 16
// waves have 64 records in flight per step and four steps per lane in the air -- the round 1-4 form walked one column per lane, 16
// records per step, and took 8-10 us for the ~1600 records of a CFFA backward.  Fixed summation order: deterministic.
__device__ __forceinline__ void reduce_records_body(const float* __restrict__ part, int nblk, int stride, int total, const RedSegs& segs, int blk,
                                                    float (*red)[64]) {
    const int lane = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c4 = blk * 64 + 4 * (lane & 15), slot = 4 * ry + (lane >> 4);
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c4 < total) {
        const float* p = part + c4;
        int b = slot;
        for (; b + 192 < nblk; b += 256) {
            const f32x4 v0 = *(const f32x4*)(p + (long)b * stride), v1 = *(const f32x4*)(p + (long)(b + 64) * stride);
            const f32x4 v2 = *(const f32x4*)(p + (long)(b + 128) * stride), v3 = *(const f32x4*)(p + (long)(b + 192) * stride);
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; b < nblk; b += 64) s0 += *(const f32x4*)(p + (long)b * stride);
    }
    *(f32x4*)(&red[slot][4 * (lane & 15)]) = (s0 + s1) + (s2 + s3);
    __syncthreads();
    const int c = blk * 64 + threadIdx.x;
    if (threadIdx.x < 64 && c < total) {
        float v = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) v += red[k][threadIdx.x];
        for (int k = 0; k < segs.nseg; ++k)
            if (c >= segs.off[k] && c < segs.off[k] + segs.width[k]) {
                float* o = segs.out[k] + (c - segs.off[k]);
                *o = segs.accumulate[k] ? *o + v : v;
            }
    }
}
__global__ void __launch_bounds__(1024) k_reduce_records(const float* __restrict__ part, int nblk, int stride, int total, RedSegs segs) {
    __shared__ float red[64][64];
    reduce_records_body(part, nblk, stride, total, segs, blockIdx.x, red);
}

// The same for up to RED_MAXJOB record sets in one launch: the block backward defers its reductions (fc1 bias / LN2 + fc2/proj bias /
// q|k|v bias) to its end, the layer backward those of the CFFA (LN1 + pooling) to the end of the range -- none of their results is read
// earlier.
#define RED_MAXJOB 10
struct RedJobs {
    int njob;
    int blk_end[RED_MAXJOB];        // exclusive prefix of 64-column workgroups per job
    const float* part[RED_MAXJOB];
    int nblk[RED_MAXJOB], stride[RED_MAXJOB], total[RED_MAXJOB];
    RedSegs segs[RED_MAXJOB];
};
__global__ void __launch_bounds__(1024) k_reduce_records_multi(RedJobs J) {
    __shared__ float red[64][64];
    int blk = blockIdx.x, j = 0;
#pragma unroll
    for (int q = 0; q < RED_MAXJOB - 1; ++q)
        if (q + 1 < J.njob && blk >= J.blk_end[q]) j = q + 1;
    if (j > 0) blk -= J.blk_end[j - 1];
    reduce_records_body(J.part[j], J.nblk[j], J.stride[j], J.total[j], J.segs[j], blk, red);
}

// dst = src in split-4 storage; n4 groups of 4 floats
__global__ void __launch_bounds__(256) k_split4(const float* __restrict__ src, float* __restrict__ dst, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) ((f32x4*)dst)[i] = split4_pack(((const f32x4*)src)[i]);
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
// dhraw = dact * gelu'(hraw + b1) (in place on dact); thread t owns columns 4t..4t+3 of the 1024, a workgroup
// owns `rows_per_block` rows and writes colsum partials part[blk][1024] (fc1 bias gradient) when part != NULL.
#define GELU_BWD_ROWS 16   // rows per workgroup (rows_per_block of the launch must equal it), read in batches of 4
__global__ void __launch_bounds__(256, 4) k_gelu_bwd(const float* __restrict__ hraw, const float* __restrict__ b1,
                                                      float* __restrict__ dact, float* __restrict__ part, long nrows,
                                                      int split /* result in split-4 storage */) {
    const int t = threadIdx.x;
    const f32x4 bb = ((const f32x4*)b1)[t];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // dact is updated in place: a load issued after a store to the same array cannot be hoisted by the compiler, so the
    // rows go in batches of 4, every row of a batch read before the first is written (8 independent 16-byte loads per lane);
    // the ~75 registers leave six waves per SIMD to cover the batch boundaries
    for (long r0 = (long)blockIdx.x * GELU_BWD_ROWS; r0 < (long)(blockIdx.x + 1) * GELU_BWD_ROWS && r0 < nrows; r0 += 4) {
        f32x4 v[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = d[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r0 + k < nrows) {
                v[k] = ((const f32x4*)(hraw + (r0 + k) * CFFM_HID))[t];
                d[k] = ((const f32x4*)(dact + (r0 + k) * CFFM_HID))[t];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (r0 + k >= nrows) break;
            const f32x4 x = v[k] + bb;
            f32x4 o = d[k];
            o[0] *= gelu_erf_grad(x[0]); o[1] *= gelu_erf_grad(x[1]); o[2] *= gelu_erf_grad(x[2]); o[3] *= gelu_erf_grad(x[3]);
            ((f32x4*)(dact + (r0 + k) * CFFM_HID))[t] = split ? split4_pack(o) : o;
            acc += o;
            sched_fence();   // one row's arithmetic at a time: keeps the register count low
        }
    }
    if (part) ((f32x4*)(part + (long)blockIdx.x * CFFM_HID))[t] = acc;
}

// --------------------------------------------------------------------------- out = x1 + (oraw + b2)
__global__ void __launch_bounds__(256) k_residual_out(const float* __restrict__ x1, const float* __restrict__ oraw,
                                                       const float* __restrict__ b2, float* __restrict__ out, long n4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
        ((f32x4*)out)[e] = ((const f32x4*)x1)[e] + ((const f32x4*)oraw)[e] + ((const f32x4*)b2)[e % (CFFM_C / 4)];
}

// --------------------------------------------------------------------------- column sums (Linear bias grads)
// part[slice][c] = sum over the slice's rows of a[r][c]; grid (ncol/256, nslices); k_reduce_partials finishes.
#define COLSUM_ROWS 32   // rows per workgroup
__global__ void __launch_bounds__(256) k_colsum_partial(const float* __restrict__ a, long nrows, int ncol, float* __restrict__ part) {
    // a workgroup owns COLSUM_ROWS rows and every column: thread t sums the 4 columns 4t.. (16-byte loads, 8 rows in flight)
    const long r0 = (long)blockIdx.x * COLSUM_ROWS;
    for (int c4 = threadIdx.x; c4 < ncol / 4; c4 += 256) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < COLSUM_ROWS; k0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (r0 + k0 + k < nrows) ? ((const f32x4*)(a + (r0 + k0 + k) * ncol))[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        ((f32x4*)(part + (long)blockIdx.x * ncol))[c4] = acc;
    }
}


// a += b (f32x4)
__global__ void __launch_bounds__(256) k_add_inplace(float* __restrict__ a, const float* __restrict__ b, long n4) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
        ((f32x4*)a)[e] += ((const f32x4*)b)[e];
}

// out[i] = sum_s part[s*n + i] (f32x4; n multiple of 4): combines the split-K partial outputs of the weight-gradient GEMMs
// out[i] = sum over the nsplit slabs part[s][i] (n floats each, n % 4 == 0): split-K partial sums of the tiled GEMMs, the per-group
// bias-gradient tiles of the attention backward.  A workgroup covers 64 float4 columns; its four waves each take every fourth slab
// (two accumulators, independent loads) and the four partial sums are added in a fixed order through LDS -- rounds 1-3 gave a thread
// all the slabs of its column: 54 dependent-latency loads per thread on 152 workgroups for the 33.6 MB of bias-gradient tiles
// (22 us, 1.5 TB/s).  grid = ceil(n / 4 / 64).
__global__ void __launch_bounds__(256) k_sum_splits(const float* __restrict__ part, int nsplit, long n, float* __restrict__ out) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const bool on = i * 4 < n;
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a;
    if (on) {
        int s = q;
        for (; s + 4 < nsplit; s += 8) { a += ((const f32x4*)(part + (long)s * n))[i]; b += ((const f32x4*)(part + (long)(s + 4) * n))[i]; }
        if (s < nsplit) a += ((const f32x4*)(part + (long)s * n))[i];
    }
    a += b;
    if (q > 0) red[q - 1][lane] = a;
    __syncthreads();
    if (q == 0 && on) ((f32x4*)out)[i] = ((a + red[0][lane]) + red[1][lane]) + red[2][lane];
}

// the same for the outputs of a grouped weight-gradient launch (k_gemm_group_tt): one launch sums every problem's partials
struct SumGroup {
    const float* part[4];
    float* out[4];
    long n[4];          // floats per output (multiple of 4)
    int nsplit[4];
    int blk_end[4];     // exclusive prefix of 256-thread blocks per problem
    int cnt;
};
__global__ void __launch_bounds__(256) k_sum_splits_group(SumGroup G) {
    int blk = blockIdx.x, p = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q + 1 < G.cnt && blk >= G.blk_end[q]) p = q + 1;
    if (p > 0) blk -= G.blk_end[p - 1];
    const long i = (long)blk * 256 + threadIdx.x, n = G.n[p];
    if (i * 4 >= n) return;
    const float* part = G.part[p];
    const int nsplit = G.nsplit[p];
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a;
    int s = 0;
    for (; s + 1 < nsplit; s += 2) { a += ((const f32x4*)(part + (long)s * n))[i]; b += ((const f32x4*)(part + (long)(s + 1) * n))[i]; }
    if (s < nsplit) a += ((const f32x4*)(part + (long)s * n))[i];
    ((f32x4*)G.out[p])[i] = a + b;
}


// --------------------------------------------------------------------------- AdamW over a table of tensor chunks
// One workgroup per chunk (<= 2048 consecutive elements of one tensor): 7 streams of 4 B per element (p, g, m, v in;
// p, m, v out), every one a 16-byte access per lane when the chunk is 16-byte aligned.  A multi-tensor launch whose
// chunks are 64 K elements leaves most of the 256 CUs idle on a 1.6 M-parameter model; 2 K-element chunks give ~800
// workgroups and the update runs at the HBM rate.
struct AdamwChunk { float* p; const float* g; float* m; float* v; long n; };
__global__ void __launch_bounds__(256) k_adamw(const AdamwChunk* __restrict__ chunks, float decay /* 1 - lr*wd */, float beta1,
                                                float beta2, float omb1 /* 1 - beta1, rounded from double */, float omb2, float eps,
                                                float step_size /* lr / bc1 */, float inv_sqrt_bc2) {
    const AdamwChunk c = chunks[blockIdx.x];
    const bool vec = (((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0;
    const long n4 = vec ? c.n / 4 : 0;
    for (long e = threadIdx.x; e < n4; e += 256) {
        f32x4 p = ((const f32x4*)c.p)[e], m = ((const f32x4*)c.m)[e], v = ((const f32x4*)c.v)[e];
        const f32x4 g = ((const f32x4*)c.g)[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = beta1 * m[k] + omb1 * g[k];
            v[k] = beta2 * v[k] + omb2 * g[k] * g[k];
            p[k] = p[k] * decay - step_size * (m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps));
        }
        ((f32x4*)c.p)[e] = p; ((f32x4*)c.m)[e] = m; ((f32x4*)c.v)[e] = v;
    }
    for (long e = 4 * n4 + threadIdx.x; e < c.n; e += 256) {
        const float g = c.g[e];
        const float m = beta1 * c.m[e] + omb1 * g, v = beta2 * c.v[e] + omb2 * g * g;
        c.m[e] = m; c.v[e] = v;
        c.p[e] = c.p[e] * decay - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

// Graph-capturable variant: the step count and the two bias-correction factors live in device memory (state[0] = t as a
// float, state[1] = lr / (1 - b1^t), state[2] = 1 / sqrt(1 - b2^t)); k_adamw_tick advances them on the stream, so a captured
// training step replays with the right t every time.  `grad_base` != NULL: the chunk table's g field is a BYTE OFFSET from
// it (the layer's gradients share one buffer whose address may change from step to step -- the table never does).
__global__ void k_adamw_tick(float* __restrict__ state, double lr, double beta1, double beta2) {
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    state[1] = (float)(lr / (1.0 - pow(beta1, t)));
    state[2] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
}
__global__ void __launch_bounds__(256) k_adamw_dev(const AdamwChunk* __restrict__ chunks, const float* __restrict__ grad_base, float decay,
                                                    float beta1, float beta2, float omb1, float omb2, float eps,
                                                    const float* __restrict__ state) {
    AdamwChunk c = chunks[blockIdx.x];
    if (grad_base) c.g = (const float*)((const char*)grad_base + (uintptr_t)c.g);
    const float step_size = state[1], inv_sqrt_bc2 = state[2];
    const bool vec = (((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0;
    const long n4 = vec ? c.n / 4 : 0;
    for (long e = threadIdx.x; e < n4; e += 256) {
        f32x4 p = ((const f32x4*)c.p)[e], m = ((const f32x4*)c.m)[e], v = ((const f32x4*)c.v)[e];
        const f32x4 g = ((const f32x4*)c.g)[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = beta1 * m[k] + omb1 * g[k];
            v[k] = beta2 * v[k] + omb2 * g[k] * g[k];
            p[k] = p[k] * decay - step_size * (m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps));
        }
        ((f32x4*)c.p)[e] = p; ((f32x4*)c.m)[e] = m; ((f32x4*)c.v)[e] = v;
    }
    for (long e = 4 * n4 + threadIdx.x; e < c.n; e += 256) {
        const float g = c.g[e];
        const float m = beta1 * c.m[e] + omb1 * g, v = beta2 * c.v[e] + omb2 * g * g;
        c.m[e] = m; c.v[e] = v;
        c.p[e] = c.p[e] * decay - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

// Multi-group form (one launch for EVERY parameter group of the optimizer): a chunk names the ROW of the hyper-parameter
// tables it is updated with.  Per row: consts[ADAMW_NCONST] (double: beta1, beta2, eps, schedule kind, max_iters, power, min_lr,
// warmup_iters, warmup_ratio, first iteration, and -- slot 10, the one WRITTEN here -- the optimizer's global iteration count), sched[2] (float: base lr, weight decay -- a device copy of a pinned host mirror,
// refreshed by a copy that is part of the captured step), state[4] (float: t, lr_t / (1 - b1^t), 1 / sqrt(1 - b2^t),
// 1 - lr_t wd; advanced by k_adamw_tick_rows).  Schedule kind 1 evaluates the reference's learning-rate schedule ON THE DEVICE
// from the step count (mmcv PolyLrUpdaterHook with linear warm-up: local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:41-45),
// so replayed graphs follow it with no host involvement at all (a host write to the mirror races with the previous replay's copy
// unless the caller orders it); kind 0 takes lr_t = the mirror's value.
#define ADAMW_NCONST 12
struct AdamwChunk2 { float* p; const float* g; float* m; float* v; int n; int row; };
// `active` (or NULL = every row): rows that own a chunk of THIS step's table.  A row whose parameters got no gradient this step keeps
// its step count -- torch's per-parameter step does not advance either -- so its bias correction and schedule iteration stay right
// when it next takes part (a parameter with gradients on 3 of 8 steps was off by 3.8e-2 when every row ticked).
// b^t for an integer-valued t >= 0 by squaring: ~40 multiplications where pow() is several hundred double-precision instructions --
// with the tick folded into the update launch every workgroup evaluates this on ONE lane before its threads can start (libm pow: the
// update launch took 27 us instead of 11).  Within ~1e-14 of pow(), far below the float rounding of the factors derived from it.
__device__ __forceinline__ double adamw_ipow(double b, double t) {
    long n = (long)t;
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}
// the four state values of row r after its next step (t, lr_t / (1 - b1^t), 1 / sqrt(1 - b2^t), 1 - lr_t wd)
__device__ __forceinline__ void adamw_next_state(const float* __restrict__ state, const float* __restrict__ sched, const double* __restrict__ consts,
                                                 int r, float out[4]) {
    const double* c = consts + (long)ADAMW_NCONST * r;
    const double t = (double)state[4 * r] + 1.0, wd = (double)sched[2 * r + 1];
    double lr = (double)sched[2 * r];
    if (c[3] == 1.0) {                      // poly decay with linear warm-up.  The iteration is the OPTIMIZER's (c[10]: steps taken so far,
                                            // advanced for every row on every step), not the row's own count t: mmcv derives every group's rate
                                            // from the runner's iteration, so a parameter that sits steps out, or first gets a gradient late
                                            // (find_unused_parameters=True), neither lags behind nor restarts the warm-up
        const double it = c[10] - c[9], max_iters = c[4], min_lr = c[6], wi = c[7];
        const double frac = it < max_iters ? 1.0 - it / max_iters : 0.0;
        lr = (lr - min_lr) * (c[5] == 1.0 ? frac : pow(frac, c[5])) + min_lr;
        if (it < wi) lr *= 1.0 - (1.0 - it / wi) * (1.0 - c[8]);
        if (it < 0.0) lr = 0.0;
    }
    out[0] = (float)t;
    out[1] = (float)(lr / (1.0 - adamw_ipow(c[0], t)));
    out[2] = (float)(1.0 / sqrt(1.0 - adamw_ipow(c[1], t)));
    out[3] = (float)(1.0 - lr * wd);
}
__global__ void k_adamw_tick_rows(float* __restrict__ state, const float* __restrict__ sched, double* __restrict__ consts, int nrows,
                                  const int* __restrict__ active) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    if (!active || active[r]) {
        float o[4];
        adamw_next_state(state, sched, consts, r, o);
        state[4 * r] = o[0]; state[4 * r + 1] = o[1]; state[4 * r + 2] = o[2]; state[4 * r + 3] = o[3];
    }
    consts[ADAMW_NCONST * r + 10] += 1.0;        // the global iteration: every row, active or not
}
// one element of the update; no FMA contraction, so that every caller rounds the same way whatever the surrounding code looks like
// (the prefetching form of k_adamw_rows_tick and the loop below must agree bit for bit)
__device__ __forceinline__ void adamw_elem(float& p, float& m, float& v, float g, float beta1, float omb1, float beta2, float omb2, float decay,
                                           float step_size, float inv_sqrt_bc2, float eps) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    m = beta1 * m + omb1 * g;
    v = beta2 * v + omb2 * g * g;
    p = p * decay - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}
__device__ __forceinline__ void adamw_update_chunk(AdamwChunk2 c, const float* __restrict__ grad_base, float step_size, float inv_sqrt_bc2,
                                                   float decay, const double* __restrict__ consts) {
    if (grad_base) c.g = (const float*)((const char*)grad_base + (uintptr_t)c.g);
    const double b1 = consts[ADAMW_NCONST * c.row], b2 = consts[ADAMW_NCONST * c.row + 1];
    const float beta1 = (float)b1, beta2 = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), eps = (float)consts[ADAMW_NCONST * c.row + 2];
    const bool vec = (((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0;
    const int n4 = vec ? c.n / 4 : 0;
    for (int e = threadIdx.x; e < n4; e += 256) {
        f32x4 p = ((const f32x4*)c.p)[e], m = ((const f32x4*)c.m)[e], v = ((const f32x4*)c.v)[e];
        const f32x4 g = ((const f32x4*)c.g)[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adamw_elem(pk, mk, vk, g[k], beta1, omb1, beta2, omb2, decay, step_size, inv_sqrt_bc2, eps);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
        ((f32x4*)c.p)[e] = p; ((f32x4*)c.m)[e] = m; ((f32x4*)c.v)[e] = v;
    }
    for (int e = 4 * n4 + threadIdx.x; e < c.n; e += 256) {
        float pk = c.p[e], mk = c.m[e], vk = c.v[e];
        adamw_elem(pk, mk, vk, c.g[e], beta1, omb1, beta2, omb2, decay, step_size, inv_sqrt_bc2, eps);
        c.m[e] = mk; c.v[e] = vk; c.p[e] = pk;
    }
}
__global__ void __launch_bounds__(256) k_adamw_rows(const AdamwChunk2* __restrict__ chunks, const float* __restrict__ grad_base,
                                                     const float* __restrict__ state, const double* __restrict__ consts) {
    const AdamwChunk2 c = chunks[blockIdx.x];
    adamw_update_chunk(c, grad_base, state[4 * c.row + 1], state[4 * c.row + 2], state[4 * c.row + 3], consts);
}
// The same update with the step-count tick folded in (one launch instead of two on the tail of a training step): every workgroup
// derives its row's factors from the state BEFORE the step (adamw_next_state: the arithmetic of k_adamw_tick_rows, so the results are
// bit-identical); the workgroup that draws the last ticket -- all others have read the state by then -- stores the advanced state of
// the active rows; the tickets are back at zero for the next launch.
__global__ void __launch_bounds__(256) k_adamw_rows_tick(const AdamwChunk2* __restrict__ chunks, const float* __restrict__ grad_base,
                                                          float* __restrict__ state, const float* __restrict__ sched,
                                                          double* __restrict__ consts, int nrows, const int* __restrict__ active,
                                                          int* __restrict__ ticket) {
    __shared__ float sc[4];
    __shared__ int last;
    AdamwChunk2 c = chunks[blockIdx.x];
    const bool full = c.n == 2048 && (((uintptr_t)c.p | (uintptr_t)(grad_base ? (const float*)((const char*)grad_base + (uintptr_t)c.g) : c.g) | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0;
    if (full) {
        // a whole aligned chunk (all but the last of a tensor): its 8 KB per operand are requested BEFORE lane 0 works out the row's
        // factors (double-precision divisions and a square root on one lane: ~2 us that every thread of the workgroup would wait out)
        if (grad_base) c.g = (const float*)((const char*)grad_base + (uintptr_t)c.g);
        const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
        f32x4 p0 = ((const f32x4*)c.p)[e0], m0 = ((const f32x4*)c.m)[e0], v0 = ((const f32x4*)c.v)[e0], p1 = ((const f32x4*)c.p)[e1],
              m1 = ((const f32x4*)c.m)[e1], v1 = ((const f32x4*)c.v)[e1];
        const f32x4 g0 = ((const f32x4*)c.g)[e0], g1 = ((const f32x4*)c.g)[e1];
        if (threadIdx.x == 0) adamw_next_state(state, sched, consts, c.row, sc);
        __syncthreads();
        const float step_size = sc[1], inv_sqrt_bc2 = sc[2], decay = sc[3];
        const double b1 = consts[ADAMW_NCONST * c.row], b2 = consts[ADAMW_NCONST * c.row + 1];
        const float beta1 = (float)b1, beta2 = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), eps = (float)consts[ADAMW_NCONST * c.row + 2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p0[k], mk = m0[k], vk = v0[k];
            adamw_elem(pk, mk, vk, g0[k], beta1, omb1, beta2, omb2, decay, step_size, inv_sqrt_bc2, eps);
            p0[k] = pk; m0[k] = mk; v0[k] = vk;
            pk = p1[k]; mk = m1[k]; vk = v1[k];
            adamw_elem(pk, mk, vk, g1[k], beta1, omb1, beta2, omb2, decay, step_size, inv_sqrt_bc2, eps);
            p1[k] = pk; m1[k] = mk; v1[k] = vk;
        }
        ((f32x4*)c.p)[e0] = p0; ((f32x4*)c.m)[e0] = m0; ((f32x4*)c.v)[e0] = v0;
        ((f32x4*)c.p)[e1] = p1; ((f32x4*)c.m)[e1] = m1; ((f32x4*)c.v)[e1] = v1;
    } else {
        if (threadIdx.x == 0) adamw_next_state(state, sched, consts, c.row, sc);
        __syncthreads();
        adamw_update_chunk(c, grad_base, sc[1], sc[2], sc[3], consts);
    }
    // (no fence: the state values were consumed above, so those loads are complete before the ticket is drawn, and nothing this
    //  workgroup wrote is read by another one -- a device-scope fence here would write back the XCD's whole L2 per workgroup: 80 us)
    // Tickets in two levels (ticket[1 + slot], slot = workgroup % 64, then ticket[0]): ~2000 device-scope atomics on ONE address cost
    // more than the update itself (the launch took 25 us instead of 11).
    if (threadIdx.x == 0) {
        const int slot = blockIdx.x & 63, quota = ((int)gridDim.x - slot + 63) / 64, nslots = (int)gridDim.x < 64 ? (int)gridDim.x : 64;
        int l = 0;
        if (atomicAdd(ticket + 1 + slot, 1) == quota - 1) {
            ticket[1 + slot] = 0;
            l = atomicAdd(ticket, 1) == nslots - 1;
        }
        last = l;
    }
    __syncthreads();
    if (!last) return;
    for (int r = threadIdx.x; r < nrows; r += 256) {
        if (!active || active[r]) {
            float o[4];
            adamw_next_state(state, sched, consts, r, o);
            state[4 * r] = o[0]; state[4 * r + 1] = o[1]; state[4 * r + 2] = o[2]; state[4 * r + 3] = o[3];
        }
        consts[ADAMW_NCONST * r + 10] += 1.0;    // the global iteration: every row, active or not
    }
    if (threadIdx.x == 0) *ticket = 0;
}
