// cffm_common.h -- shared device helpers for the CFFM hot-path kernels (gfx950 / CDNA4).
#pragma once
#ifdef CFFM_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// Tuning switches (launch shapes, stream forks, rejected forms kept for A/B runs) are read from the environment only in
// -DCFFM_EXPERIMENTS builds (scripts/): the product library is built without it, and there every switch IS its default -- what
// bench.py and the -m gpu tests run is the only configuration libcffm_hip.so has.  (The one exception is documented where it is read:
// CFFM_DW_GROUP, the two legitimate forms of the weight-gradient groups, both covered by tests.)
#include <stdlib.h>
#ifdef CFFM_EXPERIMENTS
static inline const char* cffm_tune(const char* name) { return getenv(name); }
#else
static inline const char* cffm_tune(const char*) { return nullptr; }
#endif

// ---- fixed problem constants of the reference head (cffm_head.py:74-95) ------------------------
#define CFFM_C 256         // embed_dim of every CFFM config
#define CFFM_HEADS 8
#define CFFM_HD 32
#define CFFM_WS 7
#define CFFM_WA 49         // window area
#define CFFM_NKEY 289      // 49 own + 132 ring + 25 + 49 + 25 + 9
#define CFFM_NKEY_PAD 304  // 19 MFMA key tiles of 16
#define CFFM_NQ_PAD 64
#define CFFM_NCELL 15      // pooled cells per window: 1 (target) + 1 + 4 + 9
#define CFFM_HID 1024
// the dense position bias of a block as f16 MFMA B-operand fragments [8 heads][4 waves][10 key-tile pairs][64 lanes][8] (rowops_kernels.h
// bias_assemble_body, cfm_attn_kernels.h bias_sel_frag)
#define BIASH_HALFS (CFFM_HEADS * 4 * 10 * 512)
#define CFFM_LN_EPS 1e-5f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#if defined(CFFM_EMU) && defined(CFFM_EMU_F32)
typedef float f16;  // emulator-only switch: MFMA operands keep fp32, to separate logic errors from f16 rounding
#else
typedef _Float16 f16;
#endif
typedef _Float16 h16;                                       // f16 as stored in global memory (q/k/v)
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef CFFM_EMU
#define CFFM_DYN_SMEM(name) char* name = emu::g_blk->dynsmem
#define CFFM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3 grid, dim3 block, (shmem), [=]() { kernel(__VA_ARGS__); })
#else
#define CFFM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#ifdef CFFM_NULL_LAUNCH   // profiling builds only: no kernel is launched (host-side cost of a step without the device work)
#define CFFM_LAUNCH(kernel, grid, block, shmem, stream, ...) (void)0
#else
#define CFFM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3 grid, dim3 block, (shmem), (stream), __VA_ARGS__)
#endif
#endif

// ---- wave (64 lanes) reductions ------------------------------------------------------------------
// Result in every lane.  On the GPU: 4 DPP steps inside each 16-lane row (quad swaps, half-mirror, mirror -- every
// lane ends up with its row's total), then the 4 row totals are read with v_readlane and added: ~12 cheap VALU/SALU
// instructions.  (The portable __shfl_xor butterfly lowers to 6 dependent ds_bpermute round trips through the LDS
// crossbar -- it made the LayerNorm / pooling kernels latency-bound.)
#ifndef CFFM_EMU
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
#endif
// the wave's sum, valid in lanes 48..63 ONLY (the other lanes hold partial sums): four DPP steps inside each 16-lane row, then row 0 / 2's
// total goes to rows 1 / 3 (row_bcast:15) and the lower half's to the upper (row_bcast:31) -- six VALU instructions; the round 1-4 form
// read the four row totals with v_readlane and added them (eleven)
__device__ __forceinline__ float wave_sum_hi(float v) {
#ifdef CFFM_EMU
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
#else
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);   // row_half_mirror
    v += dpp_f32<0x140>(v);   // row_mirror
    // (spelled out: from the update_dpp builtin with a row mask hipcc makes v_mov 0 + v_mov_dpp + v_add, three instructions per step;
    //  a DPP source written by the previous VALU instruction needs two wait states, which the hazard recognizer does not insert
    //  inside inline assembly)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
#endif
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef CFFM_EMU
    return wave_sum_hi(v);
#else
    return readlane_f32(wave_sum_hi(v), 63);
#endif
}
// sum over each aligned group of 16 lanes (a DPP "row"), result in every lane of the group
__device__ __forceinline__ float row16_sum(float v) {
#ifdef CFFM_EMU
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
#else
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef CFFM_EMU
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
#else
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
#endif
}

// ---- MFMA wrappers (CDNA4). Fragment maps used throughout (cdna_hip_programming.md section 3):
//   16x16x32 f16:  A: lane l holds A[i = l&15][k = 8*(l>>4) + 0..7]
//                  B: lane l holds B[k = 8*(l>>4) + 0..7][j = l&15]
//                  C/D: reg r of lane l is C[row = 4*(l>>4) + r][col = l&15]
//   Only the (l&15) <-> i/j maps and the C/D map matter for correctness: any bijection between the
//   8 k-slots of a lane group and actual k indices gives the same product as long as A and B use
//   the same one -- the PV / dK / dV contractions below exploit exactly that.
//   16x16x4 f32:   A: lane l holds A[i = l&15][k = l>>4];  B: B[k = l>>4][j = l&15];  C/D as above.
__device__ __forceinline__ f32x4 mfma16x16x32_f16(f16x8 a, f16x8 b, f32x4 c) {
#ifdef CFFM_EMU
    struct Frag { float a[8], b[8]; } mine;
    for (int j = 0; j < 8; ++j) { mine.a[j] = (float)a[j]; mine.b[j] = (float)b[j]; }
    int lane = emu::lane_linear() & 63;
    auto s = emu::deposit(&mine, sizeof(mine));
    int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            const Frag* fa = reinterpret_cast<const Frag*>(s[row + 16 * kg]);
            const Frag* fb = reinterpret_cast<const Frag*>(s[col + 16 * kg]);
            for (int j = 0; j < 8; ++j) acc += fa->a[j] * fb->b[j];
        }
        c[r] = acc;
    }
    emu::release();
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ f32x4 mfma16x16x4_f32(float a, float b, f32x4 c) {
#ifdef CFFM_EMU
    struct Frag { float a, b; } mine{a, b};
    int lane = emu::lane_linear() & 63;
    auto s = emu::deposit(&mine, sizeof(mine));
    int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const Frag* fa = reinterpret_cast<const Frag*>(s[row + 16 * k]);
            const Frag* fb = reinterpret_cast<const Frag*>(s[col + 16 * k]);
            acc = fmaf(fa->a, fb->b, acc);
        }
        c[r] = acc;
    }
    emu::release();
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// ---- bf16 hi/lo split of fp32 (operands of the Linear GEMMs) ------------------------------------------------------
typedef __bf16 bf16;
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
// "split-4" storage: 4 consecutive floats kept as the 16 bytes {bf16 hi x4, bf16 lo x4} -- exactly what split4() makes of
// them.  Tensors that only GEMMs read (the LayerNorm / GELU outputs zall, z2, act, the MLP gradient dh) and a copy of the
// weights are written in this format by their producers, so staging a tile of them is a plain copy: the split's VALU work
// (the bound of these kernels) is paid once per element instead of once per tile that re-reads it.
// (register-only bit casts: a float[4] temporary here gets "promoted" to LDS by the compiler and costs a round trip)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void unsplit4(f32x4 raw, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_bit_cast(bf16x4, (f32x2_t){raw[0], raw[1]});
    lo = __builtin_bit_cast(bf16x4, (f32x2_t){raw[2], raw[3]});
}
__device__ __forceinline__ void split4(f32x4 x, bf16x4& hi, bf16x4& lo);
__device__ __forceinline__ f32x4 split4_pack(f32x4 x) {
    bf16x4 h, l;
    split4(x, h, l);
    const f32x2_t a = __builtin_bit_cast(f32x2_t, h), b = __builtin_bit_cast(f32x2_t, l);
    return (f32x4){a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ void split4(f32x4 x, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16)x[e];
        lo[e] = (bf16)(x[e] - (float)hi[e]);
    }
}


// ---- raw buffer loads ------------------------------------------------------------------------------------------
// A buffer resource carries the extent of an array: a load whose byte offset (per-lane VGPR part + uniform SGPR part) falls
// outside returns zeros instead of faulting, so gathers with "no such row" entries need neither a branch nor 64-bit
// address arithmetic -- the lane passes BUF_OOB as its offset.  (Range check: vgpr_offset >= num_records - sgpr_offset.)
#define BUF_OOB 0xFFFFFFF0u
#ifdef CFFM_EMU
struct buf_t { const char* p; uint32_t n; };
static inline buf_t buf_make(const void* p, uint32_t bytes) { return buf_t{(const char*)p, bytes}; }
static inline void buf_ld_bytes(buf_t r, uint32_t voff, uint32_t soff, void* dst, int ndw) {
    for (int e = 0; e < ndw; ++e) {
        const uint64_t o = (uint64_t)voff + 4 * e;
        uint32_t w = 0;
        if (soff <= r.n && o + 4 <= (uint64_t)(r.n - soff)) __builtin_memcpy(&w, r.p + soff + o, 4);
        __builtin_memcpy((char*)dst + 4 * e, &w, 4);
    }
}
static inline f32x4 buf_ld16(buf_t r, uint32_t voff, uint32_t soff) {
    float t[4];
    buf_ld_bytes(r, voff, soff, t, 4);
    return (f32x4){t[0], t[1], t[2], t[3]};
}
static inline float buf_ld4(buf_t r, uint32_t voff, uint32_t soff) {
    float t;
    buf_ld_bytes(r, voff, soff, &t, 1);
    return t;
}
static inline f32x2 buf_ld8(buf_t r, uint32_t voff, uint32_t soff) {
    float t[2];
    buf_ld_bytes(r, voff, soff, t, 2);
    return (f32x2){t[0], t[1]};
}
static inline void buf_st16(buf_t r, f32x4 v, uint32_t voff, uint32_t soff) {   // out-of-range stores are dropped
    if (soff <= r.n && (uint64_t)voff + 16 <= (uint64_t)(r.n - soff)) __builtin_memcpy(const_cast<char*>(r.p) + soff + voff, &v, 16);
}
static inline void buf_st16_pair(buf_t r, f32x4 v0, f32x4 v1, uint32_t voff, uint32_t soff0, uint32_t soff1) {
    buf_st16(r, v0, voff, soff0);
    buf_st16(r, v1, voff, soff1);
}
static inline void buf_st4(buf_t r, float v, uint32_t voff, uint32_t soff) {
    if (soff <= r.n && (uint64_t)voff + 4 <= (uint64_t)(r.n - soff)) __builtin_memcpy(const_cast<char*>(r.p) + soff + voff, &v, 4);
}
static inline void buf_st8(buf_t r, f32x2 v, uint32_t voff, uint32_t soff) {
    if (soff <= r.n && (uint64_t)voff + 8 <= (uint64_t)(r.n - soff)) __builtin_memcpy(const_cast<char*>(r.p) + soff + voff, &v, 8);
}
#else
typedef __amdgpu_buffer_rsrc_t buf_t;
__device__ __forceinline__ buf_t buf_make(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld16(buf_t r, uint32_t voff, uint32_t soff) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    f32x4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}
__device__ __forceinline__ float buf_ld4(buf_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 buf_ld8(buf_t r, uint32_t voff, uint32_t soff) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __builtin_bit_cast(f32x2, v);
}
// 16-byte store: per-lane VGPR offset + uniform (scalar) offset -- no 64-bit per-lane address arithmetic, which the compiler
// otherwise hoists out of unrolled loops as one VGPR pair per distinct destination
// HAZARD (measured on MI355X, ROCm 7.2): a buffer_store_dwordx4 with an SGPR offset still reads its data VGPRs for a cycle
// or two after it issues; hipcc scheduled a v_pk_mul_f32 that overwrote them right behind the store and the stored tile was
// corrupt (the same kernel with waterfall loops around the store -- i.e. other instructions in between -- was correct).
// buf_st16_pair therefore issues its stores between two scheduling barriers with an s_nop behind them.
__device__ __forceinline__ void buf_st16(buf_t r, f32x4 v, uint32_t voff, uint32_t soff) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 w;
    __builtin_memcpy(&w, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st4(buf_t r, float v, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st8(buf_t r, f32x2 v, uint32_t voff, uint32_t soff) {   // (8-byte stores are not subject to the hazard)
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 w;
    __builtin_memcpy(&w, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st16_pair(buf_t r, f32x4 v0, f32x4 v1, uint32_t voff, uint32_t soff0, uint32_t soff1) {
    asm volatile("" : "+v"(v0), "+v"(v1));          // both tiles sit in registers of their own before the first store issues
    __builtin_amdgcn_sched_barrier(0);
    buf_st16(r, v0, voff, soff0);
    buf_st16(r, v1, voff, soff1);
    asm volatile("s_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#endif
// LDS-DMA (gfx950 `buffer_load_dwordx4 ... lds`): 16 bytes per lane go from global memory straight into LDS, without staging
// registers and without a ds_write pass.  The destination is WAVE-UNIFORM base + 16 * lane (no per-lane scatter: layouts are
// permuted through the per-lane SOURCE offset); an out-of-range source offset writes zeros.  Completion is counted on vmcnt;
// other waves may read the data after the issuing wave's wait + a barrier.
#ifdef CFFM_EMU
static inline void buf_ld16_lds(buf_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const int lane = emu::lane_linear() & 63;
    buf_ld_bytes(r, voff, soff, (char*)lds_wave_base + 16 * lane, 4);
}
#else
__device__ __forceinline__ void buf_ld16_lds(buf_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
#endif
// the same, 4 bytes per lane (a row of 64 floats)
#ifdef CFFM_EMU
static inline void buf_ld4_lds(buf_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const int lane = emu::lane_linear() & 63;
    buf_ld_bytes(r, voff, soff, (char*)lds_wave_base + 4 * lane, 1);
}
#else
__device__ __forceinline__ void buf_ld4_lds(buf_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, soff, 0, 0);
}
#endif
// LDS-DMA the compiler does not see (inline assembly, no memory clobber).  hipcc models the builtin above as a store to LDS of unknown
// extent: behind it, the first LDS read it cannot prove disjoint -- the transposed reads of the OTHER buffer of a double-buffered
// pipeline -- gets an s_waitcnt vmcnt(0) in front, i.e. the multiplication waits for the gather it was meant to hide (measured in the
// key-owner kernel of the attention backward).  The caller orders these by hand: wait_vm0() + a barrier before anything reads the
// destination, and a barrier between the last read of a buffer and the DMA that refills it.
#ifdef CFFM_EMU
typedef buf_t dma_t;
static inline dma_t dma_make(const void* p, uint32_t bytes) { return buf_make(p, bytes); }
static inline void dma_ld16(dma_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) { buf_ld16_lds(r, voff, soff, lds_wave_base); }
static inline void dma_ld4(dma_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) { buf_ld4_lds(r, voff, soff, lds_wave_base); }
#else
typedef int dma_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_t dma_make(const void* p, uint32_t bytes) {      // raw buffer descriptor: base, stride 0, extent, gfx9 flags
    const uint64_t a = (uint64_t)p;
    return (dma_t){(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma_ld16(dma_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(l), "v"(voff), "s"(r), "s"(soff) : "m0");
}
__device__ __forceinline__ void dma_ld4(dma_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(l), "v"(voff), "s"(r), "s"(soff) : "m0");
}
#endif
// 8 stored halfs through a buffer resource (zeros when out of range)
__device__ __forceinline__ f16x8 buf_ld_h8(buf_t r, uint32_t voff, uint32_t soff) {
    const f32x4 raw = buf_ld16(r, voff, soff);
    h16x8 v;
    __builtin_memcpy(&v, &raw, 16);
    f16x8 o;
    for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
    return o;
}

// 8 stored halfs (16 B) -> 8 MFMA-operand elements (a no-op copy unless the emulator's fp32-operand switch is on)
__device__ __forceinline__ f16x8 ld_h8(const h16* p) {
    const h16x8 v = *(const h16x8*)p;
    f16x8 r;
    for (int e = 0; e < 8; ++e) r[e] = (f16)v[e];
    return r;
}
__device__ __forceinline__ f16x4 to_f16x4(f32x4 v) {
    f16x4 r;
    r[0] = (f16)v[0]; r[1] = (f16)v[1]; r[2] = (f16)v[2]; r[3] = (f16)v[3];
    return r;
}
__device__ __forceinline__ f16x8 cat_f16x4(f16x4 lo, f16x4 hi) {
    f16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// LDS transpose read (gfx950 `ds_read_b64_tr_b16`): every lane passes the address of 4 contiguous 16-bit elements; inside
// each group of 16 lanes the 16 x 4 elements are transposed: lane l receives element (l & 3) of the lanes 4j + ((l & 15) >> 2),
// j = 0..3, of its group (measured on MI355X with the lanes' quads laid end to end as a 64-element block: lane l gets block
// elements l, l+16, l+32, l+48).  With lane i pointing at row (i >> 2), columns 4 (i & 3) .. +3 of a [4 rows][16 columns]
// block, lane l gets column (l & 15) of the 4 rows: row-major K / V / Q / dO images feed the MFMA operands that need the
// transposed view, so no transposed copy is ever written.  Wave-collective (all 64 lanes must call it).
__device__ __forceinline__ f16x4 lds_tr4(const f16* p) {
#ifdef CFFM_EMU
    struct Quad { f16 e[4]; } mine = {{p[0], p[1], p[2], p[3]}};
    const int lane = emu::lane_linear() & 63;
    auto s = emu::deposit(&mine, sizeof(mine));
    f16x4 r;
    for (int j = 0; j < 4; ++j) r[j] = reinterpret_cast<const Quad*>(s[(lane & 48) + 4 * j + ((lane & 15) >> 2)])->e[lane & 3];
    emu::release();
    return r;
#else
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
    f16x4 r;
    __builtin_memcpy(&r, &v, 8);
    return r;
#endif
}

// the same for bf16 data (operand images of the Linear GEMMs)
__device__ __forceinline__ bf16x4 lds_tr4_bf16(const bf16* p) {
#ifdef CFFM_EMU
    struct Quad { bf16 e[4]; } mine = {{p[0], p[1], p[2], p[3]}};
    const int lane = emu::lane_linear() & 63;
    auto s = emu::deposit(&mine, sizeof(mine));
    bf16x4 r;
    for (int j = 0; j < 4; ++j) r[j] = reinterpret_cast<const Quad*>(s[(lane & 48) + 4 * j + ((lane & 15) >> 2)])->e[lane & 3];
    emu::release();
    return r;
#else
    typedef __bf16 bf16x4_t __attribute__((__vector_size__(4 * sizeof(__bf16))));
    const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p);
    return __builtin_bit_cast(bf16x4, v);
#endif
}

// ordering point between LDS writes and reads of *other lanes of the same wave* (wave-private LDS
// scratch; LDS operations of one wave execute in order, this only pins the compiler / the emulator)
__device__ __forceinline__ void wave_lds_sync() {
#ifdef CFFM_EMU
    emu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// streaming (non-temporal) 16-byte store / load: data written once and read once by a later kernel must not
// push the L2-resident gather working set (one head's q/k/v slice per XCD) out of the 4 MiB L2
__device__ __forceinline__ void st4_stream(float* p, f32x4 v) {
#ifdef CFFM_EMU
    *(f32x4*)p = v;
#else
    __builtin_nontemporal_store(v, (f32x4*)p);
#endif
}
__device__ __forceinline__ f32x4 ld4_stream(const float* p) {
#ifdef CFFM_EMU
    return *(const f32x4*)p;
#else
    return __builtin_nontemporal_load((const f32x4*)p);
#endif
}
// keeps the instruction scheduler from interleaving across this point (bounds register pressure of unrolled loops)
__device__ __forceinline__ void sched_fence() {
#ifndef CFFM_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ float fast_exp(float x) {
#ifdef CFFM_EMU
    return expf(x);
#else
    return __expf(x);
#endif
}

// A zero the compiler cannot see through: `p + opaque_zero()` is an address it must treat as new at this point, so loads through it are
// not hoisted out of the surrounding (unrolled) loop and kept in registers -- used where re-reading LDS per iteration is cheaper than the
// registers the hoisted values would occupy.  (Laundering the POINTER itself loses its LDS address space: hipcc then emits flat accesses
// and an illegal V_CMP on src_shared_base.)
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
#ifndef CFFM_EMU
    asm volatile("" : "+v"(z));
#endif
    return z;
}

// a value that is the same in every lane of the wave, handed to the compiler as such (an SGPR): buffer-instruction scalar
// offsets and resource descriptors derived from threadIdx.x >> 6 otherwise become "waterfall" loops over the lanes
__device__ __forceinline__ int wave_uniform(int v) {
#ifdef CFFM_EMU
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// 2^x as one v_exp_f32 (callers fold the log2(e) factor into an FMA they need anyway: exp(s - m) = 2^(s*log2e - m*log2e))
#define CFFM_LOG2E 1.4426950408889634f
__device__ __forceinline__ float fast_exp2(float x) {
#ifdef CFFM_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}

// GELU (erf form, cffm_transformer.py:14 nn.GELU) and its derivative.  Phi(x) = 0.5 erfc(-x / sqrt 2) with
// erfc(y) = (a1 t + ... + a5 t^5) exp(-y^2), t = 1 / (1 + p y), y >= 0 (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 --
// three orders below the 1e-3 contract): ~14 instructions with the hardware reciprocal / exp2 instead of ~35 for erff();
// the same exp(-x^2/2) is the Gaussian of the derivative.  The fc1 epilogue evaluates 64 of these per thread.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& gauss /* exp(-x^2/2) */) {
    const float y = fabsf(x) * 0.70710678118654752f;
#ifdef CFFM_EMU
    const float t = 1.0f / (1.0f + 0.3275911f * y);
#else
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * y);
#endif
    gauss = fast_exp(-y * y);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half_erfc = 0.5f * poly * gauss;            // Phi(-|x|)
    cdf = x < 0.f ? half_erfc : 1.0f - half_erfc;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, gs;
    gelu_parts(x, cdf, gs);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, gs;
    gelu_parts(x, cdf, gs);
    return cdf + x * (0.39894228040143268f * gs);
}
