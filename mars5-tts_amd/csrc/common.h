// Shared device helpers for the MARS5 gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mars5_hip.h"

#define M5_WAVE 64

// Environment knobs (tile-configuration sweeps, first-generation kernels for A/B runs, timing ablations that compute
// WRONG results, phase clocks) exist only in the tools build (-DM5_TOOLS -> libmars5_hip_tools.so, used by tools/*.py).
// The product library never reads the environment: a stray variable cannot change what it computes.
#ifdef M5_TOOLS
#include <stdlib.h>
static inline const char* m5_tool_env(const char* name) { return getenv(name); }
#else
static inline const char* m5_tool_env(const char*) { return nullptr; }
#endif

// ---- element types ----------------------------------------------------------------
// F32T: exact fp32 operands (parity mode; MFMA 16x16x4 f32).  F16T / BF16T: 16-bit
// operands, fp32 accumulate (MFMA 16x16x32).
struct F32T {
    using storage = float;
    static constexpr int id = M5_F32;
    static constexpr int EPL = 4;  // elements per 16-byte load
    __device__ static inline float to_f32(storage v) { return v; }
    __device__ static inline storage from_f32(float f) { return f; }
};
struct F16T {
    using storage = _Float16;
    static constexpr int id = M5_F16;
    static constexpr int EPL = 8;
    __device__ static inline float to_f32(storage v) { return (float)v; }
    __device__ static inline storage from_f32(float f) { return (_Float16)f; }
};
struct BF16T {
    using storage = uint16_t;
    static constexpr int id = M5_BF16;
    static constexpr int EPL = 8;
    __device__ static inline float to_f32(storage v) { return __uint_as_float(((uint32_t)v) << 16); }
    __device__ static inline storage from_f32(float f) {     // v_cvt_pk_bf16_f32: round to nearest even
        const __bf16 b = (__bf16)f;
        return *reinterpret_cast<const uint16_t*>(&b);
    }
};

template <typename T>
__device__ inline float round_dt(float f) { return T::to_f32(T::from_f32(f)); }

// 16-byte vector of T::EPL elements, converted to fp32
template <typename T>
struct Vec16 {
    uint4 raw;
    __device__ inline void load(const void* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ inline void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ inline void to_float(float* out) const {
        const typename T::storage* e = reinterpret_cast<const typename T::storage*>(&raw);
#pragma unroll
        for (int i = 0; i < T::EPL; ++i) out[i] = T::to_f32(e[i]);
    }
};

// ---- eight 16-bit products into one fp32 accumulator ---------------------------------------------------------------------
// acc += sum_e w[e] x[e] over the 8 element pairs of two 16-byte operands, as FOUR v_dot2c_f32_{bf16,f16}
// (acc = w.lo x.lo + w.hi x.hi + acc), pairs in memory order.  Both forms of the AR decode step (ar_decode.hip's
// gemv_stream_kernel and ar_mega.hip's dot_rows) call this on the same operands in the same order: that is what keeps them
// bit-identical.  Before round 4 this was 8 unpack + 8 fmaf per 16 bytes and operand; the activation vector now lives in LDS
// in the operand type (it always held operand-rounded values), one ds_read_b128 per 8 elements instead of two.
typedef __attribute__((ext_vector_type(2))) __bf16 m5_bf2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 m5_h2_t;
template <typename T>
__device__ inline float dot8(const uint4& w, const uint4& x, float acc);
template <>
__device__ inline float dot8<BF16T>(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const m5_bf2_t*>(&w.x), *reinterpret_cast<const m5_bf2_t*>(&x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const m5_bf2_t*>(&w.y), *reinterpret_cast<const m5_bf2_t*>(&x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const m5_bf2_t*>(&w.z), *reinterpret_cast<const m5_bf2_t*>(&x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const m5_bf2_t*>(&w.w), *reinterpret_cast<const m5_bf2_t*>(&x.w), acc, false);
    return acc;
}
template <>
__device__ inline float dot8<F16T>(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2(*reinterpret_cast<const m5_h2_t*>(&w.x), *reinterpret_cast<const m5_h2_t*>(&x.x), acc, false);
    acc = __builtin_amdgcn_fdot2(*reinterpret_cast<const m5_h2_t*>(&w.y), *reinterpret_cast<const m5_h2_t*>(&x.y), acc, false);
    acc = __builtin_amdgcn_fdot2(*reinterpret_cast<const m5_h2_t*>(&w.z), *reinterpret_cast<const m5_h2_t*>(&x.z), acc, false);
    acc = __builtin_amdgcn_fdot2(*reinterpret_cast<const m5_h2_t*>(&w.w), *reinterpret_cast<const m5_h2_t*>(&x.w), acc, false);
    return acc;
}
// four fp32 values -> four operand-type values (8 bytes)
template <typename T>
__device__ inline uint2 m5_pack4(float a, float b, float c, float d) {
    typename T::storage t[4] = {T::from_f32(a), T::from_f32(b), T::from_f32(c), T::from_f32(d)};
    return *reinterpret_cast<const uint2*>(t);
}

// ---- wave / block reductions ---------------------------------------------------------
// Cross-lane exchanges without the LDS crossbar.  `__shfl_xor` compiles to ds_bpermute_b32 (address arithmetic + an
// LDS-pipe round trip + a wait, ~100 cycles per step; a reduction is a chain of 6 dependent steps).  The same partners are
// reachable with DPP modifiers (xor 1, 2: quad_perm; xor 4: quad reverse then row_half_mirror; xor 8: row_ror:8) and
// gfx950's v_permlane16_swap / v_permlane32_swap (xor 16, 32).  Same partners, same order, commutative operations: the
// results are bit-identical to the shuffle butterflies they replace (checked lane by lane on the hardware; the fp32 parity
// tests pin it).  Measured on the persistent decode step: 550 -> 520 us per token.
template <int CTRL>
__device__ inline float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ inline float lane_xor1(float v) { return dppf<0xB1>(v); }
__device__ inline float lane_xor2(float v) { return dppf<0x4E>(v); }
__device__ inline float lane_xor4(float v) { return dppf<0x141>(dppf<0x1B>(v)); }
__device__ inline float lane_xor8(float v) { return dppf<0x128>(v); }
// (All of these exchange between lanes of FULLY ACTIVE waves: DPP reads 0 from an inactive partner (bound_ctrl), the
// permlane swaps leave an inactive lane's half untouched -- unlike __shfl_xor, which returns the caller's own value.
// The half a lane keeps is chosen by its LANE id, so block shapes other than 1-D multiples of 64 work too.)
__device__ inline int m5_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ inline float lane_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((m5_lane_id() & 16) ? r[0] : r[1]);
}
__device__ inline float lane_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((m5_lane_id() & 32) ? r[0] : r[1]);
}
__device__ inline float wave_sum(float v) {                  // v += partner for lane ^ 32, 16, 8, 4, 2, 1
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    v += lane_xor8(v);
    v += lane_xor4(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}
__device__ inline float wave_max(float v) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    v = fmaxf(v, lane_xor8(v));
    v = fmaxf(v, lane_xor4(v));
    v = fmaxf(v, lane_xor2(v));
    v = fmaxf(v, lane_xor1(v));
    return v;
}
// block reductions over blockDim.x = NW*64 threads; `red` is NW floats of LDS scratch.
template <int NW>
__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ inline float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

__device__ inline float silu_f(float a) { return a / (1.0f + expf(-a)); }

#include "philox.h"      // torch's Philox draws (uniform / exponential), bit for bit

#define M5_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return M5_ERR_LAUNCH;        \
    } while (0)
