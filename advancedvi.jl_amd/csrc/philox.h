// Philox4x32-10 counter-based generator + Box-Muller: the eps stream of libmivi.
//
// Replaces `rand(rng, Normal{T}(0,1), d, M)` of src/families/location_scale.jl:76,86 (reference:
// AdvancedVI.jl).  eps[i, m] is a pure function of (seed, estimate_idx, global column m, row i):
//   block   q   = m * ceil(d/4) + i/4            (one Philox block = rows 4b..4b+3 of one column)
//   counter     = (lo32 q, hi32 q, lo32 estimate_idx, hi32 estimate_idx),  key = (lo32 seed, hi32 seed)
//   words w0..w3 -> (eps[4b], eps[4b+1]) = BoxMuller(w0, w1), (eps[4b+2], eps[4b+3]) = BoxMuller(w2, w3)
// so any shard of the columns regenerates exactly its slice of the one-GPU stream.
// Restated in numpy by oracle/oracle.py (philox_bits / box_muller_from_bits).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIVI_HD __host__ __device__ __forceinline__
#else
#define MIVI_HD inline
#endif

namespace mivi {

struct u32x4 {
  uint32_t x, y, z, w;
};

MIVI_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

MIVI_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;   // v_mad_u64_u32
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    u32x4 n;
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
    n.x = __builtin_amdgcn_bitop3_b32(hi1, c.y, k0, 0x96);   // a ^ b ^ c as ONE v_bitop3_b32 (gfx950 only; the compiler emits two v_xor_b32)
    n.z = __builtin_amdgcn_bitop3_b32(hi0, c.w, k1, 0x96);
#else
    n.x = hi1 ^ c.y ^ k0;
    n.z = hi0 ^ c.w ^ k1;
#endif
    n.y = lo1;
    n.w = lo0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// words of block q of estimate `idx`
MIVI_HD u32x4 eps_block_bits(uint64_t seed, uint64_t idx, uint64_t q) {
  u32x4 c;
  c.x = (uint32_t)q;
  c.y = (uint32_t)(q >> 32);
  c.z = (uint32_t)idx;
  c.w = (uint32_t)(idx >> 32);
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

template <typename T>
struct BoxMuller;

template <>
struct BoxMuller<float> {
  // u(w) = ((w >> 9) + 0.5) * 2^-23: exact in float32, strictly inside (0, 1)
  static MIVI_HD void pair(uint32_t wa, uint32_t wb, float &n0, float &n1) {
    const float ua = ((float)(wa >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float ub = ((float)(wb >> 9) + 0.5f) * 1.1920928955078125e-07f;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIVI_LIBM_BOXMULLER)
    // Written out for THESE arguments (u in (0, 1), never denormal; the angle 2 pi ub taken as a quarter-turn count + a remainder in
    // [-pi/4, pi/4]) instead of the general-purpose libm routines: the hardware log2 / sqrt (1 ulp each, no denormal / overflow branches),
    // Cephes' single-precision sine / cosine polynomials on the reduced angle, the quadrant applied with integer sign flips.  ~95 instead of
    // ~165 vector instructions per pair; eps within ~3 ulp of the float64 evaluation of the same uniforms (libm route: ~2; tests/test_gpu_rng.py).
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(ua));   // sqrt(-2 ln ua), ln = log2 * ln 2
    const float x4 = 4.0f * ub;                       // exact; quarter turns
    const float qf = __builtin_rintf(x4);             // 0 .. 4
    const float th = (x4 - qf) * 1.57079632679489662f;   // remainder angle in [-pi/4, pi/4] (the subtraction is exact)
    const float t2 = th * th;
    // (explicit fused multiply-adds: every kernel that draws eps must round these the same way, whatever the compiler would contract)
    const float ps = __builtin_fmaf(t2, __builtin_fmaf(t2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
    const float sp = __builtin_fmaf(th * t2, ps, th);
    const float pc = __builtin_fmaf(t2, __builtin_fmaf(t2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    const float cp = __builtin_fmaf(t2 * t2, pc, __builtin_fmaf(-0.5f, t2, 1.0f));
    const unsigned qi = (unsigned)(int)qf;            // quadrant: angle = qi * pi/2 + th
    const bool swap = (qi & 1u) != 0u;
    const unsigned ss = (qi & 2u) << 30, cs = ((qi + 1u) & 2u) << 30;   // sign bits of sin / cos in that quadrant
    const float s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, swap ? cp : sp) ^ ss);
    const float c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, swap ? sp : cp) ^ cs);
#elif defined(__HIP_DEVICE_COMPILE__)
    const float r = sqrtf(-2.0f * logf(ua));
    float s, c;
    sincospif(2.0f * ub, &s, &c);
#else
    const float r = (float)__builtin_sqrt(-2.0 * __builtin_log((double)ua));
    const double ang = 6.283185307179586476925286766559 * (double)ub;
    const float s = (float)__builtin_sin(ang), c = (float)__builtin_cos(ang);
#endif
    n0 = r * c;
    n1 = r * s;
#if defined(__HIP_DEVICE_COMPILE__)
    // (the draws are VALUES: a consumer's `x + eps` must not be contracted into fma(r, c, x) in one kernel and not in another)
    asm volatile("" : "+v"(n0), "+v"(n1));
#endif
  }
};

template <>
struct BoxMuller<double> {
  // u(w) = (w + 0.5) * 2^-32
  static MIVI_HD void pair(uint32_t wa, uint32_t wb, double &n0, double &n1) {
    const double ua = ((double)wa + 0.5) * 2.3283064365386962890625e-10;
    const double ub = ((double)wb + 0.5) * 2.3283064365386962890625e-10;
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = sqrt(-2.0 * log(ua));
    double s, c;
    sincospi(2.0 * ub, &s, &c);
#else
    const double r = __builtin_sqrt(-2.0 * __builtin_log(ua));
    const double ang = 6.283185307179586476925286766559 * ub;
    const double s = __builtin_sin(ang), c = __builtin_cos(ang);
#endif
    n0 = r * c;
    n1 = r * s;
  }
};

// the four normals of block q
template <typename T>
MIVI_HD void eps_block(uint64_t seed, uint64_t idx, uint64_t q, T out[4]) {
  const u32x4 b = eps_block_bits(seed, idx, q);
  BoxMuller<T>::pair(b.x, b.y, out[0], out[1]);
  BoxMuller<T>::pair(b.z, b.w, out[2], out[3]);
}

}  // namespace mivi
