// pack_common.h -- device helpers of the diff/threshold/pack kernels (diff_pack.hip):
// wave64 DPP scan, the 4-bytes-per-instruction compare / difference / feedback arithmetic, frame loads.
#ifndef MI355_PACK_COMMON_H_
#define MI355_PACK_COMMON_H_

#include "internal.h"
#include "lab.h"

namespace mi355 {

// ---- wave64 inclusive scan with DPP (row_shr within rows of 16, then row_bcast 15 / 31) ---------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}

// inclusive running maximum over the lanes (unsigned; lanes without a source take 0, the identity)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_inclusive_max_scan(uint32_t v) {
    v = dpp_max<0x111, 0xf>(v);
    v = dpp_max<0x112, 0xf>(v);
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    return v;
}

// v_writelane_b32: put a wave-uniform value into one lane of a VGPR (1 instruction instead of
// v_mov + v_cndmask); `lane` must be a compile-time constant here.
__device__ __forceinline__ void write_lane(uint32_t &v, uint32_t uniform_value, int lane) {
    // (readfirstlane: free when the value already lives in an SGPR; the compiler keeps some uniform values in VGPRs)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)uniform_value)), "n"(lane));
}

// ---- per-dword byte arithmetic (4 bytes per instruction) --------------------------------------------
constexpr uint32_t kH = 0x80808080u, kL = 0x7f7f7f7fu;

struct ThrConst {   // per-byte replicated constants of T' = T (T <= 127) or T - 128 (T >= 128: the HIGH form)
    uint32_t ca;    // 127 - T' : (x_l + ca) carries into bit 7  <=>  x_l >= T' + 1
    uint32_t cb;    // T'       : (x_l + cb) carries into bit 7  <=>  x_l >= 128 - T'
    uint32_t h;     // 0x80808080 (kH), for the kernels that keep their constants in vector registers (vgpr_consts)
    uint32_t w0, w1;   // v_dot4 weights that turn 0x80 flag bytes into 128 * (bit 0..3 / bit 4..7 of a byte map)
};
__host__ __device__ inline ThrConst make_thr(uint32_t thr /* 0..255 */) {
    const uint32_t t = thr & 127u;
    return ThrConst{(127u - t) * 0x01010101u, t * 0x01010101u, 0x80808080u, 0x08040201u, 0x80402010u};
}

// A wave-uniform constant as a VECTOR register.  On gfx950 a VALU instruction with a scalar-register operand issues in
// ~4.4 cycles per wave64 against ~2.7 for the all-vector form (tools/ubench/issue_rate2.hip, profiles/archive/r04n): the pack
// kernel uses its three compare constants 16 times per KiB-step.  The asm keeps the compiler from folding the value back
// into an SGPR or a literal.
__device__ __forceinline__ uint32_t vgpr_const(uint32_t v) {
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
    return r;
}
__device__ __forceinline__ ThrConst vgpr_consts(ThrConst tc) {
    return ThrConst{vgpr_const(tc.ca), vgpr_const(tc.cb), vgpr_const(tc.h), vgpr_const(tc.w0), vgpr_const(tc.w1)};
}

// v_bitop3_b32: arbitrary 3-input boolean function; the truth table is written as an expression over
// the three operand patterns TA, TB, TC (same convention as the instruction's immediate).
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
template <uint32_t TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT & 0xFFu);
}

// Exact |a.byte - s.byte| > T for the 4 bytes of a dword (kernels.cu:311-312): returns 0x80 in every
// flagged byte, 0 elsewhere; x = (a | H) - (s & L) is handed back for the diff.  10 instructions.
//   x = 128 + a_l - s_l per byte (never borrows across bytes); x7 = [a_l >= s_l], x_l = (a_l - s_l) mod 128.
//   The true difference d = 128 (a7 - s7) + x - 128 is classified by its sign and by |d| >= 128:
//     sure = (a7 ^ s7) & ~(a7 ^ x7)          |d| >= 128
//     pos  = (a7 & ~s7) | (~(a7 ^ s7) & x7)  d >= 0 (else d < 0), |d| < 128, |d| mod 128 in x_l
//     flagged = sure | (pos ? x_l >= T + 1 : x_l < 128 - T)                       T <= 127
//     flagged = sure & (pos ? x_l >= T' + 1 : x_l < 128 - T'),  T' = T - 128       T >= 128 (HIGH): |d| = 128 + x_l
//               for d >= 0 and 256 - x_l for d < 0 once |d| >= 128; |d| == 128 exactly (not "sure") is never > T
template <bool HIGH = false>
__device__ __forceinline__ uint32_t dword_flags(uint32_t a, uint32_t s, ThrConst tc, uint32_t &x) {
    x = (a | kH) - (s & kL);
    const uint32_t xl = x & kL;
    const uint32_t A = xl + tc.ca;                               // bit 7: x_l >= T + 1
    const uint32_t nB = xl + tc.cb;                              // bit 7: x_l >= 128 - T
    const uint32_t sure = bitop3<(TA ^ TB) & ~(TA ^ TC)>(a, s, x);
    const uint32_t pos = bitop3<(TA & ~TB) | (~(TA ^ TB) & TC)>(a, s, x);
    const uint32_t mag = bitop3<(TA & TB) | (~TA & ~TC)>(pos, A, nB);
    return HIGH ? bitop3<(TA & TB) & TC>(sure, mag, tc.h) : bitop3<(TA | TB) & TC>(sure, mag, tc.h);
}

// per-byte (a - s) mod 256 from x: low 7 bits are x's, bit 7 is a7 ^ s7 ^ ~x7
//                                                               (kernels.cu:314 `diff[npos] = df`)
__device__ __forceinline__ uint32_t bytes_sub_from_x(uint32_t a, uint32_t s, uint32_t x, uint32_t h = kH) {
    const uint32_t y = bitop3<(TA ^ TB ^ ~TC)>(a, s, x);
    return bitop3<(TA & TB) | (~TA & TC)>(h, y, x);
}

// v_perm selector: byte j picks byte j of the first operand where flagged, of the second otherwise
__device__ __forceinline__ uint32_t perm_select(uint32_t fh) { return (fh >> 5) | 0x03020100u; }

__device__ __forceinline__ uint4 load16_bytes(const uint8_t *p8, int valid) {
    // global, not generic: a pointer rebuilt from scalar halves (uniform_ptr) has lost its address space
    const __attribute__((address_space(1))) uint8_t *p = (const __attribute__((address_space(1))) uint8_t *)(uintptr_t)p8;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < valid) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void store16_bytes(uint8_t *p, uint4 v, int valid) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < valid) p[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}

// A stream's frames are read exactly once: non-temporal loads keep them from displacing the record
// log and the meta words (written here, read back by k_expand) from the caches.  (Pair mode keeps
// plain loads: callers often hand in overlapping cur/prev sequences that do hit.)
template <bool FAST, bool NT = false>
__device__ __forceinline__ uint4 load16(const uint8_t *p, int valid) {
    if (FAST) {
        // every frame / state pointer of this kernel is global memory: say so, or a pointer rebuilt
        // from scalar halves (uniform_ptr) would be loaded through the flat aperture
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4 *gptr;
        const gptr g = (gptr)(uintptr_t)p;
        const u32x4 v = NT ? __builtin_nontemporal_load(g) : *g;
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return load16_bytes(p, valid);
}

// Tells the compiler a pointer is wave-uniform (it is: kernel arguments and the frame counter only), so
// that it stays in SGPRs and the access uses the SGPR-base + VGPR-offset addressing form.
__device__ __forceinline__ const uint8_t *uniform_ptr(const uint8_t *p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint8_t *)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t v) {   // 0x80 per nonzero byte
    return (((v & kL) + kL) | v) & kH;
}

// 16-bit map of the nonzero (= flagged: |df| > T >= 0 is never 0) bytes of a 16-byte record
__device__ __forceinline__ uint32_t record_map16(uint4 rec) {
    const uint32_t g0 = __builtin_amdgcn_udot4(nonzero_bytes(rec.x), 0x08040201u, 0u, false);
    const uint32_t g1 = __builtin_amdgcn_udot4(nonzero_bytes(rec.y), 0x08040201u, 0u, false);
    const uint32_t g2 = __builtin_amdgcn_udot4(nonzero_bytes(rec.z), 0x08040201u, 0u, false);
    const uint32_t g3 = __builtin_amdgcn_udot4(nonzero_bytes(rec.w), 0x08040201u, 0u, false);
    return (g0 + (g1 << 4) + (g2 << 8) + (g3 << 12)) >> 7;
}

}  // namespace mi355
#endif
