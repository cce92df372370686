// TEST INFRASTRUCTURE: host-fiber implementation of the names in cyclevae-vc_amd/csrc/cvae_intrin.h.
// The MFMA follows the documented gfx950 operand maps (cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x4_f32   a: A[i=lane&15][k=lane>>4]   b: B[k=lane>>4][j=lane&15]
//                            d: D[row=4*(lane>>4)+r][col=lane&15]
// and accumulates in k order with one fp32 rounding per product (an fmaf chain), like the hardware.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CVAE_SMEM (emu::smem())

namespace emu {
f32x4 mfma_16x16x4(float a, float b, f32x4 c);
}
static inline f32x4 cvae_mfma_16x16x4(float a, float b, f32x4 c) { return emu::mfma_16x16x4(a, b, c); }
namespace emu {
f32x4 mfma_16x16x32_f16(f32x4 a, f32x4 b, f32x4 c);
unsigned short f32_to_f16_bits(float f);
float f16_bits_to_f32(unsigned short h);
}
static inline f32x4 cvae_mfma_16x16x32_f16(f32x4 a_bits, f32x4 b_bits, f32x4 c) { return emu::mfma_16x16x32_f16(a_bits, b_bits, c); }
static inline float cvae_f16_bits_to_f32(unsigned short b) { return emu::f16_bits_to_f32(b); }
static inline void cvae_split_f16(float x, unsigned short& hi, unsigned short& lo) {
    hi = emu::f32_to_f16_bits(x);
    lo = emu::f32_to_f16_bits((x - emu::f16_bits_to_f32(hi)) * 2048.0f);
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
namespace emu {
f32x16 mfma_32x32x16_f16(f32x4 a, f32x4 b, f32x16 c);
}
static inline f32x16 cvae_mfma_32x32x16_f16(f32x4 a_bits, f32x4 b_bits, f32x16 c) { return emu::mfma_32x32x16_f16(a_bits, b_bits, c); }
static inline void cvae_split3_f16(float x, unsigned short& l0, unsigned short& l1, unsigned short& l2) {
    l0 = emu::f32_to_f16_bits(x);
    const float r1 = (x - emu::f16_bits_to_f32(l0)) * 2048.0f;
    l1 = emu::f32_to_f16_bits(r1);
    const float r2 = (r1 - emu::f16_bits_to_f32(l1)) * 2048.0f;
    l2 = emu::f32_to_f16_bits(r2);
}
namespace emu {
unsigned short f32_to_f16_bits_rtz(float f);
unsigned char f32_to_bf8_bits(float f);
float bf8_bits_to_f32(unsigned char b);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CVAE_L2_SCALE 64.0f
static inline unsigned char cvae_f32_to_bf8(float v) { return emu::f32_to_bf8_bits(fminf(fmaxf(v, -57344.0f), 57344.0f)); }
static inline float cvae_bf8_to_f32(unsigned char b) { return emu::bf8_bits_to_f32(b); }
static inline void cvae_split3_f16b8(float x, unsigned short& l0, unsigned short& l1, unsigned char& l2) {
    l0 = emu::f32_to_f16_bits(x);
    const float r1 = (x - emu::f16_bits_to_f32(l0)) * 2048.0f;
    l1 = emu::f32_to_f16_bits(r1);
    const float r2 = (r1 - emu::f16_bits_to_f32(l1)) * 2048.0f;
    l2 = cvae_f32_to_bf8(r2 * CVAE_L2_SCALE);
}
static inline f32x4 cvae_bf8x8_to_h8(f32x2 raw) {
    unsigned char b[8];
    unsigned short h[8];
    memcpy(b, &raw, 8);
    for (int e = 0; e < 8; ++e) h[e] = emu::f32_to_f16_bits(emu::bf8_bits_to_f32(b[e]) * (1.0f / CVAE_L2_SCALE));
    f32x4 o;
    memcpy(&o, h, 16);
    return o;
}
static inline void cvae_split3_pack8(f32x4 va, f32x4 vb, f32x4& l0, f32x4& l1, f32x4& l2) {
    unsigned short h[3][8];
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? va[e] : vb[e - 4];
        h[0][e] = emu::f32_to_f16_bits_rtz(x);
        const float r1 = (x - emu::f16_bits_to_f32(h[0][e])) * 2048.0f;
        h[1][e] = emu::f32_to_f16_bits_rtz(r1);
        const float r2 = (r1 - emu::f16_bits_to_f32(h[1][e])) * 2048.0f;
        h[2][e] = emu::f32_to_f16_bits_rtz(r2);
    }
    memcpy(&l0, h[0], 16);
    memcpy(&l1, h[1], 16);
    memcpy(&l2, h[2], 16);
}
static inline void cvae_drain_vmem() {}
static inline void cvae_release_agent() {}
static inline void cvae_acquire_agent() {}
static inline unsigned cvae_atomic_add_agent(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned cvae_atomic_load_agent(const unsigned* p) { return *(volatile const unsigned*)p; }
static inline void cvae_atomic_store_agent(unsigned* p, unsigned v) { *(volatile unsigned*)p = v; }
static inline void cvae_sleep() { emu::yield(); }
static inline void cvae_sleep_64() { emu::yield(); }
static inline unsigned cvae_xcc_id() { return emu::cur_view->bid.x % 8; }

static inline void cvae_compiler_fence() {}
static inline void cvae_sched_fence() {}
namespace emu {
bool wave_all(bool pred);
}
static inline bool cvae_wave_all(bool pred) { return emu::wave_all(pred); }
namespace emu {
bool block_all(bool pred);
float wave_shfl(float v, int src);
int lane_id();
}
static inline bool cvae_block_all(bool pred) { return emu::block_all(pred); }
static inline float cvae_shfl(float v, int src) { return emu::wave_shfl(v, src); }
template <int J>
static inline float cvae_quad_bcast(float v) { return emu::wave_shfl(v, (emu::lane_id() & ~3) + J); }
static inline void cvae_wave_barrier() { (void)emu::wave_all(true); }
static inline float cvae_fast_exp(float x) { return expf(x); }
static inline float cvae_fast_rcp(float x) { return 1.0f / x; }
static inline long long cvae_clock() { return 0; }
static inline int cvae_uniform(int v) { return v; }

struct cvae_buf {
    const unsigned char* base;
    unsigned bytes;
};
void emu_oob(const char* what, unsigned off, unsigned bytes);
static inline cvae_buf cvae_make_buf(const void* p, unsigned bytes) { return cvae_buf{(const unsigned char*)p, bytes}; }
static inline f32x4 cvae_buf_load_f4_sc1(cvae_buf b, unsigned voff, unsigned soff) {
    if ((size_t)voff + soff + 16 > b.bytes) emu_oob("load_f4", voff + soff, b.bytes);
    f32x4 v;
    memcpy(&v, b.base + voff + soff, 16);
    return v;
}
static inline f32x4 cvae_buf_poll_f4(cvae_buf b, unsigned voff, unsigned soff) { return cvae_buf_load_f4_sc1(b, voff, soff); }
static inline f32x4 cvae_buf_load_f4(cvae_buf b, unsigned voff, unsigned soff) { return cvae_buf_load_f4_sc1(b, voff, soff); }
static inline f32x2 cvae_buf_load_f2(cvae_buf b, unsigned voff, unsigned soff) {
    if ((size_t)voff + soff + 8 > b.bytes) emu_oob("load_f2", voff + soff, b.bytes);
    f32x2 v;
    memcpy(&v, b.base + voff + soff, 8);
    return v;
}
static inline void cvae_buf_store_f2_sc1(cvae_buf b, unsigned voff, unsigned soff, f32x2 v) {
    if ((size_t)voff + soff + 8 > b.bytes) emu_oob("store_f2", voff + soff, b.bytes);
    memcpy((unsigned char*)b.base + voff + soff, &v, 8);
}
static inline float cvae_buf_load_f1_sc1(cvae_buf b, unsigned voff, unsigned soff);
static inline float cvae_buf_poll_f1(cvae_buf b, unsigned voff, unsigned soff) { return cvae_buf_load_f1_sc1(b, voff, soff); }
static inline float cvae_buf_load_f1_sc1(cvae_buf b, unsigned voff, unsigned soff) {
    if ((size_t)voff + soff + 4 > b.bytes) emu_oob("load_f1", voff + soff, b.bytes);
    float v;
    memcpy(&v, b.base + voff + soff, 4);
    return v;
}
static inline void cvae_buf_store_f4_sc1(cvae_buf b, unsigned voff, unsigned soff, f32x4 v) {
    if ((size_t)voff + soff + 16 > b.bytes) emu_oob("store_f4", voff + soff, b.bytes);
    memcpy((unsigned char*)b.base + voff + soff, &v, 16);
}

static inline f32x4 cvae_buf_poll_f4_sc0(cvae_buf b, unsigned voff, unsigned soff) { return cvae_buf_load_f4_sc1(b, voff, soff); }
static inline void cvae_buf_store_f4(cvae_buf b, unsigned voff, unsigned soff, f32x4 v) { cvae_buf_store_f4_sc1(b, voff, soff, v); }

static inline void cvae_block_map(int b, int NB, int rts, bool xcd_aware, int& c, int& ti) {
    const int nx = rts > 0 && 8 % rts == 0 ? 8 / rts : 0;
    if (xcd_aware && nx > 0 && NB % nx == 0) {
        const int x = b & 7, q = b >> 3;
        ti = x / nx;
        c = q * nx + x % nx;
    } else {
        c = b % NB;
        ti = b / NB;
    }
}

static thread_local int g_cvae_coop_launch = 0;     // (the library's launch-mode switch; the emulator has one way to launch)
template <class P>
static inline hipError_t cvae_launch_coop(void (*k)(P), dim3 g, dim3 b, size_t smem, hipStream_t, P p) {
    emu::launch([=]() { k(p); }, g, b, smem, true);
    return hipSuccess;
}
