// gfx950 intrinsics used by the CycleVAE kernels, behind short names.
//
// This header is the ONLY place that touches __builtin_amdgcn_* / inline asm.  tests/emu/ ships a
// same-named header that implements the same names on host fibers, so the kernels, the launch code and
// the C ABI can be exercised on a machine without a GPU (include path order selects the header).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// all LDS lives in one dynamic array (keeps its base 16-byte aligned: cdna guide G17)
extern __shared__ __attribute__((aligned(16))) unsigned char cvae_smem_raw[];
#define CVAE_SMEM (cvae_smem_raw)

// v_mfma_f32_16x16x4_f32: exact-f32 matrix FMA, one wave.
//   a: A[i = lane&15][k = lane>>4]     b: B[k = lane>>4][j = lane&15]
//   d: D[row = 4*(lane>>4) + r][col = lane&15], r = 0..3
__device__ __forceinline__ f32x4 cvae_mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_f16 (gfx950): 8 packed halves per lane and operand, passed around as the bits of an f32x4:
//   a: A[i = lane&15][k = 8*(lane>>4) + e]     b: B[k = 8*(lane>>4) + e][j = lane&15], e = 0..7     d: as cvae_mfma_16x16x4
// Measured on MI355X: 10.7 ns per instruction and wave under full load (13.5 ns for the 16x16x4 f32 form, which covers K = 4).
typedef _Float16 cvae_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 cvae_mfma_16x16x32_f16(f32x4 a_bits, f32x4 b_bits, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cvae_h8, a_bits), __builtin_bit_cast(cvae_h8, b_bits), c, 0, 0, 0);
}
// two-term fp16 split with a scaled tail: x = hi + lo/2048 to ~22 significant bits (hi, lo round-to-nearest halves; the
// scale keeps lo out of fp16's subnormal range for every |x| >= 2^-13)
__host__ __device__ __forceinline__ void cvae_split_f16(float x, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}
__host__ __device__ __forceinline__ float cvae_f16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// v_mfma_f32_32x32x16_f16 (gfx950): 8 packed halves per lane and operand (the bits of an f32x4), 16 accumulators per lane:
//   a: A[i = lane&31][k = 8*(lane>>5) + e]     b: B[k = 8*(lane>>5) + e][j = lane&31], e = 0..7
//   d: D[row = (q&3) + 8*(q>>2) + 4*(lane>>5)][col = lane&31], q = 0..15
// 32 matrix-pipe cycles per instruction and SIMD (MI355X_MICROARCH cycle table).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 cvae_mfma_32x32x16_f16(f32x4 a_bits, f32x4 b_bits, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cvae_h8, a_bits), __builtin_bit_cast(cvae_h8, b_bits), c, 0, 0, 0);
}
// three-term fp16 split, EXACT for fp32: x = l0 + l1/2^11 + l2/2^22 with l0 = fp16(x), l1 = fp16((x - l0)*2^11),
// l2 = fp16(((x - l0)*2^11 - l1)*2^11).  Every subtraction and scaling is exact in fp32 (the residual of a round-to-nearest
// 11-bit limb has at most 13, then 2 significant bits), so the three HALVES carry all 24 bits of x for 2^-22 <= |x| < 65504;
// below 2^-22 the limbs run into fp16's subnormal grid and the representation error is at most 2^-47 ABSOLUTE.  (That is the
// weight images, which keep l2 as a half.  Exchanged values carry l2 as a bf8 BYTE, cvae_split3_f16b8 below: exact for
// |x| >= 2^-16, absolute error <= 2^-40 below -- measured over every binade, tests/test_emu_library.py::check_limb_transport.)
__host__ __device__ __forceinline__ void cvae_split3_f16(float x, unsigned short& l0, unsigned short& l1, unsigned short& l2) {
    const _Float16 a = (_Float16)x;
    const float r1 = (x - (float)a) * 2048.0f;
    const _Float16 b = (_Float16)r1;
    const float r2 = (r1 - (float)b) * 2048.0f;
    const _Float16 c = (_Float16)r2;
    l0 = __builtin_bit_cast(unsigned short, a);
    l1 = __builtin_bit_cast(unsigned short, b);
    l2 = __builtin_bit_cast(unsigned short, c);
}

// The third limb has at most 3 significant bits, so it travels as ONE byte: bf8 (e5m2) of l2 * 2^6.  The scale keeps it a
// normal bf8 for |x| >= 2^-16 (below, the conversion leaves bf8's normal range, which costs at most 2^-40 absolute on x); the
// result is clamped to bf8's finite range, which only matters
// beyond |x| ~ 3.5e3.  Decoding multiplies by 2^-6 inside the conversion (v_cvt_scalef32_pk_f16_bf8), giving back the f16 limb.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CVAE_L2_SCALE 64.0f
__device__ __forceinline__ unsigned char cvae_f32_to_bf8(float v) {
    v = __builtin_fminf(__builtin_fmaxf(v, -57344.0f), 57344.0f);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_bf8_f32(v, v, 0u, false) & 0xffu);
}
__device__ __forceinline__ float cvae_bf8_to_f32(unsigned char b) {
    const auto h = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8((unsigned)b, 1.0f, false);
    return (float)h[0];
}
__device__ __forceinline__ void cvae_split3_f16b8(float x, unsigned short& l0, unsigned short& l1, unsigned char& l2) {
    const _Float16 a = (_Float16)x;
    const float r1 = (x - (float)a) * 2048.0f;
    const _Float16 b = (_Float16)r1;
    const float r2 = (r1 - (float)b) * 2048.0f;
    l0 = __builtin_bit_cast(unsigned short, a);
    l1 = __builtin_bit_cast(unsigned short, b);
    l2 = cvae_f32_to_bf8(r2 * CVAE_L2_SCALE);
}
// 8 third limbs (8 bytes in two registers) -> the f16 operand fragment (8 halves in four registers)
__device__ __forceinline__ f32x4 cvae_bf8x8_to_h8(f32x2 raw) {
    // (temporaries: __builtin_bit_cast of a vector ELEMENT reads element 0 with this clang -- both words decoded the low four
    //  bytes until cvae_selftest_limbs caught it)
    const float r0 = raw[0], r1 = raw[1];
    const unsigned lo = __builtin_bit_cast(unsigned, r0), hi = __builtin_bit_cast(unsigned, r1);
    f32x4 o;
    o[0] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(lo, 1.0f / CVAE_L2_SCALE, false));
    o[1] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(lo, 1.0f / CVAE_L2_SCALE, true));
    o[2] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(hi, 1.0f / CVAE_L2_SCALE, false));
    o[3] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(hi, 1.0f / CVAE_L2_SCALE, true));
    return o;
}

// The same exact split for 8 fp32 values that become one MFMA operand (8 packed halves per limb), with round-toward-zero
// limbs (v_cvt_pkrtz_f16_f32 converts and packs two values per instruction): l0 takes the top 11 bits, l1 the next 11, l2
// the last 2; residuals are exact as above.
__device__ __forceinline__ void cvae_split3_pack8(f32x4 va, f32x4 vb, f32x4& l0, f32x4& l1, f32x4& l2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = e < 2 ? va[2 * e] : vb[2 * e - 4], x1 = e < 2 ? va[2 * e + 1] : vb[2 * e - 3];
        // residual * 2^11 as ONE mixed-precision fma on the half just produced (v_fma_mix_f32): 2048*x - 2048*l is exact
        const auto a = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        const float r0 = __builtin_fmaf(-2048.0f, (float)a[0], x0 * 2048.0f), r1 = __builtin_fmaf(-2048.0f, (float)a[1], x1 * 2048.0f);
        const auto b = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        const float q0 = __builtin_fmaf(-2048.0f, (float)b[0], r0 * 2048.0f), q1 = __builtin_fmaf(-2048.0f, (float)b[1], r1 * 2048.0f);
        const auto cc = __builtin_amdgcn_cvt_pkrtz(q0, q1);
        l0[e] = __builtin_bit_cast(float, a);
        l1[e] = __builtin_bit_cast(float, b);
        l2[e] = __builtin_bit_cast(float, cc);
    }
}

__device__ __forceinline__ void cvae_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// agent-scope release: write back this XCD's dirty L2 lines; the asm wait restates the post-wbl2 wait
// where the compiler cannot drop it (MI355X_MICROARCH "Compiler hazard").
__device__ __forceinline__ void cvae_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void cvae_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

__device__ __forceinline__ unsigned cvae_atomic_add_agent(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned cvae_atomic_load_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cvae_atomic_store_agent(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cvae_sleep() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ void cvae_sleep_64() { __builtin_amdgcn_s_sleep(1); }   // ~64 cycles
__device__ __forceinline__ unsigned cvae_xcc_id() {
    return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 0xf;
}

__device__ __forceinline__ void cvae_compiler_fence() { asm volatile("" ::: "memory"); }

// hardware transcendental paths (v_exp_f32 / v_rcp_f32), ~1e-6 relative
__device__ __forceinline__ float cvae_fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float cvae_fast_rcp(float x) { return __frcp_rn(x); }

// all lanes of the wave have executed everything above (hardware: lockstep, this only pins the schedule)
__device__ __forceinline__ void cvae_wave_barrier() { __builtin_amdgcn_wave_barrier(); }

// instruction-scheduling fence: nothing is moved across it by the compiler (keeps a probe where it was written)
__device__ __forceinline__ void cvae_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// true iff the predicate holds in every lane of the wave (all 64 lanes must call it)
__device__ __forceinline__ bool cvae_wave_all(bool pred) { return __builtin_amdgcn_ballot_w64(pred) == ~0ull; }

// true iff the predicate holds in every thread of the block (all threads must call it; a block barrier)
__device__ __forceinline__ bool cvae_block_all(bool pred) { return __syncthreads_and(pred ? 1 : 0) != 0; }
// value held by member J of the caller's aligned group of four lanes (DPP quad_perm: no LDS, folds into the consuming VALU op)
template <int J>
__device__ __forceinline__ float cvae_quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}
// value held by lane `src` of the wave (every lane of the wave must call it)
__device__ __forceinline__ float cvae_shfl(float v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ long long cvae_clock() { return (long long)__builtin_readcyclecounter(); }

// value the compiler must treat as wave-uniform (threadIdx-derived wave ids are uniform but not provably so)
__device__ __forceinline__ int cvae_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Buffer descriptor + cache-policy loads/stores.  aux 16 = sc1:
//   store sc1 : write-through to memory (and dropped from this XCD's L2) -> visible to every XCD without a release fence
//   load  sc1 : bypasses this CU's L1 -> sees other CUs' write-through stores without an acquire fence
// (MI355X_MICROARCH "inter-workgroup visibility": sc1 stores AND sc1 loads on both sides is a valid hand-off form.)
typedef __amdgpu_buffer_rsrc_t cvae_buf;
typedef unsigned cvae_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cvae_buf cvae_make_buf(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 cvae_buf_load_f4_sc1(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 16));
}
// plain (L1- and L2-cached) form: only for lines that NOBODY has read since the kernel started (see k_gru_steps_v6)
__device__ __forceinline__ f32x4 cvae_buf_load_f4(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x2 cvae_buf_load_f2(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(b, (int)voff, (int)soff, 0));
}
typedef unsigned cvae_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cvae_buf_store_f2_sc1(cvae_buf b, unsigned voff, unsigned soff, f32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cvae_u32x2, v), b, (int)voff, (int)soff, 16);
}
// the same load marked volatile (aux bit 31): stays inside a polling loop, never hoisted or merged by the compiler
__device__ __forceinline__ f32x4 cvae_buf_poll_f4(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, (int)0x80000010));
}
// polling load that bypasses only this CU's L1 (sc0): for words written by plain stores of a CU of the SAME XCD (one L2)
__device__ __forceinline__ f32x4 cvae_buf_poll_f4_sc0(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, (int)0x80000001));
}
// plain store (write-back in this XCD's L2)
__device__ __forceinline__ void cvae_buf_store_f4(cvae_buf b, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvae_u32x4, v), b, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float cvae_buf_poll_f1(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)voff, (int)soff, (int)0x80000010));
}
__device__ __forceinline__ float cvae_buf_load_f1_sc1(cvae_buf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)voff, (int)soff, 16));
}
__device__ __forceinline__ void cvae_buf_store_f4_sc1(cvae_buf b, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvae_u32x4, v), b, (int)voff, (int)soff, 16);
}

// Block -> (unit octet c, row-tile group ti) for the dataflow recurrences, XCD-aware.  Workgroup b runs on XCD b % 8 (observed
// dispatch order, MI355X_MICROARCH; used for speed only, any mapping is correct).  Blocks of one row-tile group exchange state
// only among themselves, and every XCD's L2 pulls whatever its CUs read: with the plain mapping (c = b % NB, ti = b / NB) each
// of the 8 L2s fetched the state of EVERY tile group every step.  Here the rts tile groups get 8 / rts XCDs each, so an L2 pulls
// one group's state only (rts = 2: the fabric-side fetch of a launch halves).  Falls back to the plain mapping when the grid
// does not divide that way.
__device__ __forceinline__ void cvae_block_map(int b, int NB, int rts, bool xcd_aware, int& c, int& ti) {
    const int nx = rts > 0 && 8 % rts == 0 ? 8 / rts : 0;      // XCDs per tile group
    if (xcd_aware && nx > 0 && NB % nx == 0) {
        const int x = b & 7, q = b >> 3;
        ti = x / nx;
        c = q * nx + x % nx;
    } else {
        c = b % NB;
        ti = b / NB;
    }
}

// Launch of a one-struct-argument kernel whose blocks wait for each other (flags in memory, no grid.sync()): the whole grid
// has to be resident.  g_cvae_coop_launch = 1: hipLaunchCooperativeKernel, which checks that at launch time -- and, measured
// with rocprofv3 on MI355X, leaves the GPU idle for ~13 us before and ~13-18 us after every such launch (the runtime brackets
// it with barrier packets).  0 (default): the same check once per (kernel, block, LDS) through the occupancy query, then a plain
// launch: back to back with its neighbours on the stream.  A grid that does not fit is refused in both modes
// (hipErrorCooperativeLaunchTooLarge), never launched.
static thread_local int g_cvae_coop_launch = 0;   // (the running entry point's context's option coop_launch: CtxScope in cvae_lib.hip)
struct CvaeResidency {
    const void* k;
    unsigned threads;
    size_t smem;
    int dev, blocks;      // resident blocks the device holds of this kernel
};
static CvaeResidency g_cvae_residency[64];
static int g_cvae_residency_n = 0;
static std::mutex g_cvae_residency_mu;

template <class P>
static inline hipError_t cvae_launch_coop(void (*k)(P), dim3 g, dim3 b, size_t smem, hipStream_t s, P p) {
    if (g_cvae_coop_launch) {
        void* args[] = {(void*)&p};
        return hipLaunchCooperativeKernel((const void*)k, g, b, args, (unsigned)smem, s);
    }
    int dev = 0, blocks = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned threads = b.x * b.y * b.z;
    {
        std::lock_guard<std::mutex> lock(g_cvae_residency_mu);
        for (int i = 0; i < g_cvae_residency_n; ++i) {
            const CvaeResidency& r = g_cvae_residency[i];
            if (r.k == (const void*)k && r.threads == threads && r.smem == smem && r.dev == dev) blocks = r.blocks;
        }
        if (blocks < 0) {
            int per_cu = 0, cus = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, (int)threads, smem);
            if (e != hipSuccess) return e;
            e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e != hipSuccess) return e;
            blocks = per_cu * cus;
            if (g_cvae_residency_n < 64) g_cvae_residency[g_cvae_residency_n++] = CvaeResidency{(const void*)k, threads, smem, dev, blocks};
        }
    }
    if ((long)g.x * g.y * g.z > (long)blocks) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(k, g, b, smem, s, p);
    return hipGetLastError();
}
