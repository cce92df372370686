// TEST INFRASTRUCTURE: fiber scheduler behind tests/emu/hip/hip_runtime.h.
// Every GPU thread is a ucontext fiber.  __syncthreads() and the wave-wide MFMA are rendezvous points; a
// fiber that has to wait yields to a round-robin scheduler.  Blocks run one at a time unless the launch is
// cooperative (grid barriers need every block resident).
#include <hip/hip_runtime.h>
#include <cvae_intrin.h>

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

void emu_oob(const char* what, unsigned off, unsigned bytes) {
    fprintf(stderr, "emu: buffer %s out of bounds: offset %u of %u bytes\n", what, off, bytes);
    abort();
}

namespace emu {

struct Wave {
    float sh_in[64], sh_out[64];
    int sh_src[64], sh_arrived = 0;
    unsigned sh_gen = 0;
    int vote_arrived = 0, vote_true = 0;
    unsigned vote_gen = 0;
    bool vote_result = false;
    float a[64], b[64];
    f32x4 a8[64], b8[64];   // packed f16 operands of the 16x16x32 form
    f32x4 c[64], d[64];
    f32x16 c16[64], d16[64];
    int arrived = 0;
    unsigned gen = 0;
};
struct Block {
    std::vector<unsigned char> lds;
    std::vector<Wave> waves;
    int nthreads = 0, arrived = 0;
    unsigned gen = 0;
    int all_arrived = 0, all_true = 0;
    unsigned all_gen = 0;
    bool all_result = false;
};
struct Thread {
    ThreadView view;
    ucontext_t ctx;
    void* stack = nullptr;
    Block* block = nullptr;
    Wave* wave = nullptr;
    int lane = 0;
    bool done = false;
};

ThreadView* cur_view = nullptr;
static Thread* cur = nullptr;
static ucontext_t sched_ctx;
static const std::function<void()>* cur_body = nullptr;
static unsigned long progress = 0;
static const size_t STACK = 96 * 1024;

void yield() { swapcontext(&cur->ctx, &sched_ctx); }

unsigned char* smem() { return cur->block->lds.data(); }

void syncthreads() {
    Block* b = cur->block;
    const unsigned gen = b->gen;
    if (++b->arrived == b->nthreads) {
        b->arrived = 0;
        b->gen++;
        progress++;
    } else {
        while (b->gen == gen) yield();
    }
}

f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    Wave* w = cur->wave;
    const int l = cur->lane;
    const unsigned gen = w->gen;
    w->a[l] = a;
    w->b[l] = b;
    w->c[l] = c;
    if (++w->arrived == 64) {
        for (int lane = 0; lane < 64; ++lane) {
            const int col = lane & 15, rq = lane >> 4;
            f32x4 d = w->c[lane];
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r;
                float acc = d[r];
                for (int k = 0; k < 4; ++k) acc = fmaf(w->a[row + 16 * k], w->b[col + 16 * k], acc);
                d[r] = acc;
            }
            w->d[lane] = d;
        }
        w->arrived = 0;
        w->gen++;
        progress++;
    } else {
        while (w->gen == gen) yield();
    }
    return w->d[l];
}

// fp16 <-> fp32 in software (round to nearest even, subnormals, inf/nan): the emulator's half type
unsigned short f32_to_f16_bits(float f) {
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned sign = (x >> 16) & 0x8000u;
    const unsigned absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (absx > 0x7f800000u ? 0x200u : 0u));
    if (absx >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);            // rounds to >= 65520 -> inf
    if (absx < 0x33000001u) return (unsigned short)sign;                          // < 2^-25 (or exactly 2^-25 tie -> 0)
    int e = (int)(absx >> 23) - 127;
    unsigned m = (absx & 0x7fffffu) | 0x800000u;                                  // 24-bit significand
    int shift;                                                                    // bits to drop
    unsigned base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                            // subnormal half
    else { shift = 13; base = (unsigned)(e + 15) << 10; m &= 0x7fffffu; }
    const unsigned keep = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    unsigned r = keep;
    if (rem > half || (rem == half && (keep & 1u))) r++;
    return (unsigned short)(sign | (base + r));                                   // a carry out of the mantissa bumps the exponent
}
// round toward zero (v_cvt_pkrtz_f16_f32): truncate the significand, saturate at the largest finite half
unsigned short f32_to_f16_bits_rtz(float f) {
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned sign = (x >> 16) & 0x8000u;
    const unsigned absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (absx > 0x7f800000u ? 0x200u : 0u));
    if (absx >= 0x47800000u) return (unsigned short)(sign | 0x7bffu);            // >= 65536 -> 65504
    if (absx < 0x33800000u) return (unsigned short)sign;                          // < 2^-24 -> 0
    const int e = (int)(absx >> 23) - 127;
    unsigned m = (absx & 0x7fffffu) | 0x800000u;
    if (e < -14) return (unsigned short)(sign | (m >> (13 + (-14 - e))));
    return (unsigned short)(sign | (((unsigned)(e + 15) << 10) + ((m & 0x7fffffu) >> 13)));
}
// bf8 = e5m2: the upper byte of a half.  Encode with round-to-nearest-even through the half grid's coarser cousin.
float bf8_bits_to_f32(unsigned char b) { return f16_bits_to_f32((unsigned short)((unsigned)b << 8)); }
unsigned char f32_to_bf8_bits(float f) {
    if (f != f) return 0x7f;
    const unsigned sign = f < 0.0f || (f == 0.0f && 1.0f / f < 0.0f) ? 0x80u : 0u;
    const float a = fabsf(f);
    if (a >= 61440.0f) return (unsigned char)(sign | 0x7c);                       // rounds past the largest finite (57344): inf
    // candidates: floor on the bf8 grid and its successor; pick the nearer, ties to even mantissa
    unsigned lo = 0, hi = 0x7b;
    while (lo < hi) {                                                             // largest code with value <= a
        const unsigned mid = (lo + hi + 1) >> 1;
        if (bf8_bits_to_f32((unsigned char)mid) <= a) lo = mid; else hi = mid - 1;
    }
    const float vlo = bf8_bits_to_f32((unsigned char)lo), vhi = bf8_bits_to_f32((unsigned char)(lo + 1));
    unsigned code = lo;
    if (lo < 0x7b && (a - vlo > vhi - a || (a - vlo == vhi - a && (lo & 1u)))) code = lo + 1;
    return (unsigned char)(sign | code);
}
float f16_bits_to_f32(unsigned short h) {
    const unsigned sign = ((unsigned)h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400u), (int)e - 25);
    unsigned x;
    memcpy(&x, &v, 4);
    x |= sign;
    memcpy(&v, &x, 4);
    return v;
}

// v_mfma_f32_16x16x32_f16: lane l holds A[i = l & 15][k = 8*(l >> 4) + e] and B[k = 8*(l >> 4) + e][j = l & 15], e = 0..7,
// as 8 packed halves; D[row = 4*(lane >> 4) + r][col = lane & 15].  Products of two halves are exact in fp32; the sum is an
// fp32 chain in k order (the hardware's internal order differs in the last bit at most).
f32x4 mfma_16x16x32_f16(f32x4 a, f32x4 b, f32x4 c) {
    Wave* w = cur->wave;
    const int l = cur->lane;
    const unsigned gen = w->gen;
    w->a8[l] = a;
    w->b8[l] = b;
    w->c[l] = c;
    if (++w->arrived == 64) {
        static thread_local float A[16][32], B[32][16];
        for (int lane = 0; lane < 64; ++lane) {
            unsigned short ha[8], hb[8];
            memcpy(ha, &w->a8[lane], 16);
            memcpy(hb, &w->b8[lane], 16);
            for (int e = 0; e < 8; ++e) {
                A[lane & 15][8 * (lane >> 4) + e] = f16_bits_to_f32(ha[e]);
                B[8 * (lane >> 4) + e][lane & 15] = f16_bits_to_f32(hb[e]);
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int col = lane & 15, rq = lane >> 4;
            f32x4 d = w->c[lane];
            for (int r = 0; r < 4; ++r) {
                float acc = d[r];
                for (int k = 0; k < 32; ++k) acc = fmaf(A[4 * rq + r][k], B[k][col], acc);
                d[r] = acc;
            }
            w->d[lane] = d;
        }
        w->arrived = 0;
        w->gen++;
        progress++;
    } else {
        while (w->gen == gen) yield();
    }
    return w->d[l];
}

// v_mfma_f32_32x32x16_f16: lane l holds A[i = l & 31][k = 8*(l >> 5) + e] and B[k = 8*(l >> 5) + e][j = l & 31], e = 0..7;
// D[row = (q & 3) + 8*(q >> 2) + 4*(lane >> 5)][col = lane & 31], q = 0..15 (cdna_hip_programming.md section 3).
f32x16 mfma_32x32x16_f16(f32x4 a, f32x4 b, f32x16 c) {
    Wave* w = cur->wave;
    const int l = cur->lane;
    const unsigned gen = w->gen;
    w->a8[l] = a;
    w->b8[l] = b;
    w->c16[l] = c;
    if (++w->arrived == 64) {
        static thread_local float A[32][16], B[16][32];
        for (int lane = 0; lane < 64; ++lane) {
            unsigned short ha[8], hb[8];
            memcpy(ha, &w->a8[lane], 16);
            memcpy(hb, &w->b8[lane], 16);
            for (int e = 0; e < 8; ++e) {
                A[lane & 31][8 * (lane >> 5) + e] = f16_bits_to_f32(ha[e]);
                B[8 * (lane >> 5) + e][lane & 31] = f16_bits_to_f32(hb[e]);
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int col = lane & 31, hf = lane >> 5;
            f32x16 d = w->c16[lane];
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hf;
                float acc = d[q];
                for (int k = 0; k < 16; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                d[q] = acc;
            }
            w->d16[lane] = d;
        }
        w->arrived = 0;
        w->gen++;
        progress++;
    } else {
        while (w->gen == gen) yield();
    }
    return w->d16[l];
}

bool wave_all(bool pred) {
    Wave* w = cur->wave;
    const unsigned gen = w->vote_gen;
    w->vote_true += pred ? 1 : 0;
    if (++w->vote_arrived == 64) {
        w->vote_result = w->vote_true == 64;
        w->vote_arrived = 0;
        w->vote_true = 0;
        w->vote_gen++;
        progress++;
    } else {
        while (w->vote_gen == gen) yield();
    }
    return w->vote_result;
}

bool block_all(bool pred) {
    Block* b = cur->block;
    const unsigned gen = b->all_gen;
    b->all_true += pred ? 1 : 0;
    if (++b->all_arrived == b->nthreads) {
        b->all_result = b->all_true == b->nthreads;
        b->all_arrived = 0;
        b->all_true = 0;
        b->all_gen++;
        progress++;
    } else {
        while (b->all_gen == gen) yield();
    }
    return b->all_result;
}

int lane_id() { return cur->lane; }

float wave_shfl(float v, int src) {
    Wave* w = cur->wave;
    const int l = cur->lane;
    const unsigned gen = w->sh_gen;
    w->sh_in[l] = v;
    w->sh_src[l] = src;
    if (++w->sh_arrived == 64) {
        for (int lane = 0; lane < 64; ++lane) w->sh_out[lane] = w->sh_in[w->sh_src[lane] & 63];
        w->sh_arrived = 0;
        w->sh_gen++;
        progress++;
    } else {
        while (w->sh_gen == gen) yield();
    }
    return w->sh_out[l];
}

static void trampoline() {
    (*cur_body)();
    cur->done = true;
    {   // like the hardware's s_barrier, a block barrier only waits for waves that have not terminated
        Block* b = cur->block;
        b->nthreads--;
        if (b->nthreads > 0 && b->arrived == b->nthreads) {
            b->arrived = 0;
            b->gen++;
        }
        if (b->nthreads > 0 && b->all_arrived == b->nthreads) {
            b->all_result = b->all_true == b->nthreads;
            b->all_arrived = 0;
            b->all_true = 0;
            b->all_gen++;
        }
    }
    progress++;
    swapcontext(&cur->ctx, &sched_ctx);
}

static void run_blocks(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem_bytes,
                       const std::vector<unsigned>& block_ids) {
    const int nthr = block.x * block.y * block.z;
    if (nthr % 64) {
        fprintf(stderr, "emu: block size %d is not a multiple of the wave size\n", nthr);
        abort();
    }
    std::vector<Block> blocks(block_ids.size());
    std::vector<Thread> threads(block_ids.size() * (size_t)nthr);
    for (size_t bi = 0; bi < block_ids.size(); ++bi) {
        Block& B = blocks[bi];
        B.lds.assign(smem_bytes + 64, 0);
        B.waves.resize(nthr / 64);
        B.nthreads = nthr;
        const unsigned id = block_ids[bi];
        for (int t = 0; t < nthr; ++t) {
            Thread& T = threads[bi * nthr + t];
            T.view.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            T.view.bid = dim3(id % grid.x, (id / grid.x) % grid.y, id / (grid.x * grid.y));
            T.view.bdim = block;
            T.view.gdim = grid;
            T.block = &B;
            T.wave = &B.waves[t / 64];
            T.lane = t % 64;
            T.stack = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (T.stack == MAP_FAILED) {
                perror("emu: mmap");
                abort();
            }
            getcontext(&T.ctx);
            T.ctx.uc_stack.ss_sp = T.stack;
            T.ctx.uc_stack.ss_size = STACK;
            T.ctx.uc_link = &sched_ctx;
            makecontext(&T.ctx, trampoline, 0);
        }
    }
    cur_body = &body;
    size_t remaining = threads.size();
    unsigned long idle_rounds = 0;
    while (remaining) {
        const unsigned long before = progress;
        for (auto& T : threads) {
            if (T.done) continue;
            cur = &T;
            cur_view = &T.view;
            swapcontext(&sched_ctx, &T.ctx);
            if (T.done) --remaining;
        }
        idle_rounds = (progress == before) ? idle_rounds + 1 : 0;
        if (idle_rounds > 100000) {
            fprintf(stderr, "emu: deadlock (no fiber made progress)\n");
            abort();
        }
    }
    for (auto& T : threads) munmap(T.stack, STACK);
    cur = nullptr;
    cur_view = nullptr;
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem_bytes, bool all_resident) {
    const unsigned nblocks = grid.x * grid.y * grid.z;
    if (all_resident) {
        std::vector<unsigned> ids(nblocks);
        for (unsigned i = 0; i < nblocks; ++i) ids[i] = i;
        run_blocks(body, grid, block, smem_bytes, ids);
    } else {
        for (unsigned i = 0; i < nblocks; ++i) run_blocks(body, grid, block, smem_bytes, std::vector<unsigned>{i});
    }
}

}  // namespace emu
