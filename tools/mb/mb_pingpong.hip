// Microbenchmark (GPU box only): latency of a one-word hand-off between two workgroups, same XCD and different XCDs, for the
// cache-policy combinations a kernel can choose (aux bits of buffer loads / stores: 1 = sc0, 16 = sc1, 2 = nt).
//   ping-pong: A stores i to word a, B polls a then stores i to word b, A polls b.  Reported: ns per HOP (half a round trip).
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

template <int SAUX, int LAUX>
__global__ __launch_bounds__(64, 1) void k_pp(unsigned* words, int* who, int iters, int xa, int xb, long long* out, int delay) {
    const unsigned xcc = cvae_xcc_id();
    __shared__ int role;
    if (threadIdx.x == 0) {
        role = -1;
        if (xcc == (unsigned)xa && atomicCAS(&who[0], 0, 1) == 0) role = 0;
        else if (xcc == (unsigned)xb && atomicCAS(&who[1], 0, 1) == 0) role = 1;
    }
    __syncthreads();
    if (role < 0) return;
    const cvae_buf b = cvae_make_buf(words, 4096);
    // wait until both roles are taken
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(&who[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 ||
               __hip_atomic_load(&who[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
        const unsigned mine = role == 0 ? 0u : 1024u, other = role == 0 ? 1024u : 0u;
        unsigned spins = 0;
        const long long t0 = wall_clock64();
        for (int i = 1; i <= iters; ++i) {
            if (role == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(i, b, 0, (int)mine, SAUX);
                while ((int)__builtin_amdgcn_raw_buffer_load_b32(b, 0, (int)other, LAUX | (int)0x80000000) < i) { asm volatile("" ::: "memory"); if (++spins > 20000000u) { out[1] = i; return; } }
            } else {
                while ((int)__builtin_amdgcn_raw_buffer_load_b32(b, 0, (int)other, LAUX | (int)0x80000000) < i) { asm volatile("" ::: "memory"); if (++spins > 20000000u) { out[1] = i; return; } }
                if (delay) __builtin_amdgcn_s_sleep(32);   // ~2048 cycles
                __builtin_amdgcn_raw_buffer_store_b32(i, b, 0, (int)mine, SAUX);
            }
        }
        const long long t1 = wall_clock64();
        if (role == 0) out[0] = t1 - t0;
    }
}

int main() {
    unsigned* words;
    int* who;
    long long* out;
    hipMalloc(&words, 4096);
    hipMalloc(&who, 8);
    hipMalloc(&out, 16);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int xa, int xb, int delay = 0) {
        hipMemset(words, 0, 4096);
        hipMemset(who, 0, 8);
        hipMemset(out, 0, 16);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64), 0, 0, words, who, iters, xa, xb, out, delay);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long c = 0, stuck = 0;
        hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&stuck, out + 1, 8, hipMemcpyDeviceToHost);
        if (stuck) { printf("%-58s xcd %d -> %d : NO HAND-OFF (stuck at i = %lld)\n", name, xa, xb, stuck); fflush(stdout); return; }
        unsigned w[2];
        hipMemcpy(&w[0], words, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&w[1], words + 256, 4, hipMemcpyDeviceToHost);
        printf("%-58s xcd %d -> %d : %7.0f ns per hop  (ticks %lld, rate %d kHz, words %u %u, event %.0f ns per hop)\n", name, xa, xb, 1e6 * (double)c / rate / (2.0 * iters), c, rate, w[0], w[1], 1e6 * ms / (2.0 * iters)); fflush(stdout);
    };
    for (int xb = 0; xb < 2; ++xb) {
        run("store sc1, load sc1", k_pp<16, 16>, 0, xb);
        run("store sc0|sc1, load sc0|sc1", k_pp<17, 17>, 0, xb);
        run("store sc1, load sc0|sc1", k_pp<16, 17>, 0, xb);
        run("store sc1|nt, load sc1|nt", k_pp<18, 18>, 0, xb);
        run("store plain, load sc0 (same XCD only)", k_pp<0, 1>, 0, xb);
        run("store sc0, load sc0 (same XCD only)", k_pp<1, 1>, 0, xb);
    }
    run("store sc1, load sc1, replier sleeps ~2048 cycles", k_pp<16, 16>, 0, 1, 1);
    return 0;
}
