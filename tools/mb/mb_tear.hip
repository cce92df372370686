// Microbenchmark (GPU box only): are 16-byte words torn?  k_gru_steps_ll publishes a unit's state as ONE dwordx4 write-through
// store (h row 0..2, tag) and consumers accept a word when its tag matches, so a load must never see the tag of one store with
// the values of another.  One writer block stores (i, i, i, i) to 1024 words again and again; every other block polls them with
// dwordx4 sc1 loads and counts words whose four components differ.
#include <cvae_intrin.h>
#include <stdio.h>

__global__ __launch_bounds__(256, 1) void k_tear(float* words, unsigned long long* bad, unsigned long long* seen, int iters) {
    const cvae_buf b = cvae_make_buf(words, 1024 * 16);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int i = 1; i <= 40 * iters; ++i) {     // the writer outlasts the pollers
            const float v = __builtin_bit_cast(float, (unsigned)i);
#pragma unroll
            for (int q = 0; q < 4; ++q) cvae_buf_store_f4_sc1(b, (unsigned)(tid + 256 * q) * 16u, 0, (f32x4){v, v, v, v});
        }
        return;
    }
    unsigned long long nb = 0, ns = 0;
    unsigned last = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile("" ::: "memory");     // (the optimizer does not treat the builtin's volatile bit as a barrier: it hoisted the loads)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = cvae_buf_poll_f4(b, (unsigned)(tid + 256 * q) * 16u, 0);
            const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
            const unsigned u0 = __builtin_bit_cast(unsigned, a0), u1 = __builtin_bit_cast(unsigned, a1);
            const unsigned u2 = __builtin_bit_cast(unsigned, a2), u3 = __builtin_bit_cast(unsigned, a3);
            if (u0 != u1 || u0 != u2 || u0 != u3) ++nb;
            if (q == 0 && u0 != last) { ++ns; last = u0; }
        }
    }
    if (nb) atomicAdd(bad, nb);
    atomicAdd(seen, ns);
    atomicMax(seen + 1, ns);
}

int main() {
    float* words;
    unsigned long long *bad, *seen;
    hipMalloc(&words, 1024 * 16);
    hipMemset(words, 0, 1024 * 16);
    hipMalloc(&bad, 8);
    hipMalloc(&seen, 16);
    hipMemset(bad, 0, 8);
    hipMemset(seen, 0, 16);
    const int iters = 200000;
    hipLaunchKernelGGL(k_tear, dim3(256), dim3(256), 0, 0, words, bad, seen, iters);
    hipDeviceSynchronize();
    unsigned long long hb = 0, hs = 0, hm = 0;
    hipMemcpy(&hm, seen + 1, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hs, seen, 8, hipMemcpyDeviceToHost);
    printf("255 polling blocks x 256 lanes x %d iterations x 4 words = %.3g word loads; distinct values seen by the q=0 loads: %llu; most distinct values seen by one lane: %llu; torn words: %llu\n",
           iters, 255.0 * 256 * iters * 4, hs, hm, hb);
    return 0;
}
