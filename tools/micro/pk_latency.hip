// Microbenchmark (GPU box): issue interval of v_pk_fma_f32 / v_pk_add_f32 at dependency distance 1, 2, 4, 8
// for 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 pk_latency.hip -o pk_latency && ./pk_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));

template <int DIST, bool FMA>
__global__ __launch_bounds__(1024) void k(cf *out, unsigned long long *cyc, int iters)
{
    cf r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = cf{1.0f + threadIdx.x * 1e-6f + i, 0.5f};
    cf w = cf{0.9999f, 1e-4f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            const int i = u % DIST;                      // DIST independent chains
            if (FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(w));
            else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(w));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    cf s = r[0];
#pragma unroll
    for (int i = 1; i < 8; i++) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int DIST, bool FMA> void run(const char *name, int threads)
{
    cf *out; unsigned long long *cyc, h[32];
    hipMalloc(&out, sizeof(cf) * 512 * 4); hipMalloc(&cyc, 8 * 32);
    const int iters = 2000;
    k<DIST, FMA><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    k<DIST, FMA><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 8 * 32, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0, w0 = h[1] - h[0];
    for (int w = 0; w < threads / 64; w++) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    h[0] = hi - lo;
    printf("%-10s dist %d, %d waves/SIMD: wave 0 %.2f cycles per instruction; all waves done: %.2f cycles per instruction per SIMD\n",
           name, DIST, threads / 256, (double) w0 / (iters * 64.0), (double) h[0] / (iters * 64.0) / (threads / 256));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int threads : {256, 512, 1024}) {
        run<1, true>("pk_fma", threads); run<2, true>("pk_fma", threads); run<4, true>("pk_fma", threads); run<8, true>("pk_fma", threads);
        run<8, false>("pk_add", threads);
    }
    return 0;
}
