// bypass_rate.hip -- where do the R = 1 bypass kernel's 58 us of "arithmetic" go?  (86 .. 92 us for
// 168 MB, of which the knock-out builds attribute ~58 us to some six million vector instructions that should take 17.)
// The kernel's whole-block body in variants, one ingredient at a time, timed with HIP events on 2^24 samples x 2 channels.
//   hipcc --offload-arch=gfx950 -O3 -o bypass_rate tools/micro/bypass_rate.hip && ./bypass_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define TAB 10240
#define G 4
__device__ __forceinline__ int mix24(int adc, int dds) { const int m = adc * dds; return (m + 32) >> 6; }

// V: 0 full, 1 no stores, 2 no table (phase bits as the "table value"), 3 no mads (xor instead), 4 no packing (store raw words),
//    5 phase per sample from a multiply instead of the running add, 6 table index from 32-bit phase
template <int V, int NB>
__global__ __launch_bounds__(256, 4) void body(const short *__restrict__ adc, long n, const short *__restrict__ nco,
                                               short2 *__restrict__ out, long out_stride, u64 inc0, u64 inc1)
{
    __shared__ short tab[TAB];
    for (int i = threadIdx.x; i < TAB / 2; i += 256) ((u32 *) tab)[i] = ((const u32 *) nco)[i];
    __syncthreads();
    const u64 inc16[2] = {inc0 << 16, inc1 << 16};
    const long bs = G * 1024, nfb = n / bs;
    auto fetch = [&](long bk, int2 (&v)[G]) {
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = *(const int2 *) (adc + bk * bs + g * 1024 + 4 * (long) threadIdx.x);
    };
    long blk = blockIdx.x;
    if (blk >= nfb) return;
    int2 cur[G], nxt[G];
    fetch(blk, cur);
    for (; blk < nfb; blk += gridDim.x) {
        fetch(blk + gridDim.x < nfb ? blk + gridDim.x : blk, nxt);
        long bo = blk;
        asm volatile("" : "+s"(bo));
#pragma unroll
        for (int g = 0; g < G; g++) {
            const long t0 = bo * bs + g * 1024 + 4 * (long) threadIdx.x;
            const int2 v = cur[g];
            const short a[4] = {(short) v.x, (short) (v.x >> 16), (short) v.y, (short) (v.y >> 16)};
#pragma unroll
            for (int b = 0; b < NB; b++) {
                u64 ph = ((u64) t0 * (inc16[b] >> 16)) << 16;
                u32 ph32 = (u32) t0 * (u32) (inc16[b] >> 16);
                int4 w0; int *w = (int *) &w0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int ec, es;
                    if (V == 2) { ec = (int) (ph >> 51); es = (int) (ph >> 52); }
                    else if (V == 6) { ec = tab[(ph32 >> 19) + 2048]; es = tab[ph32 >> 19]; }
                    else { ec = tab[(ph >> 51) + 2048]; es = tab[ph >> 51]; }
                    int mi, mq;
                    if (V == 3) { mi = a[q] ^ ec; mq = a[q] ^ es; } else { mi = mix24(a[q], ec); mq = mix24(a[q], es); }
                    if (V == 4) w[q] = mi + mq; else w[q] = (int) (((u32) (mi >> 8) & 0xffffu) | ((u32) (mq >> 8) << 16));
                    if (V == 5) ph = ((u64) (t0 + q + 1) * (inc16[b] >> 16)) << 16; else ph += inc16[b];
                    ph32 += (u32) (inc16[b] >> 16);
                }
                if (V == 1) asm volatile("" :: "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w));
                else *(int4 *) (out + b * out_stride + t0) = w0;
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) cur[g] = nxt[g];
    }
}

// nset > 1: every launch takes another input block and another pair of output rows (nset x 160 MiB in all: past the 256 MiB
// Infinity Cache), as the bench's rotation of ADC blocks and the rest of a step's traffic make the library's kernel do
template <int V> float run(const short *adc, long n, const short *nco, short2 *out, int wgs, int nset = 1)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((body<V, 2>), dim3(wgs), dim3(256), 0, 0, adc, n, nco, out, n, 0x123456789ull, 0x0fedcba987ull);
    hipEventRecord(e0);
    const int reps = 24;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((body<V, 2>), dim3(wgs), dim3(256), 0, 0, adc + (long) (i % nset) * n, n, nco, out + (long) (i % nset) * 2 * n, n, 0x123456789ull, 0x0fedcba987ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main()
{
    const long n = 1l << 24;
    short *adc, *nco; short2 *out;
    const int NSET = 8;
    hipMalloc(&adc, n * 2 * NSET); hipMalloc(&nco, TAB * 2); hipMalloc(&out, n * 4 * 2 * NSET);
    std::vector<short> h(n); for (long i = 0; i < n; i++) h[i] = (short) (rand() & 0x3fff) - 8192;
    for (int k = 0; k < NSET; k++) hipMemcpy(adc + (long) k * n, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<short> t(TAB); for (int i = 0; i < TAB; i++) t[i] = (short) (rand() & 0x7fff) - 16384;
    hipMemcpy(nco, t.data(), TAB * 2, hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount * 4;
    printf("%s, %d workgroups, 2^24 samples x 2 channels\n", p.gcnArchName, wgs);
    printf("V0 full                         %7.1f us\n", run<0>(adc, n, nco, out, wgs));
    printf("V1 no stores                    %7.1f us\n", run<1>(adc, n, nco, out, wgs));
    printf("V2 no table reads               %7.1f us\n", run<2>(adc, n, nco, out, wgs));
    printf("V3 xor instead of the mixer mad %7.1f us\n", run<3>(adc, n, nco, out, wgs));
    printf("V4 no packing                   %7.1f us\n", run<4>(adc, n, nco, out, wgs));
    printf("V5 phase by multiply per sample %7.1f us\n", run<5>(adc, n, nco, out, wgs));
    printf("V6 32-bit phase for the index   %7.1f us\n", run<6>(adc, n, nco, out, wgs));
    printf("eight buffer sets in turn (1.3 GB: nothing of a launch's 168 MB is in a cache when it starts)\n");
    printf("V0 full                         %7.1f us\n", run<0>(adc, n, nco, out, wgs, NSET));
    printf("V1 no stores                    %7.1f us\n", run<1>(adc, n, nco, out, wgs, NSET));
    printf("V2 no table reads               %7.1f us\n", run<2>(adc, n, nco, out, wgs, NSET));
    return 0;
}
