// What the vector pipe delivers by occupancy (gfx950, MI355X): a pure stream of ONE instruction, eight independent chains per
// lane, one workgroup of 4 x W waves per CU on all 256 CUs, timed with HIP events (so at the clock the chip sustains under that load).
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define OP8(ins, tail) ins " %0, %0, %8" tail "\n\t" ins " %1, %1, %8" tail "\n\t" ins " %2, %2, %8" tail "\n\t" ins " %3, %3, %8" tail "\n\t" ins " %4, %4, %8" tail "\n\t" ins " %5, %5, %8" tail "\n\t" ins " %6, %6, %8" tail "\n\t" ins " %7, %7, %8" tail
template <int OP> __global__ void k(float *out, int iters, float seed, long long *cyc)
{
    const long long t0 = clock64();
    cf a[8], b = {seed, seed * 0.5f};
    float c[8], d = seed;
    for (int j = 0; j < 8; j++) { a[j] = cf{seed * j, threadIdx.x * 1e-3f}; c[j] = seed * j + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(asm volatile(OP8("v_pk_fma_f32", ", %0") : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile(OP8("v_pk_add_f32", "") : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile(OP8("v_pk_mul_f32", "") : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile(OP8("v_fma_f32", ", %0") : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 4) { REP16(asm volatile(OP8("v_add_f32", "") : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 5) { REP16(asm volatile(OP8("v_fmac_f32", "") : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 6) { REP16(asm volatile("v_mov_b64 %0, %8\n\tv_mov_b64 %1, %8\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %8\n\tv_mov_b64 %5, %8\n\tv_mov_b64 %6, %8\n\tv_mov_b64 %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
    }
    const long long t1 = clock64();
    if (cyc && blockIdx.x == 7 && threadIdx.x == 0) *cyc = t1 - t0;
    float s = 0;
    for (int j = 0; j < 8; j++) s += a[j].x + a[j].y + c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> static void run(const char *name, int w)
{
    const int iters = 2000, threads = 256 * w;
    float *out; hipMalloc(&out, 256 * 1024 * 4); long long *cyc; hipMalloc(&cyc, 8); long long hc = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, threads>>>(out, 10, 1.0f, nullptr); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<256, threads>>>(out, iters, 1.0f, cyc); hipEventRecord(e1); hipEventSynchronize(e1); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double) iters * 128 * w);
    printf("%-14s %d waves/SIMD: %.2f ns per wave-instruction and SIMD = %.2f cycles at 2.4 GHz; a pure stream: %.0f G wave-instructions/s chip-wide\n", name, w, ns, ns * 2.4, 1024.0 / ns);
    hipFree(out);
}
int main()
{
    run<0>("v_pk_fma_f32", 1); run<1>("v_pk_add_f32", 1); run<2>("v_pk_mul_f32", 1); run<3>("v_fma_f32", 1); run<4>("v_add_f32", 1); run<5>("v_fmac_f32", 1); run<6>("v_mov_b64", 1);
    run<0>("v_pk_fma_f32", 2); run<1>("v_pk_add_f32", 2); run<2>("v_pk_mul_f32", 2); run<3>("v_fma_f32", 2); run<4>("v_add_f32", 2); run<5>("v_fmac_f32", 2); run<6>("v_mov_b64", 2);
    run<0>("v_pk_fma_f32", 4); run<1>("v_pk_add_f32", 4); run<3>("v_fma_f32", 4); run<4>("v_add_f32", 4);
    return 0;
}
