// Issue rate of the integer adds the DDC run passes are made of (gfx950): v_add_u32, v_lshl_add_u64, the
// v_add_co_u32 / v_addc_co_u32 pair, v_mad_i32_i24, v_add3_u32.  Eight independent chains per lane, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o add64_rate add64_rate.hip && ./add64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP> __global__ void k(u64 *out, int iters, u64 seed)
{
    u64 a[8], b = seed + threadIdx.x;
    unsigned c[8], d = (unsigned) seed * 3 + threadIdx.x;
    for (int j = 0; j < 8; j++) { a[j] = seed * (j + 1) + threadIdx.x; c[j] = (unsigned) a[j]; }
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 1) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\tv_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %8, %1\n\tv_mad_i32_i24 %1, %1, %8, %2\n\tv_mad_i32_i24 %2, %2, %8, %3\n\tv_mad_i32_i24 %3, %3, %8, %4\n\tv_mad_i32_i24 %4, %4, %8, %5\n\tv_mad_i32_i24 %5, %5, %8, %6\n\tv_mad_i32_i24 %6, %6, %8, %7\n\tv_mad_i32_i24 %7, %7, %8, %0" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 3) { REP16(asm volatile("v_add3_u32 %0, %0, %8, %1\n\tv_add3_u32 %1, %1, %8, %2\n\tv_add3_u32 %2, %2, %8, %3\n\tv_add3_u32 %3, %3, %8, %4\n\tv_add3_u32 %4, %4, %8, %5\n\tv_add3_u32 %5, %5, %8, %6\n\tv_add3_u32 %6, %6, %8, %7\n\tv_add3_u32 %7, %7, %8, %0" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(d));) }
        if (OP == 4) { REP16(asm volatile("v_lshrrev_b64 %0, 3, %0\n\tv_lshrrev_b64 %1, 3, %1\n\tv_lshrrev_b64 %2, 3, %2\n\tv_lshrrev_b64 %3, 3, %3\n\tv_lshrrev_b64 %4, 3, %4\n\tv_lshrrev_b64 %5, 3, %5\n\tv_lshrrev_b64 %6, 3, %6\n\tv_lshrrev_b64 %7, 3, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(d), "v"(c[0]) : "vcc");) }
    }
    u64 s = 0;
    for (int j = 0; j < 8; j++) s += a[j] + c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> static void run(const char *name, int waves_per_simd)
{
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;          // one workgroup per CU
    u64 *out;
    hipMalloc(&out, 256 * 1024 * sizeof(u64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, threads>>>(out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256, threads>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double) iters * 16 * 8 * waves_per_simd;       // wave-instructions per SIMD
    printf("%-16s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD at 2.4 GHz (%.3f ms)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / insts, ms);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_add_u32", 1); run<1>("v_lshl_add_u64", 1); run<2>("v_mad_i32_i24", 1); run<3>("v_add3_u32", 1); run<4>("v_lshrrev_b64", 1); run<5>("v_mad_u64_u32", 1); }
        if (w == 2) { run<0>("v_add_u32", 2); run<1>("v_lshl_add_u64", 2); run<2>("v_mad_i32_i24", 2); run<3>("v_add3_u32", 2); run<4>("v_lshrrev_b64", 2); run<5>("v_mad_u64_u32", 2); }
        if (w == 4) { run<0>("v_add_u32", 4); run<1>("v_lshl_add_u64", 4); run<2>("v_mad_i32_i24", 4); run<3>("v_add3_u32", 4); run<4>("v_lshrrev_b64", 4); run<5>("v_mad_u64_u32", 4); }
    }
    return 0;
}
