// Microbenchmark (GPU box): the 4096-point complex transform of the correlators and the waterfall two ways --
//   (a) kg_subfft4096 (kg_fft.h), what the kernels run: a 256-thread workgroup, 16 points per thread, three radix-16
//       passes, two exchanges through two 32 KiB LDS tiles with a workgroup barrier each, two workgroups per CU;
//   (b) the "radix-64 two-pass" form of DESIGN.md section 7 lead 2: ONE WAVE per transform, 64 points per lane, two
//       radix-64 passes (each 8 x 8 in registers with the W_64 constants), ONE exchange through a wave-private 32 KiB
//       tile (XOR-swizzled, conflict-free), one twiddle stage W_4096^(l m) from registers, no workgroup barrier at all;
//       one wave per SIMD (the 64 + 64 points alone are 256 registers).
// Both are checked against a direct double-precision DFT on the host and timed over many back-to-back transforms on a
// full GPU (every CU busy), operands resident in registers: transforms per microsecond and CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../flydog_sdr_gps_amd/csrc fft4096_wave.hip -o fft4096_wave && ./fft4096_wave
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kg_fft.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- (a) the product transform -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_wg(const float2 *in, float2 *out, const float2 *tab4096, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tileA = smem, *tileB = smem + 4096;
    const int t = threadIdx.x;
    kg_tw4096 tw;
    kg_tw4096_load(tw, tab4096, t);
    cf x[16], y[16];
    const float2 *src = in + (size_t) blockIdx.x * 4096;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld(&src[t + 256 * j]);
    for (int it = 0; it < iters; it++) {
        kg_subfft4096<+1>(x, y, tileA, tileB, tw, t);            // y[m]: output n = t + 256 m -- the next input's layout
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = y[j] * cf{1.0f / 64.0f, 1.0f / 64.0f};     // (keeps the values finite)
    }
    float2 *dst = out + (size_t) blockIdx.x * 4096;
#pragma unroll
    for (int j = 0; j < 16; j++) kg_st(&dst[t + 256 * j], x[j]);
}

// ---- (b) one wave per transform ------------------------------------------------------------------------------------
// 64-point transform of x[j] in registers, output y[m] = sum_j x[j] W_64^(SIGN j m): j = j0 + 8 j1, m = 8 m0... as
// 8 x 8: radix-8 over j1 for each j0, the W_64^(j0 p) constants, radix-8 over j0 for each p; y[p + 8 q].
template <int SIGN> KG_DEV void radix64(cf (&x)[64], cf (&y)[64])
{
    cf u[8][8];
#pragma unroll
    for (int j0 = 0; j0 < 8; j0++) {
        cf a[8], b[8];
#pragma unroll
        for (int j1 = 0; j1 < 8; j1++) a[j1] = x[j0 + 8 * j1];
        kg_radix8<SIGN>(a, b);                                   // b[p] = sum_j1 a[j1] W_8^(j1 p)
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int e = (j0 * p) & 63;
            if (e == 0) u[j0][p] = b[p];
            else {
                const cf w = cf{KG_W64[e][0], SIGN > 0 ? KG_W64[e][1] : -KG_W64[e][1]};
                u[j0][p] = kg_cmul(b[p], w);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        cf a[8], b[8];
#pragma unroll
        for (int j0 = 0; j0 < 8; j0++) a[j0] = u[j0][p];
        kg_radix8<SIGN>(a, b);                                   // b[q] = sum_j0 a[j0] W_8^(j0 q)
#pragma unroll
        for (int q = 0; q < 8; q++) y[p + 8 * q] = b[q];
    }
}

// lane l holds X[l + 64 j]; on return y[m] is the output at n = l + 64 m (the same layout: transforms chain)
template <int SIGN> KG_DEV void fft4096_wave(cf (&x)[64], cf (&y)[64], float2 *tile, const float2 *tw, int l)
{
    radix64<SIGN>(x, y);                                         // y[m] = A[n0 = l][k0 = m]
    // twiddle W_4096^(l m), then element (n0 = l, k0 = m) -> tile[64 m + (l ^ (m & 31))]
#pragma unroll
    for (int m = 0; m < 64; m++) {
        cf v = y[m];
        if (m) { const cf w = kg_ld(&tw[64 * (m - 1) + l]); v = kg_cmul(v, SIGN > 0 ? w : cf{w.x, -w.y}); }
        kg_st(&tile[64 * m + (l ^ (m & 31))], v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane k0 = l reads n0 = 0 .. 63: tile[64 l + (n0 ^ (l & 31))]
#pragma unroll
    for (int n0 = 0; n0 < 64; n0++) x[n0] = kg_ld_tile(&tile[64 * l + (n0 ^ (l & 31))]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    radix64<SIGN>(x, y);                                         // y[k1] = X[k0 + 64 k1], k0 = l
}

__global__ __launch_bounds__(256, 1) void k_wave(const float2 *in, float2 *out, const float2 *tab4096, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float2 *tile = smem + wv * 4096;
    float2 *tw = smem + 4 * 4096;                                // the workgroup's twiddles W_4096^(l m), [m - 1][l]: 63 x 64 x 8 B, shared by its four waves
    for (int e = threadIdx.x; e < 63 * 64; e += 256) tw[e] = tab4096[((e & 63) * ((e >> 6) + 1)) & 4095];
    __syncthreads();
    cf x[64], y[64];
    const float2 *src = in + ((size_t) blockIdx.x * 4 + wv) * 4096;
#pragma unroll
    for (int j = 0; j < 64; j++) x[j] = kg_ld(&src[l + 64 * j]);
    for (int it = 0; it < iters; it++) {
        fft4096_wave<+1>(x, y, tile, tw, l);
#pragma unroll
        for (int j = 0; j < 64; j++) x[j] = y[j] * cf{1.0f / 64.0f, 1.0f / 64.0f};
    }
    float2 *dst = out + ((size_t) blockIdx.x * 4 + wv) * 4096;
#pragma unroll
    for (int j = 0; j < 64; j++) kg_st(&dst[l + 64 * j], x[j]);
}

static double check(const std::vector<float2> &in, const std::vector<float2> &out, int which)
{
    // one pass (iters = 1): out = IDFT_unnormalised(in) / 64 for transform `which`
    const float2 *x = in.data() + (size_t) which * 4096, *y = out.data() + (size_t) which * 4096;
    double worst = 0, mx = 0;
    for (int n = 0; n < 4096; n += 37) {
        double re = 0, im = 0;
        for (int k = 0; k < 4096; k++) {
            const double a = 2.0 * M_PI * (double) ((long) k * n % 4096) / 4096.0;
            re += x[k].x * cos(a) - x[k].y * sin(a);
            im += x[k].x * sin(a) + x[k].y * cos(a);
        }
        re /= 64.0; im /= 64.0;
        worst = fmax(worst, hypot(y[n].x - re, y[n].y - im));
        mx = fmax(mx, hypot(re, im));
    }
    return worst / mx;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int nwg_a = cus * 2, nwg_b = cus;                      // (a): two workgroups per CU, one transform each; (b): four waves = four transforms
    const int ntr = nwg_b * 4 > nwg_a ? nwg_b * 4 : nwg_a;
    std::vector<float2> h_in((size_t) ntr * 4096), h_out((size_t) ntr * 4096), h_tab(4096);
    srand(1);
    for (auto &v : h_in) { v.x = (float) rand() / RAND_MAX - 0.5f; v.y = (float) rand() / RAND_MAX - 0.5f; }
    for (int k = 0; k < 4096; k++) { const double a = 2.0 * M_PI * k / 4096.0; h_tab[k] = make_float2((float) cos(a), (float) sin(a)); }
    float2 *d_in, *d_out, *d_tab;
    CHECK(hipMalloc(&d_in, h_in.size() * 8)); CHECK(hipMalloc(&d_out, h_out.size() * 8)); CHECK(hipMalloc(&d_tab, 4096 * 8));
    CHECK(hipMemcpy(d_in, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tab, h_tab.data(), 4096 * 8, hipMemcpyHostToDevice));
    const size_t lds_a = 2 * 4096 * 8, lds_b = 4 * 4096 * 8 + 63 * 64 * 8;
    CHECK(hipFuncSetAttribute((const void *) k_wg, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_a));
    CHECK(hipFuncSetAttribute((const void *) k_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_b));
    printf("%s, %d CUs\n", prop.name, cus);
    // correctness: one pass each
    hipLaunchKernelGGL(k_wg, dim3(nwg_a), dim3(256), lds_a, 0, d_in, d_out, d_tab, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost));
    printf("(a) kg_subfft4096, 256 threads x 16 points:   max error / max |X| = %.2e, %.2e\n", check(h_in, h_out, 0), check(h_in, h_out, nwg_a - 1));
    hipLaunchKernelGGL(k_wave, dim3(nwg_b), dim3(256), lds_b, 0, d_in, d_out, d_tab, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost));
    printf("(b) one wave per transform, 64 x 64 points:   max error / max |X| = %.2e, %.2e\n", check(h_in, h_out, 0), check(h_in, h_out, nwg_b * 4 - 1));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int rep = 0; rep < 3; rep++) {
        float ms_a, ms_b;
        hipLaunchKernelGGL(k_wg, dim3(nwg_a), dim3(256), lds_a, 0, d_in, d_out, d_tab, 50);      // clocks up
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_wg, dim3(nwg_a), dim3(256), lds_a, 0, d_in, d_out, d_tab, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_a, e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_wave, dim3(nwg_b), dim3(256), lds_b, 0, d_in, d_out, d_tab, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_b, e0, e1));
        const double ta = (double) nwg_a * iters / (ms_a * 1e3) / cus, tb = (double) nwg_b * 4 * iters / (ms_b * 1e3) / cus;
        printf("rep %d: (a) %.3f ms = %.4f transforms/us/CU (%.0f cycles per transform and CU at 2.4 GHz)   (b) %.3f ms = %.4f transforms/us/CU (%.0f cycles)   (b)/(a) = %.3f\n",
               rep, ms_a, ta, 2400.0 / ta, ms_b, tb, 2400.0 / tb, tb / ta);
    }
    return 0;
}
