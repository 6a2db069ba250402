// Microbenchmark (GPU box): the 16-point complex transform of the correlators' passes two ways --
//   (a) kg_radix16 (kg_fft.h): one transform per LANE, 80 packed vector instructions, what the kernels run;
//   (b) as a matrix product on the MFMA pipe: Y = F X with F the 16 x 16 DFT matrix, X = 16 points x 16 transforms,
//       four real 16 x 16 x 16 products (Yr = Fr Xr - Fi Xi, Yi = Fr Xi + Fi Xr) = 16 x v_mfma_f32_16x16x4_f32 per
//       16 transforms and wave (fp32 MFMA: there is no faster fp32-accurate matrix instruction on gfx950).
// Both are checked against a direct DFT in double on the host, then timed over many repetitions with one wave per SIMD
// and with two (cycles per transform and wave from s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -I../../flydog_sdr_gps_amd/csrc mfma_dft16.hip -o mfma_dft16 && ./mfma_dft16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "kg_fft.h"

typedef float f4 __attribute__((ext_vector_type(4)));

// (a) every lane transforms its own sixteen points
__global__ __launch_bounds__(512) void k_valu(const float2 *in, float2 *out, unsigned long long *cyc, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    cf x[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld(&in[t * 16 + j]);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it + 1 < iters; it += 2) {                       // (fed back unscaled: the values overflow to inf after a
        kg_radix16<-1>(x, y);                                         // few dozen repetitions, which costs the pipes nothing)
        kg_radix16<-1>(y, x);
    }
    if (iters & 1) {
        kg_radix16<-1>(x, y);
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = y[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = x[j] * cf{0.25f, 0.25f};
#pragma unroll
    for (int j = 0; j < 16; j++) kg_st(&out[t * 16 + j], x[j]);
    if ((threadIdx.x & 63) == 0) { cyc[2 * (t >> 6)] = t0; cyc[2 * (t >> 6) + 1] = t1; }
}

// (b) a wave transforms sixteen vectors at a time: X[k][n] (point k of transform n) in the B operand layout
// (lane l holds rows 4 kb + l / 16 of column l % 16 for the four k-blocks kb), F in the A layout, Y in the C layout
// (lane l, register i: row 4 (l / 16) + i, column l % 16).  The output layout differs from the input's: the repetition
// feeds Y back as if it were in B layout (a row permutation of the data, irrelevant for the timing; the check uses one pass).
__global__ __launch_bounds__(512) void k_mfma(const float2 *in, float2 *out, unsigned long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int col = lane & 15, grp = lane >> 4;
    float fr[4], fi[4], nfi[4];             // A operands: F[row = col][k = 4 kb + grp]
    float xr[4], xi[4];                     // B operands: X[k = 4 kb + grp][n = col]
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const int k = 4 * kb + grp;
        const float a = -2.0f * 3.14159265358979323846f * (float) ((col * k) & 15) / 16.0f;
        fr[kb] = cosf(a); fi[kb] = sinf(a); nfi[kb] = -fi[kb];
        const float2 v = in[(wave * 16 + col) * 16 + k];
        xr[kb] = v.x; xi[kb] = v.y;
    }
    __syncthreads();
    f4 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        yr = f4{0, 0, 0, 0}; yi = f4{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            yr = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[kb], xr[kb], yr, 0, 0, 0);
            yr = __builtin_amdgcn_mfma_f32_16x16x4f32(nfi[kb], xi[kb], yr, 0, 0, 0);
            yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[kb], xi[kb], yi, 0, 0, 0);
            yi = __builtin_amdgcn_mfma_f32_16x16x4f32(fi[kb], xr[kb], yi, 0, 0, 0);
        }
        if (it + 1 < iters) {
#pragma unroll
            for (int kb = 0; kb < 4; kb++) { xr[kb] = yr[kb]; xi[kb] = yi[kb]; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // C layout: register i of lane l = Y[row 4 grp + i][transform col]
#pragma unroll
    for (int i = 0; i < 4; i++) out[(wave * 16 + col) * 16 + 4 * grp + i] = make_float2(yr[i], yi[i]);
    if (lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

static double spread(const std::vector<unsigned long long> &c, int nw)
{
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nw; w++) { if (c[2 * w] < lo) lo = c[2 * w]; if (c[2 * w + 1] > hi) hi = c[2 * w + 1]; }
    return (double) (hi - lo);
}

int main()
{
    const int threads = 512;                                        // 8 waves = 2 per SIMD of one CU; 256: one per SIMD
    const int nt_valu = threads, nt_mfma = (threads / 64) * 16;     // transforms per launch
    std::vector<float2> h(nt_valu * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float) ((i * 7919) % 257) / 257.0f - 0.5f, (float) ((i * 104729) % 263) / 263.0f - 0.5f);
    float2 *d_in, *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_in, sizeof(float2) * h.size()); hipMalloc(&d_out, sizeof(float2) * h.size()); hipMalloc(&d_cyc, 8 * 2 * 64);
    hipMemcpy(d_in, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice);
    std::vector<float2> o(h.size());
    std::vector<unsigned long long> c(2 * 64);
    auto check = [&](const char *name, int ntr, int scale_iters) {
        double worst = 0;
        for (int n = 0; n < ntr; n++)
            for (int m = 0; m < 16; m++) {
                double re = 0, im = 0;
                for (int j = 0; j < 16; j++) {
                    const double a = -2.0 * M_PI * (double) ((j * m) & 15) / 16.0;
                    re += h[n * 16 + j].x * cos(a) - h[n * 16 + j].y * sin(a);
                    im += h[n * 16 + j].x * sin(a) + h[n * 16 + j].y * cos(a);
                }
                const double s = scale_iters ? 0.25 : 1.0;
                const double e = fmax(fabs(o[n * 16 + m].x - s * re), fabs(o[n * 16 + m].y - s * im));
                if (e > worst) worst = e;
            }
        printf("%-6s one pass against a double DFT: max abs error %.2e over %d transforms\n", name, worst, ntr);
    };
    // numerics: one pass each
    hipLaunchKernelGGL(k_valu, dim3(1), dim3(threads), 0, 0, d_in, d_out, d_cyc, 1);
    hipMemcpy(o.data(), d_out, sizeof(float2) * o.size(), hipMemcpyDeviceToHost);
    check("valu", nt_valu, 1);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, d_in, d_out, d_cyc, 1);
    hipMemcpy(o.data(), d_out, sizeof(float2) * o.size(), hipMemcpyDeviceToHost);
    check("mfma", nt_mfma, 0);
    // timing
    const int iters = 4000;
    for (int th : {256, 512}) {
        const int nw = th / 64;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_valu, dim3(1), dim3(th), 0, 0, d_in, d_out, d_cyc, iters);
        hipMemcpy(c.data(), d_cyc, 8 * 2 * nw, hipMemcpyDeviceToHost);
        const double cv = spread(c, nw) / iters;            // cycles per repetition: every wave did 64 transforms
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_mfma, dim3(1), dim3(th), 0, 0, d_in, d_out, d_cyc, iters);
        hipMemcpy(c.data(), d_cyc, 8 * 2 * nw, hipMemcpyDeviceToHost);
        const double cm = spread(c, nw) / iters;            // every wave did 16 transforms
        const double per_simd = nw / 4.0;
        printf("%d wave(s) per SIMD: packed-vector radix-16 %.0f cycles per 64 transforms and wave = %.2f cycles per transform and SIMD;"
               " MFMA 16x16x4 f32 %.0f cycles per 16 transforms and wave = %.2f per transform and SIMD  (x%.1f)\n",
               nw / 4, cv, cv / (64.0 * per_simd), cm, cm / (16.0 * per_simd), (cm / 16.0) / (cv / 64.0));
    }
    return 0;
}
