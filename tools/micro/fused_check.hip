// GPU check of the fused butterflies of kg_fft.h against the separate twiddle + butterfly forms they replace
// (both SIGNs): kg_tw_radix16_h vs kg_twiddle16 + kg_radix16, kg_cc_radix16_h vs conj-products + kg_radix16.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../flydog_sdr_gps_amd/csrc fused_check.hip -o fused_check && ./fused_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kg_fft.h"

template <int SIGN> __global__ void k_check(const float2 *in, const float2 *tw, const float2 *cd, float2 *out)
{
    const int t = threadIdx.x;
    cf x[16], y[16], xr[16], yr[16];
    kg_tw15 w;
    for (int j = 0; j < 16; j++) x[j] = xr[j] = kg_ld(&in[t * 16 + j]);
    for (int j = 0; j < 15; j++) w.w[j] = kg_ld(&tw[t * 15 + j]);
    kg_twiddle16<SIGN>(xr, w);
    kg_radix16<SIGN>(xr, yr);
    kg_tw_radix16_h<SIGN>(x, y, w, [](int) {});
    for (int j = 0; j < 16; j++) { kg_st(&out[(t * 4 + 0) * 16 + j], yr[j]); kg_st(&out[(t * 4 + 1) * 16 + j], y[j]); }
    cf c[16], d[16], p[16];
    for (int j = 0; j < 16; j++) { c[j] = kg_ld(&in[t * 16 + j]); d[j] = kg_ld(&cd[t * 16 + j]); p[j] = kg_cmulc(c[j], d[j]); }
    kg_radix16<SIGN>(p, yr);
    kg_cc_radix16_h<SIGN>(c, d, y, [](int) {});
    for (int j = 0; j < 16; j++) { kg_st(&out[(t * 4 + 2) * 16 + j], yr[j]); kg_st(&out[(t * 4 + 3) * 16 + j], y[j]); }
}

int main()
{
    const int T = 64;
    std::vector<float2> in(T * 16), tw(T * 15), cd(T * 16), out(T * 4 * 16);
    srand(3);
    auto rnd = []() { return (float) rand() / RAND_MAX - 0.5f; };
    for (auto &v : in) v = make_float2(rnd(), rnd());
    for (auto &v : cd) v = make_float2(rnd(), rnd());
    for (auto &v : tw) { const float a = 6.2831853f * rnd(); v = make_float2(cosf(a), sinf(a)); }
    float2 *d_in, *d_tw, *d_cd, *d_out;
    hipMalloc(&d_in, in.size() * 8); hipMalloc(&d_tw, tw.size() * 8); hipMalloc(&d_cd, cd.size() * 8); hipMalloc(&d_out, out.size() * 8);
    hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_cd, cd.data(), cd.size() * 8, hipMemcpyHostToDevice);
    int bad = 0;
    for (int sign = -1; sign <= 1; sign += 2) {
        if (sign < 0) hipLaunchKernelGGL(k_check<-1>, dim3(1), dim3(T), 0, 0, d_in, d_tw, d_cd, d_out);
        else hipLaunchKernelGGL(k_check<+1>, dim3(1), dim3(T), 0, 0, d_in, d_tw, d_cd, d_out);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
        for (int which = 0; which < 2; which++) {
            double worst = 0, mx = 0;
            for (int t = 0; t < T; t++)
                for (int j = 0; j < 16; j++) {
                    const float2 a = out[(t * 4 + 2 * which) * 16 + j], b = out[(t * 4 + 2 * which + 1) * 16 + j];
                    worst = fmax(worst, hypot(a.x - b.x, a.y - b.y));
                    mx = fmax(mx, hypot(a.x, a.y));
                }
            printf("SIGN %+d %s: max |fused - separate| / max |y| = %.3e\n", sign, which ? "conj-products + radix16" : "twiddle16 + radix16     ", worst / mx);
            if (!(worst / mx < 2e-6)) bad++;
        }
    }
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
