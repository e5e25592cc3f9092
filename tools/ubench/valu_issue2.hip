// VALU issue-rate calibration for gfx950, second version (round 2): answers what round 1's valu_issue.hip could not.
//   * instruction streams are inline asm (the compiler cannot SLP-pack two scalar chains into v_pk_fma_f32 -- which is what
//     round 1's "two independent chains" variant silently measured -- and cannot add loop bookkeeping between them);
//   * 256 instructions per loop trip, so the three scalar loop instructions are < 1.5 % of the stream;
//   * time is taken INSIDE the kernel: s_memtime (shader clock ticks) for cycles and s_memrealtime (100 MHz) for the wall
//     clock, so the cycles per instruction do not depend on an assumed 2.4 GHz and the clock actually held is printed.
// Build: hipcc -O3 --offload-arch=gfx950 valu_issue2.hip -o valu_issue2
// Output: one line per (stream, waves per SIMD): shader cycles per wave-instruction per SIMD, and the clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

enum { S_FMA_DEP = 0, S_FMA_IND4, S_PK_DEP, S_PK_IND4, S_EXP_IND4, S_WALK, S_FMA_IND2, S_MUL_DEP, S_COUNT };
static const char* kNames[S_COUNT] = {"v_fma_f32 one dependent chain", "v_fma_f32 four independent chains", "v_pk_fma_f32 one dependent chain",
                                      "v_pk_fma_f32 four independent chains", "v_exp_f32 four independent", "walk-step mix (5 fma, 1 mul, 1 exp, 3 cvt_ubyte, 4 fmac-like)",
                                      "v_fma_f32 two independent chains", "v_mul_f32 one dependent chain"};
static const int kPerBlock[S_COUNT] = {16, 16, 16, 16, 16, 14, 16, 16};      // wave-instructions per asm block

#define R4(X) X X X X
#define R16(X) R4(R4(X))

template <int STREAM>
__global__ __launch_bounds__(256) void k_stream(unsigned long long* out, int iters, float a, float b)
{
    float x0 = (float)threadIdx.x, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float y0 = x0 * 0.5f, y1 = x1 * 0.5f, y2 = x2 * 0.5f, y3 = x3 * 0.5f;      // second halves of the packed pairs
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {x0, y0}, p1 = {x1, y1}, p2 = {x2, y2}, p3 = {x3, y3}, pa = {a, a}, pb = {b, b};
    unsigned int cw = threadIdx.x * 2654435761u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (STREAM == S_FMA_DEP) {
            asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
        } else if (STREAM == S_MUL_DEP) {
            asm volatile(R16("v_mul_f32 %0, %0, %1\n") : "+v"(x0) : "v"(a));
        } else if (STREAM == S_FMA_IND4) {
            asm volatile(R4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
        } else if (STREAM == S_FMA_IND2) {
            asm volatile(R4("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                         : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        } else if (STREAM == S_PK_DEP) {
            asm volatile(R16("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pa), "v"(pb));
        } else if (STREAM == S_PK_IND4) {
            asm volatile(R4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (STREAM == S_EXP_IND4) {
            asm volatile(R4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (STREAM == S_WALK) {
            // the shape of one step of k_composite's walk (dependencies as there): 4 fma geometry, r2 mul + fma, exponent fma, exp,
            // weight mul, 3 byte->float, 3 colour fmac, transmittance
            asm volatile(
                "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n"
                "v_mul_f32 %2, %0, %0\n v_fma_f32 %2, %1, %1, %2\n v_fma_f32 %2, %2, %4, %5\n v_exp_f32 %2, %2\n v_mul_f32 %2, %2, %3\n"
                "v_cvt_f32_ubyte0 %0, %6\n v_cvt_f32_ubyte1 %1, %6\n v_fmac_f32 %3, %2, %0\n v_fmac_f32 %3, %2, %1\n v_sub_f32 %3, %3, %2\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(cw));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const float s = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63u) == 0) {
        const unsigned w = blockIdx.x * 4u + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = (r1 - r0) | (s == 12345.678f ? 1ull << 63 : 0ull);
    }
}

template <int STREAM>
static void run(int waves_per_simd, int cus, unsigned long long* d, std::vector<unsigned long long>& h)
{
    const int iters = 4000;
    const int wgs = cus * waves_per_simd;          // 256-thread workgroups: 4 waves, one per SIMD of a CU
    hipLaunchKernelGGL((k_stream<STREAM>), dim3(wgs), dim3(256), 0, 0, d, 50, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_stream<STREAM>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, (size_t)wgs * 4 * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int w = 0; w < wgs * 4; w++) {
        const double c = (double)h[2 * w], r = (double)(h[2 * w + 1] & ~(1ull << 63));
        cyc.push_back(c);
        ghz.push_back(c / (r * 10.0));             // ticks / (r x 10 ns) = GHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double insts = (double)iters * kPerBlock[STREAM] * (STREAM == S_WALK ? 1 : 1);
    const double med = cyc[cyc.size() / 2];
    // a SIMD hosts waves_per_simd such waves at once: cycles per wave-instruction PER SIMD = wave cycles / (instructions x waves on the SIMD)
    printf("%-62s waves/SIMD=%d  %7.2f cycles per inst per wave  %5.2f cycles per inst per SIMD  clock %.2f GHz\n", kNames[STREAM], waves_per_simd,
           med / insts, med / insts / waves_per_simd, ghz[ghz.size() / 2]);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s CUs=%d nominal clock=%d kHz\n", p.gcnArchName, cus, p.clockRate);
    unsigned long long* d; hipMalloc(&d, (size_t)cus * 8 * 4 * 16 + 4096);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 2 + 16);
    for (int w : {1, 2, 4, 8}) run<S_FMA_DEP>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_MUL_DEP>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_FMA_IND2>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_FMA_IND4>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_PK_DEP>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_PK_IND4>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_EXP_IND4>(w, cus, d, h);
    for (int w : {1, 2, 4, 8}) run<S_WALK>(w, cus, d, h);
    return 0;
}
