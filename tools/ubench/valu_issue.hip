// VALU issue-rate calibration for gfx950: dependent / independent v_fma_f32 chains, 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b)
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = (float)threadIdx.x + (float)c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fmaf(x[c], a, b);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++) s += x[c];
    if (s == 12345.678f) out[0] = s;
}

// the same with IEEE divisions (each one a ~10-instruction dependent sequence): what k_project's canonical sequences are made of
template <int CHAINS>
__global__ __launch_bounds__(256) void k_div(float* out, int iters, float a, float b)
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = (float)threadIdx.x + (float)c + 1.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = a / x[c] + b;
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++) s += x[c];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
static void run_div(int wg_per_cu, int cus, float* d)
{
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_div<CHAINS>), dim3(cus * wg_per_cu), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_div<CHAINS>), dim3(cus * wg_per_cu), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double divs = (double)wg_per_cu * iters * 4.0 * CHAINS;
    printf("div chains=%d waves/SIMD=%d  %.3f ms  -> %.1f ns per wave-division per SIMD = %.1f cycles @2.4GHz\n", CHAINS, wg_per_cu, ms,
           ms * 1e6 / divs, ms * 1e6 / divs * 2.4);
}

template <int CHAINS>
static void run(int wg_per_cu, int cus, float* d)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_fma<CHAINS>), dim3(cus * wg_per_cu), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_fma<CHAINS>), dim3(cus * wg_per_cu), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each workgroup = 4 waves, one per SIMD: a SIMD issues wg_per_cu * iters * 16 * CHAINS wave-instructions
    const double insts = (double)wg_per_cu * iters * 16.0 * CHAINS;
    printf("chains=%d waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-inst per SIMD = %.2f cycles @2.4GHz\n", CHAINS, wg_per_cu, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, cus, p.clockRate);
    float* d; hipMalloc(&d, 4096);
    for (int w : {1, 2, 4, 8}) run<1>(w, cus, d);
    for (int w : {1, 2, 4, 8}) run<2>(w, cus, d);
    for (int w : {1, 2, 4}) run<4>(w, cus, d);
    for (int w : {4, 8}) run_div<1>(w, cus, d);
    for (int w : {2, 4}) run_div<2>(w, cus, d);
    run_div<4>(2, cus, d);
    return 0;
}
