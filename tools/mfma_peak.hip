// Register-only fp32 MFMA rate probe: what does v_mfma_f32_32x32x2_f32 sustain on this chip
// (a) with 1..4 independent accumulator chains per wave, (b) at 1..3 waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1e-3f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-3f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("acc chains %d, waves/SIMD %d, %d MFMAs/wave: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters * 8 * NACC, ms,
           flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int it : {54, 2000}) {      // 54*8 = 432 MFMAs/wave = one conv launch's worth; 2000 = steady state
        run<1>(1, it); run<2>(1, it / 2); run<4>(1, it / 4); run<1>(2, it / 2); run<1>(3, it / 3); run<4>(2, it / 8);
    }
    return 0;
}
