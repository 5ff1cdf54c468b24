// Cycles per instruction of the fp16 MFMA shapes on gfx950: the CDNA3-generation v_mfma_f32_16x16x16_f16 /
// 32x32x8_f16 against CDNA4's double-K v_mfma_f32_16x16x32_f16 / 32x32x16_f16.  One wave per SIMD, operands in registers,
// ZERO data (so that the power cap does not set the clock), s_memtime around the loop.  Diagnostic only: the question
// behind round 4's change of csrc/attention_h16.hip from the 16-deep to the 32-deep instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_rate_probe.hip -o tools/mfma_f16_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void loop(float *out, unsigned long long *cycles, int iters)
{
    half4 a4 = {}, b4 = {};
    half8 a8 = {}, b8 = {};
    f32x4 c[8] = {};
    f32x16 d[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0);
            if (KIND == 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
            if (KIND == 2) d[i & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d[i & 3], 0, 0, 0);
            if (KIND == 3) d[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[i & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += d[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        cycles[KIND] = t1 - t0;
}

int main()
{
    float *out;
    unsigned long long *cyc, h[4];
    hipMalloc(&out, 4 * 256 * 256);
    hipMalloc(&cyc, 32);
    const int iters = 100000;
    const char *names[4] = {"v_mfma_f32_16x16x16_f16 ( 8192 FLOP)", "v_mfma_f32_16x16x32_f16 (16384 FLOP)", "v_mfma_f32_32x32x8_f16  (16384 FLOP)",
                            "v_mfma_f32_32x32x16_f16 (32768 FLOP)"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 4; ++k) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(loop<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (k == 1) hipLaunchKernelGGL(loop<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (k == 2) hipLaunchKernelGGL(loop<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (k == 3) hipLaunchKernelGGL(loop<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        const double flop = (k == 0 ? 8192.0 : k == 3 ? 32768.0 : 16384.0) * 8 * iters * 1024;
        printf("%s: %.3f ms, %.1f TFLOP/s chip-wide (one wave per SIMD, zero operands); s_memtime ticks per instruction %.2f\n", names[k], ms,
               flop / (ms * 1e-3) / 1e12, (double)h[k] / (8.0 * iters));
    }
    return 0;
}
