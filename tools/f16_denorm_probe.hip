// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and does the fp32->fp16 conversion
// produce them?  Decides whether the 2 x fp16 operand split needs its inputs pre-scaled.
// Diagnostic only.  hipcc -O2 --offload-arch=gfx950 tools/f16_denorm_probe.hip -o tools/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, float tiny, float big)
{
    half8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)tiny;   // 2^-20: an fp16 subnormal (min normal 2^-14)
        b[e] = (_Float16)big;
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        out[1] = (float)a[0];
    }
}
int main()
{
    float *d, h[2];
    hipMalloc(&d, 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 9.5367431640625e-07f, 1024.0f);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("fp16(2^-20) read back as %.10g (expected 9.5367431640625e-07 if the conversion keeps subnormals)\n", h[1]);
    printf("sum over k=32 of 2^-20 * 1024 = %.10g (expected 0.03125 if the MFMA honours subnormal inputs, 0 if it flushes)\n", h[0]);
    return 0;
}
