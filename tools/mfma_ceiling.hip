// What does this chip sustain on bare bf16 MFMA loops under its power cap?  Operands in
// registers (random or zero data), no memory traffic, one or two waves per SIMD, both shapes.
// Calibrates the GEMM roofline discussion in DESIGN.md: the dense bf16 peak (2.5 PFLOP/s at
// 2.4 GHz) is not reachable on random data because the clock is pulled down under load.
// Diagnostic only.  hipcc -O3 --offload-arch=gfx950 tools/mfma_ceiling.hip -o tools/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE> // 16 or 32
__global__ __launch_bounds__(256) void loop(const bf16x8 *in, float *out, int iters)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(gid * 8 + i) & 0xffff];
        b[i] = in[(gid * 8 + 4 + i) & 0xffff];
    }
    float s = 0.0f;
    if (SHAPE == 16) {
        f32x4 c[16] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s += c[i][0] + c[i][3];
    } else {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 2; ++r)   // same MACs per iteration as the 16x16x32 arm: 8 x 16384 = 16 x 8192
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * r], b[j + 2 * r], c[i * 2 + j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s += c[i][0] + c[i][15];
    }
    out[gid] = s;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    std::vector<unsigned short> h(8 * 65536);
    bf16x8 *d_in;
    float *d_out;
    hipMalloc(&d_in, h.size() * 2);
    hipMalloc(&d_out, 4 * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int zero = 0; zero < 2; ++zero) {
        unsigned long long x = 88172645463325252ull;
        for (auto &v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            // random sign and mantissa, exponent in a narrow band around 1.0 (no inf/nan)
            v = zero ? 0 : (unsigned short)(((x >> 20) & 0x807f) | ((0x7c + ((x >> 40) & 7)) << 7));
        }
        hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
            for (int shape = 16; shape <= 32; shape += 16) {
                const int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block)
                double total_ms = 0.0;
                long launches = 0;
                while (total_ms < seconds * 1e3) {
                    hipEventRecord(e0);
                    for (int r = 0; r < 4; ++r) {
                        if (shape == 16)
                            hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
                        else
                            hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    total_ms += ms;
                    launches += 4;
                }
                const double flops = 2.0 * 8192 * 16 * (double)iters * blocks * 4;
                printf("%s data, %d wave(s)/SIMD, v_mfma_f32_%s_bf16: %8.1f TFLOP/s  (%.3f ms per launch; the last %.0f%% of a %.0f s run)\n",
                       zero ? "zero  " : "random", waves_per_simd, shape == 16 ? "16x16x32" : "32x32x16",
                       flops * launches / (total_ms * 1e-3) / 1e12, total_ms / launches, 100.0, seconds);
                fflush(stdout);
            }
    }
    return 0;
}
