// Compile-only probe: which s_waitcnt does hipcc put in front of MFMAs fed by a ring of ds_read_b128 fragments -- alone (k<0>),
// with an LDS-DMA in flight (k<1>: lgkmcnt(0) every ring instead of counted waits), with an opaque scalar (k<2>)?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only tools/waitcnt_probe.hip -o - | grep -E 'ds_read|v_mfma|s_waitcnt'
// docs/LABBOOK.md R4.14.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
template <int KIND>
__global__ __launch_bounds__(256) void k(const f32x4 *in, f32x4 *out, int n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    reinterpret_cast<f32x4 *>(smem)[lane] = in[lane];
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 b = __builtin_bit_cast(bf16x8, in[lane + 256]);
    bf16x8 ring[4];
    auto rd = [&](int f) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(smem + lane * 16 + f * 4096));
    };
    for (int it = 0; it < n; ++it) {
        if (KIND & 1)   /* an LDS-DMA in flight */
            __builtin_amdgcn_global_load_lds((gptr_t)(in + 512 + lane + 64 * it), (lptr_t)(smem + 65536 + (it & 1) * 4096), 16, 0, 0);
        if (KIND & 2) { /* opaque scalar */
            const f32x4 *p = in + it;
            asm volatile("" : "+s"(p));
            b = __builtin_bit_cast(bf16x8, p[lane]);
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) ring[f] = rd(f);
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if (f + 3 < 16) ring[(f + 3) & 3] = rd(f + 3);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[f & 3], b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND & 1) __syncthreads();
    }
    out[lane] = acc;
}
template __global__ void k<0>(const f32x4 *, f32x4 *, int);
template __global__ void k<1>(const f32x4 *, f32x4 *, int);
template __global__ void k<2>(const f32x4 *, f32x4 *, int);
