// Issue rate of v_cvt_pk_bf16_f32 (the RNE fp32 -> bf16 x 2 conversion of the three-way splits) against plain VALU operations:
// cycles per instruction per wave on one SIMD, 4 independent chains, s_memtime around 4096 instructions.
//   hipcc --offload-arch=gfx950 -O3 cvt_rate.hip -o cvt_rate.bin && ./cvt_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = 0.5f, b1 = 0.25f, b2 = 0.125f, b3 = 0.0625f;
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {  // cvt_pk: 4 independent
                u0 ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a0, b0}, bf2));
                u1 ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a1, b1}, bf2));
                u2 ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a2, b2}, bf2));
                u3 ^= __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a3, b3}, bf2));
                a0 += 1.f; a1 += 1.f; a2 += 1.f; a3 += 1.f;
            } else if (MODE == 1) {  // the same without the conversion: xor + add only
                u0 ^= __builtin_bit_cast(unsigned, a0);
                u1 ^= __builtin_bit_cast(unsigned, a1);
                u2 ^= __builtin_bit_cast(unsigned, a2);
                u3 ^= __builtin_bit_cast(unsigned, a3);
                a0 += 1.f; a1 += 1.f; a2 += 1.f; a3 += 1.f;
            } else {  // v_perm_b32 packing of the high halves (truncating split)
                u0 ^= __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b0), __builtin_bit_cast(unsigned, a0), 0x07060302u);
                u1 ^= __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b1), __builtin_bit_cast(unsigned, a1), 0x07060302u);
                u2 ^= __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b2), __builtin_bit_cast(unsigned, a2), 0x07060302u);
                u3 ^= __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b3), __builtin_bit_cast(unsigned, a3), 0x07060302u);
                a0 += 1.f; a1 += 1.f; a2 += 1.f; a3 += 1.f;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)(u0 ^ u1 ^ u2 ^ u3) + a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&cyc, 8);
    const char* names[3] = {"v_cvt_pk_bf16_f32 + xor + add", "xor + add", "v_perm_b32 + xor + add"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.f);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.f);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.f);
            (void)hipDeviceSynchronize();
        }
        long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("{\"ubench\": \"cvt_rate\", \"body\": \"%s\", \"cycles_per_group_of_4_chains_x3_ops\": %.2f}\n", names[m], (double)c / (256 * 4));
    }
    return 0;
}
