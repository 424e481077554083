// Micro-benchmark: how much non-MFMA issue hides in the shadow of dependent v_mfma_f32_32x32x16_bf16 at one wave per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, const char* gsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(0.5f + i * 0.01f); }
    f32x16 x;
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
    float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    bf16x8 ld = a;
    const char* p = smem + lane * 16;
    typedef __attribute__((ext_vector_type(4))) float f4; typedef __attribute__((ext_vector_type(2))) float f2;
    f4 ring[8]; f2 tr[8];
    for (int i = 0; i < 8; ++i) { ring[i] = f4{0,0,0,0}; tr[i] = f2{0,0}; }
    const unsigned lds_addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 3) { ld = *reinterpret_cast<const bf16x8*>(p + ((j * 1024) & 32767)); asm volatile("" ::"v"(ld)); }
            if (MODE == 2 || MODE == 3) {
                asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %0, %0, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            }
            if (MODE == 4) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"); }
            if (MODE == 5) {   // one ds_read_b128 per MFMA, consumed 8 steps later (ring), counted wait by hand
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[j & 7]) : "v"(lds_addr), "i"((j * 1024) & 32767));
                asm volatile("s_waitcnt lgkmcnt(7)");
                asm volatile("" ::"v"(ring[(j + 1) & 7]));
            }
            if (MODE == 6) {   // two ds_read_b64_tr_b16 per MFMA, ring of 4 steps
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr[(j & 3) * 2]) : "v"(lds_addr), "i"((j * 512) & 32767));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr[(j & 3) * 2 + 1]) : "v"(lds_addr), "i"(((j * 512) & 32767) + 256));
                asm volatile("s_waitcnt lgkmcnt(6)");
                asm volatile("" ::"v"(tr[((j + 1) & 3) * 2]));
            }
            if (MODE == 7 && (j & 3) == 1) {   // one LDS-DMA piece (1 KiB) per 4 MFMAs
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + ((it * 8 + (j >> 2)) & 1023) * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + (j >> 2) * 1024), 16, 0, 0);
            }
            if (MODE == 8) {   // mode 5 + mode 2
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[j & 7]) : "v"(lds_addr), "i"((j * 1024) & 32767));
                asm volatile("s_waitcnt lgkmcnt(7)");
                asm volatile("" ::"v"(ring[(j + 1) & 7]));
                asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_exp_f32 %3, %3\n\tv_add_f32 %0, %0, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    float s = v0 + v1 + v2 + v3 + (float)ld[0];
    for (int i = 0; i < 8; ++i) s += ring[i][0] + tr[i][0];
    for (int r = 0; r < 16; ++r) s += x[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(float* out, int iters, const char* gsrc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 32768, 0, out, 4, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 32768, 0, out, iters, gsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 64;   // 64*32 = 2048 MFMAs per wave
    char* gsrc; hipMalloc(&gsrc, 4 << 20); hipMemset(gsrc, 0, 4 << 20);
    const char* names[] = {"MFMA only", "MFMA + ds_read_b128 (waited at once)", "MFMA + 4 VALU (1 exp)", "MFMA + ds_read(waited) + 4 VALU", "MFMA + 20 cycles of s_nop",
                           "MFMA + ds_read_b128 (ring of 8)", "MFMA + 2 tr_b16 reads (ring of 4)", "MFMA + 1 LDS-DMA piece / 4", "MFMA + ds_read ring + 4 VALU"};
    float t[9] = {run<0>(out, iters, gsrc), run<1>(out, iters, gsrc), run<2>(out, iters, gsrc), run<3>(out, iters, gsrc), run<4>(out, iters, gsrc),
                  run<5>(out, iters, gsrc), run<6>(out, iters, gsrc), run<7>(out, iters, gsrc), run<8>(out, iters, gsrc)};
    for (int m = 0; m < 9; ++m) printf("%-40s %8.1f us   %.1f ns per MFMA step\n", names[m], t[m] * 1e3, t[m] * 1e6 / (iters * 32));
    return 0;
}
