// Microbenchmark: does VALU work issue under v_mfma_f32_16x16x4_f32 on gfx950, from the same wave or from the
// SIMD's other wave?  (the tile kernel's loss epilogue).  Prints cycles per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: both interleaved in one wave, 3: wave-specialised (even waves MFMA, odd VALU)
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){seed, seed, seed, seed};
    float a = seed + lane, b = seed * 2 + lane;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + lane;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (w < 4));
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (w >= 4));
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);     // 48 v_fma per iteration
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 1e-3f);       // 8 v_exp + 8 v_mul
        }
        if (MODE == 2) {
            // interleave: per MFMA 4 VALU
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, int blocks, int iters)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, 8 * blocks * (threads / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= h.size();
    printf("%-44s threads %4d: %8.1f cycles/iter/wave (wall %.3f ms)\n", name, threads, avg / iters, ms);
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int iters = 2000, blocks = 256;
    printf("per iteration: 16 x v_mfma_f32_16x16x4_f32 (= 512 cycles at the issue rate), 48 v_fma + 8 v_mul + 8 v_exp\n");
    for (int threads : {256, 512}) {
        run<0>("MFMA only", threads, blocks, iters);
        run<1>("VALU only", threads, blocks, iters);
        run<2>("MFMA + VALU interleaved in every wave", threads, blocks, iters);
    }
    run<3>("8 waves: 4 MFMA-only + 4 VALU-only", 512, blocks, iters);
    return 0;
}
