// Development probe: do VALU instructions of the waves of a SIMD issue in the shadow of fp32 MFMAs?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o tools/probes/mfma_valu_probe && tools/probes/mfma_valu_probe
// Every wave runs ITER trips of (M independent v_mfma_f32_16x16x4_f32, V independent v_pk_add_f32); 1 workgroup per CU with
// W waves per SIMD.  Prints cycles per trip per SIMD: max(M x 32, V x 4 x W) would be full overlap, the sum none.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int M, int V>
__global__ void k_probe(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[6];
    f32x2 v[8];
    for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){(float)threadIdx.x, 1.f};
    const float a = (float)(threadIdx.x & 3), b = (float)(threadIdx.x & 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < M; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 6], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < V; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i % 8]) : "v"(v[(i + 1) % 8]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int M, int V>
void run(int waves_per_simd) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    k_probe<M, V><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_probe<M, V><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // shader clock ~2.4 GHz: cycles per trip per SIMD from the wall time
    const double cycles = ms * 1e-3 * 2.4e9 / iters;
    printf("M=%d MFMA  V=%2d pk_add  %d waves/SIMD: %7.1f cycles per trip (all waves of a SIMD: %d MFMA = %d pipe cycles, %d VALU = %d issue cycles)\n",
           M, V, waves_per_simd, cycles, M * waves_per_simd, M * waves_per_simd * 32, V * waves_per_simd, V * waves_per_simd * 4);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4}) {
        run<6, 0>(w); run<0, 32>(w); run<6, 8>(w); run<6, 16>(w); run<6, 32>(w); run<6, 48>(w); run<2, 32>(w);
    }
    return 0;
}
