// Development probe: the rate of ds_add_f32 (no return) into an LDS table of rows, for the DistMult backward's scatter
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_atomic_probe.hip -o tools/probes/bin/lds_atomic_probe && gpurun -- tools/probes/bin/lds_atomic_probe
// A workgroup of 16 waves; a wave trip = 16 quads, each quad adds a 16-column row slice (a float4 per lane, four ds_add_f32) to a
// pseudo-random row.  STRIDE = floats per row in LDS, ROT = lane l of a quad adds element (j + l) % 4 in instruction j.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int STRIDE, int ROT, int READS, int KIND>
__global__ __launch_bounds__(1024) void k_probe(float* out, int iters, int rows, unsigned long long* cyc) {
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < rows * STRIDE; i += 1024) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, l4 = lane & 3, quad = lane >> 2;
    unsigned int s = (threadIdx.x >> 6) * 7919u + quad * 104729u + blockIdx.x * 31u + 12345u;
    float v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned int row = (s >> 10) % (unsigned)rows;
        const unsigned int base = (row * STRIDE + 4 * l4) * 4;
        if (READS) {
            float4 r;
            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(base));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += r.x + r.w;
        }
        const unsigned int o0 = base + 4 * (ROT ? ((0 + l4) & 3) : 0), o1 = base + 4 * (ROT ? ((1 + l4) & 3) : 1),
                           o2 = base + 4 * (ROT ? ((2 + l4) & 3) : 2), o3 = base + 4 * (ROT ? ((3 + l4) & 3) : 3);
        if (KIND == 0) {
            asm volatile("ds_add_f32 %0, %1" :: "v"(o0), "v"(v0) : "memory");
            asm volatile("ds_add_f32 %0, %1" :: "v"(o1), "v"(v1) : "memory");
            asm volatile("ds_add_f32 %0, %1" :: "v"(o2), "v"(v2) : "memory");
            asm volatile("ds_add_f32 %0, %1" :: "v"(o3), "v"(v3) : "memory");
        } else if (KIND == 1) {
            asm volatile("ds_add_u32 %0, %1" :: "v"(o0), "v"(v0) : "memory");
            asm volatile("ds_add_u32 %0, %1" :: "v"(o1), "v"(v1) : "memory");
            asm volatile("ds_add_u32 %0, %1" :: "v"(o2), "v"(v2) : "memory");
            asm volatile("ds_add_u32 %0, %1" :: "v"(o3), "v"(v3) : "memory");
        } else if (KIND == 2) {
            const double d0 = v0, d1 = v1;
            asm volatile("ds_add_u64 %0, %1" :: "v"(o0 & ~7u), "v"(d0) : "memory");
            asm volatile("ds_add_u64 %0, %1" :: "v"(o2 & ~7u), "v"(d1) : "memory");
        } else {
            asm volatile("ds_write_b32 %0, %1" :: "v"(o0), "v"(v0) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(o1), "v"(v1) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(o2), "v"(v2) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(o3), "v"(v3) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float sum = acc;
    for (int i = threadIdx.x; i < rows * STRIDE; i += 1024) sum += tab[i];
    out[blockIdx.x * 1024 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int STRIDE, int ROT, int READS, int KIND = 0>
void run(const char* what) {
    const int iters = 4000, rows = 645;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<STRIDE, ROT, READS, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, rows * STRIDE * 4);
    k_probe<STRIDE, ROT, READS, KIND><<<256, 1024, rows * STRIDE * 4>>>(out, iters, rows, cyc);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_probe<STRIDE, ROT, READS, KIND><<<256, 1024, rows * STRIDE * 4>>>(out, iters, rows, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // wall time at ~2.4 GHz: clocks per trip of the CU's sixteen waves together
    const double clk = ms * 1e-3 * 2.4e9 / iters / 16;
    std::printf("%-50s %7.1f clocks per wave trip and CU (%d LDS instructions)\n", what, clk, (KIND == 2 ? 2 : 4) + READS);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<16, 0, 0>("stride 16, natural order");
    run<16, 1, 0>("stride 16, rotated by lane");
    run<20, 0, 0>("stride 20, natural order");
    run<20, 1, 0>("stride 20, rotated by lane");
    run<20, 1, 1>("stride 20, rotated, with the row read first");
    run<16, 0, 0, 1>("ds_add_u32: stride 16, natural order");
    run<20, 1, 0, 1>("ds_add_u32: stride 20, rotated by lane");
    run<20, 1, 1, 1>("ds_add_u32: stride 20, rotated, row read first");
    run<20, 0, 0, 2>("ds_add_u64 (two per lane): stride 20");
    run<16, 0, 0, 3>("ds_write_b32: stride 16, natural order");
    run<20, 1, 0, 3>("ds_write_b32: stride 20, rotated by lane");
    return 0;
}
