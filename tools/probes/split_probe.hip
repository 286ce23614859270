// dev probe: is the three-term bf16 split through v_cvt_pk_bf16_f32 (round to nearest even) and v_dot2c_f32_bf16
// (residual = v - float(term), one instruction) exact?  hi + mid + lo == v bit for bit for every tested float, and every
// residual equals the plainly computed one.   hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe && ./split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float up(uint32_t h) { return __uint_as_float(h << 16); }

__global__ void k(const float* in, int n, unsigned long long* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    uint32_t c0 = 0x0000BF80u, c1 = 0xBF800000u;              // (-1, 0) and (0, -1) as packed bf16; opaque: hipcc 7.2 folds the
    asm volatile("" : "+s"(c0), "+s"(c1));                     // first into the inline constant -1.0, which the instruction reads as (0, -1)
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, c0), m1 = __builtin_bit_cast(bf16x2, c1);
    const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
    const float ra = __builtin_amdgcn_fdot2_f32_bf16(h, m0, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(h, m1, b, false);
    const bf16x2 m = __builtin_convertvector((f32x2){ra, rb}, bf16x2);
    const float sa = __builtin_amdgcn_fdot2_f32_bf16(m, m0, ra, false), sb = __builtin_amdgcn_fdot2_f32_bf16(m, m1, rb, false);
    const bf16x2 l = __builtin_convertvector((f32x2){sa, sb}, bf16x2);
    const uint32_t hb = __builtin_bit_cast(uint32_t, h), mb = __builtin_bit_cast(uint32_t, m), lb = __builtin_bit_cast(uint32_t, l);
    // reference residuals
    const float ha = up(hb & 0xffffu), hbf = __uint_as_float(hb & 0xffff0000u);
    const float ma = up(mb & 0xffffu), mbf = __uint_as_float(mb & 0xffff0000u);
    const float la = up(lb & 0xffffu), lbf = __uint_as_float(lb & 0xffff0000u);
    unsigned long long e = 0;
    if (ra != a - ha || rb != b - hbf) e |= 1;
    if (sa != ra - ma || sb != rb - mbf) e |= 2;
    if ((double)ha + (double)ma + (double)la != (double)a) e |= 4;
    if ((double)hbf + (double)mbf + (double)lbf != (double)b) e |= 8;
    if (e) {
        atomicOr(bad, e), atomicAdd(bad + 1, 1ull);
        const float big = fmaxf(fabsf(a), fabsf(b)), small = fminf(fabsf(a), fabsf(b));
        if (big < 1e37f && small > 1e-25f) atomicAdd(bad + 2, 1ull);          // neither near overflow nor with denormal residuals
    }
}

int main() {
    const int n = 1 << 24;
    std::vector<float> v(n);
    std::mt19937_64 g(7);
    for (int i = 0; i < n; ++i) {
        uint32_t bits = (uint32_t)g();
        uint32_t ex = (bits >> 23) & 0xff;
        if (i % 3 == 0) ex = 100 + ex % 60;                 // ordinary magnitudes
        if (ex == 0xff) ex = 0xfe;                          // no inf / nan
        if (i % 1000 != 0 && ex < 40) ex = 40;              // (denormal residuals: a few only)
        bits = (bits & 0x807fffffu) | (ex << 23);
        memcpy(&v[i], &bits, 4);
    }
    float* d; unsigned long long* bad;
    hipMalloc(&d, n * 4); hipMalloc(&bad, 24); hipMemset(bad, 0, 24);
    hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(d, n, bad);
    unsigned long long h[3];
    hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
    printf("tested %d floats: failure mask %llx, failing pairs %llu, of them with both values in [1e-25, 1e37]: %llu\n", n, h[0], h[1], h[2]);
    return 0;
}
