// Adam step of the training loops (torch.optim.Adam(model.parameters(), lr) + optimizer.step(), GripNet-pose.py:104,146 and
// the other drivers) over ALL parameters in one launch.  The update of torch/optim/adam.py (_single_tensor_adam, no amsgrad):
//
//   t += 1;  g += wd p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//   p -= lr / (1 - b1^t) * m / ( sqrt(v) / sqrt(1 - b2^t) + eps )
//
// The PoSE model has 14 parameter tensors of 16 to 610 K elements: torch's multi-tensor kernel takes 44 us for them inside the
// replayed training step (plus a launch for the step counters), a per-tensor loop fourteen launches.  Here the tensor table
// travels as a kernel ARGUMENT (so a captured step replays with it), a workgroup takes a 4096-element slice of one tensor,
// the step counter lives on the device and is advanced by the last workgroup to arrive (every workgroup has read it by then).
#include "common.h"

namespace {

constexpr int kAdamThreads = 256, kAdamSlice = 4096;

struct AdamTable {
    gn_adam_tensor t[GN_ADAM_MAX_TENSORS];
    int32_t first_block[GN_ADAM_MAX_TENSORS + 1];
    int n;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kAdamThreads) void k_adam(AdamTable tab, float* __restrict__ step, unsigned int* __restrict__ arrived,
                                                      float lr, float b1, float b2, float eps, float wd, int advance) {
    const int b = blockIdx.x;
    int k = 0;
    while (k + 1 < tab.n && b >= tab.first_block[k + 1]) ++k;             // (uniform: scalar registers)
    const gn_adam_tensor T = tab.t[k];
    const float t = *step + 1.0f;
    const float c1 = 1.0f - __powf(b1, t), c2 = 1.0f - __powf(b2, t);
    const float step_size = lr / c1, rs2 = 1.0f / sqrtf(c2);
    const int64_t lo = (int64_t)(b - tab.first_block[k]) * kAdamSlice, hi = min(lo + (int64_t)kAdamSlice, T.numel);
    const bool vec = ((reinterpret_cast<uintptr_t>(T.param) | reinterpret_cast<uintptr_t>(T.grad) | reinterpret_cast<uintptr_t>(T.exp_avg) |
                       reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15) == 0;
    auto update = [&](float& p, float g, float& m, float& v) {
        g += wd * p;
        m += (g - m) * (1.0f - b1);
        v = b2 * v + (1.0f - b2) * g * g;
        p -= step_size * (m / (sqrtf(v) * rs2 + eps));
    };
    if (vec) {
        // (slices start at multiples of 4096 elements: 16-byte aligned whenever the tensors are)
        for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i + 3 < hi; i += 4 * kAdamThreads) {
            f32x4 p = *reinterpret_cast<f32x4*>(T.param + i), m = *reinterpret_cast<f32x4*>(T.exp_avg + i), v = *reinterpret_cast<f32x4*>(T.exp_avg_sq + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(T.grad + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                update(pe, g[e], me, ve);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            *reinterpret_cast<f32x4*>(T.param + i) = p;
            *reinterpret_cast<f32x4*>(T.exp_avg + i) = m;
            *reinterpret_cast<f32x4*>(T.exp_avg_sq + i) = v;
        }
        const int64_t tail = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = tail + threadIdx.x; i < hi; i += kAdamThreads) update(T.param[i], T.grad[i], T.exp_avg[i], T.exp_avg_sq[i]);
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += kAdamThreads) update(T.param[i], T.grad[i], T.exp_avg[i], T.exp_avg_sq[i]);
    }
    // the step counter moves when every workgroup has read it: the last one to arrive writes it
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int seen = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == gridDim.x - 1) {
            if (advance) *step = t;
            __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream-ordered)
        }
    }
}

}  // namespace

extern "C" gn_status gn_adam_step_f32(const gn_adam_tensor* tensors, int num_tensors, float* step, void* workspace, size_t workspace_bytes,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    GN_REQUIRE(num_tensors >= 0 && (num_tensors == 0 || tensors), "tensor table is null");
    GN_REQUIRE(step && workspace && workspace_bytes >= 4 && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0,
               "step counter / workspace is null (workspace: 4 bytes, zeroed once by the caller)");
    GN_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "bad hyper-parameters");
    std::vector<gn_adam_tensor> live;
    for (int k = 0; k < num_tensors; ++k) {
        const gn_adam_tensor& T = tensors[k];
        GN_REQUIRE(T.numel >= 0 && (T.numel == 0 || (T.param && T.grad && T.exp_avg && T.exp_avg_sq)), "tensor %d has a null pointer", k);
        GN_REQUIRE(gn::ceil_div(T.numel, kAdamSlice) < (1 << 24), "tensor %d is too large for one launch", k);
        if (T.numel > 0) live.push_back(T);
    }
    // 64 tensors per launch (the table is a kernel argument); several launches of one step: only the last one moves the counter
    for (size_t done = 0; done < live.size();) {
        AdamTable tab;
        tab.n = 0;
        int blocks = 0;
        for (; done < live.size() && tab.n < GN_ADAM_MAX_TENSORS; ++done, ++tab.n) {
            tab.t[tab.n] = live[done];
            tab.first_block[tab.n] = blocks;
            blocks += (int)gn::ceil_div(live[done].numel, kAdamSlice);
        }
        tab.first_block[tab.n] = blocks;
        k_adam<<<blocks, kAdamThreads, 0, gn::as_stream(stream)>>>(tab, step, static_cast<unsigned int*>(workspace), lr, beta1, beta2,
                                                                    eps, weight_decay, done == live.size() ? 1 : 0);
        GN_LAUNCH_CHECK();
    }
    return GN_OK;
}
