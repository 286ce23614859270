// One-shot direct exchange of the relational layer's partial sums between the GPUs of one node.
//
// The sharded forward's one exchange step is an all-reduce of [n_d, 32] floats - 82,560 bytes at n_d = 645: latency-bound.
// A ring all-reduce serialises 2 (G - 1) hops over single xGMI links; xGMI is point-to-point, every GPU has a link to
// every peer, so the exchange can be ONE hop: every rank writes its partial straight into a slot of every peer's buffer
// (peer-mapped memory, hipIpc), raises a flag there, waits for its own G flags and adds the G slots in RANK ORDER - every
// rank adds the same numbers in the same order, so all ranks hold the same bits (an all-reduce's contract that ring
// algorithms do not give).  No collective library call, no host synchronisation: three small launches per exchange on
// the caller's stream.
//
//   push      every workgroup copies its slice of `src` into slot [parity][rank] of EVERY peer (system-scope stores over
//             the fabric), drains them, draws a ticket; the last workgroup to arrive stores `step` into flag [rank] of every
//             peer (system scope: a flag is only seen behind the data it announces)
//   wait      one wave: lane r spins (bounded by a wall-clock timeout, then an error flag - the grid always drains) until
//             flag [r] of this rank's own buffer has reached `step`
//   sum       dst[i] = slots[parity][0][i] + slots[parity][1][i] + ... in rank order, read with system-scope loads (the
//             lines were written from outside this GPU's caches)
// Slots alternate by step parity: a rank can be at most one step ahead of its slowest peer (it needs that peer's data of
// step s to finish step s), so the slots of step s + 2 are free when it writes them.  Flags hold step numbers and only grow.
#include "common.h"

namespace {

constexpr int kMaxRanks = 16;

struct PushArgs {
    const float* src;
    int64_t n;
    float* peer_slots[kMaxRanks];      // base of every rank's slot buffer [2][world][n] (this rank's own among them)
    int* peer_flags[kMaxRanks];        // base of every rank's flag array [world]
    unsigned int* ticket;              // this rank's own counter (zero between launches)
    int world, rank, step;
};

__global__ __launch_bounds__(256) void k_exchange_push(PushArgs a) {
    __shared__ bool last;
    const int64_t slot = ((int64_t)(a.step & 1) * a.world + a.rank) * a.n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = a.src[i];
        for (int p = 0; p < a.world; ++p)
            __hip_atomic_store(a.peer_slots[p] + slot + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();                                            // this thread's stores are visible system-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int arrived = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = arrived == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if ((int)threadIdx.x < a.world)                                    // the data of every workgroup is out: announce it
        __hip_atomic_store(a.peer_flags[threadIdx.x] + a.rank, a.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream-ordered)
}

__global__ __launch_bounds__(64) void k_exchange_wait(const int* __restrict__ flags, int world, int step, unsigned long long timeout_ticks,
                                                      int* __restrict__ err) {
    const int r = threadIdx.x;
    if (r >= world) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock
    while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {   // a peer never arrived: say so and let the grid drain
            if (err) atomicOr(err, 4);
            return;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ __launch_bounds__(256) void k_exchange_sum(const float* __restrict__ slots, int world, int64_t n, int step, float* __restrict__ dst) {
    const float* __restrict__ base = slots + (int64_t)(step & 1) * world * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = __hip_atomic_load(base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < world; ++r) s += __hip_atomic_load(base + (int64_t)r * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dst[i] = s;
    }
}

}  // namespace

extern "C" {

size_t gn_exchange_buffer_bytes(int64_t n, int world) {
    if (n <= 0 || world <= 0) return 0;
    return (size_t)2 * world * n * sizeof(float);
}

// peer_slots / peer_flags: `world` DEVICE pointers each (host arrays): every rank's slot buffer (gn_exchange_buffer_bytes, any
// contents) and flag array ([world] int32, zero before the first step), mapped into this process (hipIpc; this rank's own
// are plain pointers).  ticket: one zeroed uint32 of this rank.  step = 1, 2, 3, ... (the same on every rank).
gn_status gn_exchange_push_f32(const float* src, int64_t n, void* const* peer_slots, void* const* peer_flags, unsigned int* ticket,
                               int world, int rank, int step, void* stream) {
    GN_REQUIRE(world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world && step >= 1 && n >= 0, "bad exchange shape (world %d, rank %d, step %d)", world, rank, step);
    if (n == 0) return GN_OK;
    GN_REQUIRE(src && peer_slots && peer_flags && ticket, "null pointer");
    PushArgs a;
    a.src = src; a.n = n; a.ticket = ticket; a.world = world; a.rank = rank; a.step = step;
    for (int p = 0; p < world; ++p) {
        GN_REQUIRE(peer_slots[p] && peer_flags[p], "rank %d's buffers are not mapped", p);
        a.peer_slots[p] = static_cast<float*>(peer_slots[p]);
        a.peer_flags[p] = static_cast<int*>(peer_flags[p]);
    }
    const int grid = (int)std::min<int64_t>(64, gn::ceil_div(n, 1024));
    k_exchange_push<<<grid, 256, 0, gn::as_stream(stream)>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// dst[i] = sum over ranks, in rank order, of step `step`'s partials (dst may be the pushed vector itself).  Waits at most
// timeout_ms for the peers (then bit 2 of *error_flag is set and dst is NOT the sum); never blocks the host.
gn_status gn_exchange_wait_sum_f32(const void* my_slots, const void* my_flags, int64_t n, int world, int step, float* dst,
                                   int timeout_ms, int32_t* error_flag, void* stream) {
    GN_REQUIRE(world >= 1 && world <= kMaxRanks && step >= 1 && n >= 0 && timeout_ms > 0, "bad exchange shape");
    if (n == 0) return GN_OK;
    GN_REQUIRE(my_slots && my_flags && dst, "null pointer");
    hipStream_t st = gn::as_stream(stream);
    k_exchange_wait<<<1, 64, 0, st>>>(static_cast<const int*>(my_flags), world, step, (unsigned long long)timeout_ms * 100000ull, error_flag);
    GN_LAUNCH_CHECK();
    k_exchange_sum<<<gn::stream_grid(n, 256, 256), 256, 0, st>>>(static_cast<const float*>(my_slots), world, n, step, dst);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // extern "C"
