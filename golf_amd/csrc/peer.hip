// Push-based exchange of the synthesised audio between the GPUs of one node ("peer stores").
//
// The data-parallel path shards utterances over ranks and its only exchange is the all-gather of the audio
// (north_star / BASELINE configs[3]).  xGMI is point to point -- every GPU has its own link to each of the 7 others --
// so the natural form of that exchange on this fabric is not a ring: ONE kernel reads a step's (rows, T) block once and
// stores it into slot[rank] of every peer's receive buffer, seven links busy at once, no staging copy, no collective
// launch.  Buffers are allocated here (fine-grained, so that flags can be polled while kernels run), exported /
// opened through HIP IPC, and completion is a per-(slot, source) sequence number written with system scope after the
// stores; consumers wait for it in-stream with a bounded spin.
//
// Status: built and exercised with 2 processes on ONE GPU (tests/test_gpu_peer.py); NOT measured on a multi-GPU node
// (this pool has none) -- flag-gated (bench.py --gather-mode peer-store), RCCL's all-gather stays the default.
#include "common.h"
#include "device_common.h"
#include <cstring>

namespace golf {

constexpr int kMaxPeers = GOLF_MAX_PEERS;
struct PeerPtrs {
    void* p[kMaxPeers];
};

// grid (ceil(T / (256 * 4)), rows): a thread moves 4 elements, 256 apart (every access of a wave is one 256-byte run)
__global__ __launch_bounds__(256) void peer_store_kernel(const float* __restrict__ src, int64_t src_stride, int T,
                                                         PeerPtrs dst, int64_t dst_stride, int n_dst) {
    const int row = blockIdx.y;
    const int t0 = blockIdx.x * 1024 + threadIdx.x;
    const float* s = src + (size_t)row * src_stride;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = t0 + 256 * u < T ? s[t0 + 256 * u] : 0.f;
    for (int d = 0; d < n_dst; ++d) {
        float* o = (float*)dst.p[d] + (size_t)row * dst_stride;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + 256 * u < T) o[t0 + 256 * u] = v[u];
    }
    __threadfence_system();
}

// one thread per destination: flag = seq, visible system-wide after everything this stream stored before
__global__ void peer_signal_kernel(PeerPtrs flags, int n, unsigned seq) {
    const int i = threadIdx.x;
    if (i >= n) return;
    __threadfence_system();
    __hip_atomic_store((unsigned*)flags.p[i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane i spins until flags[i * stride] has reached seq (sequence numbers only grow; wrap-around is compared signed),
// at most `timeout_ticks` of the device's constant-rate wall clock (wall_clock64; rate from hipDeviceAttributeWallClockRate):
// a peer that never arrives must not hang the GPU
__global__ void peer_wait_kernel(const unsigned* __restrict__ flags, int n, int stride, unsigned seq,
                                 unsigned long long timeout_ticks, int* __restrict__ status) {
    const int i = threadIdx.x;
    if (i >= n) return;
    // latched: once a wait of this exchange has run out, later waits return at once instead of each spending the full
    // timeout again (a dead peer costs one timeout, not one per queued step) -- ADVICE r2
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const unsigned long long t_end = wall_clock64() + timeout_ticks;
    for (;;) {
        const unsigned v = __hip_atomic_load(flags + (size_t)i * stride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - seq) >= 0) break;
        if (wall_clock64() > t_end) {
            atomicExch(status, 1 + i);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace golf

using namespace golf;

extern "C" int golf_peer_alloc(size_t bytes, void** ptr) {
    if (!ptr || bytes == 0) return fail(GOLF_EINVAL, "peer_alloc: null pointer / zero size");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail((int)e, "peer_alloc: hipExtMallocWithFlags(%zu): %s", bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(p); return fail((int)e, "peer_alloc: hipMemset: %s", hipGetErrorString(e)); }
    *ptr = p;
    return GOLF_OK;
}

extern "C" int golf_peer_free(void* ptr) {
    if (!ptr) return GOLF_OK;
    hipError_t e = hipFree(ptr);
    return e == hipSuccess ? GOLF_OK : fail((int)e, "peer_free: %s", hipGetErrorString(e));
}

extern "C" int golf_peer_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == GOLF_PEER_HANDLE_BYTES, "handle size");
    if (!ptr || !handle64) return fail(GOLF_EINVAL, "peer_export: null pointer");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return fail((int)e, "peer_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    std::memcpy(handle64, &h, sizeof(h));
    return GOLF_OK;
}

extern "C" int golf_peer_open(const void* handle64, void** ptr) {
    if (!ptr || !handle64) return fail(GOLF_EINVAL, "peer_open: null pointer");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail((int)e, "peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *ptr = p;
    return GOLF_OK;
}

extern "C" int golf_peer_close(void* ptr) {
    if (!ptr) return GOLF_OK;
    hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? GOLF_OK : fail((int)e, "peer_close: %s", hipGetErrorString(e));
}

extern "C" int golf_peer_store_f32(const float* src, int64_t src_stride, int rows, int T, void* const* dst,
                                   int64_t dst_stride, int n_dst, void* stream) {
    if (!src || !dst) return fail(GOLF_EINVAL, "peer_store: null pointer");
    if (rows < 0 || T < 0 || n_dst < 0 || n_dst > kMaxPeers)
        return fail(GOLF_EINVAL, "peer_store: rows=%d T=%d n_dst=%d (at most %d destinations)", rows, T, n_dst, kMaxPeers);
    if (src_stride < T || dst_stride < T) return fail(GOLF_EINVAL, "peer_store: row stride < T");
    if (rows == 0 || T == 0 || n_dst == 0) return GOLF_OK;
    PeerPtrs d{};
    for (int i = 0; i < n_dst; ++i) {
        if (!dst[i]) return fail(GOLF_EINVAL, "peer_store: destination %d is null", i);
        d.p[i] = dst[i];
    }
    hipLaunchKernelGGL(peer_store_kernel, dim3((unsigned)ceil_div(T, 1024), rows), dim3(256), 0, (hipStream_t)stream, src,
                       src_stride, T, d, dst_stride, n_dst);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_peer_signal_u32(void* const* flags, int n, uint32_t seq, void* stream) {
    if (!flags || n < 0 || n > kMaxPeers) return fail(GOLF_EINVAL, "peer_signal: n=%d (at most %d)", n, kMaxPeers);
    if (n == 0) return GOLF_OK;
    PeerPtrs f{};
    for (int i = 0; i < n; ++i) {
        if (!flags[i] || ((uintptr_t)flags[i] & 3)) return fail(GOLF_EINVAL, "peer_signal: flag %d null / unaligned", i);
        f.p[i] = flags[i];
    }
    hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f, n, (unsigned)seq);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_peer_wait_u32(const uint32_t* flags, int n, int stride, uint32_t seq, int64_t timeout_us, int* status,
                                  void* stream) {
    if (!flags || !status || n < 0 || n > 64 || stride < 1)
        return fail(GOLF_EINVAL, "peer_wait: n=%d stride=%d (at most 64 flags)", n, stride);
    if (n == 0) return GOLF_OK;
    // ticks of wall_clock64(): the rate is a device attribute (kHz; 100 MHz on gfx950 today -- asked for, not assumed)
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        khz = 100000;
    const unsigned long long ticks = (unsigned long long)(timeout_us < 0 ? 0 : timeout_us) * (unsigned long long)khz / 1000ull;
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)flags, n, stride,
                       (unsigned)seq, ticks, status);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
