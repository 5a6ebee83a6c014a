// Premise test for layer-pair fusion through L2 (DESIGN.md "next" item 1): inside every XCD, 16
// "producer" workgroups stream 16 KB chunks from a big input buffer (HBM), write 16 KB chunks into a
// small per-pair ring, and 16 "consumer" workgroups on the other CUs of the same XCD read the ring
// chunks and write 16 KB to a big output buffer (HBM).  If the ring stays L2-resident, the kernel's
// FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, separate passes) equal the two big streams only; if it
// does not, they double.  Same-XCD visibility protocol: data stores -> s_waitcnt vmcnt(0) (data is in
// L2) -> flag store; consumer polls the flag and loads the data with sc1 (miss the per-CU L1).  No
// buffer_wbl2: nothing has to leave the XCD.  Every spin is bounded; a timeout sets err and exits.
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_ring_bench.hip -o /tmp/l2_ring_bench
// run:   /tmp/l2_ring_bench <slots per pair: 4|8|16|64|256> [chunks per pair = 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int CHUNK = 16 * 1024;          // bytes: one 4x32-pixel x 64-channel fp16 tile
constexpr int PAIRS = 16;                 // producer/consumer pairs per XCD
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ uint4 load_sc1(const uint4* p)
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int flag_load(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(int* p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// flags: [xcd][pair] {head, tail} on separate 128-byte lines
__global__ __launch_bounds__(256, 1) void ring(const uint4* in, uint4* out, uint4* rings, int* flags, int slots, int nchunks,
                                               int mode, int* err, unsigned* checksum)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const bool producer = slot < PAIRS;
    const int pair = slot % PAIRS;
    int* head = flags + ((xcd * PAIRS + pair) * 2 + 0) * 32;
    int* tail = flags + ((xcd * PAIRS + pair) * 2 + 1) * 32;
    uint4* myring = rings + (size_t)(xcd * PAIRS + pair) * slots * (CHUNK / 16);
    const size_t stream_base = (size_t)(xcd * PAIRS + pair) * nchunks * (CHUNK / 16);
    unsigned acc = 0;
    __shared__ int abort_s;
    for (int i = 0; i < nchunks; ++i) {
        uint4* rs = myring + (size_t)(i % slots) * (CHUNK / 16);
        if (producer) {
            // wait for a free ring slot
            if (threadIdx.x == 0) {
                int spins = 0;
                while (i - flag_load(tail) >= slots)
                    if (++spins > SPIN_LIMIT) { flag_store(err, 1); break; }
                abort_s = flag_load(err);      // one thread decides for the workgroup: the exit is uniform
            }
            __syncthreads();
            if (abort_s) return;
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = in[stream_base + (size_t)i * (CHUNK / 16) + k * 256 + threadIdx.x];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k].x += i; rs[k * 256 + threadIdx.x] = v[k]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's stores are in L2
            __syncthreads();
            if (threadIdx.x == 0) flag_store(head, i + 1);
        } else {
            if (threadIdx.x == 0) {
                int spins = 0;
                while (flag_load(head) <= i)
                    if (++spins > SPIN_LIMIT) { flag_store(err, 2); break; }
                abort_s = flag_load(err);
            }
            __syncthreads();
            if (abort_s) return;
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = load_sc1(rs + k * 256 + threadIdx.x);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc += v[k].x;
                out[stream_base + (size_t)i * (CHUNK / 16) + k * 256 + threadIdx.x] = v[k];
            }
            __syncthreads();                                       // all reads of the slot done
            if (threadIdx.x == 0) flag_store(tail, i + 1);
        }
    }
    if (!producer) atomicAdd(checksum, acc);
}

int main(int argc, char** argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 8;
    const int nchunks = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t stream_bytes = (size_t)8 * PAIRS * nchunks * CHUNK;
    const size_t ring_bytes = (size_t)8 * PAIRS * slots * CHUNK;
    uint4 *in, *out, *rings;
    int *flags, *err;
    unsigned* checksum;
    hipMalloc(&in, stream_bytes); hipMalloc(&out, stream_bytes); hipMalloc(&rings, ring_bytes);
    hipMalloc(&flags, 8 * PAIRS * 2 * 128); hipMalloc(&err, 4); hipMalloc(&checksum, 4);
    hipMemset(in, 1, stream_bytes); hipMemset(rings, 0, ring_bytes);
    printf("ring per XCD %.2f MB (%d slots x %d pairs x 16 KB), streams %.0f MB in + %.0f MB out\n",
           ring_bytes / 8 / 1048576.0, slots, PAIRS, stream_bytes / 1048576.0, stream_bytes / 1048576.0);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flags, 0, 8 * PAIRS * 2 * 128); hipMemset(err, 0, 4); hipMemset(checksum, 0, 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ring, dim3(256), dim3(256), 0, 0, in, out, rings, flags, slots, nchunks, 0, err, checksum);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int herr; unsigned hsum;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hsum, checksum, 4, hipMemcpyDeviceToHost);
        // expected checksum: every uint4.x = 0x01010101 + i, summed over all elements (mod 2^32)
        unsigned long long want = 0;
        for (int i = 0; i < nchunks; ++i) want += (unsigned long long)(0x01010101u + (unsigned)i) * (CHUNK / 16);
        want *= 8ull * PAIRS;
        printf("  %.3f ms  (%.0f GB/s per stream)  err=%d  checksum %s\n", ms, stream_bytes / (ms * 1e-3) / 1e9, herr,
               hsum == (unsigned)want ? "ok" : "MISMATCH");
    }
    return 0;
}
