// Micro-benchmark (not product code): what does a round of a PERSISTENT kernel cost on MI355X compared with a kernel
// boundary?  Every round each wave publishes 64 halfwords with agent-scope (sc1) stores, the grid synchronises, and each
// wave then reads what a wave of ANOTHER workgroup published (sc1 16-byte loads) and verifies it -- the access pattern of
// the region-voting rounds (k_voting.hip).  Variants: counter barrier (one atomic per workgroup on ONE address),
// flag barrier (one store per workgroup, all-to-all polling), and the same work as a chain of kernel launches.
// Every spin is bounded (a time-out sets an abort word and all workgroups leave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_sync.hip -o tools/ubench/grid_sync && tools/ubench/grid_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st16_agent(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 ld16B_agent(const void* p)
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ctrl[0] = arrive counter, ctrl[1] = abort, ctrl[2] = errors, ctrl[3] = timeouts; flags at ctrl + 64
template <int BAR>
__device__ __forceinline__ bool grid_barrier(int* ctrl, int round, int G)
{
    __builtin_amdgcn_s_waitcnt(0); // every store of this wave has left (vmcnt/lgkmcnt/expcnt = 0)
    __syncthreads();
    bool ok = true;
    if (BAR == 0) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctrl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int target = round * G;
            int spins = 0;
            while (ld_agent(ctrl) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20) || ld_agent(ctrl + 1)) { st_agent(ctrl + 1, 1); ok = false; break; }
            }
        }
    } else {
        if (threadIdx.x == 0) st_agent(ctrl + 64 + blockIdx.x, round);
        if (threadIdx.x < 64) { // the first wave polls all flags
            int spins = 0;
            for (;;) {
                int mn = round;
                for (int b = threadIdx.x; b < G; b += 64) mn = min(mn, ld_agent(ctrl + 64 + b));
                if (__ballot(mn < round) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 18) || ld_agent(ctrl + 1)) { st_agent(ctrl + 1, 1); ok = false; break; }
            }
        }
    }
    __syncthreads();
    return ok; // (only meaningful in the polling threads; the abort word is re-read by the caller)
}

template <int BAR, bool DATA>
__global__ __launch_bounds__(1024) void k_persist(int* ctrl, uint16_t* st, int rounds, int G, int hop)
{
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63, nwaves = G * wpb;
    int errors = 0;
    for (int r = 1; r <= rounds; r++) {
        if (DATA) st16_agent(st + (size_t)wave * 64 + lane, (uint16_t)(r * 7 + lane));
        grid_barrier<BAR>(ctrl, 2 * r - 1, G);
        if (ld_agent(ctrl + 1)) break;
        if (DATA) {
            const int other = (wave + hop) % nwaves;
            const uint4 v = ld16B_agent(st + (size_t)other * 64 + (lane & 7) * 8);
            const int l0 = (lane & 7) * 8;
            errors += (v.x & 0xffffu) != (uint16_t)(r * 7 + l0) || (v.w >> 16) != (uint16_t)(r * 7 + l0 + 7);
        }
        grid_barrier<BAR>(ctrl, 2 * r, G); // (readers done before the next round overwrites)
        if (ld_agent(ctrl + 1)) break;
    }
    if (errors) atomicAdd(ctrl + 2, errors);
}

// the same round as two kernels of a launch chain (plain stores / loads: the kernel boundary makes them visible)
__global__ __launch_bounds__(1024) void k_chain_w(uint16_t* st, int r)
{
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    st[(size_t)wave * 64 + lane] = (uint16_t)(r * 7 + lane);
}
__global__ __launch_bounds__(1024) void k_chain_r(int* ctrl, const uint16_t* st, int r, int hop)
{
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63, nwaves = gridDim.x * wpb;
    const int other = (wave + hop) % nwaves;
    const uint4 v = *reinterpret_cast<const uint4*>(st + (size_t)other * 64 + (lane & 7) * 8);
    const int l0 = (lane & 7) * 8;
    if ((v.x & 0xffffu) != (uint16_t)(r * 7 + l0) || (v.w >> 16) != (uint16_t)(r * 7 + l0 + 7)) atomicAdd(ctrl + 2, 1);
}

template <int BAR, bool DATA>
static void run(const char* name, int G, int T, int rounds, int* ctrl, uint16_t* st)
{
    CK(hipMemset(ctrl, 0, (64 + 4096) * sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int hop = (G / 2) * (T / 64) + 1; // a wave of a workgroup half a grid away
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_persist<BAR, DATA>), dim3(G), dim3(T), 0, 0, ctrl, st, rounds, G, hop);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int h[4];
    CK(hipMemcpy(h, ctrl, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-34s G %5d x %4d thr  rounds %4d  %8.2f us/round (2 barriers)  errors %d  abort %d\n", name, G, T, rounds,
           1000.0 * ms / rounds, h[2], h[1]);
    fflush(stdout);
}

int main()
{
    int* ctrl; uint16_t* st;
    CK(hipMalloc(&ctrl, (64 + 4096) * sizeof(int)));
    CK(hipMalloc(&st, (size_t)4096 * 16 * 64 * 2));
    CK(hipMemset(st, 0, (size_t)4096 * 16 * 64 * 2));
    const int R = 300;
    for (int rep = 0; rep < 2; rep++) {
        run<0, false>("counter barrier, no data", 256, 256, R, ctrl, st);
        run<0, false>("counter barrier, no data", 512, 256, R, ctrl, st);
        run<0, false>("counter barrier, no data", 1024, 256, R, ctrl, st);
        run<0, false>("counter barrier, no data", 256, 1024, R, ctrl, st);
        run<1, false>("flag barrier, no data", 256, 256, R, ctrl, st);
        run<1, false>("flag barrier, no data", 512, 256, R, ctrl, st);
        run<1, false>("flag barrier, no data", 1024, 256, R, ctrl, st);
        run<1, false>("flag barrier, no data", 256, 1024, R, ctrl, st);
        run<0, true>("counter barrier + sc1 publish/read", 256, 256, R, ctrl, st);
        run<0, true>("counter barrier + sc1 publish/read", 512, 256, R, ctrl, st);
        run<0, true>("counter barrier + sc1 publish/read", 1024, 256, R, ctrl, st);
        run<0, true>("counter barrier + sc1 publish/read", 256, 1024, R, ctrl, st);
        run<1, true>("flag barrier + sc1 publish/read", 256, 256, R, ctrl, st);
        run<1, true>("flag barrier + sc1 publish/read", 512, 256, R, ctrl, st);
        run<1, true>("flag barrier + sc1 publish/read", 256, 1024, R, ctrl, st);
        // launch chain
        for (int G : {512, 2048}) {
            CK(hipMemset(ctrl, 0, 64 * sizeof(int)));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int hop = (G / 2) * 4 + 1;
            CK(hipEventRecord(e0, 0));
            for (int r = 1; r <= R; r++) {
                hipLaunchKernelGGL(k_chain_w, dim3(G), dim3(256), 0, 0, st, r);
                hipLaunchKernelGGL(k_chain_r, dim3(G), dim3(256), 0, 0, ctrl, st, r, hop);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int h[4];
            CK(hipMemcpy(h, ctrl, sizeof(h), hipMemcpyDeviceToHost));
            printf("%-34s G %5d x  256 thr  rounds %4d  %8.2f us/round (2 launches)   errors %d\n", "launch chain, plain ld/st", G, R,
                   1000.0 * ms / R, h[2]);
        }
    }
    return 0;
}
