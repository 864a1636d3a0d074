// Micro-benchmark (not product code): does the ROW-march access shape of K4 / K5 / K6 (one wave per image row, 512 bytes per
// step, all waves at about the same x) lose bandwidth because the row pitch of the [H][W][128] f32 volume -- 1920 x 512 B =
// 983040 B = 3840 x 256 B -- maps the same x of EVERY row to the same memory channel?  Row- and column-march copies (8 loads in
// flight per wave, XCD-aware mapping, non-temporal) with the row pitch padded by 0 .. 16 pixels of 512 bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/row_pitch.hip -o tools/ubench/row_pitch && tools/ubench/row_pitch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool VERT, int DEPTH>
__global__ __launch_bounds__(64) void k_march(const f2* __restrict__ src, f2* __restrict__ dst, int W, int H, int Wp, int per_xcd, int nseg)
{
    const int b = (int)blockIdx.x, gw = (b & 7) * per_xcd + (b >> 3);
    const int nlines = VERT ? W : H;
    if ((b >> 3) >= per_xcd || gw >= nlines * nseg) return;
    const int seg = gw / nlines, line = gw - seg * nlines;
    const int N = VERT ? H : W, seg_len = (N + nseg - 1) / nseg, j0 = seg * seg_len, j1 = j0 + seg_len < N ? j0 + seg_len : N;
    const size_t estep = (VERT ? (size_t)Wp : 1) * 64; // f2 elements per step
    const size_t pix0 = VERT ? (size_t)line : (size_t)line * Wp;
    const f2* sp = src + pix0 * 64 + threadIdx.x;
    f2* dp = dst + pix0 * 64 + threadIdx.x;
    for (int j = j0; j < j1; j += DEPTH) {
        f2 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) v[u] = __builtin_nontemporal_load(&sp[(size_t)(j + u < j1 ? j + u : j1 - 1) * estep]);
#pragma unroll
        for (int u = 0; u < DEPTH; u++) if (j + u < j1) __builtin_nontemporal_store(v[u], &dp[(size_t)(j + u) * estep]);
    }
}

int main()
{
    const int W = 1920, H = 1080;
    const size_t bytes = (size_t)(W + 16) * H * 512;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("pad px | row march, 1 / 2 / 4 segments per row (TB/s) | column march (TB/s)\n");
    for (int pad : {0, 1, 2, 3, 4, 5, 8, 16}) {
        double r[4];
        int k = 0;
        for (int cfg = 0; cfg < 4; cfg++) {
            const bool vert = cfg == 3;
            const int nseg = cfg == 1 ? 2 : (cfg == 2 ? 4 : 1);
            const int nl = (vert ? W : H) * nseg, per_xcd = (nl + 7) / 8;
            float best = 1e30f;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(e0, 0));
                if (vert) hipLaunchKernelGGL((k_march<true, 8>), dim3(per_xcd * 8), dim3(64), 0, 0, (const f2*)a, (f2*)b, W, H, W + pad, per_xcd, nseg);
                else hipLaunchKernelGGL((k_march<false, 8>), dim3(per_xcd * 8), dim3(64), 0, 0, (const f2*)a, (f2*)b, W, H, W + pad, per_xcd, nseg);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            r[k++] = 2.0 * W * H * 512 / (best * 1e-3) / 1e12;
        }
        printf("%6d | %5.2f  %5.2f  %5.2f | %5.2f\n", pad, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
