// One workgroup alone on the GPU: what a dependent VALU / DPP / LDS step and a barrier of W waves cost in wall time (100 MHz counter),
// alone and beside a kernel that keeps the other compute units busy.  hipcc --offload-arch=gfx950 -O3 -o wg_probe wg_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int MODE>
__global__ void k(int iters, unsigned long long *out, unsigned *sink)
{
    __shared__ unsigned s[1024];
    s[threadIdx.x & 1023] = (threadIdx.x * 7 + 1) & 1023;
    __syncthreads();
    unsigned x = threadIdx.x;
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) lds_barrier();
            else if (MODE == 1) x = s[x & 1023];
            else if (MODE == 2) x = min(x + 1, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xF, 0xF, false));
            else if (MODE == 3) x = (x ^ (x >> 3)) + 1;
            else if (MODE == 4) { s[threadIdx.x & 1023] = x; lds_barrier(); x = s[(threadIdx.x + 1) & 1023] + 1; lds_barrier(); }
        }
    }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = w1 - w0;
    sink[threadIdx.x] = x;
}
__global__ void heater(unsigned *sink, int iters)
{
    unsigned x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    if (x == 12345) sink[0] = x;
}
template <int MODE>
static void run(int threads, int iters, unsigned long long *out, unsigned *sink, hipStream_t st, const char *what)
{
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, st, iters, out, sink);
    hipStreamSynchronize(st);
    unsigned long long h;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("  threads %4d %-28s %7.2f ns per step\n", threads, what, 10.0 * h / (16.0 * iters));
}
int main()
{
    unsigned long long *out; unsigned *sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 8192);
    hipStream_t st, st2;
    hipStreamCreate(&st); hipStreamCreate(&st2);
    const int iters = 20000;
    for (int heat = 0; heat < 2; heat++) {
        printf(heat ? "beside a kernel on 1020 other workgroups:\n" : "alone:\n");
        if (heat) hipLaunchKernelGGL(heater, dim3(1020), dim3(256), 0, st2, sink + 1024, 40000000);
        for (int threads : {64, 640}) {
            run<3>(threads, iters, out, sink, st, "dependent VALU (3 ops)");
            run<2>(threads, iters, out, sink, st, "DPP row_shr + min + add");
            run<1>(threads, iters, out, sink, st, "dependent LDS read");
            run<0>(threads, iters, out, sink, st, "barrier");
            run<4>(threads, iters, out, sink, st, "write, barrier, read, barrier");
        }
        if (heat) hipStreamSynchronize(st2);
    }
    return 0;
}
