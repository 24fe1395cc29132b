// Micro-benchmark: all-to-all exchange of one 8-byte {epoch, value} granule per workgroup among G workgroups,
//   mode 0: agent-scope relaxed atomic store / load (sc1 write-through granules -- the FPS kernel's protocol)
//   mode 1: plain store + s_waitcnt, poll = buffer_inv sc1 + plain load (only valid when all G workgroups share one XCD's L2)
// placement: workers are the blocks with blockIdx % 8 == 0 (same XCD, observed mapping) or blocks 0..G-1 (spread over XCDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(1024) void exch(unsigned long long* slots, int G, int rounds, int stride8, unsigned* xcc_out, long long* cyc_out, unsigned* sum_out, int NG, int WST)
{
    int g;
    if (stride8) { if (blockIdx.x % 8 != 0) return; g = blockIdx.x / 8; } else g = blockIdx.x;
    if (g >= G) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) xcc_out[g] = xcc_id();
    __shared__ unsigned s_val;
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int j = 1; j <= rounds; ++j) {
        const int par = j & 1;
        if (wave == 0) {
            unsigned long long* my = slots + (size_t)par * 4096 + (size_t)g * WST + lane;
            const unsigned long long v = ((unsigned long long)(unsigned)j << 32) | (unsigned)(g * 1000 + j + acc + lane);
            if (lane < NG) {
                if (MODE == 0) __hip_atomic_store(my, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(my), "v"(v) : "memory");
            }
            unsigned got = 0;
            const bool act = lane < G * NG;
            unsigned long long* gp = slots + (size_t)par * 4096 + (size_t)(lane / NG) * WST + (lane % NG);
            unsigned spins = 0;
            while (true) {
                bool ok = true;
                if (act) {
                    unsigned long long x;
                    if (MODE == 0) x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(gp) : "memory");
                    got = (unsigned)x;
                    ok = (unsigned)(x >> 32) == (unsigned)j;
                }
                if (__all(ok)) break;
                if (++spins > (1u << 22)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            unsigned m = 0;
            for (int w = 0; w < G; ++w) { unsigned q = (unsigned)__builtin_amdgcn_readlane((int)got, w * NG); m = q > m ? q : m; }
            if (lane == 0) s_val = m;
        }
        __syncthreads();
        acc = (acc + s_val) & 0xff;
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc_out[g] = t1 - t0; sum_out[g] = acc; }
}

int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 3;
    const int NG = argc > 2 ? atoi(argv[2]) : 5;
    const int WST = argc > 3 ? atoi(argv[3]) : 8;     // words between the records of two workgroups (8 = one 64-byte line each)
    const int rounds = 20000;
    unsigned long long* slots; unsigned *xcc, *sum; long long* cyc;
    hipMalloc(&slots, 2 * 4096 * 8); hipMalloc(&xcc, 64 * 4); hipMalloc(&sum, 64 * 4); hipMalloc(&cyc, 64 * 8);
    for (int stride8 = 0; stride8 < 2; ++stride8)
        for (int mode = 0; mode < 2; ++mode) {
            if (mode == 1 && !stride8) continue;    // the L2 protocol is only valid on one XCD
            hipMemset(slots, 0, 2 * 4096 * 8);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const int grid = stride8 ? 8 * G : G;
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(exch<0>, dim3(grid), dim3(1024), 0, 0, slots, G, rounds, stride8, xcc, cyc, sum, NG, WST);
            else hipLaunchKernelGGL(exch<1>, dim3(grid), dim3(1024), 0, 0, slots, G, rounds, stride8, xcc, cyc, sum, NG, WST);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<unsigned> hx(64), hs(64);
            hipMemcpy(hx.data(), xcc, 64 * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hs.data(), sum, 64 * 4, hipMemcpyDeviceToHost);
            printf("G=%d NG=%d WST=%d stride8=%d mode=%d: %.3f us/round  xcc:", G, NG, WST, stride8, mode, ms * 1e3 / rounds);
            for (int i = 0; i < G && i < 4; ++i) printf(" %u", hx[i]);
            printf("  sum0: %u", hs[0]);
            printf("\n");
        }
    return 0;
}
