// Can a kernel on stream A spin on a flag that a kernel on stream B (same process, same device) sets?  Probe for the
// in-process (virtual rank) form of the direct transport.  hipcc --offload-arch=gfx950 spin_probe.hip -o spin_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_wait(unsigned long long *flag, unsigned long long want, long long timeout, int *result)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (wall_clock64() - t0 > timeout) { *result = -1; return; }
        __builtin_amdgcn_s_sleep(4);
    }
    *result = 1;
}
__global__ void k_set(unsigned long long *flag, unsigned long long v)
{
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int run(unsigned sflags, bool finegrained, const char *name)
{
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, sflags));
    CK(hipStreamCreateWithFlags(&b, sflags));
    unsigned long long *flag;
    int *res, h = 0;
    if (finegrained) CK(hipExtMallocWithFlags((void **)&flag, 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void **)&flag, 4096));
    CK(hipMalloc((void **)&res, 4));
    CK(hipMemset(flag, 0, 4096));
    CK(hipMemset(res, 0, 4));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, a, flag, 1ull, (long long)khz * 3000, res);
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, b, flag, 1ull);
    CK(hipStreamSynchronize(b));
    CK(hipStreamSynchronize(a));
    CK(hipMemcpy(&h, res, 4, hipMemcpyDeviceToHost));
    printf("%-40s wall clock %d kHz -> %s\n", name, khz, h == 1 ? "flag seen" : "TIMEOUT");
    return 0;
}

int main()
{
    run(hipStreamDefault, true, "blocking streams, fine-grained");
    run(hipStreamNonBlocking, true, "non-blocking streams, fine-grained");
    run(hipStreamDefault, false, "blocking streams, coarse-grained");
    run(hipStreamNonBlocking, false, "non-blocking streams, coarse-grained");
    // many streams: do two of them ever share a hardware queue?
    for (int n : {4, 6, 9}) {
        hipStream_t s[16];
        for (int i = 0; i < n; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        unsigned long long *flag;
        int *res;
        hipExtMallocWithFlags((void **)&flag, 4096, hipDeviceMallocFinegrained);
        hipMalloc((void **)&res, 64);
        hipMemset(flag, 0, 4096);
        hipMemset(res, 0, 64);
        int khz = 0;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
        for (int i = 0; i < n - 1; ++i) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s[i], flag, 1ull, (long long)khz * 3000, res + i);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, s[n - 1], flag, 1ull);
        hipDeviceSynchronize();
        int h[16] = {};
        hipMemcpy(h, res, 64, hipMemcpyDeviceToHost);
        int ok = 0;
        for (int i = 0; i < n - 1; ++i) ok += h[i] == 1;
        printf("%d waiting streams + 1 setter: %d saw the flag\n", n - 1, ok);
    }
    return 0;
}
