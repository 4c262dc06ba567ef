// Does a stream of {hipMemcpyAsync H2D, a few small kernels, hipMemcpyAsync D2H, hipStreamSynchronize} stall by itself?
// (the host-pointer batched call shows one 37-43 ms call at a fixed call index, the device-resident form none)
//   hipcc --offload-arch=gfx950 -O2 tools/memcpy_stall_probe.hip -o /tmp/memcpy_stall_probe && /tmp/memcpy_stall_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void touch(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

static void run(const char* name, bool h2d, bool d2h, bool pinned_src, int iters) {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t qbytes = 256 * 384 * 4, hbytes = 256 * 10 * 16;
    float *dq, *dh, *hq, *hh;
    hipMalloc(&dq, qbytes); hipMalloc(&dh, hbytes);
    if (pinned_src) hipHostMalloc(&hq, qbytes, hipHostMallocDefault); else hq = (float*)malloc(qbytes);
    hipHostMalloc(&hh, hbytes, hipHostMallocDefault);
    for (size_t i = 0; i < qbytes / 4; ++i) hq[i] = 1.0f;
    std::vector<double> us(iters);
    for (int it = 0; it < iters; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        if (h2d) hipMemcpyAsync(dq, hq, qbytes, hipMemcpyHostToDevice, st);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(touch, dim3(96), dim3(256), 0, st, dq, 256 * 96);
        if (d2h) hipMemcpyAsync(hh, dq, hbytes, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        us[it] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::vector<int> idx(iters);
    for (int i = 0; i < iters; ++i) idx[i] = i;
    std::partial_sort(idx.begin(), idx.begin() + 4, idx.end(), [&](int a, int b) { return us[a] > us[b]; });
    std::vector<double> s(us); std::sort(s.begin(), s.end());
    printf("%-34s median %7.1f us  slowest: %d: %.0f, %d: %.0f, %d: %.0f, %d: %.0f\n", name, s[iters / 2], idx[0], us[idx[0]], idx[1], us[idx[1]],
           idx[2], us[idx[2]], idx[3], us[idx[3]]);
    hipFree(dq); hipFree(dh); hipHostFree(hh); if (pinned_src) hipHostFree(hq); else free(hq);
    hipStreamDestroy(st);
}

int main() {
    run("kernels only", false, false, true, 600);
    run("H2D pageable + kernels", true, false, false, 600);
    run("H2D pinned + kernels", true, false, true, 600);
    run("kernels + D2H pinned", false, true, true, 600);
    run("H2D pageable + kernels + D2H", true, true, false, 600);
    run("H2D pinned + kernels + D2H", true, true, true, 600);
    return 0;
}
