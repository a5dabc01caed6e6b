// Host cost of putting three small kernels on a stream, by launch API (gfx950, ROCm 7.2): what a riab_simulate call of
// the one-kernel form pays before its rate kernel can start.  hipcc -O2 --offload-arch=gfx950 tools/launch_bench.hip -o tools/exp/launch_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args { int* p; long a[12]; };   // ~100 bytes of kernel arguments, like the real kernels
__global__ void k_small(Args a) { if (a.p && threadIdx.x == 1000000) a.p[0] = (int)a.a[0]; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  int* d;
  hipMalloc(&d, 4096);
  Args a{d, {1, 2, 3}};
  hipFunction_t f;
  if (hipGetFuncBySymbol(&f, (const void*)k_small) != hipSuccess) printf("hipGetFuncBySymbol failed\n");
  // a three-node graph (two streams' worth: node 0 alone, nodes 1 -> 2)
  hipGraph_t g;
  hipGraphCreate(&g, 0);
  hipGraphNode_t n[3];
  void* kargs[] = {&a};
  hipKernelNodeParams kp{};
  kp.func = (void*)k_small; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.kernelParams = kargs;
  hipGraphAddKernelNode(&n[0], g, nullptr, 0, &kp);
  hipGraphAddKernelNode(&n[1], g, nullptr, 0, &kp);
  hipGraphAddKernelNode(&n[2], g, &n[1], 1, &kp);
  hipGraphExec_t ge;
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  auto run = [&](const char* name, auto body) {
    std::vector<double> t;
    for (int it = 0; it < 300; ++it) {
      hipDeviceSynchronize();
      const double t0 = now();
      body();
      const double t1 = now();
      hipDeviceSynchronize();
      if (it >= 50) t.push_back(t1 - t0);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s median %6.2f us  min %6.2f  p90 %6.2f\n", name, t[t.size() / 2], t[0], t[t.size() * 9 / 10]);
  };
  run("3 x hipLaunchKernelGGL (one stream)", [&] { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, a); });
  run("3 x hipLaunchKernelGGL (side, main, main)", [&] { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, a); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, a); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, a); });
  run("3 x hipLaunchKernel (args array)", [&] { for (int i = 0; i < 3; ++i) hipLaunchKernel((const void*)k_small, dim3(64), dim3(256), kargs, 0, s); });
  run("3 x hipModuleLaunchKernel", [&] { for (int i = 0; i < 3; ++i) hipModuleLaunchKernel(f, 64, 1, 1, 256, 1, 1, 0, s, kargs, nullptr); });
  {
    size_t sz = sizeof(a);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    run("3 x hipModuleLaunchKernel (argument buffer)", [&] { for (int i = 0; i < 3; ++i) hipModuleLaunchKernel(f, 64, 1, 1, 256, 1, 1, 0, s, nullptr, cfg); });
  }
  run("3 x hipExtLaunchKernelGGL (no events)", [&] { for (int i = 0; i < 3; ++i) hipExtLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0u, a); });
  run("hipGraphLaunch (3 kernel nodes)", [&] { hipGraphLaunch(ge, s); });
  run("3 x hipGraphExecKernelNodeSetParams + hipGraphLaunch", [&] { for (int i = 0; i < 3; ++i) hipGraphExecKernelNodeSetParams(ge, n[i], &kp); hipGraphLaunch(ge, s); });
  run("1 x hipLaunchKernelGGL", [&] { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, a); });
  run("hipStreamQuery (idle stream)", [&] { (void)hipStreamQuery(s); });
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  run("hipEventRecord + hipStreamWaitEvent", [&] { hipEventRecord(e, s2); hipStreamWaitEvent(s, e, 0); });
  return 0;
}
