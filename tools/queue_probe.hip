// Are some HIP streams slower than others?  (Round 4: every 3rd-4th Agent's simulate(20) region took 124 instead of 80 us
// although its kernels overlapped on the device as usual — tools/slow_mode_probe.py.)  Creates high-priority streams in
// a row, like riab_streamer_create, and times on each: a tiny kernel + hipStreamSynchronize; a tiny kernel on the null
// stream + one on the stream + hipDeviceSynchronize; the same with the stream's kernel launched FIRST.
//   hipcc --offload-arch=gfx950 -O2 -o tools/queue_probe tools/queue_probe.hip && ./tools/queue_probe [n] [keep]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void tiny(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
__global__ void spin(unsigned* p, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicAdd(p, 1u);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 24;
  const bool keep = argc > 2 && !strcmp(argv[2], "keep");
  const int prio_mode = argc > 3 ? atoi(argv[3]) : 0;  // 0 highest, 1 default
  unsigned* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  printf("priority range: least %d greatest %d; GPU_MAX_HW_QUEUES=%s\n", least, greatest, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
  std::vector<hipStream_t> held;
  for (int i = 0; i < n; ++i) {
    hipStream_t s;
    if (prio_mode == 0) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    else hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int k = 0; k < 20; ++k) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, d); }
    hipDeviceSynchronize();
    std::vector<double> a, b, c, e;
    for (int k = 0; k < 200; ++k) {
      double t0 = now_us();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
      hipStreamSynchronize(s);
      a.push_back(now_us() - t0);
      t0 = now_us();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, d);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
      hipDeviceSynchronize();
      b.push_back(now_us() - t0);
      // the pipeline's shape: a ~20 us kernel on the stream first, then a ~50 us kernel on the null stream, device sync
      t0 = now_us();
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000ull);
      hipLaunchKernelGGL(spin, dim3(512), dim3(256), 0, 0, d, 5000ull);
      hipDeviceSynchronize();
      c.push_back(now_us() - t0);
      t0 = now_us();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, d);
      hipDeviceSynchronize();
      e.push_back(now_us() - t0);
    }
    printf("stream %2d (%p): stream kernel + stream sync %6.1f us | null + stream kernels + device sync %6.1f us | 20 us on stream, 50 us on null, "
           "device sync %6.1f us | null kernel + device sync %6.1f us\n", i, (void*)s, med(a), med(b), med(c), med(e));
    fflush(stdout);
    if (keep) held.push_back(s);
    else hipStreamDestroy(s);
  }
  return 0;
}
