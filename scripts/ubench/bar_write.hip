// bar_write.hip -- can a small call's input reach the device WITHOUT a copy command?
//
// A k-NN call uploads 14 KB of queries with hipMemcpyAsync (a 4.4 us blit kernel in front of the scan kernel).  Three ways for a
// kernel to see `bytes` of fresh host data, each timed as  [host writes the data] -> launch -> hipStreamSynchronize :
//   copy : pinned staging + hipMemcpyAsync to device memory, kernel reads device memory             (what ships)
//   zc   : kernel reads the pinned host buffer itself (zero-copy over PCIe; every workgroup reads all of it)
//   bar  : device memory allocated fine-grained (hipExtMallocWithFlags), written by the HOST through the PCIe BAR mapping
//          (if the allocation is host-accessible at all: probed under a SIGSEGV / SIGBUS handler), kernel reads local memory
// The kernel stands in for the scan: `wgs` workgroups each read the whole buffer (as every scan workgroup reads every query)
// and one of them writes a checksum to pinned memory.
// usage: bar_write [bytes=14336] [wgs=977] [iters=2000]
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static sigjmp_buf probe_jmp;

__global__ void reader(const unsigned* in, int nwords, unsigned* out) {
  unsigned s = 0;
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) s += in[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) out[0] = red[0] + red[1] + red[2] + red[3];
}

int main(int argc, char** argv) {
  const int bytes = argc > 1 ? atoi(argv[1]) : 14336, wgs = argc > 2 ? atoi(argv[2]) : 977, iters = argc > 3 ? atoi(argv[3]) : 2000;
  const int nw = bytes / 4;
  CK(hipSetDevice(0));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *h_in, *h_out, *d_in, *d_fg = nullptr;
  CK(hipHostMalloc((void**)&h_in, bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_out, 64, hipHostMallocDefault));
  CK(hipMalloc((void**)&d_in, bytes));
  unsigned* src = (unsigned*)malloc(bytes);
  bool bar_ok = false;
  if (hipExtMallocWithFlags((void**)&d_fg, bytes, hipDeviceMallocFinegrained) == hipSuccess && d_fg) {
    // is the allocation writable from the host?  (a fault lands in the handler below)
    struct sigaction sa{}, o1{}, o2{};
    sa.sa_handler = [](int) { siglongjmp(probe_jmp, 1); };
    sigaction(SIGSEGV, &sa, &o1); sigaction(SIGBUS, &sa, &o2);
    if (sigsetjmp(probe_jmp, 1) == 0) { volatile unsigned* p = d_fg; p[0] = 1u; p[nw - 1] = 2u; bar_ok = p[0] == 1u; }
    sigaction(SIGSEGV, &o1, nullptr); sigaction(SIGBUS, &o2, nullptr);
    printf("fine-grained device allocation: host %s\n", bar_ok ? "can write it" : "cannot write it (fault)");
  } else {
    (void)hipGetLastError();
    printf("fine-grained device allocation: refused\n");
  }
  unsigned* d_zc = nullptr; CK(hipHostGetDevicePointer((void**)&d_zc, h_in, 0));
  unsigned* d_out = nullptr; CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
  auto run = [&](const char* name, int mode) {
    double best = 1e30, sum = 0; int bad = 0;
    for (int it = -50; it < iters; ++it) {
      unsigned want = 0;
      for (int i = 0; i < nw; ++i) { src[i] = (unsigned)(it * 131 + i * 7 + mode); want += src[i]; }
      h_out[0] = 0xdeadbeefu;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) { memcpy(h_in, src, bytes); CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, st)); hipLaunchKernelGGL(reader, dim3(wgs), dim3(256), 0, st, d_in, nw, d_out); }
      else if (mode == 1) { memcpy(h_in, src, bytes); hipLaunchKernelGGL(reader, dim3(wgs), dim3(256), 0, st, d_zc, nw, d_out); }
      else { memcpy(d_fg, src, bytes); __builtin_ia32_sfence(); hipLaunchKernelGGL(reader, dim3(wgs), dim3(256), 0, st, d_fg, nw, d_out); }
      CK(hipStreamSynchronize(st));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (it >= 0) { sum += us; if (us < best) best = us; bad += h_out[0] != want; }
    }
    printf("%-5s %6d B x %4d workgroups: mean %7.2f us  min %7.2f us  wrong %d / %d\n", name, bytes, wgs, sum / iters, best, bad, iters);
  };
  run("copy", 0);
  run("zc", 1);
  if (bar_ok) run("bar", 2);
  run("copy", 0);
  return 0;
}
