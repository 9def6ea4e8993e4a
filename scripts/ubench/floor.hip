// micro-benchmark: per-kernel floor on MI355X for the launch patterns goctr uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
struct St { unsigned g; long long b; };
__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }
__global__ void k_state(const St* st, float* out) { long long b = st->b; if (b == 12345) out[threadIdx.x] = 1.f; }
__global__ void k_lds(float* out) { extern __shared__ float sm[]; sm[threadIdx.x] = 1.f; __syncthreads(); if (sm[(threadIdx.x+1)&255] == 2.f) out[0] = 1.f; }
__global__ void k_stream(const float4* in, float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { float4 v = in[i]; v.x += 1.f; out[i] = v; } }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 64<<20)); float* d2; CK(hipMalloc(&d2, 64<<20)); St* st; CK(hipMalloc(&st, sizeof(St))); CK(hipMemset(st,0,sizeof(St)));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch, int n) {
    for (int i=0;i<50;i++) launch();
    hipStreamSynchronize(s);
    hipEventRecord(a,s); for (int i=0;i<n;i++) launch(); hipEventRecord(b,s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); printf("%-40s %8.3f us/launch (eager back-to-back)\n", name, ms*1e3/n);
    // graph of 20 launches
    hipGraph_t g; hipGraphExec_t ge; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); for (int i=0;i<20;i++) launch(); hipStreamEndCapture(s,&g); hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
    for (int i=0;i<10;i++) hipGraphLaunch(ge,s); hipStreamSynchronize(s);
    hipEventRecord(a,s); for (int i=0;i<n/20;i++) hipGraphLaunch(ge,s); hipEventRecord(b,s); hipEventSynchronize(b);
    hipEventElapsedTime(&ms,a,b); printf("%-40s %8.3f us/launch (graph x20)\n", name, ms*1e3/(n/20*20));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  };
  timeit("empty 256x256", [&]{ hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d); }, 2000);
  timeit("empty 2048x256", [&]{ hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, d); }, 2000);
  timeit("state-load 256x256", [&]{ hipLaunchKernelGGL(k_state, dim3(256), dim3(256), 0, s, st, d); }, 2000);
  timeit("lds 131KB 256x256", [&]{ hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 131*1024, s, d); }, 2000);
  timeit("lds 32KB 256x256", [&]{ hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 32*1024, s, d); }, 2000);
  timeit("stream 1MB (64K float4)", [&]{ hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, s, (const float4*)d, (float4*)d2, 65536); }, 2000);
  timeit("stream 32MB", [&]{ hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, s, (const float4*)d, (float4*)d2, 2097152); }, 1000);
  return 0;
}
