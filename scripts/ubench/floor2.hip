#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void k_load(const float4* in, float* out, int n) { int i = blockIdx.x*256+threadIdx.x; float4 v = in[i % n]; if (v.x == 12345.f) out[0] = 1.f; }
__global__ void k_store(float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) out[i] = make_float4(1,2,3,4); }
__global__ void k_copy(const float4* in, float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { float4 v = in[i]; v.x += 1.f; out[i] = v; } }
__global__ void k_copy_grid(const float4* in, float4* out, int n) { for (int i = blockIdx.x*256+threadIdx.x; i<n; i += gridDim.x*256) { float4 v = in[i]; v.x += 1.f; out[i] = v; } }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 256<<20)); float* d2; CK(hipMalloc(&d2, 256<<20));
  CK(hipMemset(d, 0, 256<<20)); CK(hipMemset(d2, 0, 256<<20)); CK(hipDeviceSynchronize());
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch, int n) {
    hipGraph_t g; hipGraphExec_t ge; (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); for (int i=0;i<20;i++) launch(); (void)hipStreamEndCapture(s,&g); (void)hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
    for (int i=0;i<20;i++) (void)hipGraphLaunch(ge,s); (void)hipStreamSynchronize(s);
    (void)hipEventRecord(a,s); for (int i=0;i<n/20;i++) (void)hipGraphLaunch(ge,s); (void)hipEventRecord(b,s); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms,a,b); printf("%-44s %8.3f us/launch\n", name, ms*1e3/(n/20*20));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  };
  for (int kb : {64, 1024, 8192, 32768, 131072}) {
    int n = kb*1024/16; int blocks = (n+255)/256; char nm[128];
    snprintf(nm, sizeof nm, "load-only  %6d KB (%d blocks)", kb, blocks);
    timeit(nm, [&]{ hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, s, (const float4*)d, d2, n); }, 1000);
    snprintf(nm, sizeof nm, "store-only %6d KB (%d blocks)", kb, blocks);
    timeit(nm, [&]{ hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, s, (float4*)d2, n); }, 1000);
    snprintf(nm, sizeof nm, "copy       %6d KB (%d blocks)", kb, blocks);
    timeit(nm, [&]{ hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)d2, n); }, 1000);
    snprintf(nm, sizeof nm, "copy-gridstride %6d KB (2048 blocks)", kb);
    timeit(nm, [&]{ hipLaunchKernelGGL(k_copy_grid, dim3(2048), dim3(256), 0, s, (const float4*)d, (float4*)d2, n); }, 1000);
  }
  return 0;
}
