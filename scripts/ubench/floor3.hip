#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void k_copy4(const float4* in, float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { float4 v = in[i]; v.x += 1.f; out[i] = v; } }
__global__ void k_copy1(const float* in, float* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { out[i] = in[i] + 1.f; } }
__global__ void k_ld_stconst(const float4* in, float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { float4 v = in[i]; if (v.x != 12345.f) out[i] = make_float4(1,2,3,4); } }
__global__ void k_copy_nt(const float4* in, float4* out, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { float4 v = in[i]; v.x += 1.f; __builtin_nontemporal_store(v.x, &out[i].x); __builtin_nontemporal_store(v.y, &out[i].y); __builtin_nontemporal_store(v.z, &out[i].z); __builtin_nontemporal_store(v.w, &out[i].w);} }
__global__ void k_two_loads(const float4* in, const float4* in2, float* out, int n) { int i = blockIdx.x*256+threadIdx.x; float4 v = in[i % n]; float4 w = in2[i % n]; if (v.x + w.x == 12345.f) out[0] = 1.f; }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 256<<20)); float* d2; CK(hipMalloc(&d2, 256<<20));
  CK(hipMemset(d, 0, 256<<20)); CK(hipMemset(d2, 0, 256<<20)); CK(hipDeviceSynchronize());
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch, int n) {
    hipGraph_t g; hipGraphExec_t ge; (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); for (int i=0;i<20;i++) launch(); (void)hipStreamEndCapture(s,&g); (void)hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
    for (int i=0;i<20;i++) (void)hipGraphLaunch(ge,s); (void)hipStreamSynchronize(s);
    (void)hipEventRecord(a,s); for (int i=0;i<n/20;i++) (void)hipGraphLaunch(ge,s); (void)hipEventRecord(b,s); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms,a,b); printf("%-50s %8.3f us/launch\n", name, ms*1e3/(n/20*20));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  };
  int n = 1024*1024/16, blocks = 256;
  timeit("copy4 d->d2 1MB", [&]{ hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)d2, n); }, 1000);
  timeit("copy4 d->d (in place) 1MB", [&]{ hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)d, n); }, 1000);
  timeit("copy4 d->d+2MB (same alloc) 1MB", [&]{ hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)(d + (2<<20)/4), n); }, 1000);
  timeit("copy1 (dword) 1MB", [&]{ hipLaunchKernelGGL(k_copy1, dim3(1024), dim3(256), 0, s, (const float*)d, d2, n*4); }, 1000);
  timeit("load + store-const 1MB", [&]{ hipLaunchKernelGGL(k_ld_stconst, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)d2, n); }, 1000);
  timeit("copy nontemporal store 1MB", [&]{ hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, s, (const float4*)d, (float4*)d2, n); }, 1000);
  timeit("two loads (d, d2) no store 1MB", [&]{ hipLaunchKernelGGL(k_two_loads, dim3(blocks), dim3(256), 0, s, (const float4*)d, (const float4*)d2, d+ (64<<20), n); }, 1000);
  timeit("copy4 128KB (32 blocks)", [&]{ hipLaunchKernelGGL(k_copy4, dim3(32), dim3(256), 0, s, (const float4*)d, (float4*)d2, 8192); }, 1000);
  timeit("copy4 256KB (64 blocks)", [&]{ hipLaunchKernelGGL(k_copy4, dim3(64), dim3(256), 0, s, (const float4*)d, (float4*)d2, 16384); }, 1000);
  timeit("copy4 512KB (128 blocks)", [&]{ hipLaunchKernelGGL(k_copy4, dim3(128), dim3(256), 0, s, (const float4*)d, (float4*)d2, 32768); }, 1000);
  return 0;
}
