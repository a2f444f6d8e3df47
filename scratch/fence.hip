#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_plain(float* out, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) out[i]=i*0.5f; }
__global__ void k_fence(float* out, int n, unsigned* counter){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) out[i]=i*0.5f;
  __syncthreads();
  if(threadIdx.x==0){ __builtin_amdgcn_fence(__ATOMIC_RELEASE,"agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); unsigned t=atomicAdd(counter,1u); if(t==gridDim.x-1){ __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"agent"); atomicExch(counter,0u);} }
}
__global__ void k_atomic_only(float* out, int n, unsigned* counter){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) out[i]=i*0.5f;
  __syncthreads();
  if(threadIdx.x==0){ unsigned t=atomicAdd(counter,1u); if(t==gridDim.x-1){ atomicExch(counter,0u);} }
}
int main(){ float* d; unsigned* c; hipMalloc(&d, 1<<24); hipMalloc(&c,4); hipMemset(c,0,4);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int blocks : {30, 133, 390, 2048}){
    int n=blocks*256; float ms;
    for(int v=0; v<3; ++v){
      for(int w=0;w<3;++w){ if(v==0) hipLaunchKernelGGL(k_plain,dim3(blocks),dim3(256),0,0,d,n); else if(v==1) hipLaunchKernelGGL(k_fence,dim3(blocks),dim3(256),0,0,d,n,c); else hipLaunchKernelGGL(k_atomic_only,dim3(blocks),dim3(256),0,0,d,n,c);}
      hipEventRecord(e0,0);
      for(int r=0;r<200;++r){ if(v==0) hipLaunchKernelGGL(k_plain,dim3(blocks),dim3(256),0,0,d,n); else if(v==1) hipLaunchKernelGGL(k_fence,dim3(blocks),dim3(256),0,0,d,n,c); else hipLaunchKernelGGL(k_atomic_only,dim3(blocks),dim3(256),0,0,d,n,c);}
      hipEventRecord(e1,0); hipEventSynchronize(e1); hipEventElapsedTime(&ms,e0,e1);
      printf("blocks %d variant %s: %.2f us/launch\n", blocks, v==0?"plain":v==1?"fence+ticket":"ticket only", ms*1000/200);
    }
  }
  return 0; }
