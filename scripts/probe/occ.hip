// probe: LDS capacity / occupancy facts of the box (build: hipcc --offload-arch=gfx950 occ.hip -o occ)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k64(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[63 - threadIdx.x]; }
__global__ void __launch_bounds__(256) k256(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("name %s CUs %d sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d maxThreadsPerMP %d clock %d\n", pr.name, pr.multiProcessorCount,
           pr.sharedMemPerBlock, pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock, pr.maxThreadsPerMultiProcessor, pr.clockRate);
    for (size_t lds : {1024, 8192, 16384, 35840, 40000, 65536, 71680, 81920, 143360}) {
        int n64 = -1, n256 = -1;
        hipFuncSetAttribute((const void*)k64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n64, k64, 64, lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n256, k256, 256, lds);
        printf("dyn LDS %6zu B: blocks/CU  64-thread %d   256-thread %d\n", lds, n64, n256);
    }
    return 0;
}
