/* TEST INFRASTRUCTURE ONLY (oracle/_ref build of the reference's own simple-knn sources; never part of the product).
 * The reference file submodules/simple-knn/simple_knn.cu is CUDA; it is compiled unmodified, from where it lies, by hipcc with
 * this directory first on the include path: the handful of CUDA runtime names it uses are mapped onto their HIP equivalents. */
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
