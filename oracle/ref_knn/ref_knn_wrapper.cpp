// TEST INFRASTRUCTURE ONLY.  C entry point around the reference's own SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:185-221,
// compiled unmodified from /root/reference by oracle/ref_knn/Makefile): what the reference's distCUDA2 (spatial.cu:15-27) calls.
// points / mean_dist2 are DEVICE pointers ([P,3] float, [P] float).  Returns 0, or the HIP error code.
#include "cuda_runtime.h"
#include "simple_knn.h"
extern "C" int ref_simple_knn(int P, const float* points, float* mean_dist2)
{
    SimpleKNN::knn(P, (float3*)points, mean_dist2);
    return (int)hipDeviceSynchronize();
}
