// Diagnostic: resident workgroups per CU the runtime reports for the z-column kernels.
#include "../aind_exaspim_neuron_segmentation_amd/csrc/conv3d.hip"
int main() {
    using namespace exaspim;
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3x3_zpair<F16Tag, 6, 8, 2, 4, 0, false>, 256, 0);
    printf("zpair<6,8> : %d workgroups per CU\n", n);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3x3_zpair<F16Tag, 4, 8, 2, 4, 0, false>, 256, 0);
    printf("zpair<4,8> : %d workgroups per CU\n", n);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3x3_zpipe<F16Tag, 6, 8, 16, 2, 4, 0, false>, 256, 0);
    printf("zpipe<6,8,16> : %d workgroups per CU\n", n);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
    return 0;
}
