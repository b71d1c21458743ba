// compile check of the conveyor's line kernel on its own (seconds instead of the minutes pipeline.hip takes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -S \
//         -I alphazero-pybind11_amd/csrc -o /tmp/cv_line.s scripts/micro/cv_line_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../experiments/conveyor_c4.h"
namespace cvn = azmi_net_dev::cv;
extern "C" __global__ __launch_bounds__(256, 1) void k_line(cvn::CvArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_cvl[];
  cvn::line_wg(a, lds_cvl);
}
