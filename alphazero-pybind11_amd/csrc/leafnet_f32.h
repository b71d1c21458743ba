// fp32 reference-precision path of the leaf net (csrc/leafnet_f32.hip): same architecture family as the
// MFMA kernels, plain fp32 arithmetic, for the 1e-5 parity tier.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/azmi.h"

namespace azmi_f32 {
size_t blob_bytes(const azmi_net_desc* d);
// returns AZMI_OK or a negative status; *err receives a static/thread-local message
int create(const azmi_net_desc* d, const void* blob, size_t bytes, int device, void** impl, const char** err);
int forward(void* impl, const float* canon, float* v, float* pi, uint32_t batch, void* stream, const char** err);
int reserve(void* impl, uint32_t batch, const char** err);
void destroy(void* impl);
void dims(void* impl, uint32_t* chw, uint32_t* p1, uint32_t* m);   // floats per canonical row, P+1, M
}  // namespace azmi_f32
