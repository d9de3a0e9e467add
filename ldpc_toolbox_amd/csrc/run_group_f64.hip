// The f64 float rules: DeviceDecoder::run_group<double> and every kernel it launches.
#include "run_group.hip.h"

namespace ldpc {
template int DeviceDecoder::run_group<double>(Workspace &, const void *, bool, size_t, uint32_t, uint8_t *, size_t, int32_t *, void *,
                                              hipStream_t, bool);
}  // namespace ldpc
