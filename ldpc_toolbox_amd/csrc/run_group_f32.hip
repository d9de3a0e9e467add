// The f32 float rules: DeviceDecoder::run_group<float> and every kernel it launches.
#ifdef LDPC_EXPERIMENTS
#define LDPC_STREAM_KERNELS_TU 1  // this translation unit compiles continuous batching's non-template kernel
#endif
#include "run_group.hip.h"
#ifdef LDPC_EXPERIMENTS
#include "decode_stream.hip.h"
#endif

namespace ldpc {
template int DeviceDecoder::run_group<float>(Workspace &, const void *, bool, size_t, uint32_t, uint8_t *, size_t, int32_t *, void *,
                                             hipStream_t, bool);
}  // namespace ldpc
