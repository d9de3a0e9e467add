// Static instruction counts of the f64 Phi function and its parts (no GPU needed):
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/mb/phi64_parts.hip -o /tmp/phi64_parts ; tools/mb/isa_loops.py /tmp/phi64_parts <kernel>
#include <hip/hip_runtime.h>
#include "../../ldpc_toolbox_amd/csrc/exact_math.h"
using namespace ldpc;
#define K(name, expr) extern "C" __global__ void name(const double *in, double *out) { const double x = in[threadIdx.x]; const double y = in[threadIdx.x + 256]; (void)y; out[threadIdx.x] = (expr); }
K(k_phi_fused, em::phi(x))
K(k_phi_three_calls, -(em::log(em::tanh(0.5 * fmax(x, 1e-30)))))
K(k_tanh, em::tanh(x))
K(k_expm1, em::expm1(x))
K(k_log, em::log(x))
K(k_div, x / y)
K(k_copy, x)
int main() { return 0; }
