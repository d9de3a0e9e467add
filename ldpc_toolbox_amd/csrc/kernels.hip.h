// HIP kernels of the batched BP decoder (gfx950 / CDNA4, wave64).
//
// Mapping used by every kernel: a LANE owns one codeword (VEC consecutive codewords in
// the streaming kernels), a WAVEFRONT owns one graph node (check row / variable) for a
// tile of 64*VEC codewords, so that
//   * all graph indices are wave-uniform (scalar loads, SGPR address math),
//   * every global access is a contiguous 256 B .. 1 KiB row segment,
//   * the per-node reductions (min1/min2/argmin/sign for min-sum, the slot-ordered
//     variable sum) run in registers with no cross-lane traffic.
// Cross-lane primitives are used where data really crosses codewords: ballots for the
// packed hard decisions and the "all codewords of my tile are finished" early-outs.
// The sum-product family stages the check row's edges in LDS ([slot][thread] columns,
// conflict-free) because those rules need random access to all d inputs.
//
// Arithmetic follows the reference rule by rule (citations at each function); compiled
// with -ffp-contract=off, comparisons instead of sign-bit tricks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "exact_math.h"
#include "fast_div.h"

// Timing-experiment switches (skip stores / gathers: WRONG results) exist only in builds made with
// EXTRA_HIPFLAGS=-DLDPC_EXPERIMENTS (tools/records_ab.py, tools/latency_probe.py); the product carries neither the
// kernel parameter nor the tests on it.
#ifdef LDPC_EXPERIMENTS
#define LDPC_DBG_PARAM(name) , uint32_t name
#define LDPC_DBG_ARG(x) , x
#else
#define LDPC_DBG_PARAM(name)
#define LDPC_DBG_ARG(x)
#endif

// The kernels, by schedule.  Every translation unit of the library includes this header (device_decoder_internal.h); a kernel is
// compiled where it is launched: the float rules in run_group_f32.hip / run_group_f64.hip, the group kernels in device_decoder.hip.
#include "kernels_common.hip.h"
#include "kernels_flooding.hip.h"
#include "kernels_layered.hip.h"
#include "kernels_group.hip.h"
#include "kernels_experiments.hip.h"
