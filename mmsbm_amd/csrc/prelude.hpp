// prelude.hpp -- what every translation unit of the library starts with, in dependency order
//
// The library is built from several translation units compiled side by side (mmsbm_amd/build.py) and linked into one
// shared object:
//   mmsbm_hip.hip  the C ABI, the context, the iteration's order of launches
//   tu_seg.hip     the two triple passes (seg_pass.hpp)
//   tu_pair.hip    the pair stage on the vector ALUs (pair_block.hpp, pair_quad.hpp)
//   tu_mfma.hip    the pair stage on the matrix cores (pair_mfma.hpp)
//   tu_etap.hip    eta_p (eta_p.hpp)
//   tu_fused.hip   the two-launch iteration of small problems (fused_small.hpp)
//   tu_once.hip    once-per-run kernels: likelihood, prod_dist / predict, omegas, the random start
//   tu_layout.hip  the layout's sorts on the device (rocPRIM)
// unity.hip includes them all into ONE unit: the diagnostic builds (-DMMSBM_STAMPS, -DMMSBM_ABLATE) and
// scripts/kernel_resources.sh use it.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mmsbm_hip.h"
#include "layout.hpp"
#include "common.hpp"
#include "shapes.hpp"
#include "stamps.hpp"
#include "rowtab.hpp"
#include "context.hpp"
#include "launch.hpp"
