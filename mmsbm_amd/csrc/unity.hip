// unity.hip -- every translation unit of the library as ONE unit (see prelude.hpp).  For the diagnostic builds, whose
// device-side state must be one object (-DMMSBM_STAMPS: g_stamps; -DMMSBM_ABLATE: phase ablation through
// mmsbm_hip_time_stage), and for scripts/kernel_resources.sh:
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DMMSBM_STAMPS -o /tmp/libstamps.so mmsbm_amd/csrc/unity.hip
//
// The product is NOT built from this file: mmsbm_amd/build.py compiles the units side by side and links them.
#include "mmsbm_hip.hip"
#include "tu_seg.hip"
#include "tu_pair.hip"
#include "tu_mfma.hip"
#include "tu_etap.hip"
#include "tu_fused.hip"
#include "tu_once.hip"
#include "tu_layout.hip"
