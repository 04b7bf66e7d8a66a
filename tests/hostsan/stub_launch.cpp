// tests/hostsan/stub_launch.cpp — TEST INFRASTRUCTURE (see hip/hip_runtime.h of this directory): the kernel launchers xr_batch.cpp
// declares, doing nothing.  Everything the sanitizer build exercises happens BEFORE a launch: validation, staging, copies, bookkeeping.
#include <hip/hip_runtime.h>

#include "../../xroute_env_amd/csrc/xr_device.h"

extern "C" {
int64_t xr_stub_alloc_limit = -1;
int64_t xr_stub_alloc_live = 0;
int64_t xr_stub_launches = 0;

hipError_t xr_launch_ingest(const uint32_t*, int16_t*, int16_t*, int64_t, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_reset(const XrBatchDev*, const uint8_t*, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_route_set_max_lds(size_t) { return hipSuccess; }
hipError_t xr_launch_route(const XrBatchDev*, const int32_t*, int, int, size_t, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_route_occupancy(int, int, size_t, int, int* per_cu, size_t* lds) { if (per_cu) *per_cu = 1; if (lds) *lds = 0; return hipSuccess; }
hipError_t xr_launch_plan(const XrBatchDev*, const int32_t*, uint32_t*, int32_t*, int*, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_route_order(const XrBatchDev*, const int32_t*, int32_t*, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_step_queue(const XrBatchDev*, const int32_t*, int, int, size_t, int, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_netplanes(const XrBatchDev*, int, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_order(const XrBatchDev*, const int32_t*, int, int32_t*, int, int, size_t, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_random_actions(const XrBatchDev*, int32_t*, uint64_t, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_obs(const XrBatchDev*, float*, int64_t, int, int, int, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_obs_records(const uint32_t*, int, int, int, const int32_t*, int, float*, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_unit_helpers(const XrBatchDev*, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_netplanes_pairs(const XrBatchDev*, const int32_t*, const int32_t*, int, float*, int64_t, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_pack_state(const XrBatchDev*, uint8_t*, int64_t, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_guide_masks(const XrBatchDev*, uint8_t*, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_ingest_state(const XrBatchDev*, const int16_t*, const uint64_t*, const int32_t*, hipStream_t) { xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_net_tower(const void*, const int32_t*, const int32_t*, int32_t, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t, const float*, const float*,
                               float*, int32_t*, int32_t, hipStream_t, int32_t* status) { *status = 0; xr_stub_launches++; return hipSuccess; }
hipError_t xr_launch_expand_state(const XrBatchDev*, const uint8_t*, int64_t, int, float*, int64_t, int32_t*, int32_t*, int, hipStream_t) { xr_stub_launches++; return hipSuccess; }
}
