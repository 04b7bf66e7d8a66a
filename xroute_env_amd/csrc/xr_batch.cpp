// xr_batch.cpp — host side of libxroute_hip.so: the C ABI declared in include/xroute_hip.h.
// Owns the device state of a batch of env slots and enqueues the gfx950 kernels of xr_kernels.hip.
// There is NO CPU fallback in this library: without a HIP device every compute entry point fails
// with XR_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/xroute_hip.h"
#include "xr_device.h"

extern "C" {
hipError_t xr_launch_ingest(const uint32_t*, int16_t*, int16_t*, int64_t, hipStream_t);
hipError_t xr_launch_reset(const XrBatchDev*, const uint8_t*, int, hipStream_t);
hipError_t xr_route_set_max_lds(size_t);
hipError_t xr_launch_route(const XrBatchDev*, const int32_t*, int, int, size_t, int, hipStream_t);
hipError_t xr_route_occupancy(int, int, size_t, int, int*, size_t*);
hipError_t xr_launch_plan(const XrBatchDev*, const int32_t*, uint32_t*, int32_t*, int*, hipStream_t);
hipError_t xr_launch_route_order(const XrBatchDev*, const int32_t*, int32_t*, hipStream_t);
hipError_t xr_launch_step_queue(const XrBatchDev*, const int32_t*, int, int, size_t, int, int, hipStream_t);
hipError_t xr_launch_netplanes(const XrBatchDev*, int, int, hipStream_t);
hipError_t xr_launch_order(const XrBatchDev*, const int32_t*, int, int32_t*, int, int, size_t, int, hipStream_t);
hipError_t xr_launch_random_actions(const XrBatchDev*, int32_t*, uint64_t, hipStream_t);
hipError_t xr_launch_obs(const XrBatchDev*, float*, int64_t, int, int, int, int, hipStream_t);
hipError_t xr_launch_obs_records(const uint32_t*, int, int, int, const int32_t*, int, float*, int, hipStream_t);
hipError_t xr_launch_unit_helpers(const XrBatchDev*, int, hipStream_t);
hipError_t xr_launch_netplanes_pairs(const XrBatchDev*, const int32_t*, const int32_t*, int, float*, int64_t, int, hipStream_t);
hipError_t xr_launch_pack_state(const XrBatchDev*, uint8_t*, int64_t, int, hipStream_t);
hipError_t xr_launch_guide_masks(const XrBatchDev*, uint8_t*, int, hipStream_t);
hipError_t xr_launch_expand_state(const XrBatchDev*, const uint8_t*, int64_t, int, float*, int64_t, int32_t*, int32_t*, int, hipStream_t);
hipError_t xr_launch_ingest_state(const XrBatchDev*, const int16_t*, const uint64_t*, const int32_t*, hipStream_t);
hipError_t xr_launch_net_tower(const void*, const int32_t*, const int32_t*, int32_t, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t, const float*, const float*,
                               float*, int32_t*, int32_t, hipStream_t, int32_t*);
}

namespace {

thread_local std::string g_err;

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define XR_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(XR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr size_t kLdsLimit = 160 * 1024;       // LDS per CU on gfx950
constexpr size_t kLdsStatic = 1280;            // static __shared__ of the route kernel (AP staging etc.)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace

struct xr_batch {
    xr_config cfg{};
    bool loaded = false;
    int n_regions = 0;
    int n_max = 0;          // padded max nodes per region (multiple of 8)
    int n_max_nodes = 0;    // true max N
    int n_lds = 0;          // padded distance-field words (odd strides), max over regions
    int tracks_max = 0;
    int lines_max = 0;
    int bits_max = 0;       // tracks + 2 x columns
    int zch = 0;            // 9 / 12 when all regions have that many layers
    int kzch = 0;           // template selector of the step kernels: zch, or -1 = bucketed-frontier router (xr_dial.h)
    int k_max = 0;
    int legal_words = 1;
    int path_cap = 0;
    int x_max = 0, y_max = 0;
    bool all_n_mult4 = true;
    bool lds_dist = true;
    bool stream_ok = false;   // ids + 2 bytes/node of the largest region fit the LDS of the observation stream form
    size_t route_lds = 0;
    int route_threads = 256;
    // regions
    DevBuf<XrRegionDev> regions;
    DevBuf<uint32_t> rg_rec;
    DevBuf<int16_t> rg_node_net, rg_owner0;
    DevBuf<int32_t> coords, net_csr, ap_node, ap_feat, net_info;
    DevBuf<uint8_t> ap_flags;
    DevBuf<int16_t> ap_pin;
    DevBuf<int32_t> guide_csr;            // XR-Maze v2, optional (xr_batch_load_guides): boxes of (region, net), indexed like net_csr
    DevBuf<int16_t> guide_box;
    DevBuf<uint8_t> guide_mask;           // XR-Maze v2 with guide_cost > 0: static "outside the guide" bitmasks of every (region, net) (build_guide_masks)
    size_t guide_mask_bytes = 0;
    std::vector<int32_t> h_net_off, h_n_nets, h_dims;   // per region: R.net_off, n_nets, (X, Y, Z) — what xr_batch_load_guides validates against
    size_t h_csr_size = 0;
    DevBuf<uint64_t> legal0;
    // envs
    DevBuf<int32_t> env_region, env_replay, nlegal, cum, delta, status, path, path_len, sweeps, touched, route_order;
    DevBuf<uint8_t> net_work, net_meas;
    int route_slots = 0;         // workgroups of the route kernel the chip holds at once (0: not asked yet)
    DevBuf<int16_t> owner;
    DevBuf<uint64_t> legal, hash;
    DevBuf<double> reward;
    DevBuf<XrStepRecord> records;
    DevBuf<uint8_t> done, cls_scratch;
    DevBuf<int64_t> env_steps;
    DevBuf<long long> phase_cycles;
    DevBuf<unsigned long long> total_steps;
    DevBuf<uint32_t> dist_scratch, dg_field, dg_masks, dg_touch, dg_path;
    bool dial_big = false;
    struct { int x = 0, y = 0, nmax = 0, margin = 0, ystep = 1; uint32_t m24_yz = 0, m24_z = 0, m24_mw = 0, s24 = 0; } win;   // window form (xr_dial3.h, WIN); x = 0: off
    DevBuf<unsigned short> list_scratch;
    // split observation
    DevBuf<int32_t> plan_region, plan_unit_net;
    DevBuf<uint32_t> plan_units, queue;         // queue: two banks of 4 counters (route tasks, units, planned units, -), alternating per call
    int queue_bank = 0;
    uint32_t* queue_last = nullptr;             // the bank of the last planning call (XR_FETCH_UNITS)
    int n_cus = 0, queue_blocks = 0, queue_blocks_sweep = 0;
    bool sweep_full = false;     // auto router: the full-rewrite queue launch of a large batch takes the line-segment sweeps
    size_t sweep_lds = 0;
    hipStream_t aux_stream = nullptr;
    std::vector<hipStream_t> region_streams;        // stream-per-region mode
    std::vector<hipEvent_t> region_events;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_w0 = nullptr, ev_w1 = nullptr;
    int last_obs_mode = 0;
    int last_obs_inplace = 0;
    int last_obs_sweeps = 0;
    const float* obs_valid_ptr = nullptr;       // buffer that holds the current observation of ALL env slots (in-place form)
    int64_t obs_valid_stride = 0;
    XrBatchDev dev{};
    ~xr_batch() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_w0) (void)hipEventDestroy(ev_w0);
        if (ev_w1) (void)hipEventDestroy(ev_w1);
        if (aux_stream) (void)hipStreamDestroy(aux_stream);
        for (hipEvent_t ev : region_events) (void)hipEventDestroy(ev);
        for (hipStream_t s : region_streams) (void)hipStreamDestroy(s);
    }
};

extern "C" {

int32_t xr_abi_version(void) { return XR_ABI_VERSION; }
const char* xr_last_error(void) { return g_err.c_str(); }

void xr_config_default(xr_config* c) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(xr_config);
    c->device = 0;
    c->n_envs = 1;
    c->via_cost = 800;
    c->drc_cost = 8;           // ispd/ispd18_test1/run-net-ordering-training.tcl:3  -drc_cost 8
    c->drc_unit = 400;
    c->max_route_count = 10;   // examples/launch_training.py:28
    c->auto_reset = 0;
    c->path_cap = 0;
    c->block_threads = 0;
    c->force_scratch_field = 0;
    c->obs_mode = 0;
    c->obs_writer_blocks = 0;
    c->obs_split_permille = 0;
    c->router = 0;
    c->dial_mult = 0;
    c->guide_cost = 0;
    c->guide_margin = 0;
    c->maze_end_iter = 1;      // ispd/ispd18_test1/run-net-ordering-training.tcl:3 runs 3; XR-Maze v1 = 1
    c->stream_per_region = 0;
    c->obs_helper_blocks = 0;
    c->launch_order = 0;
    c->debug_round_cap = 0;
    c->window = 0;
    c->w_violation = 500.0;    // baseline/DQN/train_DQN.py:99
    c->w_via = 4.0;
    c->w_wirelength = 0.5;
}

int32_t xr_device_count(int32_t* n) {
    if (!n) return fail(XR_ERR_INVALID, "xr_device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(XR_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return XR_OK;
}

int32_t xr_batch_create(const xr_config* cfg, xr_batch** out) {
    if (!cfg || !out) return fail(XR_ERR_INVALID, "xr_batch_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(xr_config))
        return fail(XR_ERR_INVALID, "xr_batch_create: xr_config.struct_size %d != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(xr_config));
    if (cfg->n_envs < 1) return fail(XR_ERR_INVALID, "xr_batch_create: n_envs must be >= 1");
    if (cfg->via_cost < 1 || cfg->drc_cost < 0 || cfg->drc_unit < 0 || cfg->max_route_count < 1)
        return fail(XR_ERR_INVALID, "xr_batch_create: via_cost >= 1, drc_cost/drc_unit >= 0, max_route_count >= 1");
    if ((int64_t)cfg->drc_cost * cfg->drc_unit >= (1 << 22) || cfg->via_cost >= (1 << 22))
        return fail(XR_ERR_RANGE, "xr_batch_create: via_cost and drc_cost*drc_unit must be < 2^22");
    if (cfg->launch_order < 0 || cfg->launch_order > 2) return fail(XR_ERR_INVALID, "xr_batch_create: launch_order must be 0, 1 or 2");
    if (cfg->obs_mode < 0 || cfg->obs_mode > XR_OBS_QUEUE || cfg->obs_writer_blocks < 0 || cfg->obs_split_permille < 0 || cfg->obs_split_permille > 1000)
        return fail(XR_ERR_INVALID, "xr_batch_create: obs_mode must be 0, XR_OBS_FUSED, XR_OBS_SPLIT or XR_OBS_QUEUE; obs_writer_blocks >= 0; obs_split_permille in 0..1000");
    if (cfg->router < 0 || cfg->router > XR_ROUTER_DIAL_R2 || cfg->dial_mult < 0 || cfg->dial_mult > 64)
        return fail(XR_ERR_INVALID, "xr_batch_create: router must be 0, XR_ROUTER_SWEEP, XR_ROUTER_DIAL or XR_ROUTER_DIAL_R2; dial_mult in 0..64");
    if (cfg->guide_cost < 0 || cfg->guide_cost >= (1 << 22) || cfg->guide_margin < 0 || cfg->maze_end_iter < 1 || cfg->maze_end_iter > 8 ||
        ((int64_t)cfg->drc_cost * cfg->drc_unit << (cfg->maze_end_iter - 1)) >= (1 << 22))
        return fail(XR_ERR_RANGE, "xr_batch_create: guide_cost in [0, 2^22), guide_margin >= 0, maze_end_iter in 1..8 with drc_cost*drc_unit << (maze_end_iter-1) < 2^22");
    if (cfg->stream_per_region < 0 || cfg->stream_per_region > 1 || (cfg->stream_per_region && cfg->n_envs > 64))
        return fail(XR_ERR_RANGE, "xr_batch_create: stream_per_region is 0 or 1 and needs n_envs <= 64");
    if (cfg->debug_round_cap < 0 || cfg->obs_helper_blocks < 0)
        return fail(XR_ERR_INVALID, "xr_batch_create: debug_round_cap and obs_helper_blocks must be >= 0");
    if (cfg->block_threads != 0 && (cfg->block_threads < 64 || cfg->block_threads > 1024 || cfg->block_threads % 64))
        return fail(XR_ERR_INVALID, "xr_batch_create: block_threads must be a multiple of 64 in [64, 1024]");
    int ndev = 0;
    XR_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(XR_ERR_HIP, "xr_batch_create: device %d not present (%d HIP devices)", cfg->device, ndev);
    XR_HIP(hipSetDevice(cfg->device));
    xr_batch* b = new (std::nothrow) xr_batch();
    if (!b) return fail(XR_ERR_NOMEM, "xr_batch_create: out of host memory");
    b->cfg = *cfg;
    *out = b;
    return XR_OK;
}

int32_t xr_batch_destroy(xr_batch* b) {
    if (!b) return XR_OK;
    (void)hipSetDevice(b->cfg.device);
    delete b;
    return XR_OK;
}

static int32_t build_guide_masks(xr_batch* b, hipStream_t st);

int32_t xr_batch_load_regions(xr_batch* b, const xr_region_desc* regs, int32_t n_regions, void* stream) {
    if (!b || !regs || n_regions < 1) return fail(XR_ERR_INVALID, "xr_batch_load_regions: bad argument");
    XR_HIP(hipSetDevice(b->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = b->cfg.n_envs;
    // a reload invalidates the batch until it has completed: a failure midway must not leave `loaded` set over
    // freed or partly reallocated device buffers
    b->loaded = false;
    b->obs_valid_ptr = nullptr;
    b->n_cus = 0;
    b->route_slots = 0;
    b->guide_csr.release(); b->guide_box.release(); b->guide_mask.release(); b->guide_mask_bytes = 0;        // guides belong to the regions they were loaded for
    memset(&b->dev, 0, sizeof(b->dev));

    std::vector<XrRegionDev> hreg(n_regions);
    std::vector<uint32_t> hrec;
    std::vector<int32_t> hcoords, hcsr, hap_node, hap_feat;
    std::vector<int16_t> hap_pin;
    std::vector<float> hwork;            // per (region, net): predicted route work (launch order of route-only launches)
    std::vector<int32_t> hinfo;          // per (region, net): static facts for xr_dial3.h (XrBatchDev::net_info)
    std::vector<uint8_t> hap_flags;      // per access point: bit 0 = its pin sits in a closed pocket
    int64_t edge_max = b->cfg.via_cost;  // longest edge of any region graph (range checks of xr_dial3.h)
    int64_t ext_max = 0;                 // widest span of a region's tracks in x or y, DBU
    std::map<uint64_t, uint64_t> magic_cache;   // (divisor, limit) -> multiplier << 8 | shift (0xFF: none)
    bool div24_all = true;               // every region has its exact 24-bit division constants
    // exact 24-bit magics (largest shift whose multiplier and products fit, then checked for every n below `lim`)
    auto magic24 = [&](uint32_t dv, uint32_t lim, uint32_t& M, uint32_t& S) -> bool {
        const uint64_t key = ((uint64_t)dv << 32) | lim;
        auto it = magic_cache.find(key);
        if (it != magic_cache.end()) { M = (uint32_t)(it->second >> 8); S = (uint32_t)(it->second & 0xFF); return S != 0xFF; }
        bool found = false;
        if (lim <= (1u << 24))
            for (int sh = 31; sh >= 0 && !found; sh--) {
                const uint64_t m = (((uint64_t)1 << sh) + dv - 1) / dv;
                if (m >= (1u << 24) || (uint64_t)(lim > 0 ? lim - 1 : 0) * m >= ((uint64_t)1 << 32)) continue;
                bool ok = true;
                for (uint32_t nn = 0; nn < lim && ok; nn++) ok = (uint32_t)(((uint64_t)nn * m) >> sh) == nn / dv;
                if (ok) { M = (uint32_t)m; S = (uint32_t)sh; found = true; }
            }
        magic_cache[key] = found ? (((uint64_t)M << 8) | S) : 0xFF;
        return found;
    };
    int n_max_nodes = 0, k_max = 0, x_max = 0, y_max = 0, n_lds = 0, tracks_max = 0, lines_max = 0, bits_max = 0, z_min = 1 << 30, z_max = 0, ncol_max = 0;
    size_t gmask_bytes = 0;        // XR-Maze v2: bytes of the static guide masks of every (region, net) (XrRegionDev::gmask_off)
    bool mult4 = true;
    for (int r = 0; r < n_regions; r++) {
        const xr_region_desc& d = regs[r];
        if (d.dim_x < 1 || d.dim_y < 1 || d.dim_z < 1 || d.dim_z > XR_MAX_LAYERS)
            return fail(XR_ERR_RANGE, "region %d: dims %dx%dx%d out of range (z <= %d)", r, d.dim_x, d.dim_y, d.dim_z,
                        XR_MAX_LAYERS);
        const int64_t n64 = (int64_t)d.dim_x * d.dim_y * d.dim_z;
        if (n64 > (int64_t)1 << 30) return fail(XR_ERR_RANGE, "region %d: too many nodes", r);
        if (!d.xs_host || !d.ys_host || !d.layer_dir_host || !d.nodes_host)
            return fail(XR_ERR_INVALID, "region %d: null array", r);
        if (d.n_nets < 0 || d.n_nets > XR_MAX_NETS) return fail(XR_ERR_RANGE, "region %d: n_nets %d", r, d.n_nets);
        // track coordinates within +-2^30 DBU: every difference of two of them (edge lengths, extents) then fits the kernels' int32 arithmetic
        for (int i = 0; i < d.dim_x; i++)
            if (d.xs_host[i] < -(1 << 30) || d.xs_host[i] > (1 << 30)) return fail(XR_ERR_RANGE, "region %d: xs[%d] = %d outside +-2^30", r, i, d.xs_host[i]);
        for (int i = 0; i < d.dim_y; i++)
            if (d.ys_host[i] < -(1 << 30) || d.ys_host[i] > (1 << 30)) return fail(XR_ERR_RANGE, "region %d: ys[%d] = %d outside +-2^30", r, i, d.ys_host[i]);
        for (int i = 1; i < d.dim_x; i++)
            if (d.xs_host[i] <= d.xs_host[i - 1]) return fail(XR_ERR_INVALID, "region %d: xs not strictly increasing", r);
        for (int i = 1; i < d.dim_y; i++)
            if (d.ys_host[i] <= d.ys_host[i - 1]) return fail(XR_ERR_INVALID, "region %d: ys not strictly increasing", r);
        const int N = (int)n64;
        XrRegionDev& R = hreg[r];
        R.X = d.dim_x; R.Y = d.dim_y; R.Z = d.dim_z; R.N = N;
        R.n_nets = d.n_nets;
        R.m0[0] = d.metrics0[0]; R.m0[1] = d.metrics0[1]; R.m0[2] = d.metrics0[2];
        R.ldir_mask = 0;
        for (int z = 0; z < d.dim_z; z++)
            if (d.layer_dir_host[z]) R.ldir_mask |= (1u << z);
        {   // bucketed-frontier router: smallest edge length, flat-index decode constants
            uint32_t wmin = (uint32_t)b->cfg.via_cost;
            for (int i = 1; i < d.dim_x; i++) wmin = std::min(wmin, (uint32_t)(d.xs_host[i] - d.xs_host[i - 1]));
            for (int i = 1; i < d.dim_y; i++) wmin = std::min(wmin, (uint32_t)(d.ys_host[i] - d.ys_host[i - 1]));
            R.w_min = std::max(1u, wmin);
            const uint32_t yz = (uint32_t)d.dim_y * (uint32_t)d.dim_z, zz = (uint32_t)d.dim_z;
            R.magic_yz = yz >= 2 ? (uint32_t)((1ULL << 32) / yz) : 0xFFFFFFFFu;
            R.magic_z = zz >= 2 ? (uint32_t)((1ULL << 32) / zz) : 0xFFFFFFFFu;
            const uint32_t mwv = (uint32_t)((n64 + 31) / 32);
            R.magic_mw = mwv >= 2 ? (uint32_t)((1ULL << 32) / mwv) : 0xFFFFFFFFu;
            uint32_t s_yz = 0, s_z = 0, s_mw = 0;
            const bool okd = n64 < 65536 && magic24(yz, (uint32_t)n64, R.m24_yz, s_yz) && magic24(zz, yz, R.m24_z, s_z) &&
                             magic24(std::max(1u, mwv), (uint32_t)n64, R.m24_mw, s_mw);
            R.s24 = s_yz | (s_z << 8) | (s_mw << 16) | ((okd ? 1u : 0u) << 24);
            if (!okd) div24_all = false;
        }
        R.xs_off = (int32_t)hcoords.size();
        hcoords.insert(hcoords.end(), d.xs_host, d.xs_host + d.dim_x);
        R.ys_off = (int32_t)hcoords.size();
        hcoords.insert(hcoords.end(), d.ys_host, d.ys_host + d.dim_y);
        // node records, padded to a multiple of 8 elements so that int16 planes stay 16-byte aligned
        R.gmask_off = (int64_t)gmask_bytes;
        R.gmask_stride = (int32_t)((((size_t)N + 7) / 8 + 15) & ~(size_t)15);
        R.pad0 = 0;
        gmask_bytes += (size_t)R.gmask_stride * (size_t)std::max(d.n_nets, 0);
        R.node_off = (int64_t)hrec.size();
        hrec.insert(hrec.end(), d.nodes_host, d.nodes_host + N);
        while (hrec.size() % 8) hrec.push_back(XR_TYPE_NORMAL);
        // per-net access-point lists (counting sort by 1-based net id; flat order inside a net)
        R.net_off = (int32_t)hcsr.size();
        R.ap_off = (int32_t)hap_node.size();
        std::vector<int32_t> cnt(d.n_nets + 2, 0);
        for (int f = 0; f < N; f++) {
            const uint32_t rec = d.nodes_host[f];
            if (XR_REC_TYPE(rec) == XR_TYPE_ACCESS) {
                const int net1 = (int)XR_REC_NET1(rec);
                if (net1 < 1 || net1 > d.n_nets)
                    return fail(XR_ERR_RANGE, "region %d node %d: ACCESS node with net id %d outside 1..%d", r, f, net1,
                                d.n_nets);
                cnt[net1 + 1]++;
            }
        }
        for (int n = 1; n <= d.n_nets + 1; n++) cnt[n] += cnt[n - 1];
        R.nlegal0 = 0;
        for (int n = 1; n <= d.n_nets; n++) {
            const int c = cnt[n + 1] - cnt[n];
            if (c > XR_MAX_AP_PER_NET)
                return fail(XR_ERR_RANGE, "region %d net %d: %d access points (max %d)", r, n, c, XR_MAX_AP_PER_NET);
            R.nlegal0 += (c > 0);
        }
        hcsr.insert(hcsr.end(), cnt.begin(), cnt.end());
        const size_t base = hap_node.size();
        hap_node.resize(base + cnt[d.n_nets + 1]);
        hap_pin.resize(base + cnt[d.n_nets + 1]);
        std::vector<int32_t> cur(cnt.begin(), cnt.end());
        for (int f = 0; f < N; f++) {
            const uint32_t rec = d.nodes_host[f];
            if (XR_REC_TYPE(rec) == XR_TYPE_ACCESS) {
                const int net1 = (int)XR_REC_NET1(rec);
                hap_node[base + cur[net1]] = f;
                hap_pin[base + cur[net1]] = (int16_t)XR_REC_PIN1(rec);
                cur[net1]++;
            }
        }
        // per access point: does it have an in-bounds axis neighbour that is an access point of the same net, any pin
        // (the reference's aliased direction planes, baseline/build_3Dgrid.py:125-138); static, so decided once here
        hap_feat.resize(hap_node.size());
        {
            const int Yd = d.dim_y, Zd = d.dim_z, YZd = Yd * Zd;
            auto net_of = [&](int f) -> int {
                const uint32_t rr = d.nodes_host[f];
                return XR_REC_TYPE(rr) == XR_TYPE_ACCESS ? (int)XR_REC_NET1(rr) : 0;
            };
            for (size_t i = base; i < hap_node.size(); i++) {
                const int f = hap_node[i], n = net_of(f);
                const int z = f % Zd, y = (f / Zd) % Yd, x = f / YZd;
                const bool adj = (x + 1 < d.dim_x && net_of(f + YZd) == n) || (y > 0 && net_of(f - Zd) == n) || (x > 0 && net_of(f - YZd) == n) ||
                                 (y + 1 < Yd && net_of(f + Zd) == n) || (z + 1 < Zd && net_of(f + 1) == n) || (z > 0 && net_of(f - 1) == n);
                hap_feat[i] = f | (adj ? (int32_t)0x80000000 : 0);
            }
        }
        // Static facts of every net for the round-3 router: lowest pin, number of distinct pins, and which pins are ISOLATED — all
        // access points of the pin sit in a pocket closed by BLOCKAGE nodes that holds no access point of another pin of the net.
        // Such a pin can never be reached (nor reach anything): XR-Maze v1 charges one violation for it, and a router that has to
        // find that out by searching explores the whole component first.  The pocket's boundary is static, so the flood (budget 64
        // nodes; a larger pocket just counts as open, the result is the same) runs here, once, not in every route.
        hinfo.resize(hcsr.size(), 0);
        hap_flags.resize(hap_node.size(), 0);
        {
            const int Xd = d.dim_x, Yd = d.dim_y, Zd = d.dim_z, YZd = Yd * Zd;
            for (int i = 1; i < Xd; i++) edge_max = std::max<int64_t>(edge_max, d.xs_host[i] - d.xs_host[i - 1]);
            for (int i = 1; i < Yd; i++) edge_max = std::max<int64_t>(edge_max, d.ys_host[i] - d.ys_host[i - 1]);
            ext_max = std::max<int64_t>(ext_max, std::max<int64_t>((int64_t)d.xs_host[Xd - 1] - d.xs_host[0], (int64_t)d.ys_host[Yd - 1] - d.ys_host[0]));
            auto blocked = [&](int f) { return XR_REC_TYPE(d.nodes_host[f]) == XR_TYPE_BLOCKAGE; };
            std::vector<int> seen_pins, pocket;
            for (int n = 1; n <= d.n_nets; n++) {
                const int lo = cnt[n], hi = cnt[n + 1];
                if (hi <= lo) continue;
                seen_pins.clear();
                for (int i = lo; i < hi; i++) {
                    const int pn = hap_pin[base + i];
                    if (std::find(seen_pins.begin(), seen_pins.end(), pn) == seen_pins.end()) seen_pins.push_back(pn);
                }
                const int first = *std::min_element(seen_pins.begin(), seen_pins.end());
                int n_iso = 0, src_iso = 0;
                for (int pn : seen_pins) {
                    pocket.clear();
                    for (int i = lo; i < hi; i++) if (hap_pin[base + i] == pn) pocket.push_back(hap_node[base + i]);
                    bool open_pocket = false;
                    for (size_t k = 0; k < pocket.size() && !open_pocket; k++) {
                        const int f = pocket[k], z = f % Zd, y = (f / Zd) % Yd, x = f / YZd;
                        const bool vert = d.layer_dir_host[z] != 0;
                        const int nb[4] = {vert ? (y + 1 < Yd ? f + Zd : -1) : (x + 1 < Xd ? f + YZd : -1),
                                           vert ? (y > 0 ? f - Zd : -1) : (x > 0 ? f - YZd : -1),
                                           z + 1 < Zd ? f + 1 : -1, z > 0 ? f - 1 : -1};
                        for (int q = 0; q < 4; q++) {
                            if (nb[q] < 0 || blocked(nb[q])) continue;
                            if (std::find(pocket.begin(), pocket.end(), nb[q]) != pocket.end()) continue;
                            if (pocket.size() >= 64) { open_pocket = true; break; }
                            pocket.push_back(nb[q]);
                        }
                    }
                    if (open_pocket) continue;
                    bool other = false;                    // an access point of another pin of the net inside the pocket: reachable
                    for (int i = lo; i < hi && !other; i++)
                        if (hap_pin[base + i] != pn && std::find(pocket.begin(), pocket.end(), hap_node[base + i]) != pocket.end()) other = true;
                    if (other) continue;
                    for (int i = lo; i < hi; i++) if (hap_pin[base + i] == pn) hap_flags[base + i] = 1;
                    if (pn == first) src_iso = 1; else n_iso++;
                }
                hinfo[R.net_off + n] = (first & 0x3FFF) | ((int)seen_pins.size() << 14) | (n_iso << 22) | (src_iso << 30);
                // heuristic slot of every access point (bits 1..2 of ap_flags): pins in ascending id order, the lowest one (the first
                // component: never a target) aside, are dealt round-robin over the three pin boxes of xr_dial3.h's heuristic
                std::sort(seen_pins.begin(), seen_pins.end());
                for (int i = lo; i < hi; i++) {
                    const int rank = (int)(std::find(seen_pins.begin(), seen_pins.end(), (int)hap_pin[base + i]) - seen_pins.begin());
                    hap_flags[base + i] |= (uint8_t)(((rank + 2) % 3) << 1);         // rank 1 -> slot 0, 2 -> 1, 3 -> 2, 4 -> 0 ...
                }
            }
        }
        // predicted work of routing net n: extent of its access points (DBU; a layer of span counted as half a via) times
        // (6 + pins) — the shape tools/lpt_probe.py fitted; only the ORDER of these numbers matters
        hwork.resize(hcsr.size(), 0.0f);
        {
            const int Yd = d.dim_y, Zd = d.dim_z, YZd = Yd * Zd;
            for (int n = 1; n <= d.n_nets; n++) {
                const int lo = cnt[n], hi = cnt[n + 1];
                if (hi <= lo) continue;
                int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1, z0 = 1 << 30, z1 = -1;
                uint64_t pins[4] = {0, 0, 0, 0};
                int npins = 0;
                for (int i = lo; i < hi; i++) {
                    const int f = hap_node[base + i], z = f % Zd, y = (f / Zd) % Yd, x = f / YZd;
                    x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
                    z0 = std::min(z0, z); z1 = std::max(z1, z);
                    const int pn = hap_pin[base + i] & 255;
                    if (!((pins[pn >> 6] >> (pn & 63)) & 1)) { pins[pn >> 6] |= 1ull << (pn & 63); npins++; }
                }
                const double ext = (double)(d.xs_host[x1] - d.xs_host[x0]) + (double)(d.ys_host[y1] - d.ys_host[y0]) +
                                   0.5 * b->cfg.via_cost * (z1 - z0) + R.w_min;
                hwork[R.net_off + n] = (float)(ext * (6 + npins));
            }
        }
        n_max_nodes = std::max(n_max_nodes, N);
        k_max = std::max(k_max, d.n_nets);
        x_max = std::max(x_max, d.dim_x);
        y_max = std::max(y_max, d.dim_y);
        if (N % 4) mult4 = false;
        {   // padded field: l = x*SX + y*SY + z, SY = Z|1, SX = (Y*SY)|1 (xr_route_kernel)
            const int64_t sy = d.dim_z | 1, sx = ((int64_t)d.dim_y * sy) | 1;
            const int64_t words = (int64_t)d.dim_x * sx;
            if (words > ((int64_t)1 << 30)) return fail(XR_ERR_RANGE, "region %d: too many nodes", r);
            n_lds = std::max(n_lds, (int)words);
            int nv = 0;
            for (int z = 0; z < d.dim_z; z++) nv += d.layer_dir_host[z] ? 1 : 0;
            tracks_max = std::max(tracks_max, (d.dim_z - nv) * d.dim_y + nv * d.dim_x);
            lines_max = std::max(lines_max, (d.dim_z - nv) * d.dim_y + nv * d.dim_x + d.dim_x * d.dim_y);
            bits_max = std::max(bits_max, (d.dim_z - nv) * d.dim_y + nv * d.dim_x + 2 * d.dim_x * d.dim_y);
            z_min = std::min(z_min, d.dim_z); z_max = std::max(z_max, d.dim_z);
            ncol_max = std::max(ncol_max, d.dim_x * d.dim_y);
        }
    }
    const int legal_words = std::max(1, (k_max + 63) / 64);
    // the observation kernels stage the ascending legal-id list in LDS (4 bytes per possible net)
    if ((size_t)legal_words * 64 * 4 + (size_t)(legal_words + 1) * 4 > 60 * 1024)
        return fail(XR_ERR_RANGE, "k_max %d too large for the observation kernel's LDS id list (max ~15000 nets)", k_max);
    std::vector<uint64_t> hlegal0((size_t)n_regions * legal_words, 0);
    for (int r = 0; r < n_regions; r++) {
        hreg[r].legal0_off = (int64_t)r * legal_words;
        const int32_t* csr = hcsr.data() + hreg[r].net_off;
        for (int n = 1; n <= hreg[r].n_nets; n++)
            if (csr[n + 1] > csr[n]) hlegal0[(size_t)r * legal_words + ((n - 1) >> 6)] |= 1ULL << ((n - 1) & 63);
    }

    b->n_regions = n_regions;
    b->n_max_nodes = n_max_nodes;
    b->n_max = (n_max_nodes + 7) & ~7;
    b->n_lds = (n_lds + 7) & ~7;
    b->tracks_max = tracks_max;
    b->lines_max = lines_max;
    b->bits_max = bits_max;
    b->zch = (z_min == z_max && (z_max == 9 || z_max == 12)) ? z_max : 0;
    b->k_max = k_max;
    b->legal_words = legal_words;
    b->x_max = x_max;
    b->y_max = y_max;
    b->all_n_mult4 = mult4;
    b->path_cap = b->cfg.path_cap > 0 ? b->cfg.path_cap : std::min(n_max_nodes, 4096);

    // route kernel placement: distance field in LDS when it fits.  Worklist items are (line, chunk of 8 nodes) pairs:
    // x-tracks * ceil(X/8) + y-tracks * ceil(Y/8) + columns * (1 when every region has 9 / 12 layers, else ceil(Z/8))
    int items_max = 0, kind_max = 0;
    for (int r = 0; r < n_regions; r++) {
        const XrRegionDev& R = hreg[r];
        int nv = 0;
        for (int z = 0; z < R.Z; z++) nv += (R.ldir_mask >> z) & 1u;
        const int chH = (R.X + 7) / 8, chV = (R.Y + 7) / 8, chC = b->zch ? 1 : (R.Z + 7) / 8;
        const int64_t itH = (int64_t)(R.Z - nv) * R.Y * chH, itV = (int64_t)nv * R.X * chV, itC = (int64_t)R.X * R.Y * chC;
        if (itH > 65536 || itV > 65536 || itC > 65536)
            return fail(XR_ERR_RANGE, "region %d: more than 65536 worklist items of one kind (%lld / %lld / %lld)", r,
                        (long long)itH, (long long)itV, (long long)itC);
        items_max = std::max(items_max, (int)(itH + itV + itC));
        kind_max = std::max(kind_max, (int)std::max(itH, std::max(itV, itC)));
    }
    lines_max = items_max;
    b->lines_max = items_max;
    // field + edge-length tables + 3 item bitmasks + worklists (u16 item ids; the claim bitmask aliases them)
    const size_t lw_max = ((size_t)items_max + 31) / 32 + 1;
    const size_t el_bytes = (size_t)(x_max + 2 + y_max + 2) * 4;
    const size_t list_bytes = std::max(((size_t)items_max * 2 + 3) & ~(size_t)3, ((size_t)b->n_lds / 32 + 1) * 4);
    const size_t lds_need = (size_t)b->n_lds * 4 + el_bytes + 3 * lw_max * 4 + list_bytes + 16;
    b->lds_dist = lds_need + kLdsStatic <= kLdsLimit && !b->cfg.force_scratch_field;
    b->route_lds = b->lds_dist ? lds_need : el_bytes + 3 * lw_max * 4;
    const bool sweep_lds_ok = b->lds_dist;
    // workgroup size of the step kernel unless the caller asks: 256 with the field in LDS (4 waves; 4 workgroups per CU
    // resident at 24x40x9), 1024 with the field in HBM scratch (latency-bound on memory: more items in flight per env)
    b->route_threads = b->cfg.block_threads ? b->cfg.block_threads : (b->lds_dist ? 256 : 1024);
    // default router: bucketed frontier expansion with field + three node bitmasks + edge tables in LDS (xr_dial.h);
    // regions that do not fit keep the sweep router's HBM-scratch form
    b->kzch = b->zch;
    {
        const size_t mw_max = (size_t)b->n_max / 32 + 1;
        const size_t dial_lds = (size_t)b->n_max * 4 + 4 * mw_max * 4 + el_bytes + 16;
        // HBM-scratch form (regions too large for LDS, or force_scratch_field): groups of 1024 nodes must fit the LDS group table
        const size_t big_lds = (size_t)XR_BIG_MAXG * 6 + (size_t)XR_BIG_CA * 4 + (size_t)XR_BIG_CN * 4 + 2 * (size_t)XR_BIG_CE * 8 + el_bytes + 16;
        const bool big_ok = ((size_t)b->n_max / 1024 + 2) <= 1024;
        b->dial_big = false;
        if (b->cfg.router != XR_ROUTER_SWEEP && !b->cfg.force_scratch_field && dial_lds + kLdsStatic <= kLdsLimit) {
            b->kzch = -1;
            b->lds_dist = true;
            b->route_lds = dial_lds;
            b->route_threads = b->cfg.block_threads ? b->cfg.block_threads : 256;
        } else if (b->cfg.router != XR_ROUTER_SWEEP && big_ok) {
            b->kzch = -1;
            b->lds_dist = false;
            b->dial_big = true;
            b->route_lds = big_lds;
            // (round 2, config 5: 256 threads 4.2-4.7 ms, 512: 3.7-3.9 ms.  Round 3, same box, ms per launch at 256 / 1024 / 4096 envs: 512 threads
            //  1.9 / 2.7 / 4.0, 1024 threads 1.8 / 2.4 / 6.5 — a wider workgroup shortens each route's rounds (wide frontiers) but only one
            //  fits a CU: it pays while the batch is at most ~4 routes per CU, i.e. while the launch is bound by its longest routes)
            int big_threads = 512;
            {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, b->cfg.device) == hipSuccess && B <= 4 * prop.multiProcessorCount) big_threads = 1024;
            }
            b->route_threads = b->cfg.block_threads ? b->cfg.block_threads : big_threads;
            // the LDS router inside a window of the region first (xr_dial3.h, WIN; xr_config.window: > 0 = that many tracks at most — the
            // largest square window <= it that fits LDS; 0 (default) and < 0 = off: measured no faster on BASELINE config 5):
            // every region must have the same layer count and hold the window, rows of the state arrays must start on 16-byte boundaries
            // wherever a window row may start, the arithmetic limits are those of the form
            b->win = {};
            const bool v2cfg = b->cfg.guide_cost > 0 || b->cfg.maze_end_iter > 1;
            const int64_t pen_w = (int64_t)b->cfg.drc_cost * b->cfg.drc_unit;
            if (b->cfg.window > 0 && !v2cfg && z_min == z_max && edge_max + pen_w < XR3_STEP_LIMIT) {
                const int Zw = z_max;
                int ystep = 1;
                while ((ystep * Zw) % 8) ystep *= 2;                 // rows start at y0 * Z elements: a multiple of 8 of them (16 bytes of int16)
                bool rows_ok = true;
                int xmin = 1 << 30, ymin = 1 << 30;
                for (int r = 0; r < n_regions; r++) {
                    rows_ok = rows_ok && ((int64_t)regs[r].dim_y * Zw) % 8 == 0;
                    xmin = std::min(xmin, (int)regs[r].dim_x); ymin = std::min(ymin, (int)regs[r].dim_y);
                }
                int w = std::min(b->cfg.window, std::min(xmin, ymin));
                for (; w >= 8 && rows_ok; w--) {
                    if ((w * Zw) % 8) continue;                          // a chunk of 8 nodes never straddles two window rows
                    const int64_t nw = (int64_t)w * w * Zw;
                    const size_t wl = XR3_LDS_BYTES((nw + 7) & ~7, w, w);
                    if (nw >= 65536 || std::max(wl, big_lds) + 2 * kLdsStatic > kLdsLimit || (int64_t)w * edge_max >= XR3_EXTENT_LIMIT) continue;
                    uint32_t myz, syz, mz, sz, mmw, smw;
                    const uint32_t mwv = (uint32_t)((nw + 31) / 32);
                    if (!magic24((uint32_t)(w * Zw), (uint32_t)nw, myz, syz) || !magic24((uint32_t)Zw, (uint32_t)std::max(w, 1) * Zw * 2, mz, sz) ||
                        !magic24(mwv, (uint32_t)nw, mmw, smw)) continue;
                    b->win.x = w; b->win.y = w; b->win.nmax = (int)((nw + 7) & ~7); b->win.ystep = ystep;
                    // tracks kept free around the net's box (the row alignment is checked per net).  A net whose box nearly fills the window
                    // floods past its faces, fails its certificate and has paid for the attempt on top of the fallback: margins of 4 / 8 /
                    // 13 / 18 tracks of a 52-track window send 27 / 33 / 48 / 68 % of BASELINE config 5's routes to the fallback
                    // (profiles/r04_l_config5_window_form.txt)
                    b->win.margin = std::max(1, std::min(4, w / 8));
                    if (const char* em = getenv("XR_WINDOW_MARGIN")) b->win.margin = std::max(1, atoi(em));      // (A/B runs)
                    b->win.m24_yz = myz; b->win.m24_z = mz; b->win.m24_mw = mmw; b->win.s24 = syz | (sz << 8) | (smw << 16);
                    b->route_lds = std::max(b->route_lds, wl);
                    break;
                }
            }
        }
        // round 3's LDS form (xr_dial3.h) where it applies: the field fits with its queues, node ids fit 16 bits, and its 27-bit distance
        // arithmetic cannot wrap: every distance that exists is below XR_DIST_CAP = 0x07F00000 (spec; a candidate at or above the cap is
        // never written), so a reached word + one edge with every penalty stays inside 32 bits while that step is < 2^20 (the x32
        // fixed point of the word); the x32 coordinate tables and the heuristic need a region that spans < 2^25 DBU
        {
            const size_t d3_lds = XR3_LDS_BYTES(b->n_max, x_max, y_max);
            const int64_t pen_max = ((int64_t)b->cfg.drc_cost * b->cfg.drc_unit) << (b->cfg.maze_end_iter - 1);
            const bool range_ok = edge_max + pen_max + b->cfg.guide_cost < XR3_STEP_LIMIT && ext_max < XR3_EXTENT_LIMIT &&
                                  (int64_t)b->cfg.via_cost * 32 < XR3_EXTENT_LIMIT;
            if (b->kzch == -1 && b->lds_dist && b->cfg.router != XR_ROUTER_DIAL_R2 && range_ok && b->n_max < 65536 && div24_all &&
                d3_lds + kLdsStatic <= kLdsLimit) {
                b->kzch = -3;
                b->route_lds = d3_lds;
            }
        }
        const bool v2 = b->cfg.guide_cost > 0 || b->cfg.maze_end_iter > 1;
        if (v2 && b->kzch == -3) b->kzch = -4;
        if (v2 && b->kzch == -1) b->kzch = -2;          // the instantiations with the XR-Maze v2 knobs compiled in (LDS and HBM-scratch form)
        if (v2 && b->kzch != -2 && b->kzch != -4)
            return fail(XR_ERR_RANGE, "xr_batch_load_regions: XR-Maze v2 (guide_cost / maze_end_iter) needs the frontier router "
                                      "(router != XR_ROUTER_SWEEP, regions within its limits)");
        if (b->kzch >= 0 && (b->cfg.router == XR_ROUTER_DIAL || b->cfg.router == XR_ROUTER_DIAL_R2)) {
            return fail(XR_ERR_RANGE, "xr_batch_load_regions: XR_ROUTER_DIAL: the largest region (%d nodes) exceeds the frontier router's limits", b->n_max);
        }
    }
    // the fused observation epilogue stages the ascending legal-id list in the same LDS
    const size_t ids_bytes = (size_t)(legal_words * 64 + ((legal_words + 1 + 3) & ~3)) * 4;
    // flat-stream observation (any N): ids + a 16-bit feature per node in LDS
    const size_t stream_bytes = (size_t)(legal_words * 64 + ((legal_words + 1 + 3) & ~3)) * 4 + (size_t)b->n_max * 2;
    b->stream_ok = stream_bytes <= 60 * 1024;
    b->route_lds = std::max(b->route_lds, b->stream_ok && !mult4 ? stream_bytes : ids_bytes);
    // router = 0 (auto) picks per entry point the scheme measured faster for it: the frontier router everywhere, except the
    // FULL-rewrite queue launch of a very large batch, where the line-segment sweeps are ahead (same box, DESIGN.md §5.1:
    // 1.727-1.730 ms against 1.773-1.775 ms per 4096-env step; at 2048 envs the frontier router wins, 0.921 against 0.945-0.954 ms)
    b->sweep_lds = std::max(lds_need, b->stream_ok && !mult4 ? stream_bytes : ids_bytes);
    // (round 3, same box: synthetic 24x40x9 regions 1.681 ms with the sweeps against 1.700 ms with the frontier router; the regions
    //  extracted from ispd18_test1 — unaligned planes, 3.5 pins per net, K up to 77: their step is bound by routing, not by the write
    //  stream — 2.40 ms against 2.05 ms: the sweeps are only chosen for aligned planes)
    b->sweep_full = b->cfg.router == 0 && (b->kzch == -1 || b->kzch == -3) && b->lds_dist && sweep_lds_ok && b->cfg.block_threads == 0 &&
                    b->cfg.n_envs >= 4096 && mult4 && b->sweep_lds + kLdsStatic <= kLdsLimit;
    if (b->route_lds + kLdsStatic > kLdsLimit)
        return fail(XR_ERR_RANGE, "route kernel needs %zu bytes of LDS (line bitmasks of the largest region)", b->route_lds);
    if (std::max(b->route_lds, b->sweep_full ? b->sweep_lds : 0) > 64 * 1024)
        XR_HIP(xr_route_set_max_lds(std::max(b->route_lds, b->sweep_full ? b->sweep_lds : 0)));

    // ---- device allocations ------------------------------------------------------------------
#define XR_ALLOC(buf, count)                                                                        \
    do {                                                                                            \
        hipError_t _e = (buf).alloc(count);                                                         \
        if (_e != hipSuccess)                                                                       \
            return fail(XR_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(count) * sizeof(*(buf).p), \
                        hipGetErrorString(_e));                                                     \
    } while (0)
    XR_ALLOC(b->regions, n_regions);
    XR_ALLOC(b->rg_rec, hrec.size());
    XR_ALLOC(b->rg_node_net, hrec.size());
    XR_ALLOC(b->rg_owner0, hrec.size());
    XR_ALLOC(b->coords, hcoords.size());
    XR_ALLOC(b->net_csr, hcsr.size());
    XR_ALLOC(b->ap_node, std::max<size_t>(1, hap_node.size()));
    XR_ALLOC(b->ap_pin, std::max<size_t>(1, hap_pin.size()));
    XR_ALLOC(b->ap_feat, std::max<size_t>(1, hap_feat.size()));
    XR_ALLOC(b->legal0, hlegal0.size());
    XR_ALLOC(b->env_region, B);
    XR_ALLOC(b->env_replay, B);
    XR_ALLOC(b->nlegal, B);
    XR_ALLOC(b->cum, (size_t)B * 3);
    XR_ALLOC(b->delta, (size_t)B * 3);
    XR_ALLOC(b->status, B);
    XR_ALLOC(b->path, (size_t)B * b->path_cap);
    XR_ALLOC(b->path_len, B);
    XR_ALLOC(b->sweeps, B);
    XR_ALLOC(b->touched, B);
    XR_ALLOC(b->route_order, B);
    XR_ALLOC(b->net_work, hcsr.size());
    XR_ALLOC(b->net_meas, hcsr.size());
    XR_ALLOC(b->net_info, hcsr.size());
    XR_ALLOC(b->ap_flags, std::max<size_t>(1, hap_flags.size()));
    XR_ALLOC(b->owner, (size_t)B * b->n_max);
    XR_ALLOC(b->legal, (size_t)B * legal_words);
    XR_ALLOC(b->hash, B);
    XR_ALLOC(b->reward, B);
    XR_ALLOC(b->records, B);
    XR_ALLOC(b->done, B);
    XR_ALLOC(b->env_steps, B);
    XR_ALLOC(b->total_steps, 1);
    XR_ALLOC(b->phase_cycles, (size_t)B * 8);
    XR_ALLOC(b->plan_region, B);
    XR_ALLOC(b->queue, 8);
    XR_ALLOC(b->plan_units, (size_t)B * std::max(1, k_max));
    XR_ALLOC(b->plan_unit_net, (size_t)B * std::max(1, k_max));
    if (b->dial_big) {
        const size_t mwg = (size_t)b->n_max / 32 + 1;
        XR_ALLOC(b->dg_field, (size_t)B * b->n_max);
        XR_ALLOC(b->dg_masks, (size_t)B * 2 * mwg);
        XR_ALLOC(b->dg_touch, (size_t)B * b->n_max);
        XR_ALLOC(b->dg_path, (size_t)B * b->n_max * 2);
    } else if (!b->lds_dist) {
        XR_ALLOC(b->dist_scratch, (size_t)B * b->n_lds);
        XR_ALLOC(b->cls_scratch, (size_t)B * b->n_lds);
        XR_ALLOC(b->list_scratch, (size_t)B * b->lines_max);
    }
#undef XR_ALLOC

    XR_HIP(hipMemcpyAsync(b->regions.p, hreg.data(), hreg.size() * sizeof(XrRegionDev), hipMemcpyHostToDevice, st));
    XR_HIP(hipMemcpyAsync(b->rg_rec.p, hrec.data(), hrec.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    XR_HIP(hipMemcpyAsync(b->coords.p, hcoords.data(), hcoords.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    XR_HIP(hipMemcpyAsync(b->net_csr.p, hcsr.data(), hcsr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    std::vector<uint8_t> hclass(hcsr.size(), 0);        // work classes 1..255 relative to the batch's largest prediction
    {
        float wmax = 1.0f;
        for (float w : hwork) wmax = std::max(wmax, w);
        for (size_t i = 0; i < hwork.size(); i++)
            if (hwork[i] > 0.0f) hclass[i] = (uint8_t)(1 + std::min(254, (int)(254.0f * hwork[i] / wmax)));
    }
    XR_HIP(hipMemcpyAsync(b->net_work.p, hclass.data(), hclass.size(), hipMemcpyHostToDevice, st));
    XR_HIP(hipMemsetAsync(b->net_meas.p, 0, hcsr.size(), st));                  // nothing measured yet: the launch orders use the geometric guess
    hinfo.resize(hcsr.size(), 0);
    XR_HIP(hipMemcpyAsync(b->net_info.p, hinfo.data(), hinfo.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!hap_flags.empty()) XR_HIP(hipMemcpyAsync(b->ap_flags.p, hap_flags.data(), hap_flags.size(), hipMemcpyHostToDevice, st));
    if (!hap_node.empty()) {
        XR_HIP(hipMemcpyAsync(b->ap_node.p, hap_node.data(), hap_node.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        XR_HIP(hipMemcpyAsync(b->ap_pin.p, hap_pin.data(), hap_pin.size() * sizeof(int16_t), hipMemcpyHostToDevice, st));
        XR_HIP(hipMemcpyAsync(b->ap_feat.p, hap_feat.data(), hap_feat.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    XR_HIP(hipMemcpyAsync(b->legal0.p, hlegal0.data(), hlegal0.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    std::vector<int32_t> henv(B);
    for (int e = 0; e < B; e++) henv[e] = e % n_regions;
    XR_HIP(hipMemcpyAsync(b->env_region.p, henv.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    XR_HIP(hipMemsetAsync(b->env_replay.p, 0, (size_t)B * sizeof(int32_t), st));
    XR_HIP(hipMemsetAsync(b->env_steps.p, 0, (size_t)B * sizeof(int64_t), st));
    XR_HIP(hipMemsetAsync(b->total_steps.p, 0, sizeof(unsigned long long), st));
    XR_HIP(hipMemsetAsync(b->phase_cycles.p, 0, (size_t)B * 8 * sizeof(long long), st));
    XR_HIP(hipMemsetAsync(b->nlegal.p, 0, (size_t)B * sizeof(int32_t), st));
    XR_HIP(hipMemsetAsync(b->touched.p, 0, (size_t)B * sizeof(int32_t), st));
    XR_HIP(hipMemsetAsync(b->route_order.p, 0, (size_t)B * sizeof(int32_t), st));
    XR_HIP(hipMemsetAsync(b->records.p, 0, (size_t)B * sizeof(XrStepRecord), st));
    XR_HIP(hipMemsetAsync(b->owner.p, 0, (size_t)B * b->n_max * sizeof(int16_t), st));
    XR_HIP(hipMemsetAsync(b->path.p, 0, (size_t)B * b->path_cap * sizeof(int32_t), st));
    if (b->dial_big) {      // the persistent CLEAN state of the scratch: field CLEAN, no open bits, word minima = infinity
        const size_t mwg = (size_t)b->n_max / 32 + 1;
        XR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->dg_field.p), (int)0xFFFFFFFEu, (size_t)B * b->n_max, st));
        for (int e = 0; e < B; e++) {
            XR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->dg_masks.p + (size_t)e * 2 * mwg), 0, mwg, st));
            XR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->dg_masks.p + (size_t)e * 2 * mwg + mwg), (int)0xFFFFFFFFu, mwg, st));
        }
    }
    std::vector<uint64_t> hhash(B, 0xcbf29ce484222325ULL);
    XR_HIP(hipMemcpyAsync(b->hash.p, hhash.data(), (size_t)B * sizeof(uint64_t), hipMemcpyHostToDevice, st));

    XR_HIP(hipMemsetAsync(b->queue.p, 0, 8 * sizeof(uint32_t), st));       // both banks start clean; from then on every plan zeroes the other bank
    b->queue_bank = 0; b->queue_last = b->queue.p;
    XR_HIP(xr_launch_ingest(b->rg_rec.p, b->rg_node_net.p, b->rg_owner0.p, (int64_t)hrec.size(), st));
    XR_HIP(hipStreamSynchronize(st));   // host staging vectors die here

    XrBatchDev& d = b->dev;
    d.regions = b->regions.p; d.rg_rec = b->rg_rec.p; d.rg_node_net = b->rg_node_net.p; d.rg_owner0 = b->rg_owner0.p;
    d.coords = b->coords.p; d.net_csr = b->net_csr.p; d.ap_node = b->ap_node.p; d.ap_pin = b->ap_pin.p; d.ap_feat = b->ap_feat.p; d.net_work = b->net_work.p; d.net_info = b->net_info.p;
    {   // measured launch order (round 5): on unless XR_NO_MEASURED_ORDER=1 (A/B switch; results never depend on the order)
        const char* off = getenv("XR_NO_MEASURED_ORDER");
        d.net_meas = (off && off[0] == '1') ? nullptr : b->net_meas.p;
        const char* ht = getenv("XR_HEAVY_CLASS"); const char* hm = getenv("XR_HEAVY_MULT");       // (experiment switches)
        d.heavy_class = std::min(255, std::max(0, ht ? atoi(ht) : 0)); d.heavy_mult = std::min(16, std::max(1, hm ? atoi(hm) : 2));   // (a width of 0 would spin a route to its round cap)
        d.meas_shift = b->lds_dist ? 13 : 15;       // class unit: 8 k cycles (LDS form: a route is 0.1-1.5 M cycles), 32 k (HBM-scratch form: up to 6 M)
    } d.ap_flags = b->ap_flags.p;
    d.legal0 = b->legal0.p; d.n_regions = n_regions;
    d.n_envs = B; d.n_max = b->n_max; d.n_lds = b->n_lds; d.lw_max = (int)lw_max; d.lines_max = lines_max; d.x_max = x_max; d.y_max = y_max; d.legal_words = legal_words; d.path_cap = b->path_cap;
    d.env_region = b->env_region.p; d.env_replay = b->env_replay.p; d.owner = b->owner.p; d.legal = b->legal.p;
    d.nlegal = b->nlegal.p; d.cum = b->cum.p; d.delta = b->delta.p; d.reward = b->reward.p; d.done = b->done.p;
    d.status = b->status.p; d.path = b->path.p; d.path_len = b->path_len.p; d.hash = b->hash.p;
    d.env_steps = b->env_steps.p; d.total_steps = b->total_steps.p; d.sweeps = b->sweeps.p; d.touched = b->touched.p; d.records = b->records.p;
    // (round 3's LDS form, same-box A/B profiles/r03_l_ab_bucket_width.txt: 12 against 8 — route-only 512 / 4096 envs 0.211 / 0.381 ->
    //  0.203 / 0.372 ms, 512-env step 0.315 -> 0.306 ms; 16 the same, 24 slower; round 2's form and the HBM-scratch form keep 8)
    d.dial_mult = b->cfg.dial_mult > 0 ? b->cfg.dial_mult : ((b->kzch == -3 || b->kzch == -4) ? 12 : 8);
    d.dial_mult_big = b->cfg.dial_mult > 0 ? b->cfg.dial_mult : 8;
    d.round_cap = b->cfg.debug_round_cap;
    d.win_x = b->dial_big ? b->win.x : 0; d.win_y = b->win.y; d.win_nmax = b->win.nmax; d.win_margin = b->win.margin; d.win_ystep = b->win.ystep;
    d.win_m24_yz = b->win.m24_yz; d.win_m24_z = b->win.m24_z; d.win_m24_mw = b->win.m24_mw; d.win_s24 = b->win.s24;
    d.guide_cost = b->cfg.guide_cost; d.guide_margin = b->cfg.guide_margin; d.maze_end_iter = b->cfg.maze_end_iter;
    d.dg_field = b->dg_field.p; d.dg_masks = b->dg_masks.p; d.dg_touch = b->dg_touch.p; d.dg_path = b->dg_path.p;
    d.dist_scratch = b->dist_scratch.p; d.cls_scratch = b->cls_scratch.p; d.list_scratch = b->list_scratch.p; d.phase_cycles = b->phase_cycles.p;
    d.obs_out = nullptr; d.obs_stride = 0; d.obs_vec4 = 0; d.obs_head_only = 0; d.obs_split_pm = 1000; d.obs_incremental = 0;
    d.obs_lds_bytes = (int32_t)std::min<size_t>(b->route_lds, 1u << 30);      // (every launch of the default router carries route_lds)
    d.env_base = 0; d.env_count = 0;
    d.plan_region = b->plan_region.p; d.plan_units = b->plan_units.p; d.plan_unit_net = b->plan_unit_net.p; d.queue = b->queue.p; d.queue_quota_pm = 750;
    if (!b->aux_stream) {
        XR_HIP(hipStreamCreateWithFlags(&b->aux_stream, hipStreamNonBlocking));
        XR_HIP(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        XR_HIP(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
        XR_HIP(hipEventCreate(&b->ev_w0));
        XR_HIP(hipEventCreate(&b->ev_w1));
    }
    d.via_cost = b->cfg.via_cost; d.pen_cost = b->cfg.drc_cost * b->cfg.drc_unit;
    d.max_route_count = b->cfg.max_route_count; d.auto_reset = b->cfg.auto_reset;
    d.w_violation = b->cfg.w_violation; d.w_via = b->cfg.w_via; d.w_wirelength = b->cfg.w_wirelength;
    b->h_net_off.resize(n_regions); b->h_n_nets.resize(n_regions); b->h_dims.resize(3 * (size_t)n_regions);
    for (int r = 0; r < n_regions; r++) {
        b->h_net_off[r] = hreg[r].net_off; b->h_n_nets[r] = hreg[r].n_nets;
        b->h_dims[3 * r] = hreg[r].X; b->h_dims[3 * r + 1] = hreg[r].Y; b->h_dims[3 * r + 2] = hreg[r].Z;
    }
    b->h_csr_size = hcsr.size();
    b->loaded = true;
    b->guide_mask_bytes = gmask_bytes;
    d.guide_csr = nullptr; d.guide_box = nullptr;
    return build_guide_masks(b, st);          // (the default guides: bounding box of every net's access points)
}

// XR-Maze v2 with guide_cost > 0 and the round-3 LDS router: the guide of every (region, net) as a static bitmask (xr_guide_mask_kernel) — from the
// boxes loaded last (xr_batch_load_guides) or the default guides.  Without it (no guide cost, another router form, or more than 2 GiB of masks)
// dev.guide_mask stays null and the router decides membership per route as before.
static int32_t build_guide_masks(xr_batch* b, hipStream_t st) {
    b->dev.guide_mask = nullptr;
    const char* off = getenv("XR_NO_GUIDE_MASK");       // (A/B switch: membership per route, round 4's form; results are the same)
    if (b->cfg.guide_cost <= 0 || b->kzch != -4 || b->guide_mask_bytes == 0 || b->guide_mask_bytes > ((size_t)1 << 31) || (off && off[0] == '1')) {
        b->guide_mask.release();
        return XR_OK;
    }
    if (b->guide_mask.n != b->guide_mask_bytes && b->guide_mask.alloc(b->guide_mask_bytes) != hipSuccess) {
        b->guide_mask.release();
        return XR_OK;                    // (no memory for the masks: the per-route form still applies)
    }
    // (a mask build that cannot run degrades like one that has no memory: the router decides membership per route, same results)
    if (xr_launch_guide_masks(&b->dev, b->guide_mask.p, b->k_max, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        b->guide_mask.release();
        return XR_OK;
    }
    b->dev.guide_mask = b->guide_mask.p;
    return XR_OK;
}

// XR-Maze v2, optional: the nets' global-route guides as boxes.  Indexed like net_csr (boxes of net n of region r:
// [guide_csr[R.net_off + n], guide_csr[R.net_off + n + 1]) into guide_box, 6 int16 per box); a net without boxes keeps the default
// guide (bounding box of its access points).  Replaces every guide loaded before; a reload of the regions drops them.
int32_t xr_batch_load_guides(xr_batch* b, const int32_t* const* box_off_host, const int16_t* const* boxes_host, void* stream) {
    if (!b) return fail(XR_ERR_INVALID, "xr_batch_load_guides: null batch");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_load_guides: load regions first");
    XR_HIP(hipSetDevice(b->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (everything is validated and staged BEFORE the batch is touched: a refused table, or one there is no device memory for, leaves
    //  the guides loaded before — boxes and static masks alike — exactly as they were)
    if (!box_off_host) {      // back to the default guides
        b->dev.guide_csr = nullptr; b->dev.guide_box = nullptr;
        b->guide_csr.release(); b->guide_box.release();
        return build_guide_masks(b, st);
    }
    std::vector<int32_t> csr(b->h_csr_size, 0);
    std::vector<int16_t> box;
    for (int r = 0; r < b->n_regions; r++) {
        const int K = b->h_n_nets[r], X = b->h_dims[3 * r], Y = b->h_dims[3 * r + 1], Z = b->h_dims[3 * r + 2];
        int32_t* c = csr.data() + b->h_net_off[r];                 // c[n] .. c[n + 1], n = 1 .. K (slot 0 unused, like net_csr)
        const int32_t* off = box_off_host[r];
        const int16_t* bx = boxes_host ? boxes_host[r] : nullptr;
        for (int n = 1; n <= K; n++) {
            c[n] = (int32_t)(box.size() / 6);
            if (!off) continue;
            const int lo = off[n - 1], hi = off[n];
            if (lo < 0 || hi < lo || hi - lo > XR_GUIDE_MAX_BOXES || (hi > lo && !bx))
                return fail(XR_ERR_RANGE, "xr_batch_load_guides: region %d net %d: %d boxes (0..%d, offsets ascending)", r, n, hi - lo, XR_GUIDE_MAX_BOXES);
            for (int i = lo; i < hi; i++) {
                const int16_t* g = bx + 6 * (size_t)i;     // x0, y0, x1, y1, z0, z1, inclusive
                if (g[0] < 0 || g[2] < g[0] || g[2] >= X || g[1] < 0 || g[3] < g[1] || g[3] >= Y || g[4] < 0 || g[5] < g[4] || g[5] >= Z)
                    return fail(XR_ERR_RANGE, "xr_batch_load_guides: region %d net %d box %d (%d,%d)-(%d,%d) layers %d..%d outside the %dx%dx%d grid",
                                r, n, i - lo, g[0], g[1], g[2], g[3], g[4], g[5], X, Y, Z);
                box.insert(box.end(), g, g + 6);
            }
        }
        c[K + 1] = (int32_t)(box.size() / 6);
    }
    DevBuf<int32_t> new_csr;
    DevBuf<int16_t> new_box;
    if (new_csr.alloc(csr.size()) != hipSuccess || new_box.alloc(std::max<size_t>(6, box.size())) != hipSuccess)
        return fail(XR_ERR_NOMEM, "xr_batch_load_guides: hipMalloc failed (the guides loaded before stay in force)");
    XR_HIP(hipMemcpyAsync(new_csr.p, csr.data(), csr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!box.empty()) XR_HIP(hipMemcpyAsync(new_box.p, box.data(), box.size() * sizeof(int16_t), hipMemcpyHostToDevice, st));
    XR_HIP(hipStreamSynchronize(st));   // host staging vectors die here; work enqueued earlier that reads the old tables has finished too
    std::swap(b->guide_csr.p, new_csr.p); std::swap(b->guide_csr.n, new_csr.n);       // (the old tables are freed when new_* go out of scope)
    std::swap(b->guide_box.p, new_box.p); std::swap(b->guide_box.n, new_box.n);
    b->dev.guide_csr = b->guide_csr.p; b->dev.guide_box = b->guide_box.p;
    return build_guide_masks(b, st);
}

int32_t xr_batch_assign(xr_batch* b, const int32_t* env_region_host) {
    if (!b || !env_region_host) return fail(XR_ERR_INVALID, "xr_batch_assign: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_assign: load regions first");
    for (int e = 0; e < b->cfg.n_envs; e++)
        if (env_region_host[e] < 0 || env_region_host[e] >= b->n_regions)
            return fail(XR_ERR_RANGE, "xr_batch_assign: env %d -> region %d outside 0..%d", e, env_region_host[e],
                        b->n_regions - 1);
    XR_HIP(hipSetDevice(b->cfg.device));
    XR_HIP(hipMemcpy(b->env_region.p, env_region_host, (size_t)b->cfg.n_envs * sizeof(int32_t), hipMemcpyHostToDevice));
    XR_HIP(hipMemset(b->env_replay.p, 0, (size_t)b->cfg.n_envs * sizeof(int32_t)));
    return XR_OK;
}

int32_t xr_batch_sizes(const xr_batch* b, int32_t* n_envs, int32_t* n_regions, int32_t* n_max, int32_t* k_max,
                       int32_t* legal_words, int32_t* path_cap, int64_t* obs_env_stride) {
    if (!b) return fail(XR_ERR_INVALID, "xr_batch_sizes: null batch");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_sizes: load regions first");
    if (n_envs) *n_envs = b->cfg.n_envs;
    if (n_regions) *n_regions = b->n_regions;
    if (n_max) *n_max = b->n_max;
    if (k_max) *k_max = b->k_max;
    if (legal_words) *legal_words = b->legal_words;
    if (path_cap) *path_cap = b->path_cap;
    // (a multiple of 32 floats: every env's row starts on a 128-byte line, which is what lets the writers of unaligned planes
    //  store whole lines — any stride >= (2+7*k_max)*n_max is accepted, this is the one to prefer)
    if (obs_env_stride) *obs_env_stride = ((int64_t)(2 + 7 * (int64_t)b->k_max) * (int64_t)b->n_max + 31) & ~(int64_t)31;
    return XR_OK;
}

int32_t xr_batch_reset(xr_batch* b, const uint8_t* mask_dev, int32_t rotate, void* stream) {
    if (!b) return fail(XR_ERR_INVALID, "xr_batch_reset: null batch");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_reset: load regions first");
    XR_HIP(hipSetDevice(b->cfg.device));
    b->obs_valid_ptr = nullptr;
    XR_HIP(xr_launch_reset(&b->dev, mask_dev, rotate ? 1 : 0, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

namespace {
// One launch over all env slots, or — stream_per_region — one single-workgroup launch per slot on a pool of internal streams
// (fork from / join to the caller's stream with events; the host never waits).
int32_t launch_route_form(xr_batch* b, const XrBatchDev& d, const int32_t* actions_dev, hipStream_t st) {
    if (!b->cfg.stream_per_region) {
        // launch order: longest predicted route first when the launch runs in more than one round of workgroups
        bool lpt = b->cfg.launch_order == 2;
        if (b->cfg.launch_order == 0) {
            if (b->route_slots == 0) {
                hipDeviceProp_t prop;
                XR_HIP(hipGetDeviceProperties(&prop, b->cfg.device));
                int per_cu = 0;
                size_t stat = 0;
                XR_HIP(xr_route_occupancy(b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, &per_cu, &stat));
                b->route_slots = std::max(1, per_cu) * prop.multiProcessorCount;
            }
            lpt = b->cfg.n_envs > b->route_slots;
        }
        if (!lpt) {
            XR_HIP(xr_launch_route(&d, actions_dev, b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, st));
            return XR_OK;
        }
        XrBatchDev dl = d;
        XR_HIP(xr_launch_route_order(&dl, actions_dev, b->route_order.p, st));
        dl.route_order = b->route_order.p;
        XR_HIP(xr_launch_route(&dl, actions_dev, b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, st));
        return XR_OK;
    }
    const int B = b->cfg.n_envs;
    if (b->region_streams.empty()) {
        const int ns = std::min(B, 16);
        for (int i = 0; i < ns; i++) {
            hipStream_t s; hipEvent_t ev;
            XR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            XR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            b->region_streams.push_back(s); b->region_events.push_back(ev);
        }
    }
    const int ns = (int)b->region_streams.size();
    XR_HIP(hipEventRecord(b->ev_fork, st));
    for (int i = 0; i < ns; i++) XR_HIP(hipStreamWaitEvent(b->region_streams[i], b->ev_fork, 0));
    for (int e = 0; e < B; e++) {
        XrBatchDev de = d;
        de.env_base = e; de.env_count = 1;
        XR_HIP(xr_launch_route(&de, actions_dev, b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, b->region_streams[e % ns]));
    }
    for (int i = 0; i < ns; i++) {
        XR_HIP(hipEventRecord(b->region_events[i], b->region_streams[i]));
        XR_HIP(hipStreamWaitEvent(st, b->region_events[i], 0));
    }
    return XR_OK;
}
}  // namespace

int32_t xr_batch_step(xr_batch* b, const int32_t* actions_dev, void* stream) {
    if (!b || !actions_dev) return fail(XR_ERR_INVALID, "xr_batch_step: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_step: load regions first");
    XR_HIP(hipSetDevice(b->cfg.device));
    b->obs_valid_ptr = nullptr;
    return launch_route_form(b, b->dev, actions_dev, static_cast<hipStream_t>(stream));
}

namespace {
int32_t step_observe_impl(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride, void* stream, bool inplace);
}

int32_t xr_batch_step_observe(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride, void* stream) {
    return step_observe_impl(b, actions_dev, out_dev, env_stride, stream, false);
}

int32_t xr_batch_step_observe_inplace(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride, void* stream) {
    return step_observe_impl(b, actions_dev, out_dev, env_stride, stream, true);
}

namespace {
int32_t step_observe_impl(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride, void* stream, bool inplace) {
    if (!b || !actions_dev || !out_dev) return fail(XR_ERR_INVALID, "xr_batch_step_observe: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_step_observe: load regions first");
    if (env_stride < (int64_t)(2 + 7 * (int64_t)b->k_max) * b->n_max_nodes)
        return fail(XR_ERR_RANGE, "xr_batch_step_observe: env_stride %lld < (2+7*k_max)*n_max = %lld", (long long)env_stride,
                    (long long)((2 + 7 * (int64_t)b->k_max) * b->n_max_nodes));
    XR_HIP(hipSetDevice(b->cfg.device));
    XrBatchDev d = b->dev;
    d.obs_out = out_dev;
    d.obs_stride = env_stride;
    // 1: aligned float4 (every N % 4 == 0), 2: shifted float4 (any N), 0: scalar (unaligned caller buffer)
    const bool aligned = (env_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    d.obs_vec4 = aligned ? (b->all_n_mult4 ? 1 : (b->stream_ok ? 2 : 0)) : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool can_split = (d.obs_vec4 == 1 || (d.obs_vec4 == 2 && b->n_max <= 60 * 1024)) && b->cfg.n_envs <= (1 << 18) && b->k_max < (1 << 14) && b->k_max >= 1;
    const bool split = can_split && b->cfg.obs_mode == XR_OBS_SPLIT && !b->cfg.stream_per_region;
    if (can_split && (b->cfg.obs_mode == XR_OBS_QUEUE || b->cfg.obs_mode == 0) && !b->cfg.stream_per_region) {      // the default: measured fastest
        // plan, then one persistent launch: as many workgroups as the chip holds (occupancy x CUs) drain the two queues
        b->last_obs_mode = XR_OBS_QUEUE;
        d.obs_head_only = 1;
        d.obs_split_pm = 1000;
        // in-place form: only when THIS buffer holds the observation of the state before the step (else: a full write)
        d.obs_incremental = (inplace && b->obs_valid_ptr == out_dev && b->obs_valid_stride == env_stride) ? 1 : 0;
        b->last_obs_inplace = d.obs_incremental;
        b->obs_valid_ptr = nullptr;                 // (set again below once every launch of this call has been enqueued without error)
        d.queue_quota_pm = b->cfg.obs_split_permille > 0 ? b->cfg.obs_split_permille : 750;
        {
            // which workgroups start with units instead of a route: bit 5 of the workgroup index.  Bit 0 (rounds 1-2) put every
            // route-first workgroup on the even XCDs (workgroup i runs on XCD i % 8).  Same box, ms per step kernel, bit 0 -> 3 -> 5 ->
            // 8 -> none: 512 envs 0.296 -> 0.291 -> 0.290 -> 0.296 -> 0.302; 1024 envs 0.500 -> 0.489 -> 0.487 -> 0.485 -> 0.525;
            // 4096 envs 1.741 -> 1.736 -> 1.730 -> 1.733 -> 1.797 (profiles/r03_r_ab_queue_unit_first_workgroups.txt)
            static const int skip_env = [] { const char* v = getenv("XR_QUEUE_SKIP_SHIFT"); return v ? atoi(v) : -2; }();      // experiments only
            d.queue_skip_shift = skip_env != -2 ? skip_env : 5;
        }
        if (b->n_cus == 0) {            // once per batch: CUs x resident workgroups per CU of the step kernel (both variants)
            hipDeviceProp_t prop;
            XR_HIP(hipGetDeviceProperties(&prop, b->cfg.device));
            int per_cu = 0;
            size_t stat = 0;
            XR_HIP(xr_route_occupancy(b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, &per_cu, &stat));
            b->n_cus = prop.multiProcessorCount;
            b->queue_blocks = std::max(1, per_cu) * b->n_cus;
            if (b->sweep_full) {
                XR_HIP(xr_route_occupancy(1, b->zch, b->sweep_lds, b->route_threads, &per_cu, &stat));
                b->queue_blocks_sweep = std::max(1, per_cu) * b->n_cus;
            }
        }
        // this call's counters / the other bank (zeroed by this call's plan for the next one: no memset per call)
        d.queue = b->queue.p + 4 * b->queue_bank;
        uint32_t* const next_queue = b->queue.p + 4 * (b->queue_bank ^ 1);
        b->queue_bank ^= 1;
        b->queue_last = d.queue;
        // Route tasks: slot order for large batches (longest-first measured SLOWER there, full rewrite 1.72 -> 1.83 ms, in-place
        // 1.06 -> 1.09 ms — the launch is bound by its write stream, and long routes up front delay the first units).  A batch of at
        // most ~2 routes per resident workgroup is different: its launch ends with the longest route, and a workgroup that starts
        // with units takes its route ~125 us late (tools/queue_timeline_probe.py, 1024 envs: the last route ends at 100 % of the
        // launch) — there the tasks are handed out longest predicted route first (xr_route_order_kernel, ~10 us).
        // Measured (profiles/r03_o_ab_queue_task_order.txt, same box, ms per step, slot order -> longest first): synthetic regions 512 /
        // 1024 / 2048 / 4096 envs 0.324 / 0.518 / 0.939 / 1.749 -> 0.318 / 0.513 / 0.945 / 1.849; the regions extracted from ispd18_test1
        // (heavier, more varied routes: their step is bound by routing) 1024 / 2048 / 4096 / 8192 slots 1.097 / 1.391 / 2.071 / 3.556 ->
        // 0.931 / 1.023 / 1.951 / 3.829.  Hence: up to 2 routes per resident workgroup, 4 for regions with unaligned planes.
        const int lpt_limit = (b->all_n_mult4 ? 2 : 4) * b->queue_blocks;
        const bool lpt_tasks = b->cfg.launch_order == 2 ||
                               (b->cfg.launch_order == 0 && b->queue_blocks > 0 && b->cfg.n_envs <= lpt_limit && b->cfg.n_envs > 64);
        int order_done = 0;       // (a batch of <= 1024 envs: plan and order are ONE launch; every dependent dispatch costs a small batch ~8 us)
        XR_HIP(xr_launch_plan(&d, actions_dev, next_queue, lpt_tasks ? b->route_order.p : nullptr, &order_done, st));
        if (lpt_tasks) {
            if (!order_done) XR_HIP(xr_launch_route_order(&d, actions_dev, b->route_order.p, st));
            d.route_order = b->route_order.p;
        }
        // which router runs the route tasks of this launch (auto: sweeps for the full rewrite of a large batch, see load)
        const bool use_sweep = b->sweep_full && !d.obs_incremental;
        b->last_obs_sweeps = use_sweep ? 1 : 0;
        const int kz = use_sweep ? b->zch : b->kzch;
        const size_t klds = use_sweep ? b->sweep_lds : b->route_lds;
        d.obs_lds_bytes = (int32_t)std::min<size_t>(klds, 1u << 30);
        const int blocks = b->cfg.obs_writer_blocks > 0 ? b->cfg.obs_writer_blocks : (use_sweep ? b->queue_blocks_sweep : b->queue_blocks);
        // helper writers (aligned planes only): LDS-free workgroups on the internal stream draining the same unit queue;
        // forked after the plan, joined before the call returns the stream (events, no host wait)
        const int helpers = d.obs_vec4 == 1 && b->cfg.obs_helper_blocks > 0 ? b->cfg.obs_helper_blocks : 0;
        const bool use_helpers = helpers > 0 && b->cfg.n_envs >= 64;
        if (use_helpers) {
            XR_HIP(hipEventRecord(b->ev_fork, st));
            XR_HIP(hipStreamWaitEvent(b->aux_stream, b->ev_fork, 0));
        }
        d.queue_grid = std::min(blocks, 4 * b->cfg.n_envs);
        XR_HIP(xr_launch_step_queue(&d, actions_dev, b->lds_dist ? 1 : 0, kz, klds, b->route_threads, d.queue_grid, st));
        if (use_helpers) {
            XR_HIP(xr_launch_unit_helpers(&d, helpers, b->aux_stream));
            XR_HIP(hipEventRecord(b->ev_join, b->aux_stream));
            XR_HIP(hipStreamWaitEvent(st, b->ev_join, 0));
        }
        b->obs_valid_ptr = out_dev; b->obs_valid_stride = env_stride;
        return XR_OK;
    }
    b->last_obs_mode = split ? XR_OBS_SPLIT : XR_OBS_FUSED;
    b->last_obs_inplace = 0;                                  // the fused and split forms always write the whole observation
    b->obs_valid_ptr = nullptr;
    if (!split) {
        const int32_t rc = launch_route_form(b, d, actions_dev, st);
        if (rc == XR_OK) { b->obs_valid_ptr = out_dev; b->obs_valid_stride = env_stride; }
        return rc;
    }
    // plan (caller's stream) -> fork: net-plane writer on the internal stream || route kernel (+ planes 0..1) on the
    // caller's stream -> join.  Everything is ordered by events; the host never waits.
    d.obs_head_only = 1;
    d.obs_split_pm = b->cfg.obs_split_permille > 0 ? b->cfg.obs_split_permille : 1000;
    d.queue = b->queue.p + 4 * b->queue_bank;
    XR_HIP(xr_launch_plan(&d, actions_dev, b->queue.p + 4 * (b->queue_bank ^ 1), nullptr, nullptr, st));
    b->queue_bank ^= 1;
    b->queue_last = d.queue;
    XR_HIP(hipEventRecord(b->ev_fork, st));
    XR_HIP(hipStreamWaitEvent(b->aux_stream, b->ev_fork, 0));
    XR_HIP(xr_launch_route(&d, actions_dev, b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, st));
    XR_HIP(hipEventRecord(b->ev_w0, b->aux_stream));
    XR_HIP(xr_launch_netplanes(&d, b->cfg.obs_writer_blocks > 0 ? b->cfg.obs_writer_blocks : 512, d.obs_vec4 == 1 ? 1 : 0,
                               b->aux_stream));
    XR_HIP(hipEventRecord(b->ev_w1, b->aux_stream));
    XR_HIP(hipStreamWaitEvent(st, b->ev_w1, 0));
    b->obs_valid_ptr = out_dev; b->obs_valid_stride = env_stride;
    return XR_OK;
}
}  // namespace

int32_t xr_batch_step_compact(xr_batch* b, const int32_t* actions_dev, float* head_out_dev, int64_t head_stride, void* stream) {
    if (!b || !actions_dev || !head_out_dev) return fail(XR_ERR_INVALID, "xr_batch_step_compact: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_step_compact: load regions first");
    if (head_stride < (int64_t)2 * b->n_max_nodes)
        return fail(XR_ERR_RANGE, "xr_batch_step_compact: head_stride %lld < 2*n_max = %lld", (long long)head_stride,
                    (long long)2 * b->n_max_nodes);
    XR_HIP(hipSetDevice(b->cfg.device));
    XrBatchDev d = b->dev;
    d.obs_out = head_out_dev;
    d.obs_stride = head_stride;
    const bool aligned = (head_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(head_out_dev) & 15) == 0);
    d.obs_vec4 = aligned ? (b->all_n_mult4 ? 1 : (b->stream_ok ? 2 : 0)) : 0;
    if (d.obs_vec4 == 0)        // (the scalar epilogue has no head-only form)
        return fail(XR_ERR_INVALID, "xr_batch_step_compact: head_out_dev must be 16-byte aligned and head_stride a multiple of 4");
    b->obs_valid_ptr = nullptr;
    d.obs_head_only = 1;          // the epilogue writes planes 0..1 and the net planes of the lowest XR_SPLIT_KEEP ranks: none
    d.obs_split_pm = 1000;
    b->last_obs_mode = XR_OBS_FUSED;
    return launch_route_form(b, d, actions_dev, static_cast<hipStream_t>(stream));
}

int32_t xr_batch_net_planes(xr_batch* b, const int32_t* pair_region_dev, const int32_t* pair_net_dev, int32_t n_pairs,
                            float* out_dev, int64_t pair_stride, void* stream) {
    if (!b || !out_dev || (n_pairs > 0 && (!pair_region_dev || !pair_net_dev))) return fail(XR_ERR_INVALID, "xr_batch_net_planes: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_net_planes: load regions first");
    if (n_pairs < 0 || pair_stride < (int64_t)7 * b->n_max_nodes)
        return fail(XR_ERR_RANGE, "xr_batch_net_planes: pair_stride %lld < 7*n_max = %lld", (long long)pair_stride,
                    (long long)7 * b->n_max_nodes);
    XR_HIP(hipSetDevice(b->cfg.device));
    const bool aligned = (pair_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    XR_HIP(xr_launch_netplanes_pairs(&b->dev, pair_region_dev, pair_net_dev, n_pairs, out_dev, pair_stride, aligned ? 1 : 0,
                                     static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_state_row_bytes(const xr_batch* b, int64_t* row_bytes) {
    if (!b || !row_bytes) return fail(XR_ERR_INVALID, "xr_batch_state_row_bytes: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_state_row_bytes: load regions first");
    *row_bytes = 16 + 8 * ((int64_t)b->legal_words + ((b->n_max + 63) >> 6));
    return XR_OK;
}

int32_t xr_batch_pack_state(xr_batch* b, uint8_t* rows_dev, int64_t row_bytes, int32_t region_base, void* stream) {
    if (!b || !rows_dev) return fail(XR_ERR_INVALID, "xr_batch_pack_state: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_pack_state: load regions first");
    const int64_t need = 16 + 8 * ((int64_t)b->legal_words + ((b->n_max + 63) >> 6));
    if (row_bytes < need || row_bytes % 8 != 0 || (reinterpret_cast<uintptr_t>(rows_dev) & 7) != 0 || region_base < 0)
        return fail(XR_ERR_RANGE, "xr_batch_pack_state: row_bytes %lld (need >= %lld, a multiple of 8, 8-byte aligned rows), region_base %d",
                    (long long)row_bytes, (long long)need, region_base);
    XR_HIP(hipSetDevice(b->cfg.device));
    XR_HIP(xr_launch_pack_state(&b->dev, rows_dev, row_bytes, region_base, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_expand_state(xr_batch* b, const uint8_t* rows_dev, int64_t row_bytes, int32_t n_rows, float* head_out_dev, int64_t head_stride,
                              int32_t* nlegal_out_dev, int32_t* region_out_dev, void* stream) {
    if (!b || (n_rows > 0 && (!rows_dev || !head_out_dev || !nlegal_out_dev || !region_out_dev)))
        return fail(XR_ERR_INVALID, "xr_batch_expand_state: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_expand_state: load regions first");
    if (n_rows < 0 || row_bytes < 16 || row_bytes % 8 != 0 || (reinterpret_cast<uintptr_t>(rows_dev) & 7) != 0 || head_stride < (int64_t)2 * b->n_max_nodes)
        return fail(XR_ERR_RANGE, "xr_batch_expand_state: n_rows %d, row_bytes %lld (a multiple of 8, 8-byte aligned rows), head_stride %lld < 2*n_max = %lld",
                    n_rows, (long long)row_bytes, (long long)head_stride, (long long)2 * b->n_max_nodes);
    XR_HIP(hipSetDevice(b->cfg.device));
    const bool aligned = (head_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(head_out_dev) & 15) == 0);
    XR_HIP(xr_launch_expand_state(&b->dev, rows_dev, row_bytes, n_rows, head_out_dev, head_stride, nlegal_out_dev, region_out_dev, aligned ? 1 : 0,
                                  static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_ingest_state(xr_batch* b, const int16_t* owner_dev, const uint64_t* legal_dev, const int32_t* cum_dev, void* stream) {
    if (!b || !owner_dev || !legal_dev || !cum_dev) return fail(XR_ERR_INVALID, "xr_batch_ingest_state: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_ingest_state: load regions first");
    if ((reinterpret_cast<uintptr_t>(owner_dev) & 15) != 0 || (reinterpret_cast<uintptr_t>(legal_dev) & 7) != 0 || (reinterpret_cast<uintptr_t>(cum_dev) & 3) != 0)
        return fail(XR_ERR_INVALID, "xr_batch_ingest_state: owner rows must be 16-byte aligned, legal words 8-byte, metrics 4-byte");
    XR_HIP(hipSetDevice(b->cfg.device));
    b->obs_valid_ptr = nullptr;                       // whatever observation a caller holds no longer describes the batch
    XR_HIP(xr_launch_ingest_state(&b->dev, owner_dev, legal_dev, cum_dev, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_net_vectors(xr_batch* b, const int32_t* pair_region_dev, const int32_t* pair_net_dev, int32_t n_pairs, int32_t D, int32_t H, int32_t W,
                             const float* weights_dev, const float* bg_dev, float* out_dev, int32_t* flags_dev, int32_t normalize, void* stream) {
    if (!b || !weights_dev || !bg_dev || (n_pairs > 0 && (!pair_region_dev || !pair_net_dev || !out_dev || !flags_dev)))
        return fail(XR_ERR_INVALID, "xr_batch_net_vectors: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_net_vectors: load regions first");
    if (n_pairs < 0 || D < 1 || H < 1 || W < 1) return fail(XR_ERR_RANGE, "xr_batch_net_vectors: n_pairs %d, grid %dx%dx%d", n_pairs, D, H, W);
    XR_HIP(hipSetDevice(b->cfg.device));
    int32_t status = XR_OK;
    XR_HIP(xr_launch_net_tower(b->regions.p, b->net_csr.p, b->ap_feat.p, b->n_regions, pair_region_dev, pair_net_dev, n_pairs, D, H, W, weights_dev, bg_dev, out_dev,
                               flags_dev, normalize, static_cast<hipStream_t>(stream), &status));
    if (status != XR_OK) return fail(status, "xr_batch_net_vectors: the fused net tower does not take a %dx%dx%d grid (the caller keeps the framework path)", D, H, W);
    return XR_OK;
}

int32_t xr_batch_route_occupancy(xr_batch* b, int32_t* workgroups_per_cu, int64_t* lds_bytes_per_workgroup) {
    if (!b || !workgroups_per_cu || !lds_bytes_per_workgroup) return fail(XR_ERR_INVALID, "xr_batch_route_occupancy: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_route_occupancy: load regions first");
    XR_HIP(hipSetDevice(b->cfg.device));
    int n = 0;
    size_t stat = 0;
    XR_HIP(xr_route_occupancy(b->lds_dist ? 1 : 0, b->kzch, b->route_lds, b->route_threads, &n, &stat));
    *workgroups_per_cu = n;
    *lds_bytes_per_workgroup = (int64_t)(b->route_lds + stat);
    return XR_OK;
}

int32_t xr_batch_observe_timing(xr_batch* b, int32_t* mode_out, float* writer_ms) {
    if (!b || !mode_out || !writer_ms) return fail(XR_ERR_INVALID, "xr_batch_observe_timing: null argument");
    *mode_out = b->last_obs_mode + (b->last_obs_mode == XR_OBS_QUEUE && b->last_obs_inplace ? 16 : 0)     // bit 4: the in-place form ran
                + (b->last_obs_mode == XR_OBS_QUEUE && b->last_obs_sweeps ? 32 : 0);                // bit 5: auto router took the sweeps
    *writer_ms = 0.f;
    if (b->last_obs_mode == XR_OBS_SPLIT) {
        XR_HIP(hipSetDevice(b->cfg.device));
        XR_HIP(hipEventSynchronize(b->ev_w1));
        XR_HIP(hipEventElapsedTime(writer_ms, b->ev_w0, b->ev_w1));
    }
    return XR_OK;
}

int32_t xr_batch_route_order(xr_batch* b, const int32_t* orders_dev, int32_t stride, int32_t* net_stats_dev, void* stream) {
    if (!b || !orders_dev) return fail(XR_ERR_INVALID, "xr_batch_route_order: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_route_order: load regions first");
    if (stride < b->k_max)
        return fail(XR_ERR_RANGE, "xr_batch_route_order: stride %d < k_max %d", stride, b->k_max);
    XR_HIP(hipSetDevice(b->cfg.device));
    b->obs_valid_ptr = nullptr;
    XR_HIP(xr_launch_order(&b->dev, orders_dev, stride, net_stats_dev, b->lds_dist ? 1 : 0, b->kzch, b->route_lds,
                           b->route_threads, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_random_actions(xr_batch* b, int32_t* actions_dev, uint64_t seed, void* stream) {
    if (!b || !actions_dev) return fail(XR_ERR_INVALID, "xr_batch_random_actions: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_random_actions: load regions first");
    XR_HIP(hipSetDevice(b->cfg.device));
    XR_HIP(xr_launch_random_actions(&b->dev, actions_dev, seed, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_observation(xr_batch* b, float* out_dev, int64_t env_stride, int32_t env_lo, int32_t env_hi,
                             void* stream) {
    if (!b || !out_dev) return fail(XR_ERR_INVALID, "xr_batch_observation: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_observation: load regions first");
    if (env_lo < 0 || env_hi > b->cfg.n_envs || env_lo > env_hi)
        return fail(XR_ERR_RANGE, "xr_batch_observation: env range [%d,%d) outside [0,%d)", env_lo, env_hi, b->cfg.n_envs);
    if (env_stride < (int64_t)2 * b->n_max_nodes)
        return fail(XR_ERR_RANGE, "xr_batch_observation: env_stride %lld too small", (long long)env_stride);
    XR_HIP(hipSetDevice(b->cfg.device));
    const bool aligned = (env_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    const int vec4 = aligned ? (b->all_n_mult4 ? 1 : (b->stream_ok ? 2 : 0)) : 0;
    XR_HIP(xr_launch_obs(&b->dev, out_dev, env_stride, env_lo, env_hi, b->n_max_nodes, vec4,
                         static_cast<hipStream_t>(stream)));
    if (env_lo == 0 && env_hi == b->cfg.n_envs && env_stride >= (int64_t)(2 + 7 * (int64_t)b->k_max) * b->n_max_nodes) {
        b->obs_valid_ptr = out_dev; b->obs_valid_stride = env_stride;      // this buffer now holds every slot's observation
    }
    return XR_OK;
}

int32_t xr_batch_fetch(xr_batch* b, int32_t what, void* dst_dev, size_t dst_bytes, void* stream) {
    if (!b || !dst_dev) return fail(XR_ERR_INVALID, "xr_batch_fetch: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_fetch: load regions first");
    const size_t B = (size_t)b->cfg.n_envs;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (what) {
    case XR_FETCH_CUM: src = b->cum.p; bytes = B * 3 * sizeof(int32_t); break;
    case XR_FETCH_DELTA: src = b->delta.p; bytes = B * 3 * sizeof(int32_t); break;
    case XR_FETCH_REWARD: src = b->reward.p; bytes = B * sizeof(double); break;
    case XR_FETCH_DONE: src = b->done.p; bytes = B; break;
    case XR_FETCH_NLEGAL: src = b->nlegal.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_STATUS: src = b->status.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_LEGAL: src = b->legal.p; bytes = B * b->legal_words * sizeof(uint64_t); break;
    case XR_FETCH_PATH_LEN: src = b->path_len.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_PATH: src = b->path.p; bytes = B * b->path_cap * sizeof(int32_t); break;
    case XR_FETCH_OWNER: src = b->owner.p; bytes = B * b->n_max * sizeof(int16_t); break;
    case XR_FETCH_HASH: src = b->hash.p; bytes = B * sizeof(uint64_t); break;
    case XR_FETCH_REGION: src = b->env_region.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_STEPS: src = b->total_steps.p; bytes = sizeof(int64_t); break;
    case XR_FETCH_SWEEPS: src = b->sweeps.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_UNITS: src = (b->queue_last ? b->queue_last : b->queue.p) + 2; bytes = sizeof(uint32_t); break;
    case XR_FETCH_ROUTE_ORDER: src = b->route_order.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_TOUCHED: src = b->touched.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_RECORD: src = b->records.p; bytes = B * sizeof(XrStepRecord); break;
    case XR_FETCH_PHASES: src = b->phase_cycles.p; bytes = B * 8 * sizeof(long long); break;
    case XR_FETCH_REPLAY: src = b->env_replay.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_ENV_STEPS: src = b->env_steps.p; bytes = B * sizeof(int64_t); break;
    default: return fail(XR_ERR_INVALID, "xr_batch_fetch: unknown selector %d", what);
    }
    if (dst_bytes < bytes)
        return fail(XR_ERR_RANGE, "xr_batch_fetch(%d): destination holds %zu bytes, need %zu", what, dst_bytes, bytes);
    XR_HIP(hipSetDevice(b->cfg.device));
    // (hipMemcpyDefault: the destination may be a device buffer or PINNED host memory — one copy straight to the host for
    // the small-batch path)
    XR_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

int32_t xr_batch_store(xr_batch* b, int32_t what, const void* src_dev, size_t src_bytes, void* stream) {
    if (!b || !src_dev) return fail(XR_ERR_INVALID, "xr_batch_store: null argument");
    if (!b->loaded) return fail(XR_ERR_STATE, "xr_batch_store: load regions first");
    const size_t B = (size_t)b->cfg.n_envs;
    void* dst = nullptr;
    size_t bytes = 0;
    switch (what) {
    case XR_FETCH_CUM: dst = b->cum.p; bytes = B * 3 * sizeof(int32_t); break;
    case XR_FETCH_DELTA: dst = b->delta.p; bytes = B * 3 * sizeof(int32_t); break;
    case XR_FETCH_REWARD: dst = b->reward.p; bytes = B * sizeof(double); break;
    case XR_FETCH_DONE: dst = b->done.p; bytes = B; break;
    case XR_FETCH_NLEGAL: dst = b->nlegal.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_STATUS: dst = b->status.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_LEGAL: dst = b->legal.p; bytes = B * b->legal_words * sizeof(uint64_t); break;
    case XR_FETCH_PATH_LEN: dst = b->path_len.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_OWNER: dst = b->owner.p; bytes = B * b->n_max * sizeof(int16_t); break;
    case XR_FETCH_HASH: dst = b->hash.p; bytes = B * sizeof(uint64_t); break;
    case XR_FETCH_REGION: dst = b->env_region.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_REPLAY: dst = b->env_replay.p; bytes = B * sizeof(int32_t); break;
    case XR_FETCH_ENV_STEPS: dst = b->env_steps.p; bytes = B * sizeof(int64_t); break;
    case XR_FETCH_RECORD: dst = b->records.p; bytes = B * sizeof(XrStepRecord); break;
    case XR_FETCH_STEPS: dst = b->total_steps.p; bytes = sizeof(int64_t); break;
    default: return fail(XR_ERR_INVALID, "xr_batch_store: selector %d is not part of the env state", what);
    }
    if (src_bytes != bytes)
        return fail(XR_ERR_RANGE, "xr_batch_store(%d): source holds %zu bytes, the array has %zu", what, src_bytes, bytes);
    XR_HIP(hipSetDevice(b->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // The arrays the kernels INDEX with are validated on the host before they reach the device (a restore is not a hot path: one small
    // staging copy + a stream sync): a region index outside the loaded regions, or legal bits / a nets-left count beyond the region's
    // nets, would make the next plan / route kernel read regions[], net_csr[] and net_work[] out of bounds — a device fault, not XR_ERR_RANGE.
    // Restore order for a consistent check: XR_FETCH_REGION before XR_FETCH_LEGAL / XR_FETCH_NLEGAL (RegionBatch.load_state_dict does).
    if (what == XR_FETCH_REGION || what == XR_FETCH_LEGAL || what == XR_FETCH_NLEGAL) {
        std::vector<unsigned char> host(bytes);
        std::vector<int32_t> reg(B);
        XR_HIP(hipMemcpyAsync(host.data(), src_dev, bytes, hipMemcpyDefault, st));
        if (what != XR_FETCH_REGION) XR_HIP(hipMemcpyAsync(reg.data(), b->env_region.p, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        XR_HIP(hipStreamSynchronize(st));
        if (what == XR_FETCH_REGION) {
            const int32_t* r = reinterpret_cast<const int32_t*>(host.data());
            for (size_t e = 0; e < B; e++)
                if (r[e] < 0 || r[e] >= b->n_regions)
                    return fail(XR_ERR_RANGE, "xr_batch_store(region): env %zu -> region %d outside 0..%d", e, r[e], b->n_regions - 1);
        } else if (what == XR_FETCH_NLEGAL) {
            const int32_t* nl = reinterpret_cast<const int32_t*>(host.data());
            for (size_t e = 0; e < B; e++)
                if (nl[e] < 0 || nl[e] > b->h_n_nets[reg[e]])
                    return fail(XR_ERR_RANGE, "xr_batch_store(nlegal): env %zu: %d nets left, its region %d has %d", e, nl[e], reg[e], b->h_n_nets[reg[e]]);
        } else {
            const uint64_t* lg = reinterpret_cast<const uint64_t*>(host.data());
            for (size_t e = 0; e < B; e++) {
                const int K = b->h_n_nets[reg[e]];
                for (int w = 0; w < b->legal_words; w++) {
                    const int lo = w * 64;
                    const uint64_t allowed = K >= lo + 64 ? ~0ULL : (K > lo ? ((1ULL << (K - lo)) - 1ULL) : 0ULL);
                    if (lg[e * b->legal_words + w] & ~allowed)
                        return fail(XR_ERR_RANGE, "xr_batch_store(legal): env %zu: net bits beyond the %d nets of its region %d", e, K, reg[e]);
                }
            }
        }
    }
    b->obs_valid_ptr = nullptr;                       // whatever observation a caller holds no longer describes the batch
    XR_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, st));
    return XR_OK;
}

int32_t xr_observation_from_records(const uint32_t* nodes_dev, int32_t X, int32_t Y, int32_t Z, const int32_t* nets_dev,
                                    int32_t k, float* out_dev, void* stream) {
    if (!nodes_dev || !out_dev || (k > 0 && !nets_dev)) return fail(XR_ERR_INVALID, "xr_observation_from_records: null argument");
    if (X < 1 || Y < 1 || Z < 1 || (int64_t)X * Y * Z > ((int64_t)1 << 30) || k < 0 || k > 15000)
        return fail(XR_ERR_RANGE, "xr_observation_from_records: dims %dx%dx%d / k %d out of range", X, Y, Z, k);
    const int N = X * Y * Z;
    const bool vec4 = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    XR_HIP(xr_launch_obs_records(nodes_dev, X, Y, Z, nets_dev, k, out_dev, vec4 ? 1 : 0, static_cast<hipStream_t>(stream)));
    return XR_OK;
}

}  // extern "C"
