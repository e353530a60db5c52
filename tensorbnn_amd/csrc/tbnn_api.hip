// libtbnn: C-ABI host driver (include/tbnn.h) for the gfx950 HMC kernels.
// One handle = one chain = one device + one stream; a whole transition is
// enqueued without a host round-trip and one small record is read back.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels_generic.hpp"
#include "kernels_hmc.hpp"
#include "kernels_hyper.hpp"
#include "kernels_fast.hpp"
#include "kernels_fast3.hpp"
#include "kernels_traj.hpp"
#include "kernels_layered.hpp"
#include "wide_api.hpp"
#include "mid_api.hpp"
#include "tall_api.hpp"
#include "fused_ops.hpp"
#include <mutex>

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)
#define NEED(h)                                                                                \
    do { if (!(h)) return fail(-1, "null handle"); } while (0)
// entry points that address ONE chain's state have no meaning on a multi-chain handle
#define ONE_CHAIN(h, what)                                                                     \
    do { if ((h)->C > 1) return fail(-1, what ": not available on a multi-chain handle (tbnn_create_multi)"); } while (0)


struct tbnn_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    int device = 0;
    float* gbuf = nullptr; size_t gbuf_floats = 0;     // library-owned gather buffer
};

struct tbnn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    NetDev nd{};
    uint32_t key0 = 0, key1 = 0, epoch = 0;
    int C = 1;                            // chains of this handle (tbnn_create_multi): every per-chain buffer is a [C][...] array, the per-chain
    uint32_t seed_hi = 0;                 // kernels run with gridDim.y = C; chain c draws from the Philox key (seed, chain_id + c)
    int kernel = TBNN_KERNEL_GENERIC;     // resolved variant
    int fast_id = -1;
    int fast_ver = 1;                     // 1: kernels_fast.hpp, 3: kernels_fast3.hpp (fringe units off the 16x16 tiles)
    int mid_id = -1;                      // >= 0: kernels_mid.hpp (mid-width fused kernel; narrow-family workspace and launch signature)
    bool lay = false; LayPlan lplan{}; float* lstore = nullptr;   // kernels_layered.hpp: run-time-shape MFMA kernels, activations through HBM
    float* lfwd = nullptr; size_t lfwd_floats = 0;                // its forward-only store (predict / metrics / ensembles): pooled, grown as needed
    std::string kernel_name;
    // data
    float* dX = nullptr; float* dY = nullptr; bool own_data = false; long n = 0;
    // validation data (network.py:47-51) and the prediction buffer of tbnn_predict / tbnn_metrics
    float* dXv = nullptr; float* dYv = nullptr; long nv = 0;
    float* fbuf = nullptr; size_t fbuf_floats = 0; double* mpart = nullptr;
    // chain state
    float *q_cur = nullptr, *g_cur = nullptr, *q = nullptr, *p = nullptr, *g = nullptr, *eta = nullptr;
    float *p0_inj = nullptr, *logu_inj = nullptr, *tmp = nullptr;
    float *gd = nullptr, *gd_cur = nullptr, *eta_prev = nullptr;   // data-term gradient of the proposal / current state; eta before a hyper step
    bool cur_valid = false;               // (logp, grad, stat) cached at q_cur for the current eta/data
    bool q_img_valid = false;             // h->qimg mirrors h->q (maintained by k_update)
    // fused-pass workspace
    int grid = 0, pitch = 0; float* slabs = nullptr; double* pstat = nullptr; float* scratch = nullptr;
    // wide-layer path (kernels_wide.hpp): a fast-kernel variant with its own workspace
    const FusedOps* jit = nullptr;         // run-time registered kernel library covering this shape (tbnn_register_kernel_lib)
    int wide_id = -1; WidePlan wplan; float* wstore = nullptr; float* wslabA = nullptr; float* wslabB = nullptr;
    int nslab = 0;                        // gradient slabs k_update reduces (wide: 1, already reduced)
    // row-sharded chain (tbnn_set_row_shard): all-reduce of the dense data-term gradient row + statistic
    tbnn_comm* shard = nullptr; long n_total = 0; float* grow = nullptr; double* pstat_red = nullptr;
    double* shard_buf = nullptr;          // the all-reduce operand: P gradient values + the statistic, as doubles
    int* imgmap = nullptr; float* qimg = nullptr; float* qimg_cur = nullptr; int img_floats = 0;   // fast kernel: padded weight images
    size_t scratchPerWG = 0;
    float* pin_dev = nullptr;              // device-side address of pin
    float* pin = nullptr;                  // pinned staging for the per-epoch state read-back (P + H floats): a 22-KB D2H copy
                                           // into pageable memory costs ~180 us, through pinned memory ~15 us
    Scal* sc = nullptr; Scal* sc_host = nullptr; Scal* sc_out = nullptr;   // sc_out: device copy for host
    double* trace = nullptr; int trace_cap = 0;
    Scal* d_recs = nullptr; Scal* h_recs = nullptr; int recs_cap = 0;     // tbnn_hmc_run: per-epoch records (pooled)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int profile = 0; long launch_no = 0;   // profile: event pair around every profile-th fwd+bwd launch
    std::vector<hipEvent_t> pev; size_t pev_used = 0;   // pooled events: created once, re-used after every drain (no allocator in the timed loop)
    std::vector<int> pev_div;                           // leapfrog steps the pair's launch covered (1: a fused pass; L: a trajectory launch)
    const char* last_path = "none";                     // which kernels ran the last transition's leapfrog steps (tbnn_last_transition_path)
    // hyper workspace
    float* hyp_ws = nullptr;
    // per-chain step control (tbnn_hmc_step_each / tbnn_hyper_step_each): [C] on the device, staged through pinned memory
    StepCtl* ctl = nullptr; StepCtl* ctl_host = nullptr; float* epsh = nullptr; float* epsh_host = nullptr;
    bool merge_ends = true;                // TBNN_MERGE_ENDS (read at tbnn_create): decision + record + commit in one k_energy launch
    bool traj = true;                      // TBNN_TRAJ (read at tbnn_create; 0: off): whole trajectories of small problems in one launch (kernels_traj.hpp)
};

extern "C" const char* tbnn_last_error(void) { return g_err.c_str(); }
extern "C" int tbnn_abi_version(void) { return TBNN_ABI_VERSION; }
// hash of the sources this library was built from (tensorbnn_amd/build.py: kernel headers, translation units, include/tbnn.h,
// compiler flags): what ties a committed rocprofv3 summary under profiles/ to the library bench.py has loaded
#ifndef TBNN_BUILD_ID
#define TBNN_BUILD_ID "unknown"
#endif
extern "C" const char* tbnn_build_id(void) { return TBNN_BUILD_ID; }
extern "C" int tbnn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(-2, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

static int build_netdev(const tbnn_net_desc* d, NetDev& nd) {
    if (!d || !d->layers) return fail(-1, "null descriptor");
    if (d->n_layers < 1 || d->n_layers > TBNN_MAX_LAYERS) return fail(-1, "n_layers must be in [1,16]");
    memset(&nd, 0, sizeof(nd));
    nd.nl = d->n_layers;
    int off = 0, so = 0, mw = 0;
    for (int l = 0; l < nd.nl; ++l) {
        const tbnn_layer_desc& L = d->layers[l];
        if (L.in_dim < 1 || L.out_dim < 1) return fail(-1, "layer dims must be >= 1");
        if (l > 0 && L.in_dim != d->layers[l - 1].out_dim) return fail(-1, "layer dims do not chain");
        if (L.act < TBNN_ACT_NONE || L.act > TBNN_ACT_ELU) return fail(-1, "unknown activation");
        if (L.prior != TBNN_PRIOR_CAUCHY && L.prior != TBNN_PRIOR_GAUSSIAN) return fail(-1, "unknown prior");
        nd.in[l] = L.in_dim; nd.out[l] = L.out_dim; nd.act[l] = L.act; nd.prior[l] = L.prior;
        nd.offW[l] = off; off += L.in_dim * L.out_dim;
        nd.offB[l] = off; off += L.out_dim;
        nd.actOff[l] = so; so += L.out_dim;
        mw = std::max(mw, std::max(L.in_dim, L.out_dim));
    }
    if (d->likelihood < TBNN_LIK_GAUSSIAN || d->likelihood > TBNN_LIK_BERNOULLI) return fail(-1, "unknown likelihood");
    nd.P = off;
    nd.lik = d->likelihood;
    nd.H = 4 * nd.nl + (nd.lik == TBNN_LIK_GAUSSIAN ? 1 : 0);
    nd.d_in = nd.in[0]; nd.d_out = nd.out[nd.nl - 1];
    nd.sumOut = so; nd.maxW = mw;
    nd.fixed_sd = d->fixed_sd;
    if (nd.lik == TBNN_LIK_FIXED_GAUSSIAN && !(d->fixed_sd > 0.f)) return fail(-1, "fixed_sd must be > 0");
    return 0;
}

// ---- kernel libraries compiled at run time for shapes outside the ahead-of-time registries ----
static std::mutex g_jit_mu;
static std::vector<FusedOps*> g_jit;              // never freed: handles keep pointers into it
static const FusedOps* find_jit(const NetDev& nd) {
    std::lock_guard<std::mutex> lk(g_jit_mu);
    for (const FusedOps* o : g_jit) if (fused_ops_match(*o, nd)) return o;
    return nullptr;
}
// a FusedOps table for this network: a run-time registered library first, then the ahead-of-time tall-fan-in instantiations
static const FusedOps* find_ops(const NetDev& nd) {
    // TBNN_REGISTERED=0 (the layered family's tests): tbnn_create ignores kernel libraries registered earlier in the process
    const bool jit_on = !(getenv("TBNN_REGISTERED") && atoi(getenv("TBNN_REGISTERED")) == 0);
    const FusedOps* o = jit_on ? find_jit(nd) : nullptr;
    if (o) return o;
    // TBNN_TALL=0 (diagnostic / A-B runs, the layered family's tests): shapes the tall-fan-in registry covers take the layered path
    const bool tall_on = !(getenv("TBNN_TALL") && atoi(getenv("TBNN_TALL")) == 0);
    return tall_on ? tall_find(nd) : nullptr;
}
extern "C" int tbnn_register_kernel_lib(const char* path) {
    if (!path) return fail(-1, "null path");
    void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(-6, std::string("dlopen: ") + dlerror());
    typedef int (*ops_fn)(FusedOps*);
    ops_fn fn = (ops_fn)dlsym(lib, "tbnn_jit_ops");
    if (!fn) { dlclose(lib); return fail(-6, std::string(path) + " does not export tbnn_jit_ops"); }
    FusedOps* o = new (std::nothrow) FusedOps();
    if (!o) return fail(-4, "out of host memory");
    memset(o, 0, sizeof(*o));
    if (fn(o) != 0 || o->abi != TBNN_JIT_ABI || (o->family != TBNN_FAMILY_NARROW && o->family != TBNN_FAMILY_WIDE)) {
        delete o; dlclose(lib);
        return fail(-6, std::string(path) + ": kernel table rejected (ABI mismatch?)");
    }
    std::lock_guard<std::mutex> lk(g_jit_mu);
    g_jit.push_back(o);
    return 0;
}
// 0: generic kernel only; 1: ahead-of-time narrow MFMA kernel; 2: ahead-of-time wide path; 3: registered library
extern "C" int tbnn_fused_kernel_available(const tbnn_net_desc* desc) {
    NetDev nd;
    int rc = build_netdev(desc, nd);
    if (rc) return rc;
    if (fast_lookup(nd) >= 0 || mid_lookup(nd) >= 0) return 1;
    if (wide_lookup(nd) >= 0) return 2;
    if (const FusedOps* o = find_ops(nd)) return o == find_jit(nd) ? 3 : 1;
    return 0;
}

static void default_eta(const NetDev& nd, std::vector<float>& eta) {
    eta.assign(nd.H, 0.f);
    for (int l = 0; l < nd.nl; ++l) {
        if (nd.prior[l] == TBNN_PRIOR_CAUCHY) {          // layer.py:136-158
            eta[4 * l + 1] = sqrtf(0.5f); eta[4 * l + 3] = sqrtf(0.5f);
        } else {                                         // layer.py:316-339
            eta[4 * l + 1] = 1.f; eta[4 * l + 3] = 1.f;
        }
    }
    if (nd.lik == TBNN_LIK_GAUSSIAN) eta[nd.H - 1] = sqrtf(0.1f);   // GaussianLikelihood(sd=0.1)
}

extern "C" int tbnn_destroy(tbnn_handle h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->own_data) { hipFree(h->dX); hipFree(h->dY); }
    float* bufs[] = {h->q_cur, h->g_cur, h->q, h->p, h->g, h->eta, h->p0_inj, h->logu_inj, h->tmp, h->gd, h->gd_cur, h->eta_prev,
                     h->slabs, h->scratch, h->hyp_ws, h->wstore, h->wslabA, h->wslabB, h->grow, h->dXv, h->dYv, h->fbuf};
    if (h->mpart) hipFree(h->mpart);
    for (float* b : bufs) if (b) hipFree(b);
    if (h->pstat) hipFree(h->pstat);
    if (h->pstat_red) hipFree(h->pstat_red);
    if (h->shard_buf) hipFree(h->shard_buf);
    if (h->imgmap) hipFree(h->imgmap);
    if (h->qimg) hipFree(h->qimg);
    if (h->qimg_cur) hipFree(h->qimg_cur);
    if (h->sc) hipFree(h->sc);
    if (h->sc_out) hipFree(h->sc_out);
    if (h->lstore) hipFree(h->lstore);
    if (h->lfwd) hipFree(h->lfwd);
    if (h->trace) hipFree(h->trace);
    if (h->d_recs) hipFree(h->d_recs);
    if (h->h_recs) hipHostFree(h->h_recs);
    if (h->sc_host) hipHostFree(h->sc_host);
    if (h->ctl) hipFree(h->ctl);
    if (h->ctl_host) hipHostFree(h->ctl_host);
    if (h->epsh) hipFree(h->epsh);
    if (h->epsh_host) hipHostFree(h->epsh_host);
    if (h->pin) hipHostFree(h->pin);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (auto e : h->pev) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

static int create_impl(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id, int n_chains, tbnn_handle* out);
extern "C" int tbnn_create(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id, tbnn_handle* out) {
    return create_impl(desc, device, seed, chain_id, 1, out);
}
// Several independent chains of one network on ONE device behind one handle (SURVEY 8(e) puts one chain on each GPU; small
// problems -- the reference's own examples -- leave most of a GPU idle and are bound by launch latency): chain c is exactly the chain
// tbnn_create(..., chain_id + c) would be, and the per-chain kernels of all of them run as ONE launch with gridDim.y = n_chains.
extern "C" int tbnn_create_multi(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id, int32_t n_chains, tbnn_handle* out) {
    if (n_chains < 1 || n_chains > 1024) return fail(-1, "n_chains must be in [1, 1024]");
    {   // element counts of the [chains][P] arrays are ints in the copy / commit kernels: refuse what would not fit
        NetDev nd;
        int rc = build_netdev(desc, nd);
        if (rc) return rc;
        if ((long long)n_chains * (long long)nd.P > 0x7FFFFFFFLL) return fail(-1, "n_chains x parameter count exceeds 2^31 - 1");
    }
    return create_impl(desc, device, seed, chain_id, n_chains, out);
}
extern "C" int tbnn_chain_count(tbnn_handle h) { NEED(h); return h->C; }
static int create_impl(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id, int n_chains, tbnn_handle* out) {
    if (!out) return fail(-1, "null out");
    *out = nullptr;
    NetDev nd;
    int rc = build_netdev(desc, nd);
    if (rc) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(-3, "no HIP device visible: libtbnn has no CPU fallback (the HMC path runs on gfx950 only)");
    if (device < 0 || device >= ndev) return fail(-1, "device index out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(-3, std::string("device is ") + prop.gcnArchName + ", libtbnn is built for gfx950 only");
    HIPCHK(hipSetDevice(device));
    tbnn_ctx* h = new (std::nothrow) tbnn_ctx();
    if (!h) return fail(-4, "out of host memory");
    h->device = device; h->nd = nd; h->C = n_chains;
    // Philox key = (seed, chain_id); the high seed word is folded into the key
    h->key0 = (uint32_t)(seed & 0xFFFFFFFFull);
    h->seed_hi = (uint32_t)(seed >> 32);
    h->key1 = chain_id ^ h->seed_hi;
    const size_t NC = (size_t)n_chains;
    auto bail = [&](int code, const std::string& m) { tbnn_destroy(h); return fail(code, m); };
#define HIPB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(-2, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
    HIPB(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t PB = NC * (size_t)nd.P * sizeof(float);               // per-chain arrays: [C][P] (one chain: [P])
    HIPB(hipMalloc(&h->q_cur, PB)); HIPB(hipMalloc(&h->g_cur, PB)); HIPB(hipMalloc(&h->q, PB));
    HIPB(hipMalloc(&h->p, PB)); HIPB(hipMalloc(&h->g, PB));
    HIPB(hipMalloc(&h->p0_inj, (size_t)std::max(nd.P, nd.H) * sizeof(float)));   // injected momentum of either transition (H > P for tiny networks)
    HIPB(hipMalloc(&h->tmp, PB + NC * (size_t)nd.H * sizeof(float)));
    HIPB(hipMalloc(&h->eta, NC * (size_t)nd.H * sizeof(float)));
    HIPB(hipMalloc(&h->eta_prev, NC * (size_t)nd.H * sizeof(float)));
    HIPB(hipMalloc(&h->gd, PB)); HIPB(hipMalloc(&h->gd_cur, PB));
    HIPB(hipMalloc(&h->logu_inj, sizeof(float)));
    HIPB(hipMalloc(&h->sc, NC * sizeof(Scal))); HIPB(hipMalloc(&h->sc_out, NC * sizeof(Scal)));
    HIPB(hipHostMalloc(&h->sc_host, NC * sizeof(Scal)));
    HIPB(hipHostMalloc(&h->pin, NC * ((size_t)nd.P + nd.H) * sizeof(float), hipHostMallocMapped));
    HIPB(hipHostGetDevicePointer((void**)&h->pin_dev, h->pin, 0));
    HIPB(hipMemset(h->sc, 0, NC * sizeof(Scal)));
    HIPB(hipMemset(h->q_cur, 0, PB));
    HIPB(hipEventCreate(&h->ev0)); HIPB(hipEventCreate(&h->ev1));
    HIPB(hipMalloc(&h->hyp_ws, NC * hyper_ws_bytes(nd)));
    HIPB(hipMalloc(&h->ctl, NC * sizeof(StepCtl))); HIPB(hipHostMalloc(&h->ctl_host, NC * sizeof(StepCtl)));
    HIPB(hipMalloc(&h->epsh, NC * sizeof(float))); HIPB(hipHostMalloc(&h->epsh_host, NC * sizeof(float)));
    std::vector<float> eta1, eta; default_eta(nd, eta1);
    for (size_t c = 0; c < NC; ++c) eta.insert(eta.end(), eta1.begin(), eta1.end());
    HIPB(hipMemcpy(h->eta, eta.data(), eta.size() * sizeof(float), hipMemcpyHostToDevice));
    // fused-kernel variant
    h->kernel = TBNN_KERNEL_GENERIC; h->kernel_name = "generic";
    const int want = desc->kernel;
    const int fid = fast_lookup(nd);
    // TBNN_MID=0 (diagnostic / A-B runs): shapes both families cover take the wide path (two kernels through HBM)
    const bool mid_on = !(getenv("TBNN_MID") && atoi(getenv("TBNN_MID")) == 0);
    const int mid = (fid < 0 && mid_on) ? mid_lookup(nd) : -1;
    const int wid = (fid < 0 && mid < 0) ? wide_lookup(nd) : -1;
    const FusedOps* jo = (fid < 0 && mid < 0 && wid < 0) ? find_ops(nd) : nullptr;
    if (want == TBNN_KERNEL_FAST && fid < 0 && mid < 0 && wid < 0 && !jo) return bail(-1, "TBNN_KERNEL_FAST requested but no specialised kernel covers this shape");
    if ((want == TBNN_KERNEL_AUTO || want == TBNN_KERNEL_FAST) && mid >= 0) {
        h->kernel = TBNN_KERNEL_FAST; h->mid_id = mid; h->kernel_name = mid_name(mid);
        h->img_floats = mid_image_floats(mid);
        std::vector<int> map(2 * (size_t)nd.P);
        mid_image_map_id(mid, map.data());
        HIPB(hipMalloc(&h->imgmap, map.size() * sizeof(int)));
        HIPB(hipMemcpy(h->imgmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPB(hipMalloc(&h->qimg, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMalloc(&h->qimg_cur, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg, 0, NC * (size_t)h->img_floats * sizeof(float)));       // padding stays zero for ever
        HIPB(hipMemset(h->qimg_cur, 0, NC * (size_t)h->img_floats * sizeof(float)));
    }
    if ((want == TBNN_KERNEL_AUTO || want == TBNN_KERNEL_FAST) && jo) {
        h->kernel = TBNN_KERNEL_FAST; h->jit = jo; h->kernel_name = jo->name;
        if (jo->family == TBNN_FAMILY_WIDE) { h->wide_id = 1000; jo->plan(16, &h->wplan); }
        h->img_floats = jo->img_floats;
        std::vector<int> map(2 * (size_t)nd.P);
        jo->image_map(map.data());
        HIPB(hipMalloc(&h->imgmap, map.size() * sizeof(int)));
        HIPB(hipMemcpy(h->imgmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPB(hipMalloc(&h->qimg, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMalloc(&h->qimg_cur, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg, 0, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg_cur, 0, NC * (size_t)h->img_floats * sizeof(float)));
    }
    if ((want == TBNN_KERNEL_AUTO || want == TBNN_KERNEL_FAST) && wid >= 0) {
        h->kernel = TBNN_KERNEL_FAST; h->wide_id = wid; h->kernel_name = wide_name(wid);
        wide_plan(wid, 16, h->wplan);
        h->img_floats = h->wplan.img_floats;
        std::vector<int> map(2 * (size_t)nd.P);
        wide_image_map_id(wid, map.data());
        HIPB(hipMalloc(&h->imgmap, map.size() * sizeof(int)));
        HIPB(hipMemcpy(h->imgmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPB(hipMalloc(&h->qimg, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMalloc(&h->qimg_cur, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg, 0, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg_cur, 0, NC * (size_t)h->img_floats * sizeof(float)));
    }
    if ((want == TBNN_KERNEL_AUTO || want == TBNN_KERNEL_FAST) && fid >= 0) {
        h->kernel = TBNN_KERNEL_FAST; h->fast_id = fid; h->kernel_name = fast_name(fid);
        {   // TBNN_FAST_VER=1|3 selects the variant (default: 3 where available)
            const char* ve = getenv("TBNN_FAST_VER");
            int want = ve ? atoi(ve) : 3;
            if (want != 3 || !fast3_available(fid)) want = 1;
            h->fast_ver = want;
        }
        if (h->fast_ver != 1) h->kernel_name = std::string("fast") + char('0' + h->fast_ver) + (h->kernel_name.c_str() + 4);
        h->img_floats = fast_image_floats(fid);
        std::vector<int> map(2 * (size_t)nd.P);
        fast_image_map(fid, map.data());
        HIPB(hipMalloc(&h->imgmap, map.size() * sizeof(int)));
        HIPB(hipMemcpy(h->imgmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPB(hipMalloc(&h->qimg, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMalloc(&h->qimg_cur, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg, 0, NC * (size_t)h->img_floats * sizeof(float)));       // padding stays zero for ever
        HIPB(hipMemset(h->qimg_cur, 0, NC * (size_t)h->img_floats * sizeof(float)));
    }
    // no shape-specialised kernel (and none registered at run time): the layered MFMA family takes any architecture
    // (TBNN_LAYERED=0: the thread-per-row kernel, as before round 3)
    const bool lay_on = !(getenv("TBNN_LAYERED") && atoi(getenv("TBNN_LAYERED")) == 0);
    if (want == TBNN_KERNEL_AUTO && fid < 0 && mid < 0 && wid < 0 && !jo && lay_on) {
        h->kernel = TBNN_KERNEL_FAST; h->lay = true;
        lay_plan_shape(nd, h->lplan);
        h->kernel_name = "layered<" + std::to_string(nd.in[0]);
        for (int l = 0; l < nd.nl; ++l) h->kernel_name += "," + std::to_string(nd.out[l]);
        h->kernel_name += ">";
        h->img_floats = h->lplan.img_floats;
        std::vector<int> map(2 * (size_t)nd.P);
        lay_image_map(nd, h->lplan, map.data());
        HIPB(hipMalloc(&h->imgmap, map.size() * sizeof(int)));
        HIPB(hipMemcpy(h->imgmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPB(hipMalloc(&h->qimg, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMalloc(&h->qimg_cur, NC * (size_t)h->img_floats * sizeof(float)));
        HIPB(hipMemset(h->qimg, 0, NC * (size_t)h->img_floats * sizeof(float)));       // padding stays zero for ever
        HIPB(hipMemset(h->qimg_cur, 0, NC * (size_t)h->img_floats * sizeof(float)));
    }
    { const char* e1 = getenv("TBNN_FAST_SINGLE"); if (e1 && atoi(e1)) h->nd.reserved_flags |= 1; }
    { const char* e2 = getenv("TBNN_MERGE_ENDS"); h->merge_ends = !(e2 && atoi(e2) == 0); }
    { const char* e3 = getenv("TBNN_TRAJ"); h->traj = !(e3 && atoi(e3) == 0); }
    const char* env = getenv("TBNN_PROFILE_FWDBWD");
    h->profile = env ? atoi(env) : 0;
#undef HIPB
    *out = h;
    return 0;
}

extern "C" int tbnn_param_count(tbnn_handle h) { NEED(h); return h->nd.P; }
extern "C" int tbnn_hyper_count(tbnn_handle h) { NEED(h); return h->nd.H; }
extern "C" const char* tbnn_kernel_name(tbnn_handle h) { return h ? h->kernel_name.c_str() : ""; }
extern "C" const char* tbnn_last_transition_path(tbnn_handle h) { return h ? h->last_path : ""; }
extern "C" int tbnn_set_profiling(tbnn_handle h, int stride) {
    NEED(h);
    h->profile = stride > 0 ? stride : 0;
    if (h->profile) {                       // fill the event pool now, outside any timed loop
        HIPCHK(hipSetDevice(h->device));
        // 256 pairs: more than any tbnn_hmc_run of bench.py profiles between two drains
        while (h->pev.size() < 512) { hipEvent_t e = nullptr; HIPCHK(hipEventCreate(&e)); h->pev.push_back(e); }
    }
    return 0;
}
extern "C" int tbnn_set_epoch(tbnn_handle h, uint32_t epoch) { NEED(h); h->epoch = epoch; return 0; }

// (re)allocate the fused-pass workspace for n rows
static int alloc_workspace(tbnn_ctx* h, long n) {
    const NetDev& nd = h->nd;
    if (h->slabs) { hipFree(h->slabs); h->slabs = nullptr; }
    if (h->pstat) { hipFree(h->pstat); h->pstat = nullptr; }
    if (h->scratch) { hipFree(h->scratch); h->scratch = nullptr; }
    for (float** b : {&h->wstore, &h->wslabA, &h->wslabB}) if (*b) { hipFree(*b); *b = nullptr; }
    int grid;
    if (h->lstore) { hipFree(h->lstore); h->lstore = nullptr; }
    if (h->lay) {
        lay_plan_rows(nd, n, h->lplan);
        grid = h->lplan.NP;                       // entries of the statistic buffer in use; the gradient slabs: lplan.NS
        h->scratchPerWG = 0;
        HIPCHK(hipMalloc(&h->lstore, (size_t)h->lplan.store_floats * sizeof(float)));
        HIPCHK(hipMemsetAsync(h->lstore, 0, (size_t)h->lplan.store_floats * sizeof(float), h->stream));     // dz padding: written once, here
        // a_0 = the rows in block form, once per data set
        const long tot = h->lplan.ntiles * h->lplan.TK[0];
        hipLaunchKernelGGL(k_lay_pack_x, dim3((int)std::min<long>(tot, 4096)), dim3(256), 0, h->stream, (const float*)h->dX, n, nd.d_in, h->lplan.TK[0],
                           h->lplan.ntiles, h->lstore + h->lplan.aOff[0]);
        HIPCHK(hipGetLastError());
    } else if (h->wide_id >= 0) {
        if (h->jit) h->jit->plan(n, &h->wplan); else wide_plan(h->wide_id, n, h->wplan);
        grid = h->wplan.gridA;
        h->scratchPerWG = 0;
        HIPCHK(hipMalloc(&h->wstore, h->wplan.store_floats * sizeof(float)));
        HIPCHK(hipMalloc(&h->wslabA, h->wplan.slabA_floats * sizeof(float)));
        HIPCHK(hipMalloc(&h->wslabB, h->wplan.slabB_floats * sizeof(float)));
    } else if (h->kernel == TBNN_KERNEL_FAST) {
        grid = h->jit ? h->jit->grid(n) : (h->mid_id >= 0 ? mid_grid_id(h->mid_id, n) : fast_grid(h->fast_id, n));
        // test hook: a smaller grid puts small row counts into the many-rounds + cooperative-tail regime of the big ones
        if (const char* ge = getenv("TBNN_FAST_GRID")) { const int gg = atoi(ge); if (gg >= 1 && gg < grid) grid = gg; }
        h->scratchPerWG = 0;
    } else {
        const long nblk = (n + GEN_RB - 1) / GEN_RB;
        grid = (int)std::min<long>(nblk, 512);
        h->scratchPerWG = generic_scratch_floats(nd);
        HIPCHK(hipMalloc(&h->scratch, h->scratchPerWG * sizeof(float) * (size_t)grid));
    }
    // tbnn_forward always uses the generic forward kernel: keep a scratch for it
    h->grid = grid;
    h->pitch = (nd.P + 3) & ~3;                  // float4-readable slabs
    h->nslab = h->wide_id >= 0 ? 1 : (h->lay ? h->lplan.NS : grid);
    HIPCHK(hipMalloc(&h->slabs, (size_t)h->C * h->nslab * h->pitch * sizeof(float)));          // [C][nslab][pitch]
    HIPCHK(hipMemset(h->slabs, 0, (size_t)h->C * h->nslab * h->pitch * sizeof(float)));
    if (grid > PSTAT_CAP) return fail(-2, "grid exceeds PSTAT_CAP");
    HIPCHK(hipMalloc(&h->pstat, (size_t)h->C * PSTAT_CAP * sizeof(double)));                   // [C][PSTAT_CAP]
    HIPCHK(hipMemset(h->pstat, 0, (size_t)h->C * PSTAT_CAP * sizeof(double)));      // entries >= grid stay zero
    if (h->pstat_red) { hipFree(h->pstat_red); h->pstat_red = nullptr; }
    HIPCHK(hipMalloc(&h->pstat_red, (size_t)PSTAT_CAP * sizeof(double)));
    HIPCHK(hipMemset(h->pstat_red, 0, (size_t)PSTAT_CAP * sizeof(double)));
    if (h->grow) { hipFree(h->grow); h->grow = nullptr; }
    HIPCHK(hipMalloc(&h->grow, (size_t)h->pitch * sizeof(float)));
    if (h->shard_buf) { hipFree(h->shard_buf); h->shard_buf = nullptr; }
    HIPCHK(hipMalloc(&h->shard_buf, ((size_t)nd.P + 1) * sizeof(double)));
    return 0;
}

extern "C" int tbnn_set_data_device(tbnn_handle h, const float* dX, const float* dY, int64_t n) {
    NEED(h);
    if (!dX || !dY || n < 1) return fail(-1, "set_data: null pointer or n < 1");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_data) { hipFree(h->dX); hipFree(h->dY); h->own_data = false; }
    h->dX = const_cast<float*>(dX); h->dY = const_cast<float*>(dY); h->n = (long)n;
    h->cur_valid = false;
    return alloc_workspace(h, (long)n);
}

extern "C" int tbnn_set_data(tbnn_handle h, const float* X, const float* Y, int64_t n) {
    NEED(h);
    if (!X || !Y || n < 1) return fail(-1, "set_data: null pointer or n < 1");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_data) { hipFree(h->dX); hipFree(h->dY); h->own_data = false; h->dX = h->dY = nullptr; }
    float *dX = nullptr, *dY = nullptr;
    HIPCHK(hipMalloc(&dX, (size_t)n * h->nd.d_in * sizeof(float)));
    HIPCHK(hipMalloc(&dY, (size_t)n * h->nd.d_out * sizeof(float)));
    HIPCHK(hipMemcpy(dX, X, (size_t)n * h->nd.d_in * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dY, Y, (size_t)n * h->nd.d_out * sizeof(float), hipMemcpyHostToDevice));
    h->dX = dX; h->dY = dY; h->own_data = true; h->n = (long)n;
    h->cur_valid = false;
    return alloc_workspace(h, (long)n);
}

extern "C" int tbnn_set_state(tbnn_handle h, const float* theta) {
    NEED(h); if (!theta) return fail(-1, "null theta");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->q_cur, theta, (size_t)h->C * h->nd.P * sizeof(float), hipMemcpyHostToDevice, h->stream));     // [C][P]
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cur_valid = false;
    return 0;
}
extern "C" int tbnn_get_state(tbnn_handle h, float* theta) {
    NEED(h); if (!theta) return fail(-1, "null theta");
    HIPCHK(hipSetDevice(h->device));
    // a kernel writes theta straight into the (device-mapped) pinned buffer: a D2H copy engine transfer of this size
    // has ~160 us of latency, the zero-copy store a few us
    const int PC = h->C * h->nd.P;
    hipLaunchKernelGGL(k_copy_f32, dim3((PC + 255) / 256), dim3(256), 0, h->stream, PC, (const float*)h->q_cur, h->pin_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(theta, h->pin, (size_t)PC * sizeof(float));
    return 0;
}
extern "C" int tbnn_set_hypers(tbnn_handle h, const float* eta) {
    NEED(h); if (!eta) return fail(-1, "null eta");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->eta, eta, (size_t)h->C * h->nd.H * sizeof(float), hipMemcpyHostToDevice, h->stream));        // [C][H]
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cur_valid = false;
    return 0;
}
extern "C" int tbnn_get_hypers(tbnn_handle h, float* eta) {
    NEED(h); if (!eta) return fail(-1, "null eta");
    HIPCHK(hipSetDevice(h->device));
    float* stage = h->pin + (size_t)h->C * h->nd.P;
    HIPCHK(hipMemcpyAsync(stage, h->eta, (size_t)h->C * h->nd.H * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(eta, stage, (size_t)h->C * h->nd.H * sizeof(float));
    return 0;
}

// ---- RCCL, resolved at run time (no link-time dependency: single-GPU users never load it) ----
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return 0;
    // TBNN_RCCL_LIB: explicit collective library (tests run the N > 1 paths on a 1-GPU box against tests/stubccl).
    // Otherwise an RCCL the process has already loaded (torch ships its own copy) is re-used before a second copy
    // is brought in: two RCCL instances in one process would each run their own bootstrap / proxy threads.
    void* lib = nullptr;
    if (const char* ov = getenv("TBNN_RCCL_LIB")) {
        lib = dlopen(ov, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(-5, std::string("TBNN_RCCL_LIB=") + ov + ": " + dlerror());
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(-5, std::string("cannot load librccl.so: ") + dlerror());
    RcclApi a; a.lib = lib;
#define RSYM(field, name) do { *(void**)(&a.field) = dlsym(lib, name); if (!a.field) return fail(-5, std::string("librccl.so lacks ") + name); } while (0)
    RSYM(GetUniqueId, "ncclGetUniqueId"); RSYM(CommInitRank, "ncclCommInitRank"); RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(CommCount, "ncclCommCount"); RSYM(AllReduce, "ncclAllReduce"); RSYM(AllGather, "ncclAllGather"); RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    g_rccl = a;
    return 0;
}
#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) return fail(-5, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == TBNN_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" int tbnn_comm_unique_id(unsigned char id[TBNN_COMM_ID_BYTES]) {
    if (!id) return fail(-1, "null id");
    int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, TBNN_COMM_ID_BYTES);
    return 0;
}
extern "C" int tbnn_comm_create(tbnn_handle h, int world, int rank, const unsigned char id[TBNN_COMM_ID_BYTES],
                                tbnn_comm_handle* out) {
    NEED(h);
    if (!out || !id) return fail(-1, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(-1, "bad world/rank");
    int rc = rccl_load(); if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId u; memcpy(&u, id, TBNN_COMM_ID_BYTES);
    tbnn_comm* c = new (std::nothrow) tbnn_comm();
    if (!c) return fail(-4, "out of host memory");
    c->world = world; c->rank = rank; c->device = h->device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(-5, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    *out = c;
    return 0;
}
// ranks in the communicator AS THE COLLECTIVE LIBRARY SEES IT (ncclCommCount): what bench.py reports per rank at N > 1
extern "C" int tbnn_comm_count(tbnn_comm_handle c) {
    if (!c) return fail(-1, "null communicator");
    int n = 0;
    NCCLCHK(g_rccl.CommCount(c->comm, &n));
    return n;
}
extern "C" int tbnn_comm_destroy(tbnn_comm_handle c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->gbuf) hipFree(c->gbuf);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}
extern "C" int tbnn_gather_samples(tbnn_handle h, tbnn_comm_handle c, float* d_out, float* host_out) {
    NEED(h); if (!c) return fail(-1, "null communicator");
    ONE_CHAIN(h, "tbnn_gather_samples");
    HIPCHK(hipSetDevice(h->device));
    const size_t per = (size_t)h->nd.P + h->nd.H, tot = per * c->world;
    if (!d_out) {
        if (c->gbuf_floats < tot) {
            if (c->gbuf) hipFree(c->gbuf);
            c->gbuf = nullptr; c->gbuf_floats = 0;
            HIPCHK(hipMalloc(&c->gbuf, tot * sizeof(float))); c->gbuf_floats = tot;
        }
        d_out = c->gbuf;
    }
    // (theta, eta) staged contiguously in tmp (P+H floats)
    HIPCHK(hipMemcpyAsync(h->tmp, h->q_cur, (size_t)h->nd.P * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->tmp + h->nd.P, h->eta, (size_t)h->nd.H * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    NCCLCHK(g_rccl.AllGather(h->tmp, d_out, per, ncclFloat, c->comm, h->stream));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, d_out, tot * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int tbnn_set_row_shard(tbnn_handle h, tbnn_comm_handle c, int64_t n_total) {
    NEED(h);
    ONE_CHAIN(h, "tbnn_set_row_shard");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (c && n_total < 1) return fail(-1, "n_total must be >= 1");
    h->shard = c; h->n_total = c ? (long)n_total : 0;
    h->cur_valid = false;
    return 0;
}
// rows that normalise the likelihood / entries of the statistic buffer to sum
static inline long rows_total(const tbnn_ctx* h) { return h->shard ? h->n_total : h->n; }
static inline int stat_entries(const tbnn_ctx* h) { return h->shard ? 1 : h->grid; }      // sharded: pstat_red[0] is the all-reduced sum
// local statistic buffer: entries >= grid stay zero; the all-reduced copy is separate (ranks may have different grids)
static inline const double* stat_ptr(const tbnn_ctx* h) { return h->shard ? h->pstat_red : h->pstat; }
// the gradient slabs k_update reduces
static inline const float* grad_slabs(const tbnn_ctx* h) { return (h->shard && h->wide_id < 0) ? h->grow : h->slabs; }
static inline int grad_nslab(const tbnn_ctx* h) { return h->shard ? 1 : h->nslab; }

// ---- launch helpers (all on h->stream, no sync) ----
// q == h->q: the leapfrog position, whose padded image h->qimg is maintained by k_update;
// any other q gets its image built here (h->qimg_cur).
static int launch_fwd_bwd(tbnn_ctx* h, const float* q, const float* eta, const StepCtl* ctl = nullptr, int t = 0) {
    const float* img = nullptr;
    const int C = h->C;
    const long imgS = h->img_floats, slabS = (long)h->nslab * h->pitch;
    if (h->kernel == TBNN_KERNEL_FAST) {
        if (q == h->q && h->q_img_valid) img = h->qimg;
        else {
            hipLaunchKernelGGL(k_make_image, dim3((h->nd.P + 255) / 256, C), dim3(256), 0, h->stream, h->nd.P, q, h->imgmap, h->qimg_cur, (long)h->nd.P, imgS);
            img = h->qimg_cur;
        }
    }
    hipEvent_t a = nullptr, b = nullptr;
    const bool prof = h->profile > 0 && (h->launch_no++ % h->profile) == 0;
    if (prof) {
        if (h->pev_used + 2 > h->pev.size()) {
            hipEventCreate(&a); hipEventCreate(&b); h->pev.push_back(a); h->pev.push_back(b);
        }
        a = h->pev[h->pev_used]; b = h->pev[h->pev_used + 1]; h->pev_used += 2;
        h->pev_div.push_back(1);
        hipEventRecord(a, h->stream);
    }
    // the one-slab-per-workgroup families take all chains of a multi-chain handle in ONE launch (gridDim.y = chain); the others
    // (two-kernel wide path, layered family, thread-per-row kernel: not what small problems run on) chain by chain through the
    // same activation store
    const ChainStride cs = {imgS, (long)h->nd.H, slabS, ctl, t};
    // (the chain-by-chain families skip a chain past its own L on the host: ctl_host mirrors ctl)
    auto done = [&](int c) { return ctl != nullptr && t > h->ctl_host[c].L; };
    if (h->lay) {
        for (int c = 0; c < C; ++c)
            if (!done(c) && lay_launch(h->nd, h->lplan, h->stream, img + c * imgS, eta + (size_t)c * h->nd.H, h->dY, h->n, h->lstore, h->slabs + c * slabS, h->pitch,
                           h->pstat + (size_t)c * PSTAT_CAP))
                return fail(-2, "layered kernel launch failed");
    } else if (h->wide_id >= 0) {
        for (int c = 0; c < C; ++c) {
            if (done(c)) continue;
            const float* ic = img + c * imgS; const float* ec = eta + (size_t)c * h->nd.H;
            double* pc = h->pstat + (size_t)c * PSTAT_CAP; float* sl = h->slabs + c * slabS;
            const int rc = h->jit ? h->jit->wlaunch(&h->wplan, h->stream, &h->nd, ic, ec, h->dX, h->dY, h->n, h->wstore, h->wslabA, h->wslabB, pc, sl)
                                  : wide_launch(h->wplan, h->stream, h->nd, ic, ec, h->dX, h->dY, h->n, h->wstore, h->wslabA, h->wslabB, pc, sl);
            if (rc) return fail(-2, "wide kernel launch failed");
        }
    } else if (h->kernel == TBNN_KERNEL_FAST && h->jit) {
        if (h->jit->launch(h->grid, h->stream, &h->nd, img, eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, C, cs))
            return fail(-2, "registered kernel launch failed");
    } else if (h->kernel == TBNN_KERNEL_FAST && h->mid_id >= 0) {
        if (mid_launch(h->mid_id, h->grid, h->stream, h->nd, img, eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, C, cs))
            return fail(-2, "mid kernel launch failed");
    } else if (h->kernel == TBNN_KERNEL_FAST) {
        int rc = h->fast_ver == 3 ? fast3_launch(h->fast_id, h->grid, h->stream, h->nd, img, eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, nullptr, C, cs)
                                  : fast_launch(h->fast_id, h->grid, h->stream, h->nd, img, eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, nullptr, C, cs);
        if (rc) return fail(-2, "fast kernel launch failed");
    } else {
        for (int c = 0; c < C; ++c)
            if (!done(c)) hipLaunchKernelGGL(k_fwd_bwd_generic, dim3(h->grid), dim3(GEN_RB), 0, h->stream, h->nd, q + (size_t)c * h->nd.P, eta + (size_t)c * h->nd.H, h->dX,
                               h->dY, h->n, h->scratch, h->scratchPerWG, h->slabs + c * slabS, h->pitch, h->pstat + (size_t)c * PSTAT_CAP);
    }
    if (h->shard) {
        // ONE collective per fused pass: the dense data-term gradient row (P values; the wide path already has one) and the
        // statistic, summed to one value on the device, travel together as P + 1 doubles (the sum over the ranks is then
        // taken in double and rounded once; at configs[1]-class step times a second small all-reduce would cost as much as
        // the step), summed over the ranks in place, unpacked into the row k_update reads and pstat_red[0]
        float* row = h->wide_id < 0 ? h->grow : h->slabs;
        const int P = h->nd.P;
        if (h->wide_id < 0)
            hipLaunchKernelGGL(k_slab_reduce, dim3((P + 63) / 64), dim3(64, 4), 0, h->stream, (const float*)h->slabs, h->nslab,
                               h->pitch, P, (float*)nullptr, h->shard_buf);
        hipLaunchKernelGGL(k_shard_pack, dim3((P + 255) / 256), dim3(256), 0, h->stream, P, h->wide_id < 0 ? (const float*)nullptr : (const float*)row,
                           (const double*)h->pstat, h->grid, h->shard_buf);
        NCCLCHK(g_rccl.AllReduce(h->shard_buf, h->shard_buf, (size_t)P + 1, ncclDouble, ncclSum, h->shard->comm, h->stream));
        hipLaunchKernelGGL(k_shard_unpack, dim3((P + 255) / 256), dim3(256), 0, h->stream, P, (const double*)h->shard_buf, row, h->pstat_red);
    }
    if (prof) hipEventRecord(b, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}
static void launch_update(tbnn_ctx* h, int mode, float eps, const float* eta, float* q, float* g, const StepCtl* ctl = nullptr, int t = 0) {
    const bool big = h->nd.P >= UPD_BIG_P;             // block geometry by parameter count (update_ops.hpp)
    const int ucols = big ? UPD_COLS_BIG : UPD_COLS;
    const int gx = (h->pitch / 4 + ucols - 1) / ucols;
    const bool img = h->kernel == TBNN_KERNEL_FAST && q == h->q;
    // the data-term gradient goes with its state: the proposal's (h->gd) or, for the bootstrap evaluation, the current one's
    float* gd = g == h->g_cur ? h->gd_cur : (g == h->g ? h->gd : nullptr);
    if (big)
        hipLaunchKernelGGL((k_update<UPD_COLS_BIG, UPD_GROUPS_BIG>), dim3(gx, h->C), dim3(UPD_COLS_BIG, UPD_GROUPS_BIG), 0, h->stream, h->nd, mode, eps, eta,
                           grad_slabs(h), grad_nslab(h), h->pitch, h->q_cur, h->g_cur, q, h->p, g, img ? h->imgmap : nullptr, h->qimg, gd, h->img_floats, ctl, t);
    else
        hipLaunchKernelGGL((k_update<UPD_COLS, UPD_GROUPS>), dim3(gx, h->C), dim3(UPD_COLS, UPD_GROUPS), 0, h->stream, h->nd, mode, eps, eta, grad_slabs(h),
                           grad_nslab(h), h->pitch, h->q_cur, h->g_cur, q, h->p, g, img ? h->imgmap : nullptr, h->qimg, gd, h->img_floats, ctl, t);
    if (img && (mode == UPD_FIRST || mode == UPD_MID)) h->q_img_valid = true;
}
static void launch_energy(tbnn_ctx* h, int which, const float* eta, const float* q, double* slot) {
    hipLaunchKernelGGL(k_energy, dim3(1, h->C), dim3(1024), 0, h->stream, h->nd, which, eta, q, h->p, h->q_cur, stat_ptr(h),
                       stat_entries(h), rows_total(h), h->sc, slot);
}

// make (logp, grad, stat) at q_cur valid
static int ensure_current(tbnn_ctx* h, double* slot) {
    if (h->cur_valid) return 0;
    int rc = launch_fwd_bwd(h, h->q_cur, h->eta);
    if (rc) return rc;
    launch_update(h, UPD_GRAD_ONLY, 0.f, h->eta, h->q_cur, h->g_cur);
    launch_energy(h, EN_CUR, h->eta, h->q_cur, slot);
    h->cur_valid = true;
    return 0;
}
// mean duration (us) of the profiled fwd+bwd launches since the last drain
static float drain_profile(tbnn_ctx* h) {
    double tot = 0.0; int cnt = 0;
    for (size_t i = 0; i + 1 < h->pev_used; i += 2) {
        float ms = 0.f;
        const int div = i / 2 < h->pev_div.size() ? h->pev_div[i / 2] : 1;
        if (hipEventElapsedTime(&ms, h->pev[i], h->pev[i + 1]) == hipSuccess) { tot += ms * 1000.0 / (div > 0 ? div : 1); ++cnt; }
    }
    h->pev_used = 0;                      // the events stay in the pool
    h->pev_div.clear();
    return cnt ? (float)(tot / cnt) : 0.f;
}

extern "C" int tbnn_logp_grad(tbnn_handle h, const float* theta, const float* eta, double* logp, float* grad,
                              double* stat) {
    NEED(h);
    ONE_CHAIN(h, "tbnn_logp_grad");
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    HIPCHK(hipSetDevice(h->device));
    const NetDev& nd = h->nd;
    // evaluation happens on scratch copies (q, tmp-eta): the chain state is untouched
    const float* dq = h->q_cur;
    const float* de = h->eta;
    if (theta) { HIPCHK(hipMemcpyAsync(h->q, theta, (size_t)nd.P * sizeof(float), hipMemcpyHostToDevice, h->stream)); dq = h->q; h->q_img_valid = false; }
    if (eta) { HIPCHK(hipMemcpyAsync(h->tmp + nd.P, eta, (size_t)nd.H * sizeof(float), hipMemcpyHostToDevice, h->stream)); de = h->tmp + nd.P; }
    int rc = launch_fwd_bwd(h, dq, de);
    if (rc) return rc;
    const int gx = (h->pitch / 4 + UPD_COLS - 1) / UPD_COLS;
    hipLaunchKernelGGL((k_update<UPD_COLS, UPD_GROUPS>), dim3(gx), dim3(UPD_COLS, UPD_GROUPS), 0, h->stream, nd, (int)UPD_GRAD_ONLY, 0.f, de, grad_slabs(h), grad_nslab(h),
                       h->pitch, h->q_cur, h->g_cur, const_cast<float*>(dq), h->p, h->tmp, (const int*)nullptr, (float*)nullptr, (float*)nullptr);
    // EN_TRACE leaves the chain's scalar record alone; stat comes from the slabs
    if (!h->trace || h->trace_cap < 2) {
        if (h->trace) hipFree(h->trace);
        HIPCHK(hipMalloc(&h->trace, 4096 * sizeof(double))); h->trace_cap = 4096;
    }
    hipLaunchKernelGGL(k_energy, dim3(1), dim3(1024), 0, h->stream, nd, (int)EN_TRACE, de, dq, h->p, h->q_cur, stat_ptr(h),
                       stat_entries(h), rows_total(h), h->sc, h->trace);
    HIPCHK(hipGetLastError());
    double lp = 0.0;
    HIPCHK(hipMemcpyAsync(&lp, h->trace, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (grad) HIPCHK(hipMemcpyAsync(grad, h->tmp, (size_t)nd.P * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    std::vector<double> ps;
    if (stat) { ps.resize(stat_entries(h)); HIPCHK(hipMemcpyAsync(ps.data(), stat_ptr(h), ps.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream)); }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->profile) drain_profile(h);
    if (logp) *logp = lp;
    if (stat) { double s = 0; for (double v : ps) s += v; *stat = s; }
    return 0;
}

// the narrow family's forward-only kernel exists for fast3 shapes (registry or run-time compiled)
static inline bool narrow_fwd_ok(const tbnn_ctx* h) {
    if (h->kernel != TBNN_KERNEL_FAST || h->wide_id >= 0) return false;
    return h->jit ? h->jit->nforward != nullptr : (h->mid_id >= 0 || h->fast_ver == 3);
}
// `nets` networks whose images lie img_stride floats apart -> outputs out_stride floats apart
static int narrow_forward(tbnn_ctx* h, int nets, const float* imgs, long img_stride, const float* dX, long n, float* dOut, long out_stride) {
    const long ntiles = (n + 15) / 16, wgs = (ntiles + FAST_WAVES - 1) / FAST_WAVES;
    // one network: fill the chip; an ensemble: the networks (grid.y) do that, fewer workgroups each re-use the image more
    const long cap = nets >= 64 ? 16 : (nets >= 8 ? 64 : 256);
    const int gx = (int)std::max<long>(1, std::min<long>(wgs, cap));
    if (h->jit) return h->jit->nforward(gx, nets, h->stream, imgs, img_stride, dX, n, dOut, out_stride);
    if (h->mid_id >= 0) return mid_forward(h->mid_id, gx, nets, h->stream, imgs, img_stride, dX, n, dOut, out_stride);
    return fast3_forward(h->fast_id, gx, nets, h->stream, imgs, img_stride, dX, n, dOut, out_stride);
}

// the layered family's forward chain (network.predict) on a store of its own: the rows are packed once, then any number of
// networks run over them
struct LayFwd { LayPlan pp{}; float* st = nullptr; long n = 0; };
// (the store is pooled on the chain, like the record buffers of tbnn_hmc_run: network.train predicts and scores every displayed
// epoch, and an allocation + a device synchronisation + a free per call is what that cost before round 4; everything runs in
// order on the chain's one stream, so the next call may overwrite it)
static int lay_fwd_prepare(tbnn_ctx* h, const float* dX, long n, LayFwd& lf) {
    const NetDev& nd = h->nd;
    lf.pp = h->lplan; lf.n = n;
    lay_plan_rows(nd, n, lf.pp);
    const size_t need = (size_t)(lf.pp.aOff[nd.nl] + lf.pp.ntiles * 256 * lf.pp.TM[nd.nl - 1]);
    if (h->lfwd_floats < need) {
        if (h->lfwd) { hipFree(h->lfwd); h->lfwd = nullptr; h->lfwd_floats = 0; }
        HIPCHK(hipMalloc(&h->lfwd, need * sizeof(float)));
        h->lfwd_floats = need;
    }
    lf.st = h->lfwd;
    hipLaunchKernelGGL(k_lay_pack_x, dim3((int)std::min<long>(lf.pp.ntiles * lf.pp.TK[0], 4096)), dim3(256), 0, h->stream, dX, n, nd.d_in, lf.pp.TK[0],
                       lf.pp.ntiles, lf.st + lf.pp.aOff[0]);
    HIPCHK(hipGetLastError());
    return 0;
}
static int lay_fwd_run(tbnn_ctx* h, const LayFwd& lf, const float* q, float* dOut) {
    const NetDev& nd = h->nd;
    hipLaunchKernelGGL(k_make_image, dim3((nd.P + 255) / 256), dim3(256), 0, h->stream, nd.P, q, h->imgmap, h->qimg_cur, 0L, 0L);
    lay_forward_chain(nd, lf.pp, h->stream, h->qimg_cur, lf.st);
    hipLaunchKernelGGL(k_lay_unpack_f, dim3((int)std::min<long>(lf.pp.ntiles, 2048)), dim3(256), 0, h->stream, (const float*)(lf.st + lf.pp.aOff[nd.nl]), lf.n,
                       lf.pp.ntiles, lf.pp.TM[nd.nl - 1], nd.d_out, dOut);
    if (hipGetLastError() != hipSuccess) return fail(-2, "layered forward launch failed");
    return 0;
}

// forward pass of the network at the weights q (device) over dX[n][d_in] -> dOut[d_out][n], on h->stream
static int launch_forward(tbnn_ctx* h, const float* q, const float* dX, long n, float* dOut) {
    const NetDev& nd = h->nd;
    const bool wide_fwd = h->wide_id >= 0 && (h->jit ? h->jit->wforward != nullptr : h->wplan.fwd_ok != 0);
    if (wide_fwd) {
        // MFMA forward (k_chain_wide<S, FWD>): needs the padded image of q
        hipLaunchKernelGGL(k_make_image, dim3((nd.P + 255) / 256), dim3(256), 0, h->stream, nd.P, q, h->imgmap, h->qimg_cur);
        const int rc = h->jit ? h->jit->wforward(h->stream, &h->nd, h->qimg_cur, dX, n, dOut)
                              : wide_forward(h->wide_id, h->stream, h->nd, h->qimg_cur, dX, n, dOut);
        if (rc) return fail(-2, "wide forward launch failed");
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->lay) {
        LayFwd lf;
        int rc = lay_fwd_prepare(h, dX, n, lf);
        if (!rc) rc = lay_fwd_run(h, lf, q, dOut);
        return rc;
    }
    if (narrow_fwd_ok(h)) {
        // MFMA forward of the narrow family (k_forward_fast3): image of q, one network
        hipLaunchKernelGGL(k_make_image, dim3((nd.P + 255) / 256), dim3(256), 0, h->stream, nd.P, q, h->imgmap, h->qimg_cur, 0L, 0L);
        if (narrow_forward(h, 1, h->qimg_cur, 0, dX, n, dOut, 0)) return fail(-2, "fast3 forward launch failed");
        HIPCHK(hipGetLastError());
        return 0;
    }
    const long nblk = (n + GEN_RB - 1) / GEN_RB;
    const int grid = (int)std::min<long>(nblk, 512);
    const size_t per = generic_scratch_floats(nd);
    float* scr = nullptr;
    HIPCHK(hipMalloc(&scr, per * sizeof(float) * grid));
    hipLaunchKernelGGL(k_forward_generic, dim3(grid), dim3(GEN_RB), 0, h->stream, nd, q, dX, n, scr, per, dOut);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(scr);
    return 0;
}

extern "C" int tbnn_forward(tbnn_handle h, const float* theta, const float* X, int64_t n, float* out) {
    NEED(h);
    if (!theta) ONE_CHAIN(h, "tbnn_forward at the chain's own state");
    if (!X || !out || n < 1) return fail(-1, "forward: null pointer or n < 1");
    HIPCHK(hipSetDevice(h->device));
    const NetDev& nd = h->nd;
    const float* dq = h->q_cur;
    if (theta) { HIPCHK(hipMemcpyAsync(h->q, theta, (size_t)nd.P * sizeof(float), hipMemcpyHostToDevice, h->stream)); dq = h->q; h->q_img_valid = false; }
    float *dXf = nullptr, *dOut = nullptr;
    HIPCHK(hipMalloc(&dXf, (size_t)n * nd.d_in * sizeof(float)));
    HIPCHK(hipMalloc(&dOut, (size_t)n * nd.d_out * sizeof(float)));
    HIPCHK(hipMemcpyAsync(dXf, X, (size_t)n * nd.d_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
    int rc = launch_forward(h, dq, dXf, (long)n, dOut);
    if (!rc) {
        HIPCHK(hipMemcpyAsync(out, dOut, (size_t)n * nd.d_out * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    hipFree(dXf); hipFree(dOut);
    return rc;
}

extern "C" int tbnn_set_validation(tbnn_handle h, const float* X, const float* Y, int64_t n) {
    NEED(h);
    if (!X || !Y || n < 1) return fail(-1, "set_validation: null pointer or n < 1");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->dXv) { hipFree(h->dXv); h->dXv = nullptr; }
    if (h->dYv) { hipFree(h->dYv); h->dYv = nullptr; }
    HIPCHK(hipMalloc(&h->dXv, (size_t)n * h->nd.d_in * sizeof(float)));
    HIPCHK(hipMalloc(&h->dYv, (size_t)n * h->nd.d_out * sizeof(float)));
    HIPCHK(hipMemcpy(h->dXv, X, (size_t)n * h->nd.d_in * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->dYv, Y, (size_t)n * h->nd.d_out * sizeof(float), hipMemcpyHostToDevice));
    h->nv = (long)n;
    return 0;
}

// predictions over the staged training (which = 0) or validation (1) rows into h->fbuf[d_out][n]
static int predict_resident(tbnn_ctx* h, int which, const float* theta, long* n_out) {
    const NetDev& nd = h->nd;
    if (!theta) ONE_CHAIN(h, "tbnn_predict / tbnn_metrics at the chain's own state");
    const float* dX = which ? h->dXv : h->dX;
    const long n = which ? h->nv : h->n;
    if (!dX || n < 1) return fail(-1, which ? "tbnn_set_validation has not been called" : "tbnn_set_data has not been called");
    const float* dq = h->q_cur;
    if (theta) { HIPCHK(hipMemcpyAsync(h->q, theta, (size_t)nd.P * sizeof(float), hipMemcpyHostToDevice, h->stream)); dq = h->q; h->q_img_valid = false; }
    const size_t need = (size_t)n * nd.d_out;
    if (h->fbuf_floats < need) {
        if (h->fbuf) hipFree(h->fbuf);
        h->fbuf = nullptr; h->fbuf_floats = 0;
        HIPCHK(hipMalloc(&h->fbuf, need * sizeof(float))); h->fbuf_floats = need;
    }
    *n_out = n;
    return launch_forward(h, dq, dX, n, h->fbuf);
}

extern "C" int tbnn_predict(tbnn_handle h, int which, const float* theta, float* out) {
    NEED(h);
    if (which != 0 && which != 1) return fail(-1, "which must be 0 (training rows) or 1 (validation rows)");
    HIPCHK(hipSetDevice(h->device));
    long n = 0;
    int rc = predict_resident(h, which, theta, &n);
    if (rc) return rc;
    if (out) HIPCHK(hipMemcpyAsync(out, h->fbuf, (size_t)n * h->nd.d_out * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// Ensemble prediction (predictor.py:132-155): m networks, theta_i = thetas + i * theta_stride, over the same rows.
// X == null: the staged rows selected by `which` (0 training, 1 validation); else n host rows.  out[m][d_out][n].
extern "C" int tbnn_forward_many(tbnn_handle h, const float* thetas, int32_t m, int64_t theta_stride, int which, const float* X,
                                 int64_t n, float* out) {
    NEED(h);
    const NetDev& nd = h->nd;
    if (!thetas || !out || m < 1 || theta_stride < nd.P) return fail(-1, "forward_many: null pointer, m < 1 or theta_stride < P");
    HIPCHK(hipSetDevice(h->device));
    const float* dX = nullptr;
    float* dXown = nullptr;
    long rows = 0;
    if (X) {
        if (n < 1) return fail(-1, "forward_many: n < 1");
        rows = (long)n;
        HIPCHK(hipMalloc(&dXown, (size_t)rows * nd.d_in * sizeof(float)));
        HIPCHK(hipMemcpyAsync(dXown, X, (size_t)rows * nd.d_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
        dX = dXown;
    } else {
        if (which != 0 && which != 1) return fail(-1, "which must be 0 (training rows) or 1 (validation rows)");
        dX = which ? h->dXv : h->dX; rows = which ? h->nv : h->n;
        if (!dX || rows < 1) return fail(-1, which ? "tbnn_set_validation has not been called" : "tbnn_set_data has not been called");
    }
    const size_t per_net = (size_t)rows * nd.d_out;
    // networks per pass: the output chunk stays below 1 GiB
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)m, ((size_t)1 << 28) / std::max<size_t>(per_net, 1)));
    float *dTh = nullptr, *dOut = nullptr, *dImg = nullptr;
    int rc = 0;
    auto cleanup = [&]() { if (dTh) hipFree(dTh); if (dOut) hipFree(dOut); if (dImg) hipFree(dImg); if (dXown) hipFree(dXown); };
#define FM_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(-2, hipGetErrorString(e_)); } } while (0)
    FM_CHK(hipMalloc(&dTh, (size_t)chunk * nd.P * sizeof(float)));
    FM_CHK(hipMalloc(&dOut, (size_t)chunk * per_net * sizeof(float)));
    const bool batched = narrow_fwd_ok(h);
    if (batched) {
        FM_CHK(hipMalloc(&dImg, (size_t)chunk * h->img_floats * sizeof(float)));
        FM_CHK(hipMemsetAsync(dImg, 0, (size_t)chunk * h->img_floats * sizeof(float), h->stream));
    }
    for (int i0 = 0; i0 < m && !rc; i0 += chunk) {
        const int c = std::min(chunk, m - i0);
        FM_CHK(hipMemcpy2DAsync(dTh, (size_t)nd.P * sizeof(float), thetas + (size_t)i0 * theta_stride, (size_t)theta_stride * sizeof(float),
                                (size_t)nd.P * sizeof(float), (size_t)c, hipMemcpyHostToDevice, h->stream));
        if (batched) {
            hipLaunchKernelGGL(k_make_image, dim3((nd.P + 255) / 256, c), dim3(256), 0, h->stream, nd.P, (const float*)dTh, h->imgmap, dImg,
                               (long)nd.P, (long)h->img_floats);
            rc = narrow_forward(h, c, dImg, h->img_floats, dX, rows, dOut, (long)per_net);
            if (rc) rc = fail(-2, "fast3 ensemble forward launch failed");
        } else if (h->lay) {
            LayFwd lf;
            rc = lay_fwd_prepare(h, dX, rows, lf);
            for (int i = 0; i < c && !rc; ++i) rc = lay_fwd_run(h, lf, dTh + (size_t)i * nd.P, dOut + (size_t)i * per_net);
        } else {
            for (int i = 0; i < c && !rc; ++i) rc = launch_forward(h, dTh + (size_t)i * nd.P, dX, rows, dOut + (size_t)i * per_net);
        }
        if (!rc) {
            FM_CHK(hipGetLastError());
            FM_CHK(hipMemcpyAsync(out + (size_t)i0 * per_net, dOut, (size_t)c * per_net * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            FM_CHK(hipStreamSynchronize(h->stream));
        }
    }
#undef FM_CHK
    hipStreamSynchronize(h->stream);
    cleanup();
    return rc;
}

extern "C" int tbnn_metrics(tbnn_handle h, int which, const float* theta, float mean, float sd, int exp_pred, int exp_real,
                            double out3[3]) {
    NEED(h);
    if (!out3) return fail(-1, "null out3");
    if (which != 0 && which != 1) return fail(-1, "which must be 0 (training rows) or 1 (validation rows)");
    HIPCHK(hipSetDevice(h->device));
    long n = 0;
    int rc = predict_resident(h, which, theta, &n);
    if (rc) return rc;
    const int MB = 256;
    if (!h->mpart) HIPCHK(hipMalloc(&h->mpart, 3 * MB * sizeof(double)));
    const long tot = n * h->nd.d_out;
    const int grid = (int)std::min<long>(MB, (tot + 255) / 256);
    hipLaunchKernelGGL(k_metrics, dim3(grid), dim3(256), 0, h->stream, (const float*)h->fbuf, (const float*)(which ? h->dYv : h->dY), n,
                       h->nd.d_out, mean, sd, exp_pred, exp_real, h->mpart);
    HIPCHK(hipGetLastError());
    std::vector<double> part(3 * (size_t)grid);
    HIPCHK(hipMemcpyAsync(part.data(), h->mpart, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double a[3] = {0, 0, 0};
    for (int b = 0; b < grid; ++b) for (int k = 0; k < 3; ++k) a[k] += part[3 * b + k];
    for (int k = 0; k < 3; ++k) out3[k] = a[k] / (double)tot;
    return 0;
}

// enqueue one transition (no sync).  trace: device array of L+1 doubles or null.
// ctl (device, mirrored in h->ctl_host) != null: every chain at its own (eps, L) -- L is then max_c L_c and eps unused
static int enqueue_transition(tbnn_ctx* h, float eps, int L, const float* d_p0, const float* d_logu, double* d_trace,
                              Scal* d_out, const StepCtl* ctl = nullptr) {
    const NetDev& nd = h->nd;
    // logp0 of the trace: a fresh bootstrap evaluation writes it; a cached one (cur_valid) is copied from the chain's
    // scalar record -- h->pstat then holds the statistic of the LAST fused launch (a rejected proposal, a
    // tbnn_logp_grad probe), not of q_cur, so EN_CUR must not be re-derived from it
    const bool cached = h->cur_valid;
    int rc = ensure_current(h, d_trace);
    if (rc) return rc;
    if (d_trace && cached) hipLaunchKernelGGL(k_trace_logp_cur, dim3(1), dim3(64), 0, h->stream, (const Scal*)h->sc, d_trace);
    hipLaunchKernelGGL(k_begin, dim3(1, h->C), dim3(1024), 0, h->stream, nd, d_p0, d_logu, h->epoch, h->key0, h->key1, h->p, h->sc, h->seed_hi);
    launch_update(h, UPD_FIRST, eps, h->eta, h->q, h->g, ctl, 0);
    // a small problem on an ahead-of-time narrow kernel: the L leapfrog steps in ONE launch, one workgroup per chain (kernels_traj.hpp).
    // Not for a traced transition (per-step energies) or a sharded gradient.  Profiling (tbnn_set_profiling) does NOT change the path (round 6;
    // ADVICE round 5): a profiled trajectory launch is bracketed by one event pair and reported per leapfrog step (its time / L).
    const bool traj = h->traj && !d_trace && L >= 1 && h->kernel == TBNN_KERNEL_FAST && h->wide_id < 0 && h->mid_id < 0 && !h->lay && !h->shard &&
                      (h->jit ? (h->jit->family == TBNN_FAMILY_NARROW && h->jit->traj != nullptr && h->n <= h->jit->traj_max_rows)
                              : (h->fast_ver == 3 && h->n <= fast3_traj_max_rows(h->fast_id)));
    h->last_path = traj ? "trajectory" : "per-step";
    if (traj) {
        hipEvent_t ea = nullptr, eb = nullptr;
        const bool prof = h->profile > 0 && (h->launch_no++ % h->profile) == 0;
        if (prof) {
            if (h->pev_used + 2 > h->pev.size()) { hipEventCreate(&ea); hipEventCreate(&eb); h->pev.push_back(ea); h->pev.push_back(eb); }
            ea = h->pev[h->pev_used]; eb = h->pev[h->pev_used + 1]; h->pev_used += 2;
            h->pev_div.push_back(L);
            hipEventRecord(ea, h->stream);
        }
        const int trc = h->jit ? h->jit->traj(h->C, h->stream, &nd, h->qimg, (long)h->img_floats, h->eta, h->dX, h->dY, h->n, h->q, h->p, h->g, h->gd, h->imgmap,
                                              h->pstat, stat_entries(h), eps, L, ctl)
                               : fast3_traj_launch(h->fast_id, h->C, h->stream, nd, h->qimg, (long)h->img_floats, h->eta, h->dX, h->dY, h->n, h->q, h->p, h->g,
                                                   h->gd, h->imgmap, h->pstat, stat_entries(h), eps, L, ctl);
        if (prof) hipEventRecord(eb, h->stream);
        if (trc)
            return fail(-2, "trajectory kernel launch failed");
        h->q_img_valid = false;            // the images the kernel advanced lived in LDS
    }
    for (int t = 1; t <= L && !traj; ++t) {
        rc = launch_fwd_bwd(h, h->q, h->eta, ctl, t);
        if (rc) return rc;
        if (d_trace && t < L) {
            // logp at q_t needs the prior at q_t: evaluate before the drift
            launch_update(h, UPD_GRAD_ONLY, 0.f, h->eta, h->q, h->g);
            launch_energy(h, EN_TRACE, h->eta, h->q, d_trace + t);
        }
        launch_update(h, t < L ? UPD_MID : UPD_LAST, eps, h->eta, h->q, h->g, ctl, t);
    }
    // the Metropolis decision, the host record and the commit in ONE single-workgroup launch for networks whose state that
    // workgroup copies in a few trips (TBNN_MERGE_ENDS=0: three launches, as before round 3)
    if (h->merge_ends && nd.P <= 32768) {
        hipLaunchKernelGGL(k_energy, dim3(1, h->C), dim3(1024), 0, h->stream, h->nd, (int)EN_NEW, (const float*)h->eta, (const float*)h->q, (const float*)h->p,
                           (const float*)h->q_cur, stat_ptr(h), stat_entries(h), rows_total(h), h->sc, d_trace ? d_trace + L : (double*)nullptr,
                           d_out, (const float*)h->g, h->q_cur, h->g_cur, (const float*)h->gd, h->gd_cur);
    } else {
        launch_energy(h, EN_NEW, h->eta, h->q, d_trace ? d_trace + L : nullptr);
        hipLaunchKernelGGL(k_commit, dim3((nd.P + 255) / 256, h->C), dim3(256), 0, h->stream, nd.P, (const Scal*)h->sc, (const float*)h->q, (const float*)h->g,
                           h->q_cur, h->g_cur, (const float*)h->gd, h->gd_cur);
        hipLaunchKernelGGL(k_commit_scal, dim3(h->C), dim3(64), 0, h->stream, h->sc, d_out);
    }
    HIPCHK(hipGetLastError());
    h->epoch += 1;
    return 0;
}

static void fill_out(const Scal& s, int L, float dev_us, float fb_us, tbnn_step_out* o) {
    o->accepted = s.accepted; o->n_leapfrog = L;
    o->log_accept_ratio = (float)s.lar;
    o->accept_prob = s.lar < 0 ? (float)exp(s.lar) : 1.f;      // network.py:410-411
    o->logp_old = s.logp_cur; o->logp_new = s.logp_new;
    o->kinetic_old = s.k0; o->kinetic_new = s.k1;
    o->sjd = s.sjd; o->device_us = dev_us; o->fwdbwd_us = fb_us;
}

extern "C" int tbnn_hmc_step(tbnn_handle h, float eps, int32_t L, const float* p0, const float* log_u,
                             tbnn_step_out* out, double* trace_logp) {
    NEED(h);
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    if (L < 1) return fail(-1, "L must be >= 1");
    if (h->C > 1 && (p0 || log_u || trace_logp)) return fail(-1, "hmc_step on a multi-chain handle: no injected draws, no trace (out: one record per chain)");
    HIPCHK(hipSetDevice(h->device));
    const NetDev& nd = h->nd;
    const float* d_p0 = nullptr; const float* d_lu = nullptr;
    if (p0) { HIPCHK(hipMemcpyAsync(h->p0_inj, p0, (size_t)nd.P * sizeof(float), hipMemcpyHostToDevice, h->stream)); d_p0 = h->p0_inj; }
    if (log_u) { HIPCHK(hipMemcpyAsync(h->logu_inj, log_u, sizeof(float), hipMemcpyHostToDevice, h->stream)); d_lu = h->logu_inj; }
    double* d_trace = nullptr;
    if (trace_logp) {
        if (h->trace_cap < L + 1) {
            if (h->trace) hipFree(h->trace);
            h->trace = nullptr; h->trace_cap = 0;
            HIPCHK(hipMalloc(&h->trace, (size_t)(L + 1 + 4096) * sizeof(double))); h->trace_cap = L + 1 + 4096;
        }
        d_trace = h->trace;
    }
    // the bootstrap evaluation (Q10: the reference pays it every epoch) is outside the timed events only
    // when it is cached; ensure_current is part of enqueue_transition.
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    int rc = enqueue_transition(h, eps, L, d_p0, d_lu, d_trace, h->sc_out);
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipMemcpyAsync(h->sc_host, h->sc_out, (size_t)h->C * sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
    if (trace_logp) HIPCHK(hipMemcpyAsync(trace_logp, h->trace, (size_t)(L + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f; hipEventElapsedTime(&ms, h->ev0, h->ev1);
    const float fb = h->profile ? drain_profile(h) : 0.f;
    if (out)
        for (int c = 0; c < h->C; ++c) fill_out(h->sc_host[c], L, ms * 1000.f, fb, out + c);
    return 0;
}

extern "C" int tbnn_hmc_run(tbnn_handle h, float eps, int32_t L, int32_t n_epochs, tbnn_step_out* outs) {
    NEED(h);
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    if (L < 1 || n_epochs < 1) return fail(-1, "L and n_epochs must be >= 1");
    if ((long long)n_epochs * (long long)h->C > (1LL << 24)) return fail(-1, "n_epochs x chains exceeds 2^24 records per call: split the run");
    HIPCHK(hipSetDevice(h->device));
    // per-epoch records: a pooled device buffer and a pooled pinned host mirror (no allocator call -- hipFree synchronises
    // the device -- inside a caller's timed loop once the pool has grown to the largest n_epochs seen)
    const int C = h->C;                                  // records: [epoch][chain] on the device, outs[chain][epoch] for the caller
    if (h->recs_cap < n_epochs * C) {
        if (h->d_recs) hipFree(h->d_recs);
        if (h->h_recs) hipHostFree(h->h_recs);
        h->d_recs = nullptr; h->h_recs = nullptr; h->recs_cap = 0;
        const int cap = std::max(n_epochs * C, 64);
        HIPCHK(hipMalloc(&h->d_recs, (size_t)cap * sizeof(Scal)));
        HIPCHK(hipHostMalloc(&h->h_recs, (size_t)cap * sizeof(Scal)));
        h->recs_cap = cap;
    }
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int e = 0; e < n_epochs; ++e) {
        int rc = enqueue_transition(h, eps, L, nullptr, nullptr, nullptr, h->d_recs + (size_t)e * C);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_recs, h->d_recs, (size_t)n_epochs * C * sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f; hipEventElapsedTime(&ms, h->ev0, h->ev1);
    const float fb = h->profile ? drain_profile(h) : 0.f;
    if (outs)
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < n_epochs; ++e) fill_out(h->h_recs[(size_t)e * C + c], L, ms * 1000.f / n_epochs, fb, outs + (size_t)c * n_epochs + e);
    return 0;
}

// every chain of the handle at its own (eps, L): stage the control block, run max L lockstep steps (enqueue_transition)
static int stage_ctl(tbnn_ctx* h, const float* eps, const int32_t* L, int* Lmax) {
    if (!eps || !L) return fail(-1, "null eps / L array");
    int m = 0;
    for (int c = 0; c < h->C; ++c) {
        if (L[c] < 1) return fail(-1, "every L[c] must be >= 1");
        h->ctl_host[c].eps = eps[c]; h->ctl_host[c].L = L[c];
        m = std::max(m, (int)L[c]);
    }
    HIPCHK(hipMemcpyAsync(h->ctl, h->ctl_host, (size_t)h->C * sizeof(StepCtl), hipMemcpyHostToDevice, h->stream));
    *Lmax = m;
    return 0;
}
extern "C" int tbnn_hmc_run_each(tbnn_handle h, const float* eps, const int32_t* L, int32_t n_epochs, tbnn_step_out* outs) {
    NEED(h);
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    if (n_epochs < 1) return fail(-1, "n_epochs must be >= 1");
    if ((long long)n_epochs * (long long)h->C > (1LL << 24)) return fail(-1, "n_epochs x chains exceeds 2^24 records per call: split the run");
    HIPCHK(hipSetDevice(h->device));
    const int C = h->C;
    // (the stream is idle here -- every entry point returns after its work has completed -- so re-staging ctl_host is safe)
    int Lmax = 0;
    int rc = stage_ctl(h, eps, L, &Lmax);
    if (rc) return rc;
    if (h->recs_cap < n_epochs * C) {
        if (h->d_recs) hipFree(h->d_recs);
        if (h->h_recs) hipHostFree(h->h_recs);
        h->d_recs = nullptr; h->h_recs = nullptr; h->recs_cap = 0;
        const int cap = std::max(n_epochs * C, 64);
        HIPCHK(hipMalloc(&h->d_recs, (size_t)cap * sizeof(Scal)));
        HIPCHK(hipHostMalloc(&h->h_recs, (size_t)cap * sizeof(Scal)));
        h->recs_cap = cap;
    }
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int e = 0; e < n_epochs; ++e) {
        rc = enqueue_transition(h, 0.f, Lmax, nullptr, nullptr, nullptr, h->d_recs + (size_t)e * C, h->ctl);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_recs, h->d_recs, (size_t)n_epochs * C * sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f; hipEventElapsedTime(&ms, h->ev0, h->ev1);
    const float fb = h->profile ? drain_profile(h) : 0.f;
    if (outs)
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < n_epochs; ++e) fill_out(h->h_recs[(size_t)e * C + c], L[c], ms * 1000.f / n_epochs, fb, outs + (size_t)c * n_epochs + e);
    return 0;
}
extern "C" int tbnn_hmc_step_each(tbnn_handle h, const float* eps, const int32_t* L, tbnn_step_out* out) {
    return tbnn_hmc_run_each(h, eps, L, 1, out);
}

extern "C" int tbnn_hyper_logp_grad(tbnn_handle h, const float* eta, double* logp, float* grad) {
    NEED(h);
    ONE_CHAIN(h, "tbnn_hyper_logp_grad");
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    HIPCHK(hipSetDevice(h->device));
    const NetDev& nd = h->nd;
    int rc = ensure_current(h, nullptr);   // stat_cur = sum (y-f)^2 at q_cur (closed-form data term)
    if (rc) return rc;
    float* de = h->eta;
    if (eta) { HIPCHK(hipMemcpyAsync(h->tmp + nd.P, eta, (size_t)nd.H * sizeof(float), hipMemcpyHostToDevice, h->stream)); de = h->tmp + nd.P; }
    hipLaunchKernelGGL(k_hyper, dim3(1), dim3(HYP_THREADS), 0, h->stream, nd, (int)HYP_EVAL, 0.f, 0, de, h->q_cur, rows_total(h),
                       (const float*)nullptr, (const float*)nullptr, 0u, h->key0, h->key1, h->sc, h->hyp_ws, h->sc_out);
    HIPCHK(hipGetLastError());
    std::vector<float> ws(hyper_ws_bytes(nd) / sizeof(float));
    HIPCHK(hipMemcpyAsync(ws.data(), h->hyp_ws, hyper_ws_bytes(nd), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->sc_host, h->sc_out, sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (logp) *logp = h->sc_host->logp_new;
    if (grad) memcpy(grad, ws.data() + HYP_WS_GRAD * nd.H, (size_t)nd.H * sizeof(float));
    return 0;
}

static int hyper_step_impl(tbnn_handle h, float eps_h, const float* eps_each, int32_t L_h, const float* p0, const float* log_u, tbnn_step_out* out);
extern "C" int tbnn_hyper_step(tbnn_handle h, float eps_h, int32_t L_h, const float* p0, const float* log_u,
                               tbnn_step_out* out) {
    return hyper_step_impl(h, eps_h, nullptr, L_h, p0, log_u, out);
}
extern "C" int tbnn_hyper_step_each(tbnn_handle h, const float* eps_h, int32_t L_h, tbnn_step_out* out) {
    NEED(h);
    if (!eps_h) return fail(-1, "null eps_h array");
    return hyper_step_impl(h, 0.f, eps_h, L_h, nullptr, nullptr, out);
}
// eps_each (host, [chains]) != null: chain c's hyper transition at eps_each[c]
static int hyper_step_impl(tbnn_handle h, float eps_h, const float* eps_each, int32_t L_h, const float* p0, const float* log_u, tbnn_step_out* out) {
    NEED(h);
    if (!h->dX) return fail(-1, "tbnn_set_data has not been called");
    if (L_h < 1) return fail(-1, "L_h must be >= 1");
    if (h->C > 1 && (p0 || log_u)) return fail(-1, "hyper_step on a multi-chain handle: no injected draws (out: one record per chain)");
    HIPCHK(hipSetDevice(h->device));
    const NetDev& nd = h->nd;
    const int C = h->C;
    int rc = ensure_current(h, nullptr);
    if (rc) return rc;
    const float* d_p0 = nullptr; const float* d_lu = nullptr;
    if (p0) { HIPCHK(hipMemcpyAsync(h->p0_inj, p0, (size_t)nd.H * sizeof(float), hipMemcpyHostToDevice, h->stream)); d_p0 = h->p0_inj; }
    if (log_u) { HIPCHK(hipMemcpyAsync(h->logu_inj, log_u, sizeof(float), hipMemcpyHostToDevice, h->stream)); d_lu = h->logu_inj; }
    HIPCHK(hipMemcpyAsync(h->eta_prev, h->eta, (size_t)C * nd.H * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    const float* d_each = nullptr;
    if (eps_each) {
        memcpy(h->epsh_host, eps_each, (size_t)C * sizeof(float));
        HIPCHK(hipMemcpyAsync(h->epsh, h->epsh_host, (size_t)C * sizeof(float), hipMemcpyHostToDevice, h->stream));
        d_each = h->epsh;
    }
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    // the hyper transition uses the epoch counter of the weight transition that preceded it
    const uint32_t ep = h->epoch > 0 ? h->epoch - 1 : 0;
    // (one workgroup runs one chain's whole hyper transition: the chains of a multi-chain handle side by side, gridDim.x = chain)
    static_assert(sizeof(float) * 4 == 16, "hyper work space: 4 H floats per chain (hyper_ws_bytes)");
    hipLaunchKernelGGL(k_hyper, dim3(C), dim3(HYP_THREADS), 0, h->stream, nd, (int)HYP_STEP, eps_h, (int)L_h, h->eta, (const float*)h->q_cur, rows_total(h),
                       d_p0, d_lu, ep, h->key0, h->key1, h->sc, h->hyp_ws, h->sc_out, h->seed_hi, d_each);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipMemcpyAsync(h->sc_host, h->sc_out, (size_t)C * sizeof(Scal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f; hipEventElapsedTime(&ms, h->ev0, h->ev1);
    if (out)
        for (int c = 0; c < C; ++c) fill_out(h->sc_host[c], L_h, ms * 1000.f, 0.f, out + c);
    // eta changed => the weight target changed.  The prediction does not depend on eta: the cached statistic stays, the
    // data-term gradient rescales with sigma, the prior terms are recomputed -- O(P), no pass over the rows
    // (TBNN_HYPER_FULL_REFRESH=1: the round-1 behaviour, a whole bootstrap evaluation)
    bool any = false;
    for (int c = 0; c < C; ++c) any = any || h->sc_host[c].accepted;
    if (any) {
        static const bool full = getenv("TBNN_HYPER_FULL_REFRESH") && atoi(getenv("TBNN_HYPER_FULL_REFRESH"));
        if (full || !h->cur_valid) h->cur_valid = false;
        else {
            // one launch each for all chains (gridDim.y = chain); a chain whose hyper proposal was rejected returns at once, on the
            // device, by its record in sc_out (it keeps its cached gradient bit for bit)
            const Scal* gate = C > 1 ? (const Scal*)h->sc_out : (const Scal*)nullptr;
            hipLaunchKernelGGL(k_refresh_grad_after_hyper, dim3((nd.P + 255) / 256, C), dim3(256), 0, h->stream, nd,
                               (const float*)h->eta, (const float*)h->q_cur, (const float*)h->gd_cur, h->g_cur, gate);
            hipLaunchKernelGGL(k_energy, dim3(1, C), dim3(1024), 0, h->stream, nd, (int)EN_REFRESH, (const float*)h->eta,
                               (const float*)h->q_cur, (const float*)h->p, (const float*)h->q_cur,
                               stat_ptr(h), stat_entries(h), rows_total(h), h->sc, (double*)nullptr,
                               (Scal*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, (float*)nullptr, gate);
            HIPCHK(hipGetLastError());
        }
    }
    return 0;
}

// predictor.trainProbs / reweight (predictor.py:188-206, :248-266): sum over the dense layers of calculateHyperProbs for m saved
// (theta, eta) pairs in ONE launch (grid.x = network).  priors: nl prior families to judge the layers under, or null (the chain's)
extern "C" int tbnn_hyper_probs_many(tbnn_handle h, const int32_t* priors, const float* thetas, int64_t theta_stride,
                                     const float* etas, int64_t eta_stride, int32_t m, double* out) {
    NEED(h);
    const NetDev& nd0 = h->nd;
    if (!thetas || !etas || !out || m < 1 || theta_stride < nd0.P || eta_stride < 4 * nd0.nl)
        return fail(-1, "hyper_probs_many: null pointer, m < 1, theta_stride < P or eta_stride < 4 * layers");
    NetDev nd = nd0;
    if (priors)
        for (int l = 0; l < nd.nl; ++l) {
            if (priors[l] != TBNN_PRIOR_CAUCHY && priors[l] != TBNN_PRIOR_GAUSSIAN) return fail(-1, "hyper_probs_many: unknown prior");
            nd.prior[l] = priors[l];
        }
    HIPCHK(hipSetDevice(h->device));
    float *dTh = nullptr, *dEt = nullptr; double* dOut = nullptr;
    auto cleanup = [&]() { if (dTh) hipFree(dTh); if (dEt) hipFree(dEt); if (dOut) hipFree(dOut); };
#define HP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(-2, hipGetErrorString(e_)); } } while (0)
    const int chunk = (int)std::min<size_t>((size_t)m, std::max<size_t>(1, ((size_t)1 << 28) / (size_t)nd.P));      // <= 1 GiB of weights per pass
    HP_CHK(hipMalloc(&dTh, (size_t)chunk * nd.P * sizeof(float)));
    HP_CHK(hipMalloc(&dEt, (size_t)chunk * 4 * nd.nl * sizeof(float)));
    HP_CHK(hipMalloc(&dOut, (size_t)chunk * sizeof(double)));
    for (int i0 = 0; i0 < m; i0 += chunk) {
        const int c = std::min(chunk, m - i0);
        HP_CHK(hipMemcpy2DAsync(dTh, (size_t)nd.P * sizeof(float), thetas + (size_t)i0 * theta_stride, (size_t)theta_stride * sizeof(float),
                                (size_t)nd.P * sizeof(float), (size_t)c, hipMemcpyHostToDevice, h->stream));
        HP_CHK(hipMemcpy2DAsync(dEt, (size_t)4 * nd.nl * sizeof(float), etas + (size_t)i0 * eta_stride, (size_t)eta_stride * sizeof(float),
                                (size_t)4 * nd.nl * sizeof(float), (size_t)c, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_hyper_probs, dim3(c), dim3(256), 0, h->stream, nd, (const float*)dTh, (long)nd.P, (const float*)dEt, (long)(4 * nd.nl), dOut);
        HP_CHK(hipGetLastError());
        HP_CHK(hipMemcpyAsync(out + i0, dOut, (size_t)c * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HP_CHK(hipStreamSynchronize(h->stream));
    }
#undef HP_CHK
    cleanup();
    return 0;
}

extern "C" int tbnn_export_sample_device(tbnn_handle h, float* d_out) {
    NEED(h); if (!d_out) return fail(-1, "null d_out");
    ONE_CHAIN(h, "tbnn_export_sample_device");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(d_out, h->q_cur, (size_t)h->nd.P * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_out + h->nd.P, h->eta, (size_t)h->nd.H * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// diagnostic (tests): the momentum the last trajectory ENDED with (p_L, after the closing half kick), whether or not the proposal
// was accepted -- what a time-reversal test needs (include/tbnn.h)
extern "C" int tbnn_debug_momentum(tbnn_handle h, float* p_out) {
    NEED(h); if (!p_out) return fail(-1, "null p_out");
    ONE_CHAIN(h, "tbnn_debug_momentum");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(p_out, h->p, (size_t)h->nd.P * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int tbnn_debug_draw(tbnn_handle h, uint32_t epoch, uint32_t purpose, int32_t n, float* out_normals,
                               float* out_log_u) {
    NEED(h);
    if (n < 1 || !out_normals) return fail(-1, "debug_draw: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)(n + 1) * sizeof(float)));
    hipLaunchKernelGGL(k_debug_draw, dim3((n + 255) / 256), dim3(256), 0, h->stream, epoch, purpose, h->key0, h->key1, (int)n, d, d + n);
    HIPCHK(hipGetLastError());
    std::vector<float> host(n + 1);
    HIPCHK(hipMemcpyAsync(host.data(), d, (size_t)(n + 1) * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(d);
    memcpy(out_normals, host.data(), (size_t)n * sizeof(float));
    if (out_log_u) *out_log_u = host[n];
    return 0;
}

// diagnostic: stamps of workgroup 0 inside one fused launch (include/tbnn.h)
extern "C" int tbnn_debug_stamps(tbnn_handle h, uint64_t* out5) {
    if (!out5) return fail(-1, "null out16");
    NEED(h);
    if (h->kernel != TBNN_KERNEL_FAST || h->wide_id >= 0 || h->mid_id >= 0 || h->jit || h->lay || !h->dX || h->C > 1) return fail(-1, "debug_stamps: narrow fast kernel + data required");
    HIPCHK(hipSetDevice(h->device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 16 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, 16 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_make_image, dim3((h->nd.P + 255) / 256), dim3(256), 0, h->stream, h->nd.P, h->q_cur, h->imgmap, h->qimg_cur);
    for (int rep = 0; rep < 3; ++rep) {
        if (h->fast_ver == 3) fast3_launch(h->fast_id, h->grid, h->stream, h->nd, h->qimg_cur, h->eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, d);
        else fast_launch(h->fast_id, h->grid, h->stream, h->nd, h->qimg_cur, h->eta, h->dX, h->dY, h->n, h->slabs, h->pitch, h->pstat, d);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out5, d, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    hipFree(d);
    return 0;
}

extern "C" int tbnn_debug_fused_burst(tbnn_handle h, int32_t reps, float* us_per_pass) {
    if (!us_per_pass || reps < 1) return fail(-1, "fused_burst: reps >= 1 and an output pointer required");
    NEED(h);
    if (!h->dX) return fail(-1, "fused_burst: no data");
    if (h->shard) return fail(-1, "fused_burst: not on a row-sharded handle (every pass would be a collective; the other ranks are not in this call)");
    HIPCHK(hipSetDevice(h->device));
    const size_t PB = (size_t)h->C * h->nd.P * sizeof(float);
    // the proposal buffers (free between two transitions) take a copy of the current state; its image is built once
    HIPCHK(hipMemcpyAsync(h->q, h->q_cur, PB, hipMemcpyDeviceToDevice, h->stream));
    if (h->kernel == TBNN_KERNEL_FAST) {
        hipLaunchKernelGGL(k_make_image, dim3((h->nd.P + 255) / 256, h->C), dim3(256), 0, h->stream, h->nd.P, (const float*)h->q, h->imgmap, h->qimg, (long)h->nd.P,
                           (long)h->img_floats);
        h->q_img_valid = true;
    }
    struct ProfileOff {                                    // restored on EVERY exit (HIPCHK returns early)
        tbnn_ctx* h; int saved;
        explicit ProfileOff(tbnn_ctx* h_) : h(h_), saved(h_->profile) { h->profile = 0; }
        ~ProfileOff() { h->profile = saved; h->q_img_valid = false; }
    } guard(h);
    int rc = launch_fwd_bwd(h, h->q, h->eta);              // one untimed pass (first touch)
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps && !rc; ++r) rc = launch_fwd_bwd(h, h->q, h->eta);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *us_per_pass = ms * 1000.f / (float)reps;
    return 0;
}

#ifdef TBNN_TILE_STAMPS
extern "C" int tbnn_debug_tile_stamps(tbnn_handle h, uint64_t* out64) {
    NEED(h);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tile_stamps), 64 * sizeof(uint64_t)));
    return 0;
}
#endif
