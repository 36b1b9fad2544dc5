// C-ABI host side of the MI355X N2 integrator (see include/rmt_n2.h for the contract and the
// reference interfaces each entry point replaces).  Host code only: the device code is the
// template in kernels/*.inc (concatenated by embed.py in the order of kernels/ORDER), specialised by a generated prelude and compiled with hipRTC.
#include "rmt_n2.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static const char* const k_template =
#include "n2_kernels_embed.h"
    ;

static thread_local std::string g_err;

// workgroups per CU the single-evaluation kernel rmt_n2_rhs is launched with at most (it loops over the reactors): enough
// to fill every wave slot the kernel's registers allow (4 per SIMD at 256 threads), with a second set queued behind
#define RMT_N2_RHS_WGS_PER_CU 8
// calls of rmt_n2_rk4 that run the plain on-chip stepper alone after a cached launch lost half of its reactors to it
#define RMT_N2_PLAIN_LAUNCHES 8

static int fail(const char* fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_OK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fail("%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct rmt_n2_handle {
    int S = 0, V = 0, NU = 0, N = 0, E = 0, fp32 = 0, block = 0, npt = 0, mode = 0, device = 0;
    int ros_nb = 0;                          // mesh nodes per workgroup of the stiff stepper (block, or block / 4 in the quad layout)
    size_t real_size = 8;
    hipModule_t module = nullptr;
    hipFunction_t f_rhs = nullptr, f_rk4_reg = nullptr, f_rk4_mem = nullptr, f_rk45_reg = nullptr,
                  f_rk45_mem = nullptr, f_multistep = nullptr, f_rk4_chain = nullptr, f_ros4 = nullptr, f_n1 = nullptr,
                  f_ros4_chain = nullptr, f_rk45_chain = nullptr, f_rk4_redo = nullptr, f_rk4_chain_redo = nullptr;
    unsigned long long* d_rings = nullptr;   // tagged-word links of the chained stiff stepper: rings, decision slots, abort words
    size_t ring_bytes = 0;
    double* d_members1 = nullptr;
    unsigned* d_mask = nullptr;
    size_t mask_elems = 0;
    unsigned long long* d_sync = nullptr;
    double* d_slots = nullptr;
    size_t chain_links = 0;
    int n_cus = 0;
    double* d_members = nullptr;
    unsigned* d_flags = nullptr;
    void* d_work = nullptr;
    void* d_backup = nullptr;          // chained cached RK4 stepper: the launch's input, and its redo / status words
    unsigned* d_redo = nullptr;
    size_t work_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // cached one-workgroup RK4 stepper: the fallback counter is copied back behind every cached launch (pinned word +
    // event, never waited for); when a launch lost at least half of its reactors to the plain stepper, the next
    // RMT_N2_PLAIN_LAUNCHES calls run the plain stepper alone
    unsigned* fb_host = nullptr;
    hipEvent_t ev_fb = nullptr;
    bool fb_pending = false;
    unsigned fb_seen = 0;
    int plain_left = 0;
    bool timed = false;
    int last_chunks = 1, last_teams = 0;     // geometry of the last stepper launch (rmt_n2_last_geometry)
};

// Every entry point runs on the device the handle was created on: workspace allocations, event
// records and module launches all act on the CURRENT device, which the caller may have changed
// since rmt_n2_create (multi-GPU hosts).  The guard switches to the handle's device and restores the
// caller's on scope exit.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(const rmt_n2_handle* h) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != h->device) {
            err = hipSetDevice(h->device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(h)                                                                        \
    DeviceGuard guard_(h);                                                                  \
    if (guard_.err != hipSuccess)                                                           \
        return fail("cannot switch to device %d of this handle: %s", (h)->device, hipGetErrorString(guard_.err))

extern "C" const char* rmt_n2_last_error(void) { return g_err.c_str(); }
extern "C" int rmt_n2_abi_version(void) { return RMT_N2_ABI_VERSION; }
extern "C" const char* rmt_n2_kernel_template(void) { return k_template; }
extern "C" const char* rmt_n2_hiprtc_path(void) {
    // the shared object hiprtcCompileProgram resolves to in THIS process
    Dl_info info;
    if (dladdr((void*)&hiprtcCompileProgram, &info) && info.dli_fname) return info.dli_fname;
    return "";
}
extern "C" void rmt_n2_free(void* p) { free(p); }
// Options every kernel is compiled with.  -disable-machine-licm: the machine-level loop-invariant code motion
// hoists the materialisation of fp64 literals (s_mov pairs) and compare masks out of the time-step loop into
// SGPRs, runs out of them and spills to VGPR lanes (v_writelane / v_readlane, full VALU slots) and, for the
// VGPR-resident ones, to scratch; without it the step loop of rmt_n2_rk45_reg has 2135 instead of 2294 VALU
// instructions and 4 instead of 13 scratch accesses (measured: +16 %; rmt_n2_rk4_reg +4 %, profiles/round2_issue_model.md).
static const char* const k_default_opts = "-O3 -std=c++17 -mllvm -disable-machine-licm";
extern "C" const char* rmt_n2_compile_options(void) { return k_default_opts; }

extern "C" int rmt_n2_compile(const char* source, const char* arch, const char* extra_opts,
                              void** code, size_t* code_size, char** log) {
    if (!source || !code || !code_size) return fail("rmt_n2_compile: null argument");
    *code = nullptr;
    *code_size = 0;
    if (log) *log = nullptr;
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, source, "rmt_n2_generated.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return fail("hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    std::string archopt = std::string("--offload-arch=") + (arch && *arch ? arch : "gfx950");
    std::vector<std::string> store = {archopt};
    // an -mllvm switch may be given only once: a caller that sets machine-licm itself replaces the default
    const bool own_licm = extra_opts && strstr(extra_opts, "machine-licm");
    for (std::string s : {std::string(own_licm ? "-O3 -std=c++17" : k_default_opts), std::string(extra_opts ? extra_opts : "")}) {
        size_t pos = 0;
        while (pos < s.size()) {
            size_t sp = s.find(' ', pos);
            if (sp == std::string::npos) sp = s.size();
            if (sp > pos) store.push_back(s.substr(pos, sp - pos));
            pos = sp + 1;
        }
    }
    std::vector<const char*> opts;
    for (auto& s : store) opts.push_back(s.c_str());
    r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t logsz = 0;
    hiprtcGetProgramLogSize(prog, &logsz);
    std::string logs(logsz, '\0');
    if (logsz) hiprtcGetProgramLog(prog, &logs[0]);
    if (log && logsz) {
        *log = (char*)malloc(logsz + 1);
        memcpy(*log, logs.c_str(), logsz);
        (*log)[logsz] = 0;
    }
    if (r != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        return fail("hiprtcCompileProgram: %s\n%.1500s", hiprtcGetErrorString(r), logs.c_str());
    }
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    void* buf = malloc(sz);
    if (!buf) {
        hiprtcDestroyProgram(&prog);
        return fail("out of host memory for %zu byte code object", sz);
    }
    hiprtcGetCode(prog, (char*)buf);
    hiprtcDestroyProgram(&prog);
    *code = buf;
    *code_size = sz;
    return 0;
}

extern "C" int rmt_n2_create(const rmt_n2_plan* p, rmt_n2_handle** out) {
    if (!p || !out) return fail("rmt_n2_create: null argument");
    *out = nullptr;
    if (p->abi_version != RMT_N2_ABI_VERSION)
        return fail("ABI version mismatch: plan %d, library %d", p->abi_version, RMT_N2_ABI_VERSION);
    if (p->n_species < 1 || p->n_nodes < 2 || p->n_members < 1)
        return fail("bad plan sizes S=%d N=%d E=%d", p->n_species, p->n_nodes, p->n_members);
    if (p->n_vars != p->n_species && p->n_vars != p->n_species + 1)
        return fail("n_vars must be S or S+1 (got %d for S=%d)", p->n_vars, p->n_species);
    if (p->block < 64 || p->block > 1024 || p->block % 64)
        return fail("block must be a multiple of 64 in [64,1024] (got %d)", p->block);
    if (p->nodes_per_thread < 1) return fail("nodes_per_thread must be >= 1");
    if (p->n_user_params < 0 || p->n_user_params > 64) return fail("n_user_params must be in [0,64] (got %d)", p->n_user_params);
    if (p->ros4_nodes_per_block != 0 && p->ros4_nodes_per_block != p->block && p->ros4_nodes_per_block * 4 != p->block)
        return fail("ros4_nodes_per_block must be 0, block or block / 4 (got %d for block %d)", p->ros4_nodes_per_block, p->block);
    if (!p->code_object || !p->code_size || !p->members) return fail("plan lacks code object/members");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: the N2 integrator has no CPU fallback");
    rmt_n2_handle* h = new rmt_n2_handle();
    h->S = p->n_species;
    h->V = p->n_vars;
    h->NU = p->n_user_params;
    h->ros_nb = p->ros4_nodes_per_block > 0 ? p->ros4_nodes_per_block : p->block;
    h->N = p->n_nodes;
    h->E = p->n_members;
    h->fp32 = p->fp32;
    h->block = p->block;
    h->npt = p->nodes_per_thread;
    h->real_size = p->fp32 ? 4 : 8;
#define CREATE_OK(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fail("%s failed: %s", #call, hipGetErrorString(e_));                            \
            rmt_n2_destroy(h);                                                              \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
    CREATE_OK(hipGetDevice(&h->device));
    CREATE_OK(hipModuleLoadData(&h->module, p->code_object));
    CREATE_OK(hipModuleGetFunction(&h->f_rhs, h->module, "rmt_n2_rhs"));
    CREATE_OK(hipModuleGetFunction(&h->f_rk4_reg, h->module, "rmt_n2_rk4_reg"));
    CREATE_OK(hipModuleGetFunction(&h->f_rk4_mem, h->module, "rmt_n2_rk4_mem"));
    // code objects whose on-chip RK4 stepper caches the temperature-only rate constants (RMT_KCACHE) carry the plain
    // stepper as a second kernel: it re-integrates the reactors the cached one gave up on
    if (hipModuleGetFunction(&h->f_rk4_redo, h->module, "rmt_n2_rk4_reg_redo") != hipSuccess)
        h->f_rk4_redo = nullptr;
    if (hipModuleGetFunction(&h->f_rk45_reg, h->module, "rmt_n2_rk45_reg") != hipSuccess)
        h->f_rk45_reg = nullptr;
    if (hipModuleGetFunction(&h->f_rk45_mem, h->module, "rmt_n2_rk45_mem") != hipSuccess)
        h->f_rk45_mem = nullptr;
    if (hipModuleGetFunction(&h->f_rk45_chain, h->module, "rmt_n2_rk45_chain") != hipSuccess)
        h->f_rk45_chain = nullptr;
    if (hipModuleGetFunction(&h->f_multistep, h->module, "rmt_n2_multistep_mem") != hipSuccess)
        h->f_multistep = nullptr;
    if (hipModuleGetFunction(&h->f_rk4_chain, h->module, "rmt_n2_rk4_chain") != hipSuccess)
        h->f_rk4_chain = nullptr;
    if (hipModuleGetFunction(&h->f_rk4_chain_redo, h->module, "rmt_n2_rk4_chain_redo") != hipSuccess)
        h->f_rk4_chain_redo = nullptr;
    if (hipModuleGetFunction(&h->f_ros4, h->module, "rmt_n2_ros4_mem") != hipSuccess)
        h->f_ros4 = nullptr;
    if (hipModuleGetFunction(&h->f_n1, h->module, "rmt_n1_ros4") != hipSuccess) h->f_n1 = nullptr;
    if (hipModuleGetFunction(&h->f_ros4_chain, h->module, "rmt_n2_ros4_chain") != hipSuccess) h->f_ros4_chain = nullptr;
    (void)hipGetLastError();
    CREATE_OK(hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const size_t mbytes = (size_t)h->E * (RMT_N2_MEMBER_FIXED + h->S + h->NU) * sizeof(double);
    CREATE_OK(hipMalloc((void**)&h->d_members, mbytes));
    CREATE_OK(hipMemcpy(h->d_members, p->members, mbytes, hipMemcpyHostToDevice));
    // one status word per reactor + one word behind them: how many reactor-launches the cached RK4 steppers handed to
    // their plain twins so far (rmt_n2_fallbacks)
    // (a second word behind them: non-zero = rmt_n2_rk4_reg_redo integrates EVERY reactor, the cached stepper is skipped)
    CREATE_OK(hipMalloc((void**)&h->d_flags, ((size_t)h->E + 2) * sizeof(unsigned)));
    CREATE_OK(hipMemset(h->d_flags, 0, ((size_t)h->E + 2) * sizeof(unsigned)));
    CREATE_OK(hipHostMalloc((void**)&h->fb_host, 2 * sizeof(unsigned), hipHostMallocDefault));
    h->fb_host[0] = h->fb_host[1] = 0u;
    CREATE_OK(hipEventCreateWithFlags(&h->ev_fb, hipEventDisableTiming));
    CREATE_OK(hipEventCreate(&h->ev0));
    CREATE_OK(hipEventCreate(&h->ev1));
#undef CREATE_OK
    *out = h;
    return 0;
}

extern "C" void rmt_n2_destroy(rmt_n2_handle* h) {
    if (!h) return;
    DeviceGuard guard_(h);
    if (h->fb_pending && h->ev_fb) (void)hipEventSynchronize(h->ev_fb);       // the counter copy into fb_host has landed
    if (h->d_members) (void)hipFree(h->d_members);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_work) (void)hipFree(h->d_work);
    if (h->d_backup) (void)hipFree(h->d_backup);
    if (h->d_redo) (void)hipFree(h->d_redo);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_members1) (void)hipFree(h->d_members1);
    if (h->d_sync) (void)hipFree(h->d_sync);
    if (h->d_slots) (void)hipFree(h->d_slots);
    if (h->d_rings) (void)hipFree(h->d_rings);
    if (h->ev_fb) (void)hipEventDestroy(h->ev_fb);
    if (h->fb_host) (void)hipHostFree(h->fb_host);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->module) (void)hipModuleUnload(h->module);
    delete h;
}

extern "C" int rmt_n2_set_stream(rmt_n2_handle* h, void* s) {
    if (!h) return fail("null handle");
    h->stream = (hipStream_t)s;
    return 0;
}

extern "C" int rmt_n2_set_mode(rmt_n2_handle* h, int mode) {
    if (!h) return fail("null handle");
    if (mode < 0 || mode > 3) return fail("mode must be 0 (auto), 1 (on-chip), 2 (memory) or 3 (chained)");
    h->mode = mode;
    return 0;
}

extern "C" int rmt_n2_set_members(rmt_n2_handle* h, const double* members) {
    if (!h || !members) return fail("null argument");
    ON_DEVICE(h);
    const size_t mbytes = (size_t)h->E * (RMT_N2_MEMBER_FIXED + h->S + h->NU) * sizeof(double);
    HIP_OK(hipMemcpyAsync(h->d_members, members, mbytes, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}

static int ensure_work(rmt_n2_handle* h, size_t arrays, size_t extra_bytes = 0) {
    const size_t need = arrays * (size_t)h->E * h->V * h->N * h->real_size + extra_bytes;
    if (h->work_bytes >= need) return 0;
    if (h->d_work) {
        HIP_OK(hipStreamSynchronize(h->stream));
        HIP_OK(hipFree(h->d_work));
        h->d_work = nullptr;
        h->work_bytes = 0;
    }
    HIP_OK(hipMalloc(&h->d_work, need));
    h->work_bytes = need;
    return 0;
}

// tagged-word links of the chained steppers: [links] rings of RMT_N2_RING entries of 2(V+1) words, then 2 decision
// words and one abort word per team; cleared before every launch (tag 0 never matches)
static int ensure_rings(rmt_n2_handle* h, int T, int C, unsigned long long** decision, unsigned** abort_words) {
    const size_t words = 2 * (size_t)(h->V + 1);
    const size_t ring_words = (size_t)T * C * RMT_N2_RING * words;
    const size_t need = (ring_words + 2 * (size_t)T) * sizeof(unsigned long long) + (size_t)T * sizeof(unsigned);
    if (h->ring_bytes < need) {
        if (h->d_rings) { HIP_OK(hipStreamSynchronize(h->stream)); HIP_OK(hipFree(h->d_rings)); }
        h->d_rings = nullptr; h->ring_bytes = 0;
        HIP_OK(hipMalloc((void**)&h->d_rings, need));
        h->ring_bytes = need;
    }
    HIP_OK(hipMemsetAsync(h->d_rings, 0, need, h->stream));
    *decision = h->d_rings + ring_words;
    *abort_words = (unsigned*)(*decision + 2 * (size_t)T);
    return 0;
}

static int launch(rmt_n2_handle* h, hipFunction_t f, void** args, int grid = -1, int chunks = 1, int teams = 0,
                  hipFunction_t then = nullptr) {
    h->last_chunks = chunks;
    h->last_teams = teams > 0 ? teams : h->E;
    HIP_OK(hipEventRecord(h->ev0, h->stream));
    HIP_OK(hipModuleLaunchKernel(f, (unsigned)(grid > 0 ? grid : h->E), 1, 1, (unsigned)h->block, 1, 1, 0,
                                 h->stream, args, nullptr));
    if (then)          // a follow-up kernel with the same arguments and geometry, inside the timed region
        HIP_OK(hipModuleLaunchKernel(then, (unsigned)(grid > 0 ? grid : h->E), 1, 1, (unsigned)h->block, 1, 1, 0,
                                     h->stream, args, nullptr));
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

extern "C" int rmt_n2_last_geometry(rmt_n2_handle* h, int* chunks, int* teams) {
    if (!h || !chunks || !teams) return fail("null argument");
    *chunks = h->last_chunks;
    *teams = h->last_teams;
    return 0;
}

extern "C" int rmt_n2_last_kernel_ms(rmt_n2_handle* h, float* ms) {
    if (!h || !ms) return fail("null argument");
    if (!h->timed) return fail("no launch recorded yet");
    ON_DEVICE(h);
    HIP_OK(hipEventSynchronize(h->ev1));
    HIP_OK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

extern "C" int rmt_n2_rhs(rmt_n2_handle* h, double t, const void* y, void* dydt) {
    (void)t; /* the N2 right-hand side is autonomous (pbHomoReactor.py:3706: t unused) */
    if (!h || !y || !dydt) return fail("null argument");
    ON_DEVICE(h);
    int N = h->N, E = h->E;
    void* args[] = {(void*)&y, (void*)&dydt, (void*)&h->d_members, (void*)&N, (void*)&E, (void*)&h->d_flags};
    // persistent workgroups: at most RMT_N2_RHS_WGS_PER_CU per CU, each walking reactors e, e + grid, ...
    static const int per_cu = [] {               // tuning override, read once
        const char* v = getenv("RMT_N2_RHS_WGS_PER_CU");
        const int n = v ? atoi(v) : 0;
        return n > 0 ? n : RMT_N2_RHS_WGS_PER_CU;
    }();
    const long long cap = (long long)h->n_cus * per_cu;
    return launch(h, h->f_rhs, args, h->E < cap ? h->E : (int)cap);
}

static bool fits_registers(const rmt_n2_handle* h) { return h->N <= h->block * h->npt; }

// Caching RK4 steppers: did the last cached launch lose at least half of its reactors to the plain stepper (a transient
// faster than the cache's range serves)?  Then the cached pass would only be run in vain for a while: true = this call
// runs the plain stepper alone.  The counter behind the status words is copied back behind every cached launch (pinned
// word + event, never waited for).
static bool plain_this_call(rmt_n2_handle* h) {
    if (h->fb_pending && hipEventQuery(h->ev_fb) == hipSuccess) {
        h->fb_pending = false;
        const unsigned lost = h->fb_host[0] - h->fb_seen;
        h->fb_seen = h->fb_host[0];
        if (2u * lost >= (unsigned)h->E) h->plain_left = RMT_N2_PLAIN_LAUNCHES;
    }
    if (h->plain_left <= 0) return false;
    --h->plain_left;
    return true;
}

static int copy_back_fallbacks(rmt_n2_handle* h) {
    if (h->fb_pending) return 0;
    HIP_OK(hipMemcpyAsync(h->fb_host, h->d_flags + h->E, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipEventRecord(h->ev_fb, h->stream));
    h->fb_pending = true;
    return 0;
}

static bool capturing(rmt_n2_handle* h) {      // (the policy reads an event and copies to host memory: not under capture)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return h->stream && hipStreamIsCapturing(h->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}

extern "C" int rmt_n2_rk4(rmt_n2_handle* h, void* y, double t0, double dt, int64_t nsteps) {
    (void)t0;
    if (!h || !y) return fail("null argument");
    if (!(dt > 0) || nsteps < 0) return fail("rk4 needs dt > 0 and nsteps >= 0");
    ON_DEVICE(h);
    int N = h->N, E = h->E;
    long long ns = (long long)nsteps;
    const bool reg = h->mode == 1 || (h->mode == 0 && fits_registers(h));
    if (reg) {
        if (!fits_registers(h))
            return fail("register-resident stepper holds at most %d nodes per reactor (N=%d)",
                        h->block * h->npt, h->N);
        void* args[] = {(void*)&y, (void*)&h->d_members, (void*)&N, (void*)&dt, (void*)&ns,
                        (void*)&h->d_flags};
        if (!h->f_rk4_redo) return launch(h, h->f_rk4_reg, args);
        // (a stream that is being captured into a graph gets the two kernels and nothing else)
        if (capturing(h)) {
            HIP_OK(hipModuleLaunchKernel(h->f_rk4_reg, (unsigned)h->E, 1, 1, (unsigned)h->block, 1, 1, 0, h->stream, args, nullptr));
            HIP_OK(hipModuleLaunchKernel(h->f_rk4_redo, (unsigned)h->E, 1, 1, (unsigned)h->block, 1, 1, 0, h->stream, args, nullptr));
            return 0;
        }
        if (plain_this_call(h)) {        // flags[E + 1] != 0: rmt_n2_rk4_reg_redo integrates every reactor
            HIP_OK(hipMemsetAsync(h->d_flags + h->E + 1, 1, sizeof(unsigned), h->stream));
            const int rc = launch(h, h->f_rk4_redo, args);
            HIP_OK(hipMemsetAsync(h->d_flags + h->E + 1, 0, sizeof(unsigned), h->stream));
            return rc;
        }
        if (launch(h, h->f_rk4_reg, args, -1, 1, 0, h->f_rk4_redo)) return 1;
        return copy_back_fallbacks(h);
    }
    // chained workgroups: C chunks per reactor, T teams, every workgroup resident (T*C <= #CUs)
    const int W = h->block * h->npt;
    int C = (h->N + W - 1) / W;
    const bool can_chain = h->f_rk4_chain && C >= 2 && C <= h->n_cus;
    if (h->mode == 3 && !can_chain)
        return fail("chained stepper needs 2 <= chunks (%d) <= CUs (%d)", C, h->n_cus);
    if (h->mode == 3 || (h->mode == 0 && can_chain)) {
        int T = h->n_cus / C;
        if (T > h->E) T = h->E;
        const size_t links = (size_t)T * C;
        if (h->chain_links < links) {
            if (h->d_sync) { HIP_OK(hipStreamSynchronize(h->stream)); HIP_OK(hipFree(h->d_sync)); HIP_OK(hipFree(h->d_slots)); }
            h->d_sync = nullptr; h->d_slots = nullptr; h->chain_links = 0;
            HIP_OK(hipMalloc((void**)&h->d_sync, links * 32 * sizeof(unsigned long long)));
            HIP_OK(hipMalloc((void**)&h->d_slots, links * 8 * (size_t)(h->V + 1) * sizeof(double)));
            h->chain_links = links;
        }
        // a chained stepper that caches the temperature-only rate constants (RMT_KCACHE_CHAIN) saves the launch's input
        // and is followed by rmt_n2_rk4_chain_redo on the same grid, see kernels/50_rk4.inc rmt_rk4_chain_body
        void* backup = nullptr;
        unsigned *redo = nullptr, *falt = nullptr;
        if (h->f_rk4_chain_redo) {
            const size_t state = (size_t)h->E * h->V * h->N * (h->fp32 ? 4 : 8);
            if (!h->d_backup) {
                HIP_OK(hipMalloc(&h->d_backup, state));
                HIP_OK(hipMalloc((void**)&h->d_redo, 2 * (size_t)h->E * sizeof(unsigned)));
            }
            backup = h->d_backup;
            redo = h->d_redo;
            falt = h->d_redo + h->E;
            HIP_OK(hipMemsetAsync(h->d_redo, 0, 2 * (size_t)h->E * sizeof(unsigned), h->stream));
        }
        // the same policy as for the one-workgroup stepper: after a launch that lost half of its reactors the plain
        // chained stepper runs alone for a while - every redo word set (to a value the fallback counter does not count),
        // the input copied to the backup buffer the plain stepper reads
        const bool plain = h->f_rk4_chain_redo && !capturing(h) && plain_this_call(h);
        if (plain) {
            const size_t state = (size_t)h->E * h->V * h->N * (h->fp32 ? 4 : 8);
            HIP_OK(hipMemcpyAsync(h->d_backup, y, state, hipMemcpyDeviceToDevice, h->stream));
            HIP_OK(hipMemsetAsync(h->d_redo, 2, (size_t)h->E * sizeof(unsigned), h->stream));
        }
        unsigned long long* decision = nullptr;
        unsigned* abort_words = nullptr;
        void* args[] = {(void*)&y, (void*)&h->d_members, (void*)&N, (void*)&E, (void*)&C, (void*)&T,
                        (void*)&dt, (void*)&ns, (void*)&h->d_sync, (void*)&h->d_slots, (void*)&h->d_flags,
                        (void*)&h->d_rings, (void*)&abort_words, (void*)&backup, (void*)&redo, (void*)&falt};
        h->last_chunks = C;
        h->last_teams = T;
        HIP_OK(hipEventRecord(h->ev0, h->stream));
        for (hipFunction_t f : {plain ? (hipFunction_t) nullptr : h->f_rk4_chain, h->f_rk4_chain_redo}) {
            if (!f) continue;
            HIP_OK(hipMemsetAsync(h->d_sync, 0, links * 32 * sizeof(unsigned long long), h->stream));
            if (ensure_rings(h, T, C, &decision, &abort_words)) return 1;        // (clears the rings)
            HIP_OK(hipModuleLaunchKernel(f, (unsigned)(T * C), 1, 1, (unsigned)h->block, 1, 1, 0, h->stream, args, nullptr));
        }
        HIP_OK(hipEventRecord(h->ev1, h->stream));
        h->timed = true;
        if (h->f_rk4_chain_redo && !plain && !capturing(h)) return copy_back_fallbacks(h);
        return 0;
    }
    if (ensure_work(h, 3)) return 1;
    void* args[] = {(void*)&y, (void*)&h->d_work, (void*)&h->d_members, (void*)&N, (void*)&E,
                    (void*)&dt, (void*)&ns, (void*)&h->d_flags};
    return launch(h, h->f_rk4_mem, args);
}

extern "C" int rmt_n2_multistep(rmt_n2_handle* h, void* y, double t0, double dt, int64_t nsteps,
                                int method) {
    (void)t0;
    if (!h || !y) return fail("null argument");
    if (!(dt > 0) || nsteps < 3) return fail("multistep needs dt > 0 and nsteps >= 3");
    if (method != 0 && method != 1) return fail("method must be 0 (AdBash3) or 1 (PreCorr3)");
    if (!h->f_multistep) return fail("code object has no multistep kernel");
    ON_DEVICE(h);
    if (ensure_work(h, 8)) return 1;
    int N = h->N, E = h->E;
    long long ns = (long long)nsteps;
    void* args[] = {(void*)&y, (void*)&h->d_work, (void*)&h->d_members, (void*)&N, (void*)&E,
                    (void*)&dt, (void*)&ns, (void*)&method, (void*)&h->d_flags};
    return launch(h, h->f_multistep, args);
}

extern "C" int rmt_n2_rk45(rmt_n2_handle* h, void* y, double t0, double t1, double rtol, double atol,
                           double h0, int64_t max_steps, rmt_n2_stats* stats) {
    if (!h || !y || !stats) return fail("null argument");
    if (!(t1 > t0) || !(rtol > 0) || !(atol >= 0) || !(h0 != 0)) return fail("bad rk45 arguments");
    ON_DEVICE(h);
    int N = h->N, E = h->E;
    long long ms = (long long)max_steps;
    const bool reg = (h->mode == 1 || (h->mode == 0 && fits_registers(h))) && h->f_rk45_reg;
    if (reg) {
        if (!fits_registers(h)) return fail("register-resident rk45 does not fit N=%d", h->N);
        void* args[] = {(void*)&y, (void*)&h->d_members, (void*)&N, (void*)&t0, (void*)&t1,
                        (void*)&rtol, (void*)&atol, (void*)&h0, (void*)&ms, (void*)&stats,
                        (void*)&h->d_flags};
        return launch(h, h->f_rk45_reg, args);
    }
    // Reactors longer than one workgroup's on-chip capacity: C chunks per reactor on C CUs, T teams, every
    // workgroup resident (rmt_n2_rk45_chain; the code object has it when the on-chip slots fit the geometry).
    {
        const int W = h->block * h->npt;
        int C = (h->N + W - 1) / W;
        const bool can_chain = h->f_rk45_chain && C >= 2 && C <= h->n_cus && C <= RMT_N2_MAX_CHUNKS;
        if (h->mode == 3 && !can_chain)
            return fail("chained rk45 needs the on-chip stepper in the code object and 2 <= chunks (%d) <= %d", C,
                        h->n_cus < RMT_N2_MAX_CHUNKS ? h->n_cus : RMT_N2_MAX_CHUNKS);
        if (can_chain && (h->mode == 3 || h->mode == 0)) {
            int T = h->n_cus / C;
            if (T > h->E) T = h->E;
            unsigned long long* decision = nullptr;
            unsigned* abort_words = nullptr;
            if (ensure_rings(h, T, C, &decision, &abort_words)) return 1;
            void* args[] = {(void*)&y, (void*)&h->d_members, (void*)&N, (void*)&E, (void*)&C, (void*)&T, (void*)&t0,
                            (void*)&t1, (void*)&rtol, (void*)&atol, (void*)&h0, (void*)&ms, (void*)&stats,
                            (void*)&h->d_flags, (void*)&h->d_rings, (void*)&decision, (void*)&abort_words};
            return launch(h, h->f_rk45_chain, args, T * C, C, T);
        }
    }
    if (!h->f_rk45_mem) return fail("code object has no rk45 kernel");
    if (ensure_work(h, 10)) return 1;      // K_1..K_7 (when they do not fit in LDS), y_new, K_1/K_7 of the FSAL swap
    void* args[] = {(void*)&y, (void*)&h->d_work, (void*)&h->d_members, (void*)&N, (void*)&E,
                    (void*)&t0, (void*)&t1, (void*)&rtol, (void*)&atol, (void*)&h0, (void*)&ms,
                    (void*)&stats, (void*)&h->d_flags};
    return launch(h, h->f_rk45_mem, args);
}

extern "C" int rmt_n2_ros4(rmt_n2_handle* h, void* y, double t0, double t1, double rtol, double atol,
                           double h0, int64_t max_steps, rmt_n2_stats* stats) {
    if (!h || !y || !stats) return fail("null argument");
    if (!(t1 > t0) || !(rtol > 0) || !(atol >= 0) || !(h0 != 0)) return fail("bad ros4 arguments");
    if (!h->f_ros4) return fail("code object has no ros4 kernel");
    ON_DEVICE(h);
    if (h->block > 512)
        return fail("the Rosenbrock kernel holds a VxV matrix per lane: generate the code object with "
                    "block <= 512 (got %d)", h->block);
    // One reactor over several CUs (rmt_n2_ros4_chain): C chunks of W nodes per reactor, T teams, every
    // workgroup resident (T*C <= #CUs).  Auto mode chains when the ensemble alone cannot fill the device.
    const int nblocks = (h->N + h->ros_nb - 1) / h->ros_nb;      // node blocks per reactor
    int C = 1;
    if (h->f_ros4_chain && h->npt == 1 && nblocks >= 2 && h->mode != 2) {
        // Chunks of ONE node block pipeline stage by stage (a chunk lags its upstream neighbour by a message
        // latency, ~0.3 stage-block times); chunks of b > 1 blocks run all six stages of a block before the
        // next one (the block's stage vectors live in LDS), so a downstream chunk starts 6(b-1) stage-block
        // times late.  Estimated step time in stage-block units, minimised over the chunk count:
        // The T = #CUs / C teams of a launch work through the reactors in rounds of T, so the job costs
        // ceil(E / T) rounds of that step time - against ceil(E / #CUs) rounds of 6 nblocks for one workgroup per
        // reactor.  (Few chunks of several blocks lose to many one-block chunks run in more rounds: 8 reactors of
        // 16384 nodes take 2.99 s as 32 chunks x 8 teams and 0.36 s as 64 chunks x 4 teams x 2 rounds.)
        int cmax = nblocks < h->n_cus ? nblocks : h->n_cus;
        if (cmax > RMT_N2_MAX_CHUNKS) cmax = RMT_N2_MAX_CHUNKS;
        double best = 6.0 * nblocks * ((h->E + h->n_cus - 1) / h->n_cus);
        for (int c = 2; c <= cmax; ++c) {
            const int b = (nblocks + c - 1) / c;
            const int cc = (nblocks + b - 1) / b;
            int teams = h->n_cus / cc;
            if (teams > h->E) teams = h->E;
            const int rounds = (h->E + teams - 1) / teams;
            // (x 1.35: a chained stage-block is slower than a resident one - its sweeps exchange the chunk
            // boundary over the links; calibrated on the 4096-node reactor, 97 us per step in 16 chunks)
            const double cost = 1.35 * rounds * (6.0 * b + (cc - 1) * (6.0 * (b - 1) + 0.3));
            if (cost < best || (h->mode == 3 && C < 2)) { best = cost; C = cc; }
        }
        if (const char* force = getenv("RMT_N2_ROS4_CHUNKS")) {       // tuning experiments: chunk count by hand
            const int c = atoi(force);
            if (c >= 1 && c <= cmax) C = c;
        }
    }
    if (h->mode == 3 && C < 2)
        return fail("chained stiff stepper needs >= 2 node blocks per reactor (N=%d block=%d E=%d CUs=%d)",
                    h->N, h->block, h->E, h->n_cus);
    if (C >= 2) {
        const int bpc = (nblocks + C - 1) / C;          // node blocks per chunk
        C = (nblocks + bpc - 1) / bpc;
        int W = bpc * h->ros_nb;
        int T = h->n_cus / C;
        if (T > h->E) T = h->E;
        if (ensure_work(h, 1)) return 1;                // y_new
        unsigned long long* decision = nullptr;
        unsigned* abort_words = nullptr;
        if (ensure_rings(h, T, C, &decision, &abort_words)) return 1;
        int N = h->N, E = h->E;
        long long ms = (long long)max_steps;
        void* args[] = {(void*)&y, (void*)&h->d_work, (void*)&h->d_members, (void*)&N, (void*)&E, (void*)&C, (void*)&T,
                        (void*)&W, (void*)&t0, (void*)&t1, (void*)&rtol, (void*)&atol, (void*)&h0, (void*)&ms,
                        (void*)&stats, (void*)&h->d_flags, (void*)&h->d_rings, (void*)&decision, (void*)&abort_words};
        return launch(h, h->f_ros4_chain, args, T * C, C, T);
    }
    // 7 vector arrays + the VxV inverse per node (= V more "vector arrays")
    if (ensure_work(h, 8 + (size_t)h->V)) return 1;   // 7 stage arrays + VxV inverses + upwind coupling (model M2)
    const size_t nmask = (size_t)h->E * h->N;
    if (h->mask_elems < nmask) {
        if (h->d_mask) { HIP_OK(hipStreamSynchronize(h->stream)); HIP_OK(hipFree(h->d_mask)); }
        h->d_mask = nullptr; h->mask_elems = 0;
        HIP_OK(hipMalloc((void**)&h->d_mask, nmask * sizeof(unsigned)));
        h->mask_elems = nmask;
    }
    int N = h->N, E = h->E;
    long long ms = (long long)max_steps;
    void* args[] = {(void*)&y, (void*)&h->d_work, (void*)&h->d_mask, (void*)&h->d_members, (void*)&N,
                    (void*)&E, (void*)&t0, (void*)&t1, (void*)&rtol, (void*)&atol, (void*)&h0,
                    (void*)&ms, (void*)&stats, (void*)&h->d_flags};
    return launch(h, h->f_ros4, args);
}

extern "C" int rmt_n1_profile(rmt_n2_handle* h, const double* members1, void* out, int nout, double rtol,
                              double atol, double h0, int64_t max_steps, rmt_n2_stats* stats) {
    if (!h || !members1 || !out || !stats) return fail("null argument");
    if (nout < 2 || !(rtol > 0) || !(atol >= 0) || !(h0 > 0)) return fail("bad N1 arguments");
    if (!h->f_n1) return fail("code object has no N1 kernel");
    ON_DEVICE(h);
    const size_t mbytes = (size_t)h->E * (RMT_N2_MEMBER_FIXED + h->S + h->NU) * sizeof(double);
    if (!h->d_members1) HIP_OK(hipMalloc((void**)&h->d_members1, mbytes));
    HIP_OK(hipMemcpyAsync(h->d_members1, members1, mbytes, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    int E = h->E;
    long long ms = (long long)max_steps;
    void* args[] = {(void*)&h->d_members1, (void*)&out, (void*)&E, (void*)&nout, (void*)&rtol, (void*)&atol,
                    (void*)&h0, (void*)&ms, (void*)&stats, (void*)&h->d_flags};
    HIP_OK(hipEventRecord(h->ev0, h->stream));
    HIP_OK(hipModuleLaunchKernel(h->f_n1, (unsigned)((E + 63) / 64), 1, 1, 64, 1, 1, 0, h->stream, args, nullptr));
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

extern "C" int rmt_n2_fallbacks(rmt_n2_handle* h, uint64_t* count) {
    if (!h || !count) return fail("null argument");
    ON_DEVICE(h);
    unsigned c = 0;
    HIP_OK(hipMemcpyAsync(&c, h->d_flags + h->E, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    *count = c;
    return 0;
}

extern "C" int rmt_n2_status(rmt_n2_handle* h, uint32_t* flags_host) {
    if (!h || !flags_host) return fail("null argument");
    ON_DEVICE(h);
    HIP_OK(hipMemcpyAsync(flags_host, h->d_flags, (size_t)h->E * sizeof(unsigned), hipMemcpyDeviceToHost,
                          h->stream));
    HIP_OK(hipMemsetAsync(h->d_flags, 0, (size_t)h->E * sizeof(unsigned), h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return 0;
}
