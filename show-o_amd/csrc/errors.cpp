// error reporting for the C ABI
#include "common.h"
#include "../../include/showo_hip.h"
#include <cstdio>
#include <cstring>

namespace showo {
static thread_local char g_err[512] = "";
int set_error_hip(hipError_t e, const char* what, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return 1000 + (int)e;
}
int set_error_msg(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
}  // namespace showo

extern "C" const char* showo_last_error(void) { return showo::g_err; }
extern "C" int showo_abi_version(void) { return SHOWO_ABI_VERSION; }
extern "C" int showo_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return showo::set_error_hip(e, "hipGetDevice", __FILE__, __LINE__);
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return showo::set_error_hip(e, "hipGetDeviceProperties", __FILE__, __LINE__);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

// ---- per-launch event timing -------------------------------------------------------------------------
#include "prof.h"
#include <vector>
namespace showo {
struct ProfRec { int kind; double work; hipEvent_t a, b; };
static bool g_prof_on = false;
static int g_prof_stride = 1;            // time every stride-th launch of a kind (systematic sample; 1 = all)
static int64_t g_prof_seen[PROF_KINDS] = {0, 0, 0};
static double g_prof_work_all[PROF_KINDS] = {0, 0, 0};
static std::vector<ProfRec> g_prof;
ProfScope::ProfScope(int kind, double work, hipStream_t stream) : idx(-1), s(stream) {
    if (!g_prof_on) return;
    g_prof_work_all[kind] += work;
    if ((g_prof_seen[kind]++ % g_prof_stride) != 0) return;
    ProfRec r;
    r.kind = kind; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
}
bool g_prof_on_query() { return g_prof_on; }
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, s);
}
}  // namespace showo

extern "C" int showo_prof_enable(int on) {
    showo::g_prof_on = on != 0;
    return 0;
}
extern "C" int showo_prof_reset(void) {
    for (auto& r : showo::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    showo::g_prof.clear();
    for (int k = 0; k < showo::PROF_KINDS; ++k) { showo::g_prof_seen[k] = 0; showo::g_prof_work_all[k] = 0; }
    return 0;
}
// time only every stride-th launch of each kind (a stride coprime to the per-layer launch pattern samples every shape)
extern "C" int showo_prof_set_stride(int stride) {
    showo::g_prof_stride = stride > 0 ? stride : 1;
    return 0;
}
// launches seen / work submitted (timed or not) since the last reset
extern "C" int showo_prof_totals(int kind, int64_t* launches, double* work) {
    if (kind < 0 || kind >= showo::PROF_KINDS) return showo::set_error_msg(1, "prof_totals: bad kind");
    if (launches) *launches = showo::g_prof_seen[kind];
    if (work) *work = showo::g_prof_work_all[kind];
    return 0;
}
// kind: 0 gemm, 1 attention, 2 conv.  Returns summed elapsed ms, launch count and summed work (flops).
extern "C" int showo_prof_read(int kind, double* total_ms, int64_t* launches, double* work) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return showo::set_error_hip(e, "hipDeviceSynchronize", __FILE__, __LINE__);
    double ms = 0, w = 0;
    int64_t n = 0;
    for (auto& r : showo::g_prof) {
        if (r.kind != kind) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms += t; w += r.work; n++; }
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (work) *work = w;
    return 0;
}

// ---- compute stream that leaves CUs to a concurrent collective -------------------------------------------------------------------
// Data-parallel training overlaps the RCCL all-reduce of a finished gradient bucket with the backward of the next block.  The GEMM /
// conv kernels launch one 512-thread block per CU on all 256 CUs; RCCL's channel kernels need CUs of their own, and on a full chip they
// queue behind whole GEMM tiles.  A stream created with a CU mask keeps `reserve` CUs (spread evenly over the 8 XCDs: every 256/reserve-th
// CU id) out of every kernel launched on it; RCCL's own streams are unmasked and find those CUs idle.  reserve = 0: plain stream.
// Live masked streams of this process (stream -> reserve).  Everything that sizes a grid from the CU count asks showo_cu_usable(stream):
// split-K targets of gemm2p / gemm_tn / the split-precision conv, the tile-height model, the residency test of the cooperative split-K
// reduction -- ONE place, so a launch on a masked stream is sized for the CUs it may actually use (grids sized to 256
// CUs fell into a second round on 240 and cost 20-45 %).
#include <map>
#include <mutex>
static std::mutex g_mask_mu;
static std::map<hipStream_t, int> g_masked;
extern "C" int showo_cu_reserved_max(void) {
    std::lock_guard<std::mutex> lock(g_mask_mu);
    int m = 0;
    for (auto& kv : g_masked) m = kv.second > m ? kv.second : m;
    return m;
}
static int device_cus() {
    static int cus = 0;
    if (!cus && showo_device_info(&cus, nullptr, nullptr, 0)) cus = 256;
    return cus;
}
extern "C" int showo_cu_usable(void* stream) {
    std::lock_guard<std::mutex> lock(g_mask_mu);
    auto it = g_masked.find((hipStream_t)stream);
    return device_cus() - (it == g_masked.end() ? 0 : it->second);
}
extern "C" int showo_stream_create_cu_mask(int reserve, void** out) {
    if (!out || reserve < 0) return showo::set_error_msg(1, "stream_create_cu_mask: bad argument");
    const int cus = device_cus();
    if (reserve >= cus) return showo::set_error_msg(1, "stream_create_cu_mask: reserve must be smaller than the CU count");
    constexpr int XCC = 8;  // MI355X: 8 XCDs x 32 CUs; kernels are dealt to the XCDs round-robin, so every XCD must lose the same number
    if (reserve % XCC || cus % XCC) return showo::set_error_msg(1, "stream_create_cu_mask: reserve must be a multiple of 8 (the same number of CUs per XCD)");
    hipStream_t s = nullptr;
    hipError_t e;
    if (reserve == 0) {
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    } else {
        // KFD interleaves the mask bits across the XCCs: bit i is CU (i / 8) of XCC (i % 8).  Each XCC loses reserve / 8 CUs, spread
        // over its 32 (clearing every (256 / reserve)-th bit took them all from XCC 0).
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
        for (int i = 0; i < cus; ++i) mask[i >> 5] |= 1u << (i & 31);
        const int per = reserve / XCC, cpx = cus / XCC;
        for (int x = 0; x < XCC; ++x)
            for (int j = 0; j < per; ++j) {
                const int i = x + XCC * (int)(((int64_t)j * cpx) / per + cpx / (2 * per));
                mask[i >> 5] &= ~(1u << (i & 31));
            }
        // (a stream created this way is a BLOCKING stream: it synchronises with the legacy null stream, unlike the reserve = 0 form)
        e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    }
    if (e != hipSuccess) return showo::set_error_hip(e, "stream create (CU mask)", __FILE__, __LINE__);
    if (reserve > 0) {
        std::lock_guard<std::mutex> lock(g_mask_mu);
        g_masked[s] = reserve;
    }
    *out = (void*)s;
    return 0;
}
extern "C" int showo_stream_destroy(void* stream) {
    if (!stream) return 0;
    {
        std::lock_guard<std::mutex> lock(g_mask_mu);
        g_masked.erase((hipStream_t)stream);
    }
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? 0 : showo::set_error_hip(e, "hipStreamDestroy", __FILE__, __LINE__);
}
