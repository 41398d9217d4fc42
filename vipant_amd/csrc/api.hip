// Error plumbing, version and device checks of libvipant_hip.so.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

static thread_local char g_err[512] = "";

void vipant_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vipant_last_error(void) { return g_err; }

extern "C" int32_t vipant_version(void) { return 100; }  // 0.1.0

extern "C" int32_t vipant_device_check(void) {
    int dev = 0;
    VIPANT_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    VIPANT_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vipant_set_error("device %d is %s; libvipant_hip.so is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return VIPANT_EHIP;
    }
    return VIPANT_OK;
}

// The ticket block of a (device, stream) pair (common.h): allocated and zeroed on the pair's first persistent launch, kept for the
// life of the process.  Contract: the alternation of the two counter sets follows the order in which launches REACH the
// stream, so ticket launches on one stream must be issued by one host thread at a time (the library's own callers are: one Python
// thread per process), and such a launch cannot be captured into a hipGraph and replayed -- a replay would reuse a counter set nobody
// zeroed.  VIPANT_GEMM_VARIANT bit 22 (static walk) lifts both restrictions.  Returns the counter set this launch uses and, in *other, the set it has to zero for the stream's next one.
uint32_t* vipant_ticket_block(hipStream_t stream, uint32_t** other) {
    struct Block { uint32_t* base; int turn; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Block> blocks;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        vipant_set_error("ticket block: hipGetDevice failed");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = blocks.find({dev, stream});
    if (it == blocks.end()) {
        uint32_t* p = nullptr;
        if (hipMalloc((void**)&p, VIPANT_TICKET_WORDS * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(p, 0, VIPANT_TICKET_WORDS * sizeof(uint32_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            vipant_set_error("ticket block: allocation of %zu bytes failed", VIPANT_TICKET_WORDS * sizeof(uint32_t));
            return nullptr;
        }
        it = blocks.emplace(std::make_pair(dev, stream), Block{p, 0}).first;
    }
    Block& b = it->second;
    uint32_t* mine = b.base + 8 * b.turn;
    *other = b.base + 8 * (b.turn ^ 1);
    b.turn ^= 1;
    return mine;
}
