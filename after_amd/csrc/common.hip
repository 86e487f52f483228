#include "common.h"

namespace after {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int Arena::init(size_t bytes) {
    release();
    bytes = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        base = nullptr;
        return AFTER_E_NOMEM;
    }
    cap = bytes;
    off = 0;
    return AFTER_OK;
}

void Arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = off = 0;
}

int KernelTimer::enable(bool on) {
    if (on && !ev) {
        ev = new hipEvent_t[2 * kMax];
        for (int i = 0; i < 2 * kMax; ++i) {
            if (hipEventCreate(&ev[i]) != hipSuccess) {
                set_error("hipEventCreate failed");
                return AFTER_E_HIP;
            }
        }
    }
    enabled = on;
    n = 0;
    flops = bytes = 0;
    return AFTER_OK;
}

int KernelTimer::collect(double* ms, long long* launches, double* fl, double* by) {
    double tot = 0;
    for (int i = 0; i < n; ++i) {
        float t = 0;
        hipError_t e = hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]);
        if (e != hipSuccess) {
            set_error("hipEventElapsedTime failed: %s (sync the stream first)",
                      hipGetErrorString(e));
            return AFTER_E_HIP;
        }
        tot += t;
    }
    if (ms) *ms = tot;
    if (launches) *launches = n;
    if (fl) *fl = flops;
    if (by) *by = bytes;
    n = 0;
    flops = bytes = 0;
    return AFTER_OK;
}

void KernelTimer::destroy() {
    if (ev) {
        for (int i = 0; i < 2 * kMax; ++i) (void)hipEventDestroy(ev[i]);
        delete[] ev;
        ev = nullptr;
    }
    enabled = false;
}

}  // namespace after

extern "C" const char* after_last_error(void) { return after::g_err; }
extern "C" const char* after_version(void) { return "after_hip gfx950 r1"; }
