// Stream / hipGraph / event plumbing behind the C ABI (no torch types; everything is a hipStream_t).
#include "dc_common.h"
#include "dcrafter_hip.h"

extern "C" int dc_stream_create(void** out) {
    if (!out) return DC_ERR_ARG;
    hipStream_t s;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    *out = (void*)s;
    return 0;
}
extern "C" int dc_stream_destroy(void* s) { return (int)hipStreamDestroy((hipStream_t)s); }
extern "C" int dc_stream_sync(void* s) { return (int)hipStreamSynchronize((hipStream_t)s); }

extern "C" int dc_graph_begin_capture(void* s) {
    return (int)hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal);
}
extern "C" int dc_graph_end_capture(void* s, void** exec_out) {
    if (!exec_out) return DC_ERR_ARG;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)s, &graph);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *exec_out = (void*)exec;
    return 0;
}
extern "C" int dc_graph_launch(void* exec, void* s) { return (int)hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)s); }
extern "C" int dc_graph_destroy(void* exec) { return (int)hipGraphExecDestroy((hipGraphExec_t)exec); }

extern "C" int dc_event_create(void** out) {
    if (!out) return DC_ERR_ARG;
    hipEvent_t ev;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return (int)e;
    *out = (void*)ev;
    return 0;
}
extern "C" int dc_event_record(void* ev, void* s) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)s); }
extern "C" int dc_event_elapsed_ms(void* a, void* b, float* ms) {
    if (!ms) return DC_ERR_ARG;
    hipError_t e = hipEventSynchronize((hipEvent_t)b);
    if (e != hipSuccess) return (int)e;
    return (int)hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b);
}
extern "C" int dc_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

// ---- error word (dc_common.h): a __device__ int of this code object; its address per device is looked up once
__device__ int g_dc_error_word = 0;

int* dc_error_word_device() {
    static std::atomic<int*> addr[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    int* p = addr[dev].load(std::memory_order_acquire);
    if (!p) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_dc_error_word)) != hipSuccess || !sym) return nullptr;
        p = static_cast<int*>(sym);
        addr[dev].store(p, std::memory_order_release);
    }
    return p;
}

extern "C" int dc_error_word_read(int* out, int reset) {
    if (!out) return DC_ERR_ARG;
    int* const p = dc_error_word_device();
    if (!p) return DC_ERR_ARG;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(out, p, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (reset && *out) e = hipMemset(p, 0, sizeof(int));
    return (int)e;
}

extern "C" const char* dc_version(void) { return "dcrafter_hip 0.1 (gfx950)"; }
